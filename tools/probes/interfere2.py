#!/usr/bin/env python
"""What slows the student chain down when another stream is busy?  The chain alone, then beside (a) ~800 tiny kernels per step
on a second stream (launch-rate / cache-maintenance interference, no CU pressure), (b) the same number of 4 MB fills (HBM
traffic bursts), (c) the real teacher forward.   python tools/probes/interfere2.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import torch
import bench as Bn
from fpd_amd import executor as E, synth
from fpd_amd.lib.models import hourglass
dev = torch.device('cuda', 0)
B, J, H, W = 32, 16, 256, 256
torch.manual_seed(1); student = hourglass.get_pose_net(Bn.make_cfg(128, 4, J, 'bf16'), True).to(dev)
torch.manual_seed(2); teacher = hourglass.get_pose_net(Bn.make_cfg(256, 8, J, 'bf16'), False).to(dev)
x, tg, tw = synth.make_batch(1000, B, J, (W, H), (W // 4, H // 4))
step = E.FusedFPDStep(student.device_state(), student.cfg_hg, teacher.device_state(), teacher.cfg_hg, B, H, W, alpha=0.5)
step.set_batch(x, tg, tw)
step.run_pipelined(3)
torch.cuda.synchronize()
s = step.student
side = torch.cuda.Stream()
tiny = torch.zeros(64, device=dev)
big = torch.zeros(1 << 20, device=dev)          # 4 MB
mm_a = torch.randn(2048, 2048, device=dev, dtype=torch.bfloat16)

def chain():
    for ph in ('prep', 'fwd', 'mid', 'bwd', 'adam'):
        s.run(ph)

def timed(bg, n=10):
    def one():
        if bg is not None:
            with torch.cuda.stream(side):
                bg()
        chain()
    one(); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n):
        one()
    torch.cuda.synchronize()
    return (time.time() - t0) / n * 1e3

def tiny_bg(k):
    def f():
        for _ in range(k):
            tiny.add_(1.0)
    return f
def fill_bg(k):
    def f():
        for _ in range(k):
            big.add_(1.0)
    return f
def mm_bg(k):
    def f():
        for _ in range(k):
            torch.mm(mm_a, mm_a)
    return f
print('chain alone                      %.2f ms' % timed(None))
print('chain + 400 tiny kernels         %.2f ms' % timed(tiny_bg(400)))
print('chain + 1600 tiny kernels        %.2f ms' % timed(tiny_bg(1600)))
print('chain + 400 x 4 MB add_          %.2f ms' % timed(fill_bg(400)))
print('chain + 40 x bf16 mm 2048^3      %.2f ms' % timed(mm_bg(40)))
print('chain + teacher fwd              %.2f ms' % timed(lambda: step.teachers[0].run('fwd')))
print('teacher fwd alone                %.2f ms' % (lambda: (torch.cuda.synchronize(), time.time(), [step.teachers[0].run('fwd') for _ in range(10)], torch.cuda.synchronize(), time.time()))().__getitem__(0) if False else '')
t0 = time.time()
for _ in range(10): step.teachers[0].run('fwd')
torch.cuda.synchronize()
print('teacher fwd alone                %.2f ms' % ((time.time() - t0) / 10 * 1e3))
