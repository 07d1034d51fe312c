#!/usr/bin/env python
"""Where a step's time goes: the student chain alone (per phase), the frozen teacher alone, and both pipelined (bench.py's mode).
   python tools/probes/phase_times.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import torch
import bench as Bn
from fpd_amd import executor as E, runtime as R, synth
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
def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / n * 1e3
s = step.student
print('teacher fwd alone      %.2f ms' % timed(lambda: step.teachers[0].run('fwd')))
for ph in ('prep', 'fwd', 'mid', 'bwd', 'adam'):
    print('student %-5s alone    %.2f ms' % (ph, timed(lambda: s.run(ph))))
def chain():
    for ph in ('prep', 'fwd', 'mid', 'bwd', 'adam'): s.run(ph)
print('student chain alone    %.2f ms' % timed(chain))
os.environ['X'] = '1'
print('pipelined step         %.2f ms' % (timed(lambda: step.run_pipelined(10), 3) / 10))
