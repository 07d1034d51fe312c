#!/usr/bin/env python
"""How much does a concurrent load on ANOTHER stream slow the student chain (alone: ~9.4 ms)?
   LOAD=none|teacher|copy|mfma  python tools/probes/interference.py      (teacher grid cap via FPD_BNECK_BLOCKS / FPD_HEAD_BLOCKS)"""
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
s = step.student
load = os.environ.get('LOAD', 'none')
side = torch.cuda.Stream()
big_a = torch.empty(256 << 20, dtype=torch.uint8, device=dev); big_b = torch.empty_like(big_a)
def chain():
    for ph in ('prep', 'fwd', 'mid', 'bwd', 'adam'): s.run(ph)
def enqueue_load(n):
    with torch.cuda.stream(side):
        for _ in range(n):
            if load == 'teacher': step.teachers[0].run('fwd')
            elif load == 'copy': big_b.copy_(big_a)
N = 12
chain(); torch.cuda.synchronize()
# enough side work to cover the whole measurement
enqueue_load({'teacher': 3 * N, 'copy': 400 * N}.get(load, 0))
t0 = time.time()
for _ in range(N): chain()
torch.cuda.current_stream().synchronize()
dt = (time.time() - t0) / N * 1e3
torch.cuda.synchronize()
print('load=%-8s bneck cap %-5s  student chain %.2f ms' % (load, os.environ.get('FPD_BNECK_BLOCKS', '160'), dt))
