#!/usr/bin/env python
"""What about a concurrent kernel slows the student chain down?  The chain alone, then beside background kernels on a second
stream that hold `blocks` compute units for ~6.4 ms per step the way the fused teacher Bottleneck does (512-thread
persistent blocks owning their CU's LDS) while they (0) only spin, (1) stream an L2-resident buffer, (2) stream HBM,
(3) keep the matrix pipes busy -- and beside the real teacher forward.  Builds tools/probes/occupy_probe.hip on the box.
   python tools/probes/occupy_probe.py"""
import ctypes as C
import os
import subprocess
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import torch
import bench as Bn
from fpd_amd import executor as E, synth
from fpd_amd.lib.models import hourglass
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(os.environ.get('TMPDIR', '/tmp'), 'occupy_probe.so')
subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-O2', '-shared', '-fPIC', os.path.join(here, 'occupy_probe.hip'), '-o', so])
lib = C.CDLL(so)
lib.occupy_launch.argtypes = [C.c_int, C.c_int, C.c_int, C.c_longlong, C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p]
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
small = torch.zeros(512 * 1024 // 4, dtype=torch.float32, device=dev)          # 512 KB: stays in every XCD's L2
large = torch.zeros(1 << 28, dtype=torch.float32, device=dev)                   # 1 GB: HBM (larger than the 256 MB MALL)
sink = torch.zeros(4, dtype=torch.float32, device=dev)


def chain():
    for ph in ('prep', 'fwd', 'mid', 'bwd', 'adam'):
        s.run(ph)


def bg(mode, blocks, lds, ms):
    buf = small if mode == 1 else large
    def f():
        rc = lib.occupy_launch(mode, blocks, lds, int(ms * 1e5), buf.data_ptr(), buf.numel() // 4, sink.data_ptr(), side.cuda_stream)
        assert rc == 0, rc
    return f


def timed(bgf, n=10):
    def one():
        if bgf is not None:
            side.wait_stream(torch.cuda.current_stream())
            bgf()
        chain()
    one(); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n):
        one()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    return (time.time() - t0) / n * 1e3


print('chain alone                                   %.2f ms' % timed(None))
names = {0: 'spin', 1: 'L2 stream', 2: 'HBM stream', 3: 'MFMA loop'}
for blocks, lds in ((128, 150 * 1024), (128, 16 * 1024), (64, 150 * 1024), (256, 16 * 1024)):
    for mode in (0, 1, 2, 3):
        print('chain + %-10s %3d blocks, %3d KB LDS, 6.4 ms  %.2f ms' % (names[mode], blocks, lds // 1024, timed(bg(mode, blocks, lds, 6.4))))


def teacher_bg():
    with torch.cuda.stream(side):
        step.teachers[0].run('fwd')
print('chain + teacher fwd                           %.2f ms' % timed(teacher_bg))
