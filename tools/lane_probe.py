"""Dev probe: host enqueue time vs GPU time of the student/teacher phases, per lane configuration."""
import os, sys, time
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import make_cfg
from fpd_amd import executor as E, runtime as R, synth
from fpd_amd.lib.models import hourglass
R.lib()
dev = torch.device('cuda', 0)
B = 32
torch.manual_seed(1)
student = hourglass.get_pose_net(make_cfg(128, 4, 16, 'bf16'), is_train=True).to(dev)
torch.manual_seed(2)
teacher = hourglass.get_pose_net(make_cfg(256, 8, 16, 'bf16'), is_train=False).to(dev)
x, tg, tw = synth.make_batch(1000, B, 16, (256, 256), (64, 64))
step = E.FusedFPDStep(student.device_state(), student.cfg_hg, teacher.device_state(), teacher.cfg_hg, B, 256, 256, alpha=0.5)
step.set_batch(x, tg, tw)
step.run_pipelined(2)
torch.cuda.synchronize()
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n): fn()
    th = time.time() - t0
    torch.cuda.synchronize()
    return th / n * 1e3, (time.time() - t0) / n * 1e3
s, t = step.student, step.teacher
for name, fn in [('teacher fwd', lambda: t.run('fwd')), ('student fwd', lambda: s.run('fwd')), ('student bwd', lambda: s.run('bwd'))]:
    h, g = timeit(fn)
    print('%-12s host %.2f ms   total %.2f ms' % (name, h, g), flush=True)
if os.environ.get('GRAPH', '0') == '1':
    step.enable_graphs()
    for name, fn in [('teacher fwd', lambda: t.run('fwd')), ('student fwd', lambda: s.run('fwd')), ('student bwd', lambda: s.run('bwd'))]:
        h, g = timeit(fn)
        print('graph %-12s host %.2f ms   total %.2f ms' % (name, h, g), flush=True)
