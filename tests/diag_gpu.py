"""GPU diagnostic (not a test): per-parameter gradient error of the fused FPD step vs the fp64 oracle, next to the
reference's own fp32 error.  python tests/diag_gpu.py [tiny|cfg1]"""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import _cases
from tests.test_model_gpu import build_models

def main(name):
    from fpd_amd import executor as E
    c, gold, student, teacher = build_models(name)
    tr = _cases.truth64(name)
    step = E.FusedFPDStep(student.device_state(), student.cfg_hg, teacher.device_state(), teacher.cfg_hg,
                          c['batch'], c['image'][1], c['image'][0], alpha=0.5)
    x, tg, tw = _cases.batch(name, 0)
    step.set_batch(x, tg, tw)
    step.step()
    torch.cuda.synchronize()
    student._attach_grads()
    stride = int(gold['grad_stride'])
    rows, off = [], 0
    named = dict(student.named_parameters())
    for k in student.table.trainable_keys():
        t = tr['grads'][k].reshape(-1).numpy()
        o = named[k].grad.reshape(-1).cpu().numpy().astype(np.float64)
        n = t.size
        line = [np.abs(o - t).max(), np.abs(t).max(), k]
        if stride == 1:
            r = gold['grad_flat'][off:off + n]
            line.insert(1, np.abs(r - t).max())
        else:
            line.insert(1, float('nan'))
        off += n
        rows.append(line)
    rows.sort(key=lambda r: -r[0])
    print('%s: top gradient errors (ours | reference fp32 | |grad|max)' % name)
    for e, re_, m, k in rows[:15]:
        print('  %-38s %.3e  %.3e  %.3e' % (k, e, re_, m))
    for i in range(len(step.student.g.outputs)):
        o = step.student.output_view(i).permute(0, 3, 1, 2).float().cpu().numpy()
        t = tr['outputs'][i].numpy()
        print('  map %d: ours-vs-fp64 %.3e  ref32-vs-fp64 %.3e' % (i, np.abs(o - t).max(), np.abs(gold['output%d' % i] - t).max()))

if __name__ == '__main__':
    for n in (sys.argv[1:] or ['tiny', 'cfg1']):
        main(n)
