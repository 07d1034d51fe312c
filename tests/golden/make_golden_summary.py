#!/usr/bin/env python
"""Golden `get_model_summary` text of the REFERENCE (lib/utils/utils.py:86-202 over lib/models/hourglass.py) for the
hg S=2 F=64 network at 1x3x256x256, terse and verbose form (build container only):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_summary.py
"""
import importlib.util
import os
import sys

import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))


def load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


class AD(dict):
    __getattr__ = dict.__getitem__


def main():
    hg = load('ref_hourglass', '/root/reference/lib/models/hourglass.py')
    ut = load('ref_utils', '/root/reference/lib/utils/utils.py')
    cfg = AD(MODEL=AD(NUM_JOINTS=16, EXTRA=AD(NUM_FEATURES=64, NUM_STACKS=2, NUM_BLOCKS=1)))
    model = hg.get_pose_net(cfg, is_train=True)
    x = torch.rand(1, 3, 256, 256)
    with open(os.path.join(HERE, 'summary_hg2x64.txt'), 'w') as f:
        f.write(ut.get_model_summary(model, x))
        f.write('\n=====VERBOSE=====\n')
        f.write(ut.get_model_summary(model, x, verbose=True))
    print(ut.get_model_summary(model, x))


if __name__ == '__main__':
    main()
