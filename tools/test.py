#!/usr/bin/env python
"""Validation entry point with the reference's command line (/root/reference/tools/test.py:38-135):

    python tools/test.py --cfg S.yaml [TEST.MODEL_FILE model.pth] [KEY VALUE ...]

Same flow as the reference's main(): config merge, `models.<NAME>.get_pose_net(cfg, is_train=False)`, weights from
TEST.MODEL_FILE (strict=False like the reference, :88-90) or <output dir>/final_state.pth (:91-96), JointsMSELoss,
validation loader, `core.function.validate` (flip test / heat-map shift / post-processing per cfg.TEST).  What differs: the
model runs on the HIP path of one GPU; DATASET.DATASET 'synthetic' (MPII / COCO archives are not available offline) feeds
the seeded synthetic validation set of tools/fpd_train.py."""
import argparse
import logging
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

import torch  # noqa: E402
import torch.utils.data  # noqa: E402

from fpd_amd import dist as fdist  # noqa: E402
from fpd_amd.lib import models  # noqa: E402,F401
from fpd_amd.lib.config import cfg, update_config  # noqa: E402
from fpd_amd.lib.core.function import validate  # noqa: E402
from fpd_amd.lib.core.loss import JointsMSELoss  # noqa: E402
from fpd_amd.lib.dataset import SyntheticPose  # noqa: E402
from fpd_amd.lib.utils.utils import load_checkpoint  # noqa: E402


def parse_args():
    p = argparse.ArgumentParser(description='Test keypoints network')
    p.add_argument('--cfg', help='experiment configure file name', required=True, type=str)
    p.add_argument('opts', help='Modify config options using the command-line', default=None, nargs=argparse.REMAINDER)
    p.add_argument('--modelDir', default='', type=str)
    p.add_argument('--logDir', default='', type=str)
    p.add_argument('--dataDir', default='', type=str)
    return p.parse_args()


def main():
    args = parse_args()
    update_config(cfg, args)
    logging.basicConfig(level=logging.INFO, format='%(asctime)-15s %(message)s')
    logger = logging.getLogger()
    torch.cuda.set_device(0)
    dev = torch.device('cuda', 0)
    out_dir = os.path.join(cfg.OUTPUT_DIR, cfg.DATASET.DATASET, cfg.MODEL.NAME, os.path.basename(args.cfg).split('.')[0])
    os.makedirs(out_dir, exist_ok=True)

    torch.manual_seed(1)
    model = eval('models.' + cfg.MODEL.NAME + '.get_pose_net')(cfg, is_train=False)          # :84-86
    model_file = cfg.TEST.MODEL_FILE or os.path.join(out_dir, 'final_state.pth')              # :88-96
    logger.info('=> loading model from %s', model_file)
    load_checkpoint(model_file, model, strict=not cfg.TEST.MODEL_FILE, model_info=cfg.MODEL.NAME)
    model = fdist.DataParallelReplica(model.to(dev))
    criterion = JointsMSELoss(use_target_weight=cfg.LOSS.USE_TARGET_WEIGHT).to(dev)          # :101-103

    if cfg.DATASET.DATASET != 'synthetic':
        sys.exit('dataset %r is not available here (CPU-side MPII/COCO pipeline is out of scope); use DATASET.DATASET synthetic'
                 % cfg.DATASET.DATASET)
    valid_set = SyntheticPose(cfg, cfg.DATASET.NUM_VALID_SAMPLES, seed=1009)
    valid_loader = torch.utils.data.DataLoader(valid_set, batch_size=cfg.TEST.BATCH_SIZE_PER_GPU, shuffle=False, num_workers=0,
                                               pin_memory=cfg.PIN_MEMORY, collate_fn=valid_set.collate)
    perf = validate(cfg, valid_loader, valid_set, model, criterion, out_dir, cfg.LOG_DIR)    # :130-132
    logger.info('=> validation done: perf indicator %.4f, loss %.5f, accuracy %.4f', perf, validate.last['loss'], validate.last['acc'])


if __name__ == '__main__':
    main()
