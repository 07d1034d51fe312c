#!/usr/bin/env python
"""FPD training entry point with the reference's command line (/root/reference/tools/fpd_train.py:44-83):

    python tools/fpd_train.py --cfg S.yaml --tcfg T.yaml [KEY VALUE ...]
    python -m torch.distributed.run --nproc-per-node 8 tools/fpd_train.py --cfg ... --tcfg ...      (data parallel)

Same flow as the reference's main() (:96-294): config merge, student `models.<NAME>.get_pose_net(cfg, is_train=True)`,
teacher from the cloned config merged with --tcfg, strict teacher-checkpoint load, two JointsMSELoss criteria,
Adam + MultiStepLR, epoch loop -> core.function.fpd_train -> checkpoint.  What differs: one process per GPU with an
RCCL gradient all-reduce instead of nn.DataParallel; the models run on the HIP path; DATASET.DATASET 'synthetic'
(the default -- MPII/COCO are not available offline) feeds seeded synthetic crops, and KD.TEACHER 'synthetic' builds a
random teacher with calibrated BN statistics instead of loading a checkpoint.
"""
import argparse
import logging
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('GPU_MAX_HW_QUEUES', '6' if int(os.environ.get('WORLD_SIZE', '1')) > 1 else '8')      # before torch loads the HIP runtime; see bench.py / DESIGN.md section 4

import torch  # noqa: E402
import torch.utils.data  # noqa: E402

from fpd_amd import dist as fdist, executor as E, synth  # noqa: E402
from fpd_amd.lib import models  # noqa: E402,F401
from fpd_amd.lib.config import cfg, update_config  # noqa: E402
from fpd_amd.lib.core.function import fpd_train, train, validate  # noqa: E402
from fpd_amd.lib.dataset import SyntheticPose  # noqa: E402
from fpd_amd.lib.core.loss import JointsMSELoss  # noqa: E402
from fpd_amd.lib.utils.utils import (get_model_summary, get_optimizer, load_checkpoint, multistep_lr,  # noqa: E402
                                     save_checkpoint)


def parse_args():
    p = argparse.ArgumentParser(description='Train keypoints network (FPD)')
    p.add_argument('--cfg', help='student experiment configure file name', required=True, type=str)
    p.add_argument('--tcfg', help='teacher experiment configure file name', default='', type=str)
    p.add_argument('opts', help='Modify config options using the command-line', default=None, nargs=argparse.REMAINDER)
    p.add_argument('--modelDir', default='', type=str)
    p.add_argument('--logDir', default='', type=str)
    p.add_argument('--dataDir', default='', type=str)
    p.add_argument('--max-iters', type=int, default=0, help='stop every epoch after this many iterations (smoke runs)')
    return p.parse_args()


def get_train_type(train_type, checkpoint):
    """tools/fpd_train.py:85-94."""
    if train_type == 'NORMAL':
        return train_type
    if train_type == 'FPD' and (checkpoint == 'synthetic' or (checkpoint and os.path.exists(checkpoint))):
        return 'FPD'
    if train_type == 'FPD':
        sys.exit('ERROR: teacher checkpoint is not existed.')
    sys.exit('ERROR: please change train type {} to NORMAL or FPD.'.format(train_type))


def main():
    args = parse_args()
    update_config(cfg, args)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    logging.basicConfig(level=logging.INFO if rank == 0 else logging.WARNING, format='%(asctime)-15s %(message)s')
    logger = logging.getLogger()
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
    train_type = get_train_type(cfg.KD.TRAIN_TYPE, cfg.KD.TEACHER)
    out_dir = os.path.join(cfg.OUTPUT_DIR, cfg.DATASET.DATASET, cfg.MODEL.NAME,
                           os.path.basename(args.cfg).split('.')[0])
    os.makedirs(out_dir, exist_ok=True)

    torch.manual_seed(1)
    model = eval('models.' + cfg.MODEL.NAME + '.get_pose_net')(cfg, is_train=True)          # :122-124
    tcfg = cfg.clone()                                                                        # :128-131
    if args.tcfg:
        tcfg.merge_from_file(args.tcfg)
    if rank == 0:                                                                             # :162-167 (shape-only walk)
        logger.info(get_model_summary(model, torch.empty(1, 3, cfg.MODEL.IMAGE_SIZE[1], cfg.MODEL.IMAGE_SIZE[0], device='meta')))
    tmodel = None
    if train_type == 'FPD':
        torch.manual_seed(2)
        tmodel = eval('models.' + tcfg.MODEL.NAME + '.get_pose_net')(tcfg, is_train=False)   # :135-137
        if cfg.KD.TEACHER != 'synthetic':
            load_checkpoint(cfg.KD.TEACHER, tmodel, strict=True, model_info='teacher_' + tcfg.MODEL.NAME)   # :139-141
        tmodel = fdist.DataParallelReplica(tmodel.to(dev))
    if cfg.TRAIN.CHECKPOINT:
        load_checkpoint(cfg.TRAIN.CHECKPOINT, model, strict=True, model_info='student_' + cfg.MODEL.NAME)
    model = fdist.DataParallelReplica(model.to(dev))
    if world > 1:
        fdist.broadcast_state(dist, model.module)
        if tmodel is not None:
            fdist.broadcast_state(dist, tmodel.module)

    pose_criterion = JointsMSELoss(use_target_weight=cfg.LOSS.USE_TARGET_WEIGHT).to(dev)     # :145-147,177-179
    kd_pose_criterion = JointsMSELoss(use_target_weight=tcfg.LOSS.USE_TARGET_WEIGHT).to(dev)

    if cfg.DATASET.DATASET != 'synthetic':
        sys.exit('dataset %r is not available here (CPU-side MPII/COCO pipeline is out of scope); use DATASET.DATASET synthetic'
                 % cfg.DATASET.DATASET)
    bs = cfg.TRAIN.BATCH_SIZE_PER_GPU
    train_set = SyntheticPose(cfg, cfg.DATASET.NUM_SAMPLES, seed=rank)
    loader = torch.utils.data.DataLoader(train_set, batch_size=bs, shuffle=cfg.TRAIN.SHUFFLE, num_workers=0,
                                         pin_memory=cfg.PIN_MEMORY, drop_last=True, collate_fn=train_set.collate)
    valid_set = SyntheticPose(cfg, cfg.DATASET.NUM_VALID_SAMPLES, seed=1009)
    valid_loader = torch.utils.data.DataLoader(valid_set, batch_size=cfg.TEST.BATCH_SIZE_PER_GPU, shuffle=False, num_workers=0,
                                               pin_memory=cfg.PIN_MEMORY, collate_fn=valid_set.collate)
    if args.max_iters:
        import itertools
        full = loader

        class _Limited(object):
            def __iter__(self):
                return itertools.islice(iter(full), args.max_iters)

            def __len__(self):
                return min(len(full), args.max_iters)
        loader = _Limited()
    if tmodel is not None and cfg.KD.TEACHER == 'synthetic':          # give the random teacher sane eval-mode BN statistics once
        x0 = next(iter(loader))[0]
        t = tmodel.module
        cal = E.GraphInstance(t.device_state(), t.cfg_hg, x0.shape[0], x0.shape[2], x0.shape[3], train=True).finalize()
        cal.image().copy_(x0)
        cal.run('prep'); cal.run('fwd')
        cal.calibrate_running_stats()
        del cal

    optimizer = get_optimizer(cfg, model)                                                     # :218
    begin_epoch, best_perf = cfg.TRAIN.BEGIN_EPOCH, 0.0
    ckpt = os.path.join(out_dir, 'checkpoint.pth')
    if cfg.AUTO_RESUME and os.path.exists(ckpt):                                              # :224-234
        state = torch.load(ckpt, map_location=dev)
        begin_epoch, best_perf = state['epoch'], state['perf']
        model.load_state_dict(state['state_dict'])
        optimizer.load_state_dict(state['optimizer'])
        logger.info("=> loaded checkpoint '%s' (epoch %d)", ckpt, state['epoch'])
    base_lr = float(optimizer.param_groups[0].get('initial_lr', cfg.TRAIN.LR))               # :236-239 MultiStepLR
    optimizer.param_groups[0]['initial_lr'] = base_lr
    allreduce = fdist.make_allreduce(dist) if world > 1 else None
    writer_dict = {'writer': None, 'train_global_steps': 0, 'valid_global_steps': 0}
    if rank == 0:
        # :243-250: before the first epoch the reference evaluates the teacher and then the student on the validation set
        # (the two "Test:" blocks in front of epoch 0 in its logs).  Both go through core.function.validate; rank 0 only, like
        # the per-epoch validation below.
        if tmodel is not None:
            validate(cfg, valid_loader, valid_set, tmodel, pose_criterion, out_dir, cfg.LOG_DIR, writer_dict)
        validate(cfg, valid_loader, valid_set, model, pose_criterion, out_dir, cfg.LOG_DIR, writer_dict)
    if world > 1:
        dist.barrier()
    for epoch in range(begin_epoch, cfg.TRAIN.END_EPOCH):                                     # :252-286
        t0 = time.time()
        optimizer.param_groups[0]['lr'] = multistep_lr(base_lr, cfg.TRAIN.LR_STEP, cfg.TRAIN.LR_FACTOR, epoch)   # :253
        if train_type == 'FPD':                                                               # :255-264
            loss = fpd_train(cfg, loader, model, tmodel, pose_criterion, kd_pose_criterion, optimizer, epoch, out_dir,
                             cfg.LOG_DIR, writer_dict, allreduce=allreduce, world_size=world)
        else:
            loss = train(cfg, loader, model, pose_criterion, optimizer, epoch, out_dir, cfg.LOG_DIR, writer_dict,
                         allreduce=allreduce, world_size=world)
        torch.cuda.synchronize()
        logger.info('=> epoch %d done in %.1fs, %.1f samples/s, last logged loss %.5f', epoch, time.time() - t0,
                    len(loader) * bs * world / max(time.time() - t0, 1e-9), loss)
        if rank == 0:
            # :266-285: evaluate on the validation set (flip test etc. per cfg.TEST), keep the best model
            perf_indicator = validate(cfg, valid_loader, valid_set, model, pose_criterion, out_dir, cfg.LOG_DIR, writer_dict)
            best_model = perf_indicator >= best_perf
            best_perf = max(best_perf, perf_indicator)
            save_checkpoint({'epoch': epoch + 1, 'model': cfg.MODEL.NAME, 'state_dict': model.state_dict(),
                             'best_state_dict': model.module.state_dict(), 'perf': perf_indicator,
                             'optimizer': optimizer.state_dict()}, best_model, out_dir)
        if world > 1:
            dist.barrier()       # ranks != 0 wait for rank 0's validation + checkpoint HERE, explicitly, rather than inside
                                 # the first gradient all-reduce of the next epoch
    if rank == 0:
        torch.save(model.module.state_dict(), os.path.join(out_dir, 'final_state.pth'))      # :288-294
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
