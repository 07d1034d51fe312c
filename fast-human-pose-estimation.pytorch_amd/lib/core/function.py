"""`core.function.fpd_train` with the reference's signature (/root/reference/lib/core/function.py:99-187), driving
the fused MI355X step (executor.FusedFPDStep): per batch the frozen-teacher forward, the student forward/backward, the
pose + distillation JointsMSELoss of every stack, the data-parallel gradient exchange and Adam run as recorded HIP
plans with no host synchronisation; the teacher runs one batch ahead on its own stream (it does not depend on the
student weights); losses / accuracy are read back only when a log line is due (PRINT_FREQ)."""
import logging
import time

import torch

from ... import executor as E
from ..utils.utils import FusedAdam
from .evaluate import accuracy

logger = logging.getLogger(__name__)


class AverageMeter(object):
    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count if self.count != 0 else 0


_STEPS = {}


def _unwrap(m):
    return m.module if hasattr(m, 'module') else m


def fused_step_for(model, tmodel, optimizer, batch_shape, alpha, world_size=1):
    """One FusedFPDStep per (student, teacher, batch shape); shares Adam state with a FusedAdam optimizer."""
    s, t = _unwrap(model), _unwrap(tmodel)
    key = (id(s), id(t), tuple(batch_shape), float(alpha), world_size)
    if key not in _STEPS:
        n, _, h, w = batch_shape
        step = E.FusedFPDStep(s.device_state(), s.cfg_hg, t.device_state(), t.cfg_hg, n, h, w, alpha,
                              lr=float(optimizer.param_groups[0]['lr']), world_size=world_size,
                              adam=optimizer if isinstance(optimizer, FusedAdam) else None)
        _STEPS[key] = step
    return _STEPS[key]


def fpd_train(config, train_loader, model, tmodel, pose_criterion, kd_pose_criterion, optimizer, epoch,
              output_dir, tb_log_dir, writer_dict, allreduce=None, world_size=1):
    batch_time, data_time = AverageMeter(), AverageMeter()
    losses, pose_losses, kd_pose_losses, acc = AverageMeter(), AverageMeter(), AverageMeter(), AverageMeter()
    alpha = config.KD.ALPHA
    model.train()          # function.py:110-111
    tmodel.eval()
    step = None
    end = time.time()
    it = iter(train_loader)
    nxt = next(it, None)
    i = -1
    while nxt is not None:
        i += 1
        inp, target, target_weight, meta = nxt
        data_time.update(time.time() - end)
        if step is None or tuple(inp.shape) != tuple(step.student.image().shape):
            if step is not None:
                step.flush()
            step = fused_step_for(model, tmodel, optimizer, inp.shape, alpha, world_size)
            metric = step.enable_metric()              # PCK of every iteration, accumulated on the device
            metric.drain()
            step.teacher_async(inp)                    # pipeline prologue: teacher forward of the first batch
        if isinstance(optimizer, FusedAdam):
            optimizer.sync_lr()
        else:
            step.set_lr(float(optimizer.param_groups[0]['lr']))
        step.set_batch(inp, target, target_weight)
        nxt = next(it, None)
        if nxt is not None and tuple(nxt[0].shape) == tuple(inp.shape):
            step.teacher_async(nxt[0])                 # teacher runs one batch ahead, overlapping this student step
        elif nxt is not None:
            pass                                       # shape change: the new step object primes itself above
        step.student_step(allreduce)
        if i % config.PRINT_FREQ == 0:
            pose, kd, loss = step.losses()                    # the only host sync of the loop
            n = inp.size(0)
            pose_losses.update(pose, n); kd_pose_losses.update(kd, n); losses.update(loss, n)
            for avg_acc, cnt in metric.drain():               # one entry per iteration since the last log line,
                acc.update(avg_acc, cnt)                      # exactly what function.py:154-155 feeds its meter
            batch_time.update(time.time() - end)
            msg = 'Epoch: [{0}][{1}/{2}]\t' \
                  'Time {batch_time.val:.3f}s ({batch_time.avg:.3f}s)\t' \
                  'Speed {speed:.1f} samples/s\t' \
                  'Data {data_time.val:.3f}s ({data_time.avg:.3f}s)\t' \
                  'POSE_Loss {pose_loss.val:.5f} ({pose_loss.avg:.5f})\t' \
                  'KD_POSE_Loss {kd_pose_loss.val:.5f} ({kd_pose_loss.avg:.5f})\t' \
                  'Loss {loss.val:.5f} ({loss.avg:.5f})\t' \
                  'Accuracy {acc.val:.3f} ({acc.avg:.3f})'.format(
                      epoch, i, len(train_loader), batch_time=batch_time,
                      speed=inp.size(0) * world_size / max(batch_time.val, 1e-9), data_time=data_time, pose_loss=pose_losses,
                      kd_pose_loss=kd_pose_losses, loss=losses, acc=acc)
            logger.info(msg)
            if writer_dict is not None and writer_dict.get('writer') is not None:
                w, gs = writer_dict['writer'], writer_dict['train_global_steps']
                w.add_scalar('train_pose_loss', pose_losses.val, gs)
                w.add_scalar('train_kd_pose_loss', kd_pose_losses.val, gs)
                w.add_scalar('train_loss', losses.val, gs)
                w.add_scalar('train_acc', acc.val, gs)
                writer_dict['train_global_steps'] = gs + 1
        else:
            batch_time.update(time.time() - end)
        end = time.time()
    if step is not None:
        step.flush()
        for avg_acc, cnt in metric.drain():
            acc.update(avg_acc, cnt)
    return losses.avg
