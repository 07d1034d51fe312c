"""`core.function.fpd_train` with the reference's signature (/root/reference/lib/core/function.py:99-187), driving
the fused MI355X step (executor.FusedFPDStep): per batch the frozen-teacher forward, the student forward/backward, the
pose + distillation JointsMSELoss of every stack, the data-parallel gradient exchange and Adam run as recorded HIP
plans with no host synchronisation; the teacher runs one batch ahead on its own stream (it does not depend on the
student weights).  Like the reference (function.py:150-155) EVERY iteration feeds the loss and accuracy meters: the
device appends {avg_acc, cnt, pose, kd} of each iteration to a ring (csrc/pck.hip) that is drained -- the only host
synchronisation of the loop -- when a log line is due (PRINT_FREQ) and at the end of the epoch.

Not silently substituted: the fused step implements Adam (lib.utils.utils.FusedAdam) and JointsMSELoss criteria; any
other optimizer / criterion object raises instead of being ignored."""
import logging
import time

import torch

from ... import executor as E
from ... import runtime as R
from ..utils.utils import FusedAdam
from .evaluate import accuracy  # noqa: F401  (re-exported like the reference's core.function namespace)
from .loss import JointsMSELoss

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


def _unwrap(m):
    return m.module if hasattr(m, 'module') else m


def _check_supported(optimizer, pose_criterion, kd_pose_criterion):
    """The fused step runs ITS Adam and ITS JointsMSELoss: refuse objects it would otherwise silently ignore."""
    if not isinstance(optimizer, FusedAdam):
        raise R.FpdError('fpd_train: the fused MI355X step implements Adam only (utils.get_optimizer with TRAIN.OPTIMIZER '
                         "'adam' -> FusedAdam); got %s -- its update rule and state would be ignored" % type(optimizer).__name__)
    for name, c in (('pose_criterion', pose_criterion), ('kd_pose_criterion', kd_pose_criterion)):
        if not isinstance(c, JointsMSELoss):
            raise R.FpdError('fpd_train: %s must be a core.loss.JointsMSELoss (the fused loss kernel evaluates exactly that '
                             'criterion), got %s' % (name, type(c).__name__))
    return bool(pose_criterion.use_target_weight), bool(kd_pose_criterion.use_target_weight)


def fused_step_for(model, tmodel, optimizer, batch_shape, alpha, world_size=1, use_target_weight=(True, True)):
    """One FusedFPDStep per (student, teacher, batch shape); shares Adam state with the FusedAdam optimizer.
    tmodel None = plain (non-distillation) training: no teacher graph, alpha must be 0."""
    s, t = _unwrap(model), (_unwrap(tmodel) if tmodel is not None else None)
    # The cache lives ON the student module (not in a module-level dict keyed by id(): ids are recycled once an object
    # is collected, and a global would keep every plan and its arenas alive for the life of the process).  The teacher
    # and optimizer are compared by identity through the references the entry holds, so they cannot be recycled either.
    cache = s.__dict__.setdefault('_fused_steps', {})
    key = (tuple(batch_shape), float(alpha), world_size, tuple(use_target_weight))
    hit = cache.get(key)
    if hit is not None and hit[0] is t and hit[1] is optimizer:
        return hit[2]
    n, _, h, w = batch_shape
    assert t is not None or alpha == 0.0
    step = E.FusedFPDStep(s.device_state(), s.cfg_hg, t.device_state() if t is not None else None,
                          t.cfg_hg if t is not None else None, n, h, w, alpha,
                          lr=float(optimizer.param_groups[0]['lr']), world_size=world_size,
                          adam=optimizer, use_target_weight=use_target_weight)
    cache[key] = (t, optimizer, step)
    return step


def _run_epoch(config, train_loader, model, tmodel, use_w, optimizer, epoch, writer_dict, allreduce, world_size, alpha):
    """The loop shared by fpd_train (function.py:99-187) and train (function.py:28-96; tmodel None, alpha 0)."""
    kd_mode = tmodel is not None
    batch_time, data_time = AverageMeter(), AverageMeter()
    losses, pose_losses, kd_pose_losses, acc = AverageMeter(), AverageMeter(), AverageMeter(), AverageMeter()
    model.train()          # function.py:110-111
    if kd_mode:
        tmodel.eval()
    step = metric = None
    end = time.time()
    it = iter(train_loader)
    nxt = next(it, None)
    i = -1
    n_img = 0

    def drain():
        """Feed the meters with every iteration since the last drain (function.py:150-155); synchronises."""
        for avg_acc, cnt, pose, kd in metric.drain(full=True):
            pose_losses.update(pose, n_img); kd_pose_losses.update(kd, n_img)
            losses.update((1 - alpha) * pose + alpha * kd, n_img)
            acc.update(avg_acc, cnt)

    while nxt is not None:
        i += 1
        inp, target, target_weight, meta = nxt
        data_time.update(time.time() - end)
        if step is None or tuple(inp.shape) != tuple(step.student.image().shape):
            if step is not None:                       # batch shape changed (last, smaller batch): finish the old step
                step.flush()
                drain()
            step = fused_step_for(model, tmodel, optimizer, inp.shape, alpha, world_size, use_w)
            metric = step.enable_metric(min_slots=config.PRINT_FREQ + 1)   # PCK + losses of every iteration, logged on the device
            metric.drain()
            n_img = inp.size(0)
            step.teacher_async(inp)                    # pipeline prologue: teacher forward of the first batch
        optimizer.sync_lr()
        step.set_batch(inp, target, target_weight)
        nxt = next(it, None)
        if nxt is not None and tuple(nxt[0].shape) == tuple(inp.shape):
            step.teacher_async(nxt[0])                 # teacher runs one batch ahead, overlapping this student step
        step.student_step(allreduce)
        if i % config.PRINT_FREQ == 0:
            drain()                                    # the only host sync of the loop
            batch_time.update(time.time() - end)
            speed = inp.size(0) * world_size / max(batch_time.val, 1e-9)
            if kd_mode:
                msg = 'Epoch: [{0}][{1}/{2}]\t' \
                      'Time {batch_time.val:.3f}s ({batch_time.avg:.3f}s)\t' \
                      'Speed {speed:.1f} samples/s\t' \
                      'Data {data_time.val:.3f}s ({data_time.avg:.3f}s)\t' \
                      'POSE_Loss {pose_loss.val:.5f} ({pose_loss.avg:.5f})\t' \
                      'KD_POSE_Loss {kd_pose_loss.val:.5f} ({kd_pose_loss.avg:.5f})\t' \
                      'Loss {loss.val:.5f} ({loss.avg:.5f})\t' \
                      'Accuracy {acc.val:.3f} ({acc.avg:.3f})'.format(
                          epoch, i, len(train_loader), batch_time=batch_time, speed=speed, data_time=data_time,
                          pose_loss=pose_losses, kd_pose_loss=kd_pose_losses, loss=losses, acc=acc)
            else:
                msg = 'Epoch: [{0}][{1}/{2}]\t' \
                      'Time {batch_time.val:.3f}s ({batch_time.avg:.3f}s)\t' \
                      'Speed {speed:.1f} samples/s\t' \
                      'Data {data_time.val:.3f}s ({data_time.avg:.3f}s)\t' \
                      'Loss {loss.val:.5f} ({loss.avg:.5f})\t' \
                      'Accuracy {acc.val:.3f} ({acc.avg:.3f})'.format(
                          epoch, i, len(train_loader), batch_time=batch_time, speed=speed, data_time=data_time,
                          loss=losses, acc=acc)
            logger.info(msg)
            if writer_dict is not None and writer_dict.get('writer') is not None:
                w, gs = writer_dict['writer'], writer_dict['train_global_steps']
                if kd_mode:
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
        drain()
    return {'loss': losses.avg, 'pose_loss': pose_losses.avg, 'kd_pose_loss': kd_pose_losses.avg, 'acc': acc.avg,
            'iterations': losses.count // max(n_img, 1) if n_img else 0}


def fpd_train(config, train_loader, model, tmodel, pose_criterion, kd_pose_criterion, optimizer, epoch,
              output_dir, tb_log_dir, writer_dict, allreduce=None, world_size=1):
    """function.py:99-187 (same positional signature; allreduce / world_size are the data-parallel extras).  Returns the
    epoch's average loss (the reference returns None)."""
    use_w = _check_supported(optimizer, pose_criterion, kd_pose_criterion)
    return _run_epoch(config, train_loader, model, tmodel, use_w, optimizer, epoch, writer_dict, allreduce, world_size,
                      float(config.KD.ALPHA))['loss']


def train(config, train_loader, model, criterion, optimizer, epoch, output_dir, tb_log_dir, writer_dict,
          allreduce=None, world_size=1):
    """function.py:28-96: plain (non-distillation) training = the same fused step without a teacher graph (alpha 0)."""
    use_w = _check_supported(optimizer, criterion, criterion)
    return _run_epoch(config, train_loader, model, None, use_w, optimizer, epoch, writer_dict, allreduce, world_size, 0.0)['loss']


def _to_numpy(v):
    import numpy as np
    return v.numpy() if torch.is_tensor(v) else np.asarray(v)


def validate(config, val_loader, val_dataset, model, criterion, output_dir, tb_log_dir, writer_dict=None):
    """function.py:189-332 with the reference's signature and return value (the dataset's perf indicator).

    Per batch: eval-mode forward (the folded-BN plan: fused frozen Bottlenecks / heads in bf16, the parity kernels in
    fp32), TEST.FLIP_TEST second forward on the width-flipped batch, flip_back + TEST.SHIFT_HEATMAP shift + average
    (one kernel), JointsMSELoss, PCK accuracy (csrc/pck.hip), and get_final_preds (arg-max, TEST.POST_PROCESS
    quarter-pixel shift, affine map to image coordinates: one kernel).  The reference moves every one of these tensors
    to the host and back (np.flip / flip_back / accuracy / get_final_preds are numpy); here only the final
    [N,J,2]+[N,J,1] predictions and two scalars per batch cross PCIe.  The result arrays handed to
    `val_dataset.evaluate(...)` have the reference's layout (all_preds [num,J,3], all_boxes [num,6])."""
    import numpy as np

    from ..utils.transforms import flip_input, flip_merge, get_affine_transform
    from .evaluate import DeviceAccuracy
    from .inference import final_preds_device
    batch_time, losses, acc = AverageMeter(), AverageMeter(), AverageMeter()
    model.eval()
    net = _unwrap(model)
    dev = net._flat['param'].device
    num_samples = len(val_dataset)
    all_preds = np.zeros((num_samples, config.MODEL.NUM_JOINTS, 3), dtype=np.float32)
    all_boxes = np.zeros((num_samples, 6))
    image_path, filenames, imgnums = [], [], []
    idx = 0
    metric = None
    with torch.no_grad():
        end = time.time()
        for i, (inp, target, target_weight, meta) in enumerate(val_loader):
            inp = inp.to(dev, non_blocking=True)
            outputs = model(inp)
            output = outputs[-1] if isinstance(outputs, list) else outputs
            if config.TEST.FLIP_TEST:
                outputs_flipped = model(flip_input(inp))
                output_flipped = outputs_flipped[-1] if isinstance(outputs_flipped, list) else outputs_flipped
                output = flip_merge(output, output_flipped, val_dataset.flip_pairs, config.TEST.SHIFT_HEATMAP)
            target = target.to(dev, non_blocking=True)
            target_weight = target_weight.to(dev, non_blocking=True)
            loss = criterion(output, target, target_weight)
            num_images = inp.size(0)
            n, j, h, w = output.shape
            # accuracy(output, target) on the device: NHWC copy of the merged map -> pck kernels -> log ring
            if metric is None or metric.args.B != n:
                metric = DeviceAccuracy(n, j, h, w, R.F32, dev, slots=16)
                nhwc = torch.empty((n, h, w, j), dtype=torch.float32, device=dev)
            R.check(R.lib().fpd_nchw_to_nhwc(output.data_ptr(), nhwc.data_ptr(), n, j, h, w, R.F32, R.current_stream()))
            tgt = target.float().contiguous()
            metric.bind(nhwc.data_ptr(), tgt.data_ptr()).enqueue()
            c, s = _to_numpy(meta['center']), _to_numpy(meta['scale'])
            score = _to_numpy(meta['score'])
            trans = np.stack([get_affine_transform(c[k], s[k], 0, [w, h], inv=1) for k in range(n)])
            _, preds, maxvals = final_preds_device(output, torch.from_numpy(trans).to(dev), config.TEST.POST_PROCESS)
            # the one synchronisation of the iteration: results + the two scalars
            preds, maxvals = preds.cpu().numpy(), maxvals.cpu().numpy()
            losses.update(loss.item(), num_images)
            (avg_acc, cnt), = metric.drain()
            acc.update(avg_acc, cnt)
            batch_time.update(time.time() - end)
            end = time.time()
            all_preds[idx:idx + num_images, :, 0:2] = preds[:, :, 0:2]
            all_preds[idx:idx + num_images, :, 2:3] = maxvals
            all_boxes[idx:idx + num_images, 0:2] = c[:, 0:2]
            all_boxes[idx:idx + num_images, 2:4] = s[:, 0:2]
            all_boxes[idx:idx + num_images, 4] = np.prod(s * 200, 1)
            all_boxes[idx:idx + num_images, 5] = score
            image_path.extend(meta['image'])
            idx += num_images
            if i % config.PRINT_FREQ == 0:
                logger.info('Test: [{0}/{1}]\t'
                            'Time {batch_time.val:.3f} ({batch_time.avg:.3f})\t'
                            'Loss {loss.val:.4f} ({loss.avg:.4f})\t'
                            'Accuracy {acc.val:.3f} ({acc.avg:.3f})'.format(i, len(val_loader), batch_time=batch_time,
                                                                             loss=losses, acc=acc))
        name_values, perf_indicator = val_dataset.evaluate(config, all_preds, output_dir, all_boxes, image_path,
                                                           filenames, imgnums)
        for nv in (name_values if isinstance(name_values, list) else [name_values]):
            _print_name_value(nv, config.MODEL.NAME)
        if writer_dict and writer_dict.get('writer') is not None:
            writer, gs = writer_dict['writer'], writer_dict['valid_global_steps']
            writer.add_scalar('valid_loss', losses.avg, gs)
            writer.add_scalar('valid_acc', acc.avg, gs)
            for nv in (name_values if isinstance(name_values, list) else [name_values]):
                writer.add_scalars('valid', dict(nv), gs)
            writer_dict['valid_global_steps'] = gs + 1
    validate.last = {'loss': losses.avg, 'acc': acc.avg, 'all_preds': all_preds, 'all_boxes': all_boxes}
    return perf_indicator


def _print_name_value(name_value, full_arch_name):
    """function.py:335-353: markdown table of the dataset's metrics."""
    names, values = list(name_value.keys()), list(name_value.values())
    logger.info('| Arch ' + ' '.join('| {}'.format(n) for n in names) + ' |')
    logger.info('|---' * (len(names) + 1) + '|')
    arch = full_arch_name if len(full_arch_name) <= 15 else full_arch_name[:8] + '...'
    logger.info('| ' + arch + ' ' + ' '.join('| {:.3f}'.format(v) for v in values) + ' |')
