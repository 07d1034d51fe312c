"""Data-parallel replicas: one process per GPU, student gradients summed with ONE RCCL all-reduce of the flat
gradient arena per step (torch.distributed backend 'nccl' == RCCL over xGMI on ROCm).

Replaces the reference's single-process nn.DataParallel (/root/reference/tools/fpd_train.py:143,173): no
per-step parameter re-broadcast, no output gather; each rank runs the whole FPD step on its shard with
per-replica BatchNorm statistics (the same semantics DataParallel has), the loss kernel scales gradients by
1/world so the summed gradient is the gradient of the global-batch mean (equal shards).  The collective is
issued asynchronously right after the backward and waited for only before Adam, i.e. it overlaps the NEXT
step's teacher forward, which does not depend on the student weights (executor.FusedFPDStep.step)."""
import os

import torch


def make_allreduce(dist, group=None):
    """Returns hook(flat_grad, buckets=None) -> wait() for executor.FusedFPDStep.student_step(allreduce=...).

    Default: ONE all-reduce of the whole arena, issued behind the backward on the student's stream and waited for right
    before Adam -- it overlaps the next step's teacher forward, and no stream beyond RCCL's own is created.
    FPD_ALLREDUCE_BUCKETS=1 (opt-in): one collective per gradient bucket from a side stream, so the early buckets'
    collectives also overlap the rest of the backward.  It is NOT the default because of a measured pathology of this
    ROCm stack (profiles/README.md, round 3): the side stream is the sixth active hardware queue of the process, and with
    GPU_MAX_HW_QUEUES=8 (what the three hot streams need so that they never share a queue) the step then takes 26.7 ms
    instead of 10.8 (with 6 queues: 10.8 either way, with 4: 15.6 either way).

    `buckets` = [(begin, end, wait)] (executor.FusedFPDStep.grad_buckets): the flat gradient arena is laid out in the
    order gradients complete during the backward (graph.ParamTable buckets: last stack first), so each bucket is one
    contiguous slice and gets its OWN asynchronous all-reduce, issued from a side stream that waits -- on the device --
    for the plan op after which that slice is final.  The collectives of the early buckets therefore run while the rest
    of the backward is still executing; only the last bucket's is exposed, and even that overlaps the next step's
    teacher forward (the student stream waits for the collectives right before Adam)."""
    comm = {}
    spans = []                          # (event before, event after) around the student stream's waits: exposed collective time

    def timed(wait):
        def w():
            if not torch.cuda.is_available():
                return wait()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            wait()
            e1.record()
            spans.append((e0, e1))
            if len(spans) > 256:
                del spans[:128]
        return w

    def exposed_us():
        """Mean time per step the consuming stream sat in front of Adam waiting for the collectives (synchronises)."""
        if not spans:
            return 0.0
        torch.cuda.synchronize()
        return 1e3 * sum(a.elapsed_time(b) for a, b in spans) / len(spans)

    def hook(flat_grad, buckets=None):
        if not buckets or not flat_grad.is_cuda or os.environ.get('FPD_ALLREDUCE_BUCKETS', '0') != '1':
            work = dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=group, async_op=True)
            return timed(work.wait) if flat_grad.is_cuda else work.wait     # current stream waits for the collective; no host sync on RCCL
        if 's' not in comm:
            comm['s'] = torch.cuda.Stream(device=flat_grad.device)
        works = []
        cur = torch.cuda.current_stream()
        for lo, hi, wait in buckets:
            if wait is None:
                works.append(dist.all_reduce(flat_grad[lo:hi], op=dist.ReduceOp.SUM, group=group, async_op=True))
                continue
            with torch.cuda.stream(comm['s']):
                wait(comm['s'])         # RCCL's stream orders itself behind the stream the collective is issued from
                works.append(dist.all_reduce(flat_grad[lo:hi], op=dist.ReduceOp.SUM, group=group, async_op=True))

        def wait_all():
            for w in works:
                w.wait()                # current stream waits for each collective (device-side)
            return None
        del cur
        return timed(wait_all)
    hook.exposed_us = exposed_us
    return hook


def broadcast_state(dist, model, src=0, group=None):
    """Make every replica start from rank `src`'s parameters and BN buffers (flat arenas: 3 collectives)."""
    for name in ('param', 'rstat', 'nbt'):
        dist.broadcast(model._flat[name], src=src, group=group)


def shard(batch_tensors, rank, world):
    """Rank r takes samples [r*B/world, (r+1)*B/world) of each tensor -- DataParallel's dim-0 scatter
    (tools/fpd_train.py:202: loader batch = BATCH_SIZE_PER_GPU * len(GPUS))."""
    out = []
    for t in batch_tensors:
        n = t.shape[0]
        assert n % world == 0, 'global batch %d not divisible by world size %d' % (n, world)
        per = n // world
        out.append(t[rank * per:(rank + 1) * per])
    return out


class DataParallelReplica(torch.nn.Module):
    """`.module` / `.parameters()` / `.state_dict()` surface the reference scripts use on their DataParallel
    wrapper (tools/fpd_train.py:279-294; state_dict keys carry the 'module.' prefix like the reference's)."""

    def __init__(self, module):
        super().__init__()
        self.module = module

    def forward(self, *a, **kw):
        return self.module(*a, **kw)
