#!/usr/bin/env python
"""FPD train-step benchmark (BASELINE.json metric): images/s of
    teacher forward (hg S=8 F=256, eval BN) + student forward/backward (hg S=4 F=128, train BN)
    + fused pose/KD JointsMSELoss + [RCCL gradient all-reduce] + Adam
on synthetic 256x256 crops, batch 32 per GPU.

  python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run, one rank per GPU)

Prints ONE JSON line on rank 0.  `roofline` prices one whole step (the unit the plan replays) against the dense
bf16 MFMA peak with the algorithmic conv FLOPs of SURVEY.md section 8(d) (79.478 GFLOP per image);
`cpu_baseline` times the CPU oracle (restatement of the reference loop, teacher under no_grad) on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

GFLOP_PER_IMAGE = {'hg4x128<-hg8x256': 79.478}       # SURVEY.md section 8(d), conv 2*MAC: t-fwd + s-fwd + s-bwd
PEAK_TFLOPS = {'bf16': 2500.0, 'fp32': 157.3}          # MI355X dense MFMA peaks (MI355X_MICROARCH.md)


class AD(dict):
    __getattr__ = dict.__getitem__


def make_cfg(feats, stacks, joints, dtype):
    return AD(MODEL=AD(NUM_JOINTS=joints, DTYPE=dtype, EXTRA=AD(NUM_FEATURES=feats, NUM_STACKS=stacks, NUM_BLOCKS=1)))


def cpu_baseline(batch, steps, seed=0):
    """CPU oracle (port of lib/core/function.py:114-147 over the restated hourglass) on the host cores."""
    from oracle import fpd_ref, hourglass_ref
    torch.manual_seed(seed)
    s_sd = fpd_ref.synth_state_dict(hourglass_ref.hourglass_keys(128, 4, 16), 1)
    t_sd = fpd_ref.synth_state_dict(hourglass_ref.hourglass_keys(256, 8, 16), 2)
    x, tg, tw = fpd_ref.synth_batch(100, batch, 16)
    adam = {}
    fpd_ref.fpd_step(s_sd, t_sd, 4, 8, x, tg, tw, 0.5, adam_state=adam)          # warm-up
    t0 = time.time()
    for _ in range(steps):
        fpd_ref.fpd_step(s_sd, t_sd, 4, 8, x, tg, tw, 0.5, adam_state=adam)
    dt = (time.time() - t0) / steps
    return batch / dt, dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=32, help='per-GPU batch')
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'fp32'])
    ap.add_argument('--backend', default='mfma', choices=['mfma', 'naive'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-graphs', action='store_true', help='launch kernel by kernel instead of replaying hipGraphs')
    ap.add_argument('--cpu-batch', type=int, default=4)
    ap.add_argument('--cpu-steps', type=int, default=2)
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit('launch with: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d ...' % (args.gpus, args.gpus))
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)

    from fpd_amd import executor as E, runtime as R, synth
    from fpd_amd.lib.models import hourglass
    from fpd_amd import dist as fdist
    R.lib()
    R.set_backend(R.BACKEND_NAIVE if args.backend == 'naive' else R.BACKEND_MFMA)

    B, J, H, W = args.batch, 16, 256, 256
    torch.manual_seed(1)                       # identical weights on every rank
    student = hourglass.get_pose_net(make_cfg(128, 4, J, args.dtype), is_train=True).to(dev)
    torch.manual_seed(2)
    teacher = hourglass.get_pose_net(make_cfg(256, 8, J, args.dtype), is_train=False).to(dev)

    # synthetic teacher: calibrate BN running statistics once so eval-mode activations are sane
    x, tg, tw = synth.make_batch(1000 + rank, B, J, (W, H), (W // 4, H // 4))
    cal = E.GraphInstance(teacher.device_state(), teacher.cfg_hg, B, H, W, train=True).finalize()
    cal.image().copy_(x)
    cal.run('prep'); cal.run('fwd')
    cal.calibrate_running_stats()
    del cal
    torch.cuda.empty_cache()

    step = E.FusedFPDStep(student.device_state(), student.cfg_hg, teacher.device_state(), teacher.cfg_hg, B, H, W,
                          alpha=0.5, lr=2.5e-4, world_size=world)
    step.set_batch(x, tg, tw)                  # data resident in HBM before timing
    allreduce = fdist.make_allreduce(dist) if world > 1 else None
    if world > 1:
        fdist.broadcast_state(dist, student)
        fdist.broadcast_state(dist, teacher)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    step.run_pipelined(1, allreduce)                 # one eager step (sets one-time kernel attributes) before capture
    if not args.no_graphs:
        step.enable_graphs()
    step.run_pipelined(args.warmup, allreduce)
    barrier()
    l = R.lib()
    ev0, ev1 = l.fpd_event_create(), l.fpd_event_create()
    st = R.current_stream()
    t0 = time.time()
    l.fpd_event_record(ev0, st)
    step.run_pipelined(args.steps, allreduce)      # K teacher forwards + K student steps, pipeline starts/ends empty
    l.fpd_event_record(ev1, st)
    barrier()
    wall = time.time() - t0
    ev_ms = l.fpd_event_elapsed_ms(ev0, ev1)
    pose, kd, loss = step.losses()
    if world > 1:
        t = torch.tensor([wall], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    ms_per_step = wall / args.steps * 1e3
    value = world * B * args.steps / wall
    flop_step = GFLOP_PER_IMAGE['hg4x128<-hg8x256'] * 1e9 * B
    achieved = flop_step / (ev_ms / args.steps * 1e-3) / 1e12
    peak = PEAK_TFLOPS[args.dtype]
    out = {
        'metric': 'images/sec FPD train step (4-stack HG student, 8-stack teacher) 256x256',
        'value': round(value, 2), 'unit': 'images/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(ms_per_step, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': args.dtype, 'data': 'synthetic',
        'config': {'workload': 'configs[1]: hourglass student S=4 F=128 + teacher S=8 F=256, 256x256, batch %d/GPU, '
                               'fused FPD step incl. Adam, teacher forward one batch ahead on a 2nd stream%s' % (B, ' + RCCL all-reduce' if world > 1 else ''),
                   'global_batch': world * B, 'parallelism': 'dp%d' % world, 'backend': args.backend,
                   'launch': 'eager' if args.no_graphs else 'hipGraph replay per phase',
                   'loss_last_step': round(loss, 6), 'finite': bool(loss == loss and abs(loss) < 1e6)},
        'roofline': {'bound': 'mfma', 'achieved': round(achieved, 2), 'peak': peak, 'unit': 'TFLOP/s',
                     'frac': round(achieved / peak, 4), 'traffic': None,
                     'note': 'one launch = one replay of the FPD step plan on one GPU: %.3f TFLOP algorithmic conv work '
                             '(79.478 GFLOP/image x %d), timed with HIP events on the launch stream: %.3f ms/step' % (
                                 flop_step / 1e12, B, ev_ms / args.steps)},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        v, dt = cpu_baseline(args.cpu_batch, args.cpu_steps)
        out['cpu_baseline'] = {'value': round(v, 3), 'unit': 'images/s', 'cores': torch.get_num_threads(), 'kind': 'port',
                               'sample': 'same FPD pair (hg4x128 <- hg8x256, 256x256) at batch %d, %d timed steps after 1 '
                                         'warm-up, torch CPU fp32 oracle, teacher under no_grad (%.2f s/step)' % (
                                             args.cpu_batch, args.cpu_steps, dt)}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
