"""Dev probe: cost of one torch.distributed (RCCL) all-reduce of the flat gradient arena with world size 1."""
import os, time, torch, torch.distributed as dist
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29544')
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
g = torch.zeros(3287936, device='cuda')
for n in (3287936, 65536):
    t = g[:n]
    for _ in range(3): dist.all_reduce(t)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.time(); e0.record()
    for _ in range(20): dist.all_reduce(t)
    e1.record(); th = time.time() - t0; torch.cuda.synchronize()
    print('sync  all_reduce n=%d: %.1f us GPU, %.1f us host per call' % (n, e0.elapsed_time(e1) * 50, th * 5e4))
    e0.record()
    for _ in range(20):
        w = dist.all_reduce(t, async_op=True); w.wait()
    e1.record(); torch.cuda.synchronize()
    print('async all_reduce n=%d: %.1f us GPU per call' % (n, e0.elapsed_time(e1) * 50))
dist.destroy_process_group()
