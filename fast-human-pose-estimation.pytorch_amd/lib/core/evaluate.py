"""Training-time metric of the reference (arg-max PCK@0.5, /root/reference/lib/core/evaluate.py:16-71 and
lib/core/inference.py:18-46), written with torch ops so it runs on the device the heat-maps live on and is only
evaluated when a log line is printed (the reference does a D2H copy + numpy every iteration, function.py:154-155)."""
import torch


def get_max_preds(hm):
    """hm [B,J,h,w] -> (coords [B,J,2] (x,y) float, maxvals [B,J,1]); coords are zeroed where maxval <= 0."""
    b, j, h, w = hm.shape
    flat = hm.reshape(b, j, -1)
    maxvals, idx = flat.max(dim=2, keepdim=True)
    x = (idx % w).float()
    y = torch.floor(idx.float() / w)
    coords = torch.cat([x, y], dim=2)
    coords = coords * (maxvals > 0).float()
    return coords, maxvals


def accuracy(output, target, thr=0.5):
    """Returns (per-joint acc [J+1] tensor with the average first, avg_acc float, cnt int, pred coords)."""
    pred, _ = get_max_preds(output)
    gt, _ = get_max_preds(target)
    h, w = output.shape[2], output.shape[3]
    norm = torch.tensor([w, h], dtype=torch.float32, device=output.device) / 10.0
    valid = (gt[..., 0] > 1) & (gt[..., 1] > 1)
    d = torch.linalg.norm((pred - gt) / norm, dim=2)
    d = torch.where(valid, d, torch.full_like(d, -1.0))           # [B,J]
    n_valid = (d >= 0).sum(0).float()                             # [J]
    hit = ((d < thr) & (d >= 0)).sum(0).float()
    acc_j = torch.where(n_valid > 0, hit / n_valid.clamp_min(1), torch.full_like(hit, -1.0))
    has = acc_j >= 0
    cnt = int(has.sum().item())
    avg = float((acc_j * has.float()).sum().item() / cnt) if cnt else 0.0
    return torch.cat([torch.tensor([avg], device=output.device), acc_j]), avg, cnt, pred


class DeviceAccuracy:
    """The same metric as `accuracy()` computed by the HIP kernels of csrc/pck.hip on the NHWC prediction the fused step
    keeps in its arena: `enqueue()` is asynchronous (one entry {avg_acc, cnt} per call appended to a device ring),
    `drain()` returns the entries appended since the last drain -- the only point that synchronises."""

    def __init__(self, batch, joints, height, width, dtype, device, slots=4096, thr=0.5):
        from ... import runtime as R
        self.R = R
        self.slots = slots
        self.counts = torch.zeros(joints * 2, dtype=torch.float32, device=device)
        self.log = torch.zeros(slots * 2, dtype=torch.float32, device=device)
        self.cursor = torch.zeros(1, dtype=torch.int64, device=device)
        self.read = 0
        a = R.PckT()
        a.B, a.J, a.H, a.W, a.dtype, a.log_slots, a.thr = batch, joints, height, width, dtype, slots, thr
        a.counts, a.log, a.cursor = self.counts.data_ptr(), self.log.data_ptr(), self.cursor.data_ptr()
        self.args = a

    def bind(self, out_ptr, target_ptr):
        self.args.out, self.args.target = out_ptr, target_ptr
        return self

    def enqueue(self, stream=None):
        R = self.R
        R.check(R.lib().fpd_pck(self.args, stream if stream is not None else R.current_stream()), 'fpd_pck')

    def drain(self):
        n = int(self.cursor.item())                       # synchronises with the stream the kernels ran on
        assert n - self.read <= self.slots, 'metric ring overflow: drain() more often'
        log = self.log.view(self.slots, 2).cpu()
        out = [(float(log[k % self.slots, 0]), int(log[k % self.slots, 1])) for k in range(self.read, n)]
        self.read = n
        return out
