"""Training-time metric of the reference (arg-max PCK@0.5, /root/reference/lib/core/evaluate.py:16-71 and
lib/core/inference.py:18-46).

`DeviceAccuracy` is the product path: the HIP kernels of csrc/pck.hip evaluate the metric on the NHWC prediction the fused
step keeps in its arena and append {avg_acc, cnt, pose_loss, kd_loss} to a device ring every iteration; the host drains
the ring when a log line is due (the reference does a D2H copy + numpy every iteration, function.py:154-155).
`accuracy()` keeps the reference's function signature for module-API users (tensor in, same 4-tuple out); it evaluates
the same definition with torch ops on whatever device the maps live on.  Both follow the reference bit for bit, quirk
included: the normaliser is `[h, w] / 10` applied to `(x, y)` (evaluate.py:55) and distances/averages are float64.
Pinned to the reference's own functions by tests/golden/pck_ref.npz."""
import torch


def get_max_preds(hm):
    """inference.py:18-46.  hm [B,J,h,w] -> (coords [B,J,2] (x,y) float32, maxvals [B,J,1]); coords are zeroed where
    maxval <= 0; the first maximum wins (numpy.argmax)."""
    b, j, h, w = hm.shape
    flat = hm.reshape(b, j, -1)
    maxvals = flat.max(dim=2, keepdim=True)[0]
    # first index attaining the maximum (torch.max's index is not guaranteed to be the first one on every backend)
    pos = torch.arange(flat.shape[2], device=hm.device).expand_as(flat)
    idx = torch.where(flat == maxvals, pos, torch.full_like(pos, flat.shape[2])).min(dim=2, keepdim=True)[0]
    x = (idx % w).float()
    y = torch.floor(idx.float() / w)
    coords = torch.cat([x, y], dim=2)
    coords = coords * (maxvals > 0).float()
    return coords, maxvals


def accuracy(output, target, hm_type='gaussian', thr=0.5):
    """evaluate.py:42-71.  Returns (acc [J+1] float64 tensor with the average first, avg_acc float, cnt int, pred)."""
    assert hm_type == 'gaussian'
    pred, _ = get_max_preds(output)
    gt, _ = get_max_preds(target)
    h, w = output.shape[2], output.shape[3]
    norm = torch.tensor([h, w], dtype=torch.float64, device=output.device) / 10        # [h, w]: the reference's order
    valid = (gt[..., 0] > 1) & (gt[..., 1] > 1)                                        # evaluate.py:22
    diff = pred.double() / norm - gt.double() / norm
    d = torch.sqrt((diff * diff).sum(dim=2))                                           # [B,J] float64
    n_valid = valid.sum(0).double()                                                    # [J]
    hit = ((d < thr) & valid).sum(0).double()
    acc_j = torch.where(n_valid > 0, hit * 1.0 / n_valid.clamp_min(1), torch.full_like(hit, -1.0))
    has = acc_j >= 0
    cnt = int(has.sum().item())
    avg = 0.0
    for v in acc_j[has].tolist():                                                      # evaluate.py:62-66: running sum
        avg = avg + v
    avg = avg / cnt if cnt != 0 else 0
    acc = torch.cat([torch.tensor([avg if cnt else 0.0], dtype=torch.float64, device=output.device), acc_j])
    return acc, avg, cnt, pred


class DeviceAccuracy:
    """fpd_pck (include/fpd_amd.h) bound to fixed buffers: `enqueue()` is asynchronous (one entry {avg_acc, cnt, pose, kd}
    per call appended to a device ring), `drain()` returns the entries appended since the last drain -- the only point
    that synchronises."""

    def __init__(self, batch, joints, height, width, dtype, device, slots=4096, thr=0.5):
        from ... import runtime as R
        self.R = R
        self.slots = slots
        self.counts = torch.zeros(joints * 2, dtype=torch.float32, device=device)
        self.log = torch.zeros(slots * 4, dtype=torch.float64, device=device)
        self.cursor = torch.zeros(1, dtype=torch.int64, device=device)
        self.read = 0
        a = R.PckT()
        a.B, a.J, a.H, a.W, a.dtype, a.log_slots, a.thr = batch, joints, height, width, dtype, slots, thr
        a.counts, a.log, a.cursor = self.counts.data_ptr(), self.log.data_ptr(), self.cursor.data_ptr()
        a.losses = None
        self.args = a

    def bind(self, out_ptr, target_ptr, losses_ptr=None):
        self.args.out, self.args.target, self.args.losses = out_ptr, target_ptr, losses_ptr
        return self

    def enqueue(self, stream=None):
        R = self.R
        R.check(R.lib().fpd_pck(self.args, stream if stream is not None else R.current_stream()), 'fpd_pck')

    def pending(self):
        return int(self.cursor.item()) - self.read

    def drain(self, full=False):
        """[(avg_acc, cnt)] per iteration since the last drain; full=True: [(avg_acc, cnt, pose, kd)]."""
        n = int(self.cursor.item())                       # synchronises with the stream the kernels ran on
        assert n - self.read <= self.slots, 'metric ring overflow: drain() more often'
        log = self.log.view(self.slots, 4).cpu()
        out = []
        for k in range(self.read, n):
            e = log[k % self.slots]
            out.append((float(e[0]), int(e[1]), float(e[2]), float(e[3])) if full else (float(e[0]), int(e[1])))
        self.read = n
        return out
