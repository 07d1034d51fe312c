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
