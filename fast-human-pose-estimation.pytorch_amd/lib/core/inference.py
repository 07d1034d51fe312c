"""`core.inference` with the reference's function names (/root/reference/lib/core/inference.py), evaluated on the device.

get_final_preds(config, batch_heatmaps, center, scale) -> (preds [N,J,2], maxvals [N,J,1]) numpy arrays like the
reference (its caller stores them into numpy result arrays, function.py:264-270); `batch_heatmaps` is a CUDA tensor
[N,J,h,w] (a numpy array is uploaded).  One kernel does the arg-max, the TEST.POST_PROCESS quarter-pixel shift and the
affine map back to image coordinates (csrc/infer.hip); the per-sample 2x3 matrices are host numpy
(utils.transforms.get_affine_transform, a dozen flops each)."""
import numpy as np
import torch

from ... import runtime as R
from ..utils.transforms import get_affine_transform
from .evaluate import get_max_preds  # noqa: F401  (same name as the reference's core.inference.get_max_preds)


def final_preds_device(batch_heatmaps, trans, post_process):
    """(coords, preds, maxvals) CUDA tensors.  trans: [N,2,3] float64 CUDA tensor or None (heat-map coordinates only)."""
    hm = batch_heatmaps
    if not hm.is_cuda:
        raise R.FpdError('get_final_preds works on CUDA (ROCm) tensors; there is no CPU path')
    hm = hm.detach().float().contiguous()
    n, j, h, w = hm.shape
    dev = hm.device
    coords = torch.empty((n, j, 2), dtype=torch.float32, device=dev)
    maxvals = torch.empty((n, j, 1), dtype=torch.float32, device=dev)
    preds = torch.empty((n, j, 2), dtype=torch.float32, device=dev) if trans is not None else None
    a = R.FinalPredsT()
    a.N, a.J, a.H, a.W, a.post_process = n, j, h, w, int(bool(post_process))
    a.hm, a.coords, a.maxvals = hm.data_ptr(), coords.data_ptr(), maxvals.data_ptr()
    if trans is not None:
        assert trans.dtype == torch.float64 and tuple(trans.shape) == (n, 2, 3) and trans.is_cuda
        trans = trans.contiguous()
        a.trans, a.preds = trans.data_ptr(), preds.data_ptr()
    R.check(R.lib().fpd_final_preds(a, R.current_stream()), 'fpd_final_preds')
    return coords, preds, maxvals


def get_final_preds(config, batch_heatmaps, center, scale):
    """inference.py:49-79."""
    hm = batch_heatmaps if torch.is_tensor(batch_heatmaps) else torch.from_numpy(np.ascontiguousarray(batch_heatmaps)).cuda()
    n, _, h, w = hm.shape
    center, scale = np.asarray(center), np.asarray(scale)
    trans = np.stack([get_affine_transform(center[i], scale[i], 0, [w, h], inv=1) for i in range(n)])
    _, preds, maxvals = final_preds_device(hm, torch.from_numpy(trans).to(hm.device), config.TEST.POST_PROCESS)
    return preds.cpu().numpy(), maxvals.cpu().numpy()
