"""`utils.transforms` with the reference's function names (/root/reference/lib/utils/transforms.py).

Heat-map work runs on the device (csrc/infer.hip): `flip_back` takes and returns CUDA tensors.  The per-sample 2x3 affine
matrices (`get_affine_transform`: three point pairs -> matrix, a dozen flops per sample) are host numpy like the
reference's, with cv2.getAffineTransform replaced by the float64 3-point solve it performs; applying them to heat-map
coordinates (transform_preds) or to images (crop) is device work (core/inference.py, dataset/device_pipeline.py)."""
import numpy as np
import torch

from ... import runtime as R


def channel_sources(num_joints, matched_parts):
    """src[j] = the channel that ends up in channel j after flip_back's sequential pair swaps (transforms.py:23-27)."""
    src = list(range(num_joints))
    for a, b in matched_parts:
        src[a], src[b] = src[b], src[a]
    return src


def flip_merge(output, output_flipped, matched_parts, shift):
    """(output + shift(flip_back(output_flipped))) * 0.5 in one kernel (function.py:229-238); output None = no average."""
    f = output_flipped
    if not f.is_cuda:
        raise R.FpdError('flip_back/flip_merge work on CUDA (ROCm) tensors; there is no CPU path')
    assert f.dim() == 4, 'output_flipped should be [batch_size, num_joints, height, width]'
    f = f.detach().float().contiguous()
    n, j, h, w = f.shape
    a = R.FlipMergeT()
    a.N, a.J, a.H, a.W, a.shift = n, j, h, w, int(bool(shift))
    y = torch.empty_like(f)
    if output is not None:
        o = output.detach().float().contiguous()
        assert o.shape == f.shape
        a.a = o.data_ptr()
    a.b, a.y = f.data_ptr(), y.data_ptr()
    for k, s in enumerate(channel_sources(j, matched_parts)):
        a.src[k] = s
    R.check(R.lib().fpd_flip_merge(a, R.current_stream()), 'fpd_flip_merge')
    return y


def flip_back(output_flipped, matched_parts):
    """transforms.py:15-29 on a CUDA tensor [N,J,h,w]: width reversed, left/right joint channels swapped."""
    return flip_merge(None, output_flipped, matched_parts, False)


def flip_input(x):
    """input[:, :, :, ::-1] of an [N,C,H,W] fp32 CUDA batch (function.py:217-221)."""
    if not x.is_cuda:
        raise R.FpdError('flip_input works on CUDA (ROCm) tensors; there is no CPU path')
    x = x.float().contiguous()
    y = torch.empty_like(x)
    R.check(R.lib().fpd_flip_w(x.data_ptr(), y.data_ptr(), x.numel() // x.shape[-1], x.shape[-1], R.current_stream()), 'fpd_flip_w')
    return y


def fliplr_joints(joints, joints_vis, width, matched_parts):
    """transforms.py:32-47: mirror the x coordinates and exchange left/right joints (in place, like the reference)."""
    joints[:, 0] = width - joints[:, 0] - 1
    for a, b in matched_parts:
        joints[[a, b]] = joints[[b, a]]
        joints_vis[[a, b]] = joints_vis[[b, a]]
    return joints * joints_vis, joints_vis


def _third_point(a, b):
    d = a - b
    return b + np.array([-d[1], d[0]], dtype=np.float32)


def _three_point_affine(src, dst):
    """The 2x3 matrix taking src[i] to dst[i], i < 3 (what cv2.getAffineTransform computes), float64."""
    lhs = np.concatenate([np.asarray(src, np.float64), np.ones((3, 1))], axis=1)
    return np.ascontiguousarray(np.linalg.solve(lhs, np.asarray(dst, np.float64)).T)


def get_affine_transform(center, scale, rot, output_size, shift=np.array([0, 0], dtype=np.float32), inv=0):
    """transforms.py:57-92: the crop's similarity transform from (centre, scale*200 px, rotation) to the output window,
    defined by three float32 point pairs; inv=1 returns the output->image map."""
    if not isinstance(scale, (np.ndarray, list)):
        scale = np.array([scale, scale])
    box = np.asarray(scale) * 200.0
    src_w, dst_w, dst_h = box[0], output_size[0], output_size[1]
    ang = np.pi * rot / 180
    sn, cs = np.sin(ang), np.cos(ang)
    up = -0.5 * src_w
    src_dir = [0 * cs - up * sn, 0 * sn + up * cs]
    src = np.zeros((3, 2), np.float32)
    dst = np.zeros((3, 2), np.float32)
    src[0] = center + box * shift
    src[1] = center + src_dir + box * shift
    dst[0] = [dst_w * 0.5, dst_h * 0.5]
    dst[1] = np.array([dst_w * 0.5, dst_h * 0.5]) + np.array([0, dst_w * -0.5], np.float32)
    src[2] = _third_point(src[0], src[1])
    dst[2] = _third_point(dst[0], dst[1])
    return _three_point_affine(dst, src) if inv else _three_point_affine(src, dst)


def affine_transform(pt, t):
    """transforms.py:99-102."""
    return np.dot(t, np.array([pt[0], pt[1], 1.]))[:2]


def invert_affine(m):
    """dst->src map of a 2x3 matrix (cv2.warpAffine inverts its argument the same way before sampling)."""
    m = np.asarray(m, np.float64)
    det = m[0, 0] * m[1, 1] - m[0, 1] * m[1, 0]
    d = 1.0 / det if det != 0 else 0.0
    a11, a22, a12, a21 = m[1, 1] * d, m[0, 0] * d, -m[0, 1] * d, -m[1, 0] * d
    return np.array([[a11, a12, -a11 * m[0, 2] - a12 * m[1, 2]], [a21, a22, -a21 * m[0, 2] - a22 * m[1, 2]]])
