"""TEST INFRASTRUCTURE (oracle) -- numpy restatement of the reference's validate / flip-test post-processing and of the
per-sample data pipeline.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Follows, line by line:
  flip_back                      /root/reference/lib/utils/transforms.py:15-29
  flip-test merge                /root/reference/lib/core/function.py:216-238
  get_final_preds                /root/reference/lib/core/inference.py:49-79   (get_max_preds: oracle/pck_ref.py)
  get_affine_transform & co      /root/reference/lib/utils/transforms.py:50-108
  generate_target                /root/reference/lib/dataset/JointsDataset.py:233-289
  warp_affine_u8                 cv2.warpAffine(INTER_LINEAR) as called at JointsDataset.py:160-165

Pinned: everything except warp_affine_u8 is checked against the reference's OWN functions through
tests/golden/infer_ref.npz (tests/golden/make_golden_infer.py imports them; cv2.getAffineTransform -- the only cv2 call
on those paths -- is stubbed with the 6x6 float64 system OpenCV solves).
PARITY UNPINNED for warp_affine_u8: OpenCV is not installed in the build container, so the fixed-point bilinear scheme is
restated from OpenCV's published algorithm (imgwarp.cpp: AB_BITS 10, INTER_BITS 5, INTER_REMAP_COEF_BITS 15) and checked
only against a float64 bilinear interpolation (<= 1 grey level) and exact-copy / integer-shift cases.
"""
import numpy as np

from .pck_ref import get_max_preds


def flip_back(output_flipped, matched_parts):
    """transforms.py:15-29 (operates on a reversed VIEW, swaps write through it)."""
    assert output_flipped.ndim == 4
    output_flipped = output_flipped[:, :, :, ::-1]
    for pair in matched_parts:
        tmp = output_flipped[:, pair[0], :, :].copy()
        output_flipped[:, pair[0], :, :] = output_flipped[:, pair[1], :, :]
        output_flipped[:, pair[1], :, :] = tmp
    return output_flipped


def flip_merge(output, output_flipped, matched_parts, shift):
    """function.py:229-238 on float32 arrays: flip back, optional one-pixel shift, average."""
    f = flip_back(output_flipped.copy(), matched_parts).copy()
    if shift:
        f[:, :, :, 1:] = f.copy()[:, :, :, 0:-1]
    return ((output + f) * np.float32(0.5)).astype(np.float32)


def get_3rd_point(a, b):
    direct = a - b
    return b + np.array([-direct[1], direct[0]], dtype=np.float32)


def get_dir(src_point, rot_rad):
    sn, cs = np.sin(rot_rad), np.cos(rot_rad)
    return [src_point[0] * cs - src_point[1] * sn, src_point[0] * sn + src_point[1] * cs]


def solve_affine(src, dst):
    """cv2.getAffineTransform(src, dst): the 2x3 map taking three points to three points, float64."""
    src, dst = np.asarray(src, np.float64), np.asarray(dst, np.float64)
    a = np.concatenate([src, np.ones((3, 1))], axis=1)
    return np.linalg.solve(a, dst).T


def get_affine_transform(center, scale, rot, output_size, shift=np.array([0, 0], dtype=np.float32), inv=0):
    """transforms.py:57-92 (points are built in float32 like the reference)."""
    if not isinstance(scale, np.ndarray) and not isinstance(scale, list):
        scale = np.array([scale, scale])
    scale_tmp = scale * 200.0
    src_w = scale_tmp[0]
    dst_w, dst_h = output_size[0], output_size[1]
    rot_rad = np.pi * rot / 180
    src_dir = get_dir([0, src_w * -0.5], rot_rad)
    dst_dir = np.array([0, dst_w * -0.5], np.float32)
    src = np.zeros((3, 2), dtype=np.float32)
    dst = np.zeros((3, 2), dtype=np.float32)
    src[0, :] = center + scale_tmp * shift
    src[1, :] = center + src_dir + scale_tmp * shift
    dst[0, :] = [dst_w * 0.5, dst_h * 0.5]
    dst[1, :] = np.array([dst_w * 0.5, dst_h * 0.5]) + dst_dir
    src[2:, :] = get_3rd_point(src[0, :], src[1, :])
    dst[2:, :] = get_3rd_point(dst[0, :], dst[1, :])
    return solve_affine(dst, src) if inv else solve_affine(src, dst)


def affine_transform(pt, t):
    """transforms.py:99-102."""
    return np.dot(t, np.array([pt[0], pt[1], 1.]).T)[:2]


def transform_preds(coords, center, scale, output_size):
    """transforms.py:50-55."""
    target_coords = np.zeros(coords.shape)
    trans = get_affine_transform(center, scale, 0, output_size, inv=1)
    for p in range(coords.shape[0]):
        target_coords[p, 0:2] = affine_transform(coords[p, 0:2], trans)
    return target_coords


def get_final_preds(post_process, batch_heatmaps, center, scale):
    """inference.py:49-79."""
    coords, maxvals = get_max_preds(batch_heatmaps)
    h, w = batch_heatmaps.shape[2], batch_heatmaps.shape[3]
    if post_process:
        for n in range(coords.shape[0]):
            for p in range(coords.shape[1]):
                hm = batch_heatmaps[n][p]
                px = int(np.floor(coords[n][p][0] + 0.5))
                py = int(np.floor(coords[n][p][1] + 0.5))
                if 1 < px < w - 1 and 1 < py < h - 1:
                    diff = np.array([hm[py][px + 1] - hm[py][px - 1], hm[py + 1][px] - hm[py - 1][px]])
                    coords[n][p] += np.sign(diff) * .25
    preds = coords.copy()
    for i in range(coords.shape[0]):
        preds[i] = transform_preds(coords[i], center[i], scale[i], [w, h])
    return preds, maxvals, coords


def gaussian_patch(sigma):
    """JointsDataset.py:266-271: the (6 sigma + 1)^2 patch in the reference's float32 numpy expression."""
    tmp_size = sigma * 3
    size = 2 * tmp_size + 1
    x = np.arange(0, size, 1, np.float32)
    y = x[:, np.newaxis]
    x0 = y0 = size // 2
    return np.exp(- ((x - x0) ** 2 + (y - y0) ** 2) / (2 * sigma ** 2))


def generate_target(joints, joints_vis, image_size, heatmap_size, sigma):
    """JointsDataset.py:233-289 for one sample.  joints [J,3], joints_vis [J,3]; sizes are (w, h) arrays."""
    num_joints = joints.shape[0]
    image_size, heatmap_size = np.asarray(image_size), np.asarray(heatmap_size)
    target_weight = np.ones((num_joints, 1), dtype=np.float32)
    target_weight[:, 0] = joints_vis[:, 0]
    target = np.zeros((num_joints, heatmap_size[1], heatmap_size[0]), dtype=np.float32)
    tmp_size = sigma * 3
    g = gaussian_patch(sigma)
    for joint_id in range(num_joints):
        feat_stride = image_size / heatmap_size
        mu_x = int(joints[joint_id][0] / feat_stride[0] + 0.5)
        mu_y = int(joints[joint_id][1] / feat_stride[1] + 0.5)
        ul = [int(mu_x - tmp_size), int(mu_y - tmp_size)]
        br = [int(mu_x + tmp_size + 1), int(mu_y + tmp_size + 1)]
        if ul[0] >= heatmap_size[0] or ul[1] >= heatmap_size[1] or br[0] < 0 or br[1] < 0:
            target_weight[joint_id] = 0
            continue
        g_x = max(0, -ul[0]), min(br[0], heatmap_size[0]) - ul[0]
        g_y = max(0, -ul[1]), min(br[1], heatmap_size[1]) - ul[1]
        img_x = max(0, ul[0]), min(br[0], heatmap_size[0])
        img_y = max(0, ul[1]), min(br[1], heatmap_size[1])
        if target_weight[joint_id] > 0.5:
            target[joint_id][img_y[0]:img_y[1], img_x[0]:img_x[1]] = g[g_y[0]:g_y[1], g_x[0]:g_x[1]]
    return target, target_weight


def invert_affine(m):
    """cv::invertAffineTransform (what cv2.warpAffine applies to its matrix unless WARP_INVERSE_MAP), float64."""
    m = np.asarray(m, np.float64)
    d = m[0, 0] * m[1, 1] - m[0, 1] * m[1, 0]
    d = 1.0 / d if d != 0 else 0.0
    a11, a22 = m[1, 1] * d, m[0, 0] * d
    a12, a21 = -m[0, 1] * d, -m[1, 0] * d
    b1 = -a11 * m[0, 2] - a12 * m[1, 2]
    b2 = -a21 * m[0, 2] - a22 * m[1, 2]
    return np.array([[a11, a12, b1], [a21, a22, b2]])


def warp_affine_u8(img, minv, out_w, out_h):
    """cv2.warpAffine(img, M, (out_w, out_h), flags=INTER_LINEAR), border constant 0, for uint8 HxWx3 images, given
    minv = invert_affine(M).  OpenCV's fixed-point scheme (PARITY UNPINNED, see the module docstring)."""
    h, w = img.shape[:2]
    xs, ys = np.arange(out_w, dtype=np.float64), np.arange(out_h, dtype=np.float64)
    adelta = np.rint(minv[0, 0] * xs * 1024).astype(np.int64)
    bdelta = np.rint(minv[1, 0] * xs * 1024).astype(np.int64)
    x0 = np.rint((minv[0, 1] * ys + minv[0, 2]) * 1024).astype(np.int64) + 16
    y0 = np.rint((minv[1, 1] * ys + minv[1, 2]) * 1024).astype(np.int64) + 16
    X = (x0[:, None] + adelta[None, :]) >> 5
    Y = (y0[:, None] + bdelta[None, :]) >> 5
    sx, sy = np.clip(X >> 5, -32768, 32767), np.clip(Y >> 5, -32768, 32767)
    fx, fy = X & 31, Y & 31
    w00 = np.minimum((32 - fx) * (32 - fy) * 32, 32767)
    w01, w10, w11 = fx * (32 - fy) * 32, (32 - fx) * fy * 32, fx * fy * 32
    src = img.astype(np.int64)

    def tap(yy, xx):
        ok = (yy >= 0) & (yy < h) & (xx >= 0) & (xx < w)
        v = src[np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)]
        return v * ok[..., None]
    acc = tap(sy, sx) * w00[..., None] + tap(sy, sx + 1) * w01[..., None] + tap(sy + 1, sx) * w10[..., None] + \
        tap(sy + 1, sx + 1) * w11[..., None]
    return np.clip((acc + (1 << 14)) >> 15, 0, 255).astype(np.uint8)


def to_tensor_normalize(img_u8, mean, std):
    """torchvision ToTensor + Normalize (tools/fpd_train.py:182-193) in float32: HWC uint8 -> CHW (v/255 - mean)/std."""
    t = img_u8.astype(np.float32).transpose(2, 0, 1) / np.float32(255)
    return (t - np.asarray(mean, np.float32)[:, None, None]) / np.asarray(std, np.float32)[:, None, None]
