"""The per-sample work of `JointsDataset.__getitem__` (/root/reference/lib/dataset/JointsDataset.py:160-176,233-289) as
batch kernels on the device (csrc/data.hip): a loader hands over decoded 8-bit images, the augmentation parameters it
drew (centre, scale, rotation -> utils.transforms.get_affine_transform) and the joint annotations; the crop
(cv2.warpAffine + ToTensor + Normalize), the joint transform and the Gaussian targets are produced where the training
step consumes them.  At >10 k images/s per GPU the reference's CPU workers (cv2 + numpy per sample) cannot feed the
step; this can."""
import ctypes as C

import numpy as np
import torch

from ... import runtime as R
from ..utils.transforms import invert_affine


def gaussian_patch(sigma):
    """The (6 sigma + 1)^2 un-normalised Gaussian of generate_target (JointsDataset.py:266-271), in the float32 numpy
    arithmetic the reference uses -- the device only places it."""
    size = 2 * (sigma * 3) + 1
    x = np.arange(0, size, 1, np.float32)
    y = x[:, np.newaxis]
    x0 = y0 = size // 2
    return np.exp(-((x - x0) ** 2 + (y - y0) ** 2) / (2 * sigma ** 2))


class DevicePipeline:
    def __init__(self, image_size, heatmap_size, sigma, device, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
        """image_size / heatmap_size are (w, h) like cfg.MODEL.IMAGE_SIZE / HEATMAP_SIZE; mean/std are the Normalize
        constants of tools/fpd_train.py:182-184, in the channel order of the images handed to crop()."""
        self.image_size, self.heatmap_size, self.sigma = tuple(int(v) for v in image_size), tuple(int(v) for v in heatmap_size), int(sigma)
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise R.FpdError('DevicePipeline runs on a CUDA (ROCm) device; there is no CPU path')
        self.g = torch.from_numpy(np.ascontiguousarray(gaussian_patch(self.sigma), np.float32)).to(self.device)
        self.mean, self.std = tuple(float(v) for v in mean), tuple(float(v) for v in std)

    def generate_target(self, joints, joints_vis):
        """joints [B,J,3] (network-input pixels, after the affine transform), joints_vis [B,J,3] or [B,J] ->
        (target [B,J,h,w] fp32, target_weight [B,J,1] fp32) on the device."""
        jt = torch.as_tensor(joints, dtype=torch.float64).to(self.device).contiguous()
        vis = torch.as_tensor(joints_vis, dtype=torch.float32)
        vis = (vis[..., 0] if vis.dim() == 3 else vis).to(self.device).contiguous()
        b, j = vis.shape
        w, h = self.heatmap_size
        target = torch.empty((b, j, h, w), dtype=torch.float32, device=self.device)
        weight = torch.empty((b, j, 1), dtype=torch.float32, device=self.device)
        a = R.TargetsT()
        a.B, a.J, a.H, a.W, a.patch = b, j, h, w, self.g.shape[0]
        a.stride_x, a.stride_y = self.image_size[0] / self.heatmap_size[0], self.image_size[1] / self.heatmap_size[1]
        a.joints, a.vis, a.g = jt.data_ptr(), vis.data_ptr(), self.g.data_ptr()
        a.target, a.weight = target.data_ptr(), weight.data_ptr()
        R.check(R.lib().fpd_render_targets(a, R.current_stream()), 'fpd_render_targets')
        return target, weight

    def transform_joints(self, joints, joints_vis, trans):
        """joints[i, 0:2] <- trans . [x, y, 1] for visible joints (JointsDataset.py:170-172), batched float64 on the host
        arrays' device (B*J*2 values: not worth a kernel)."""
        jt = torch.as_tensor(joints, dtype=torch.float64).clone()
        t = torch.as_tensor(np.asarray(trans), dtype=torch.float64)
        vis = torch.as_tensor(joints_vis, dtype=torch.float64)
        vis = vis[..., 0] if vis.dim() == 3 else vis
        xy1 = torch.cat([jt[..., 0:2], torch.ones_like(jt[..., 0:1])], dim=-1)
        new = torch.einsum('bik,bjk->bji', t, xy1)
        jt[..., 0:2] = torch.where(vis[..., None] > 0.0, new, jt[..., 0:2])
        return jt

    def crop(self, images, trans):
        """images: list of B uint8 [h_i, w_i, 3] CUDA tensors (decoded, any size); trans [B,2,3]: the SRC->DST matrices
        get_affine_transform returns (what the reference passes to cv2.warpAffine).  -> input [B,3,H,W] fp32, normalised."""
        b = len(images)
        w, h = self.image_size
        table = (R.WarpSrcT * b)()
        keep = []
        for i, im in enumerate(images):
            if not (im.is_cuda and im.dtype == torch.uint8 and im.dim() == 3 and im.shape[2] == 3):
                raise R.FpdError('crop: image %d must be a uint8 [h,w,3] CUDA tensor' % i)
            im = im.contiguous()
            keep.append(im)
            table[i].img, table[i].h, table[i].w, table[i].row_bytes = im.data_ptr(), im.shape[0], im.shape[1], im.shape[1] * 3
            minv = invert_affine(trans[i])
            for k in range(6):
                table[i].minv[k] = float(minv.reshape(-1)[k])
        raw = torch.frombuffer(bytearray(bytes(table)), dtype=torch.uint8).to(self.device)
        out = torch.empty((b, 3, h, w), dtype=torch.float32, device=self.device)
        a = R.WarpT()
        a.B, a.H, a.W = b, h, w
        a.src, a.out = raw.data_ptr(), out.data_ptr()
        for c in range(3):
            a.mean[c], a.std[c] = self.mean[c], self.std[c]
        R.check(R.lib().fpd_warp_affine(a, R.current_stream()), 'fpd_warp_affine')
        raw.record_stream(torch.cuda.current_stream())
        del keep
        return out


assert C.sizeof(R.WarpSrcT) == 72
