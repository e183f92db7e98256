"""GridMask train-time image augmentation (reference: projects/mmdet3d_plugin/models/utils/grid_mask.py:70-124,
instantiated by BEVFormerOcc as GridMask(True, True, rotate=1, offset=False, ratio=0.5, mode=1, prob=0.7),
bevformer_occ.py:52-53, applied in extract_img_feat when `use_grid_mask`, :81-82).

Draws from numpy's global RNG in the reference's order (apply?, period d, row start, column start, angle), so
a run seeded like the reference masks the same pixels.  The mask is built on the host (one (1.5h, 1.5w) byte
plane per call) and applied on the device."""
import numpy as np
import torch
import torch.nn as nn


class GridMask(nn.Module):
    def __init__(self, use_h, use_w, rotate=1, offset=False, ratio=0.5, mode=0, prob=1.):
        super().__init__()
        self.use_h, self.use_w, self.rotate, self.offset = use_h, use_w, rotate, offset
        self.ratio, self.mode, self.st_prob, self.prob = ratio, mode, prob, prob
        self.fp16_enable = False

    def set_prob(self, epoch, max_epoch):
        self.prob = self.st_prob * epoch / max_epoch

    @staticmethod
    def _rotate_nearest(mask, degrees):
        """Counter-clockwise rotation about the image centre, nearest neighbour, zero fill (PIL's
        Image.rotate defaults)."""
        if degrees % 360 == 0:
            return mask
        hh, ww = mask.shape
        a = np.deg2rad(degrees)
        cy, cx = hh / 2.0, ww / 2.0
        y, x = np.mgrid[0:hh, 0:ww].astype(np.float64)
        xs, ys = x + 0.5 - cx, y + 0.5 - cy
        sx = np.cos(a) * xs - np.sin(a) * ys + cx
        sy = np.sin(a) * xs + np.cos(a) * ys + cy
        ix, iy = np.floor(sx).astype(np.int64), np.floor(sy).astype(np.int64)
        ok = (ix >= 0) & (ix < ww) & (iy >= 0) & (iy < hh)
        out = np.zeros_like(mask)
        out[ok] = mask[iy[ok], ix[ok]]
        return out

    def make_mask(self, h, w):
        """The (h, w) float32 multiplier of one call (host array), consuming the RNG like the reference."""
        hh, ww = int(1.5 * h), int(1.5 * w)
        d = np.random.randint(2, h)
        self.l = min(max(int(d * self.ratio + 0.5), 1), d - 1)
        mask = np.ones((hh, ww), np.float32)
        st_h, st_w = np.random.randint(d), np.random.randint(d)
        if self.use_h:
            for i in range(hh // d):
                s = d * i + st_h
                mask[s:min(s + self.l, hh), :] = 0
        if self.use_w:
            for i in range(ww // d):
                s = d * i + st_w
                mask[:, s:min(s + self.l, ww)] = 0
        r = np.random.randint(self.rotate)
        mask = self._rotate_nearest(np.uint8(mask), r).astype(np.float32)
        mask = mask[(hh - h) // 2:(hh - h) // 2 + h, (ww - w) // 2:(ww - w) // 2 + w]
        return 1 - mask if self.mode == 1 else mask

    def forward(self, x):
        if np.random.rand() > self.prob or not self.training:
            return x
        n, c, h, w = x.size()
        mask = torch.from_numpy(np.ascontiguousarray(self.make_mask(h, w))).to(device=x.device, dtype=x.dtype)
        if self.offset:
            off = torch.from_numpy(2 * (np.random.rand(h, w) - 0.5)).to(device=x.device, dtype=x.dtype)
            return x * mask + off * (1 - mask)
        return x * mask
