"""BEVFormerOcc detector shell: images -> ResNet/FPN features (B, N, C, H, W) -> BEVFormerOccHead.

Mirror of the reference's projects/mmdet3d_plugin/bevformer/detectors/bevformer_occ.py (registry
name, constructor kwargs of the MVXTwoStageDetector call shape, `img_backbone` / `img_neck` /
`pts_bbox_head` attribute names, forward(return_loss=...) / forward_train / forward_test /
simple_test / obtain_history_bev contracts).  As in mmdet3d's MVXTwoStageDetector, the `pts` part of
train_cfg / test_cfg is injected into the head's config.
"""
import torch

from .. import cache_epoch, ext
from .bricks import BaseModule
from .grid_mask import GridMask
from .registry import DETECTORS, build_backbone, build_head, build_neck


@DETECTORS.register_module()
class BEVFormerOcc(BaseModule):

    def __init__(self, use_grid_mask=False, pts_voxel_layer=None, pts_voxel_encoder=None,
                 pts_middle_encoder=None, pts_fusion_layer=None, img_backbone=None,
                 pts_backbone=None, img_neck=None, pts_neck=None, pts_bbox_head=None,
                 img_roi_head=None, img_rpn_head=None, train_cfg=None, test_cfg=None,
                 pretrained=None, video_test_mode=False):
        super().__init__()
        for name, v in (('pts_voxel_layer', pts_voxel_layer), ('pts_voxel_encoder', pts_voxel_encoder),
                        ('pts_middle_encoder', pts_middle_encoder), ('pts_fusion_layer', pts_fusion_layer),
                        ('pts_backbone', pts_backbone), ('pts_neck', pts_neck),
                        ('img_roi_head', img_roi_head), ('img_rpn_head', img_rpn_head)):
            if v is not None:
                raise NotImplementedError(f'{name}: the occupancy model is camera-only')
        if pts_bbox_head:
            pts_bbox_head = dict(pts_bbox_head)
            pts_bbox_head.update(train_cfg=train_cfg.get('pts') if train_cfg else None)
            pts_bbox_head.update(test_cfg=test_cfg.get('pts') if test_cfg else None)
            self.pts_bbox_head = build_head(pts_bbox_head)
        if img_backbone:
            self.img_backbone = build_backbone(img_backbone)
        if img_neck is not None:
            self.img_neck = build_neck(img_neck)
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        # train-time image augmentation, applied in extract_img_feat when use_grid_mask (reference :52-53,81-82)
        self.grid_mask = GridMask(True, True, rotate=1, offset=False, ratio=0.5, mode=1, prob=0.7)
        self.use_grid_mask = use_grid_mask
        self.fp16_enabled = False
        # mixed-precision knob of THIS implementation (the reference is fp32 throughout): when set (e.g.
        # torch.bfloat16) the stock ResNet/FPN modules of the training / autograd path run under torch.autocast
        # and the FPN maps are handed to the fp32 hot path as float32.  It lives inside the detector so that
        # a DDP-wrapped model is driven through DDP.forward (the reducer must see the forward).
        self.backbone_autocast_dtype = None
        self.pretrained = pretrained
        self.video_test_mode = video_test_mode
        self.prev_frame_info = {'prev_bev': None, 'scene_token': None, 'prev_pos': 0, 'prev_angle': 0}

    def train(self, mode=True):
        """Mode switch + cache safety net.  Every derived-weight cache (packed Linear / chain weights, folded backbone
        plan, decoder / heads packs) is keyed on (address, _version) of its sources; writes through `.data`
        (param.data.copy_: mmcv's EMAHook swap before validation, manual loads) do not bump _version.  On a REAL
        train -> eval transition the contents are therefore fingerprinted (per tensor: 2-norm and a position-weighted sum,
        compared element by element; one host sync) and the cache epoch is bumped when they changed since the last check
        (the first check counts as a change); the eval <-> train flips inside obtain_history_bev (twice per training step)
        skip it.  NOT covered: a `.data` write while the model already is in eval mode (no transition
        to hang the check on) — call occnet_amd.invalidate_caches() after such a write."""
        was_training = self.training
        # parameters / BatchNorm statistics may change while training, and an eval() entry is where stale plans would be
        # served from: re-check the folded plan's sources at the next forward in either case
        object.__setattr__(self, '_plan_dirty', True)
        out = super().train(mode)
        if was_training and not mode and not getattr(self, '_in_history', False):
            fp = self._content_fingerprint()
            old = getattr(self, '_content_fp', None)
            if fp is not None and (old is None or old.shape != fp.shape or not torch.equal(old, fp)):
                from .bricks import _bump_cache_epoch
                _bump_cache_epoch()
                object.__setattr__(self, '_content_fp', fp)
        return out

    def _content_fingerprint(self):
        """(n_tensors, 2) float64 host tensor — per device parameter / buffer its 2-norm and its sum weighted by a fixed
        pseudo-random ramp over the element index (None on the host or without floating-point tensors): changes whenever a
        weight is rewritten, however it was written.  The weighted sum is ORDER-sensitive: a sign flip, a permutation or two
        equal-norm tensors trading places, which a sum of norms cannot see, all move it."""
        ts = [t.detach() for t in list(self.parameters()) + list(self.buffers()) if t.is_cuda and t.is_floating_point()]
        if not ts:
            return None
        n_max = max(t.numel() for t in ts)
        ramp = getattr(self, '_fp_ramp', None)
        if ramp is None or ramp.numel() < n_max or ramp.device != ts[0].device:
            idx = torch.arange(n_max, device=ts[0].device, dtype=torch.float32)
            ramp = torch.frac(torch.sin(idx * 12.9898 + 0.5) * 43758.5453) + 0.5           # in (-0.5, 1.5), fixed
            object.__setattr__(self, '_fp_ramp', ramp)
        norms = torch._foreach_norm(ts)
        dots = [torch.dot(t.reshape(-1).float(), ramp[:t.numel()]) for t in ts]
        return torch.stack([torch.stack([n.double() for n in norms]), torch.stack([d.double() for d in dots])], 1).cpu()

    def _backbone_signature(self):
        mods = [m for m in (getattr(self, 'img_backbone', None), getattr(self, 'img_neck', None)) if m is not None]
        return tuple((t.data_ptr(), t._version) for m in mods for t in list(m.parameters()) + list(m.buffers()))

    def _current_plan(self):
        """The folded inference backbone, rebuilt when its source parameters changed since the fold."""
        plan = getattr(self, '_inference_backbone', None)
        if plan is None:
            return None
        stale = plan.built_epoch != cache_epoch()
        if not stale and getattr(self, '_plan_dirty', False):
            stale = plan.signature != self._backbone_signature()
            object.__setattr__(self, '_plan_dirty', self.training)
        if stale:
            self.enable_fused_backbone(**self._inference_backbone_args)
            plan = self._inference_backbone
        return plan

    @property
    def with_img_neck(self):
        return hasattr(self, 'img_neck') and self.img_neck is not None

    def init_weights(self):
        for m in (getattr(self, 'img_backbone', None), getattr(self, 'pts_bbox_head', None)):
            if m is not None:
                m.init_weights()

    def extract_img_feat(self, img, img_metas=None, len_queue=None):
        """img (B, N, 3, H, W) -> list of (B, N, C, h, w) (or (B/len_queue, len_queue, N, C, h, w))."""
        if img is None:
            return None
        B = img.size(0)
        if img.dim() == 5:
            B, N, C, H, W = img.size()
            img = img.reshape(B * N, C, H, W)
        plan = None
        if not self.training and not torch.is_grad_enabled():
            plan = self._current_plan()     # refolded if the parameters were reloaded / trained since the fold
        if plan is not None:
            img_feats = plan(img)       # BN-folded, NHWC, own bf16 kernels
        else:
            if self.use_grid_mask:
                img = self.grid_mask(img)
            ac = self.backbone_autocast_dtype
            with torch.autocast(img.device.type, dtype=ac or torch.bfloat16, enabled=ac is not None):
                img_feats = self.img_backbone(img)
                if isinstance(img_feats, dict):
                    img_feats = list(img_feats.values())
                if self.with_img_neck:
                    img_feats = self.img_neck(img_feats)
            if ac is not None:
                img_feats = [f.float() for f in img_feats]
        out = []
        for f in img_feats:
            BN, C, H, W = f.size()
            if len_queue is not None:
                out.append(f.view(int(B / len_queue), len_queue, int(BN / B), C, H, W))
            else:
                out.append(f.view(B, int(BN / B), C, H, W))
                # max|x| over the maps, accumulated by the plan's FPN output convolutions (backbone.py): rides on the views
                am = ext.absmax_of(f)
                if am is not None:
                    ext.attach_absmax(out[-1], am)
        return out

    def enable_fused_backbone(self, dtype=torch.bfloat16, fused_ops=False, hip_tail=True, use_graph=False,
                              fused_bottleneck=True):
        """Inference-only: run ResNet+FPN through FusedInferenceBackbone (eval BN folded into the
        convolutions, NHWC, MIOpen's fused conv+bias(+add)+ReLU).  Call again after changing backbone
        weights; pass dtype=None to disable."""
        from .backbone import FusedInferenceBackbone
        object.__setattr__(self, '_inference_backbone', None)
        object.__setattr__(self, '_inference_backbone_args', dict(
            dtype=dtype, fused_ops=fused_ops, hip_tail=hip_tail, use_graph=use_graph,
            fused_bottleneck=fused_bottleneck))
        # folded while training (no explicit .train() call needed for that): the sources may move before the first forward
        object.__setattr__(self, '_plan_dirty', bool(self.training))
        if dtype is not None:
            plan = FusedInferenceBackbone(self.img_backbone, self.img_neck, dtype=dtype,
                                          fused_ops=fused_ops, hip_tail=hip_tail,
                                          fused_bottleneck=fused_bottleneck)
            plan.use_graph = use_graph
            plan.built_epoch = cache_epoch()
            plan.signature = self._backbone_signature()
            object.__setattr__(self, '_inference_backbone', plan)   # not a sub-module: owns copies
        return self

    def extract_feat(self, img, img_metas=None, len_queue=None):
        return self.extract_img_feat(img, img_metas, len_queue=len_queue)

    def extract_feat_u8(self, img_u8, img_norm_cfg, size_divisor=32):
        """Device-side input path: img_u8 (B, N, Hs, Ws, 3) uint8 HWC raw camera images on
        the device.  NormalizeMultiviewImage(**img_norm_cfg) + PadMultiViewImage(size_divisor) + the layout
        change of DefaultFormatBundle3D (reference transform_3d.py:31-45,82-94) run inside the stem kernel of
        the inference plan, or as torch device ops without it.  -> (list of (B, N, C, h, w) maps, padded (H, W))."""
        B, N, Hs, Ws, _ = img_u8.shape
        mean, std = img_norm_cfg['mean'], img_norm_cfg['std']
        to_rgb = bool(img_norm_cfg.get('to_rgb', True))
        plan = None if self.training else self._current_plan()
        if plan is not None and getattr(plan, '_stem_fused', False):
            feats, hw = plan.forward_u8(img_u8.reshape(B * N, Hs, Ws, 3), mean, std, to_rgb, size_divisor)
        else:
            x = img_u8.reshape(B * N, Hs, Ws, 3).float()
            if to_rgb:
                x = x.flip(-1)
            stdinv = [float(1.0 / float(torch.tensor(v, dtype=torch.float32))) for v in std]   # mmcv: * (1 / std)
            x = (x - x.new_tensor(mean)) * x.new_tensor(stdinv)
            d = int(size_divisor)
            H, W = (Hs + d - 1) // d * d, (Ws + d - 1) // d * d
            x = torch.nn.functional.pad(x.permute(0, 3, 1, 2), (0, W - Ws, 0, H - Hs))
            return self.extract_img_feat(x.reshape(B, N, 3, H, W).contiguous()), (H, W)
        out = []
        for f in feats:
            BN, C, H, W = f.size()
            out.append(f.view(B, N, C, H, W))
            am = ext.absmax_of(f)
            if am is not None:
                ext.attach_absmax(out[-1], am)
        return out, hw

    def load_checkpoint(self, path_or_state, strict=False, map_location='cpu'):
        """Load an mmcv-format checkpoint ({'state_dict': ..., 'meta': ...}, what the reference's
        `load_checkpoint(model, ckpt, map_location='cpu')` reads, tools/test.py:213) or a bare state_dict; a
        'module.' prefix (saved from a DDP wrapper) is stripped.  Every derived-weight cache is invalidated.
        -> (missing_keys, unexpected_keys)."""
        from .. import invalidate_caches
        ckpt = path_or_state
        if isinstance(ckpt, (str, bytes)) or hasattr(ckpt, '__fspath__'):
            ckpt = torch.load(ckpt, map_location=map_location, weights_only=False)
        sd = ckpt.get('state_dict', ckpt) if isinstance(ckpt, dict) else ckpt
        sd = {(k[7:] if k.startswith('module.') else k): v for k, v in sd.items()}
        res = self.load_state_dict(sd, strict=strict)
        invalidate_caches(self)
        return list(res.missing_keys), list(res.unexpected_keys)

    def forward_pts_train(self, pts_feats, gt_bboxes_3d, gt_labels_3d, voxel_semantics, voxel_flow,
                          mask_camera, img_metas, gt_bboxes_ignore=None, prev_bev=None):
        outs = self.pts_bbox_head(pts_feats, img_metas, prev_bev)
        return self.pts_bbox_head.loss(voxel_semantics, voxel_flow, mask_camera, outs,
                                       img_metas=img_metas)

    def forward(self, return_loss=True, **kwargs):
        if return_loss:
            return self.forward_train(**kwargs)
        return self.forward_test(**kwargs)

    def obtain_history_bev(self, imgs_queue, img_metas_list):
        """BEV of the history frames, iteratively, without gradients
        (imgs_queue (bs, len_queue, N, 3, H, W))."""
        was_training = self.training
        object.__setattr__(self, '_in_history', True)      # no content fingerprint for this eval <-> train flip
        self.eval()
        object.__setattr__(self, '_in_history', False)
        with torch.no_grad():
            prev_bev = None
            bs, len_queue, num_cams, C, H, W = imgs_queue.shape
            imgs_queue = imgs_queue.reshape(bs * len_queue, num_cams, C, H, W)
            img_feats_list = self.extract_feat(img=imgs_queue, len_queue=len_queue)
            for i in range(len_queue):
                img_metas = [each[i] for each in img_metas_list]
                if not img_metas[0]['prev_bev_exists']:
                    prev_bev = None
                img_feats = [each_scale[:, i] for each_scale in img_feats_list]
                prev_bev = self.pts_bbox_head(img_feats, img_metas, prev_bev, only_bev=True)
        if was_training:
            self.train()
        return prev_bev

    def forward_train(self, points=None, img_metas=None, gt_bboxes_3d=None, gt_labels_3d=None,
                      voxel_semantics=None, voxel_flow=None, mask_lidar=None, mask_camera=None,
                      gt_labels=None, gt_bboxes=None, img=None, proposals=None,
                      gt_bboxes_ignore=None, img_depth=None, img_mask=None):
        img_feats = self.extract_feat(img=img, img_metas=img_metas)
        losses = dict()
        losses.update(self.forward_pts_train(img_feats, gt_bboxes_3d, gt_labels_3d, voxel_semantics,
                                             voxel_flow, mask_camera, img_metas, gt_bboxes_ignore,
                                             prev_bev=None))
        return losses

    def forward_test(self, img_metas, img=None, voxel_semantics=None, mask_lidar=None,
                     mask_camera=None, **kwargs):
        if not isinstance(img_metas, list):
            raise TypeError('img_metas must be a list, but got {}'.format(type(img_metas)))
        img = [img] if img is None else img
        new_prev_bev, occ_results, flow_results = self.simple_test(img_metas[0], img[0],
                                                                   prev_bev=None, **kwargs)
        return {'occ_results': occ_results.cpu(), 'flow_results': flow_results.cpu()}

    def simple_test_pts(self, x, img_metas, prev_bev=None, rescale=False):
        outs = self.pts_bbox_head(x, img_metas, prev_bev=prev_bev, test=True)
        occ, flow = self.pts_bbox_head.get_occ(outs, img_metas, rescale=rescale)
        return outs['bev_embed'], occ, flow

    def simple_test(self, img_metas, img=None, prev_bev=None, rescale=False):
        img_feats = self.extract_feat(img=img, img_metas=img_metas)
        return self.simple_test_pts(img_feats, img_metas, prev_bev, rescale=rescale)
