"""BEVFormerOccHead — owner of the BEV queries and positional encoding, thin caller of TransformerOcc.

Mirror of the reference's projects/mmdet3d_plugin/bevformer/dense_heads/bevformer_occ_head.py
(registry name, constructor kwargs — swallowing in_channels / sync_cls_avg_factor / train_cfg /
test_cfg through **kwargs and reading kwargs['num_classes'] —, `bev_embedding`,
`positional_encoding`, `transformer` attribute names, forward / loss / get_occ contracts).
"""
import torch
import torch.nn as nn

from .. import cache_epoch
from .bricks import BaseModule
from .registry import HEADS, build_loss, build_positional_encoding, build_transformer


@HEADS.register_module()
class BEVFormerOccHead(BaseModule):

    def __init__(self, *args, with_box_refine=False, as_two_stage=False, transformer=None,
                 bbox_coder=None, num_cls_fcs=2, code_weights=None,
                 pc_range=[-40, -40, -1.0, 40, 40, 5.4], bev_h=30, bev_w=30, loss_occ=None,
                 loss_flow=None, use_mask=False, positional_encoding=None, **kwargs):
        super().__init__()
        self.bev_h = bev_h
        self.bev_w = bev_w
        self.fp16_enabled = False
        self.num_classes = kwargs['num_classes']
        self.use_mask = use_mask
        self.with_box_refine = with_box_refine
        self.as_two_stage = as_two_stage
        if self.as_two_stage:
            transformer['as_two_stage'] = self.as_two_stage
        self.pc_range = pc_range
        self.real_w = self.pc_range[3] - self.pc_range[0]
        self.real_h = self.pc_range[4] - self.pc_range[1]
        self.num_cls_fcs = num_cls_fcs - 1
        self.loss_occ = build_loss(loss_occ)
        self.loss_flow = build_loss(loss_flow)
        self.positional_encoding = build_positional_encoding(positional_encoding)
        self.transformer = build_transformer(transformer)
        self.embed_dims = self.transformer.embed_dims
        if not self.as_two_stage:
            self.bev_embedding = nn.Embedding(self.bev_h * self.bev_w, self.embed_dims)

    def init_weights(self):
        self.transformer.init_weights()

    def _bev_pos(self, bs, device, dtype):
        """Learned positional encoding of the BEV plane, (bs, C, bev_h, bev_w).  It depends only on the two
        embedding tables, so without autograd the SAME tensor is handed out until they change: downstream
        caches (the encoder's query-major copy, the TSA position term) key on its identity."""
        pe = self.positional_encoding
        needs_grad = torch.is_grad_enabled() and any(p.requires_grad for p in pe.parameters())
        key = None
        if not needs_grad:
            key = (bs, str(device), dtype, cache_epoch()) + tuple((p.data_ptr(), p._version) for p in pe.parameters())
            if getattr(self, '_pos_key', None) == key:
                return self._pos_val
        bev_mask = torch.zeros((bs, self.bev_h, self.bev_w), device=device).to(dtype)
        bev_pos = pe(bev_mask).to(dtype)
        if key is not None:
            bev_pos = bev_pos.detach()
            object.__setattr__(self, '_pos_key', key)
            object.__setattr__(self, '_pos_val', bev_pos)
        return bev_pos

    def forward(self, mlvl_feats, img_metas, prev_bev=None, only_bev=False, test=False):
        """mlvl_feats: list of (B, N, C, H, W) -> {'bev_embed','occ','flow'}; with only_bev the
        (bs, H*W, C) BEV embedding alone (history frames)."""
        bs = mlvl_feats[0].shape[0]
        # the reference takes the feature dtype (bevformer_occ_head.py:118); the MI355X hot path always
        # computes in fp32 (a half-precision backbone's maps are widened when they are flattened)
        dtype = mlvl_feats[0].dtype if mlvl_feats[0].dtype == torch.float64 else torch.float32
        bev_queries = self.bev_embedding.weight.to(dtype)
        bev_pos = self._bev_pos(bs, bev_queries.device, dtype)
        grid_length = (self.real_h / self.bev_h, self.real_w / self.bev_w)
        if only_bev:
            return self.transformer.get_bev_features(
                mlvl_feats, bev_queries, self.bev_h, self.bev_w, grid_length=grid_length,
                bev_pos=bev_pos, img_metas=img_metas, prev_bev=prev_bev)
        bev_embed, occ_outs, flow_outs = self.transformer(
            mlvl_feats, bev_queries, None, self.bev_h, self.bev_w, grid_length=grid_length,
            bev_pos=bev_pos, reg_branches=None, cls_branches=None, img_metas=img_metas,
            prev_bev=prev_bev)
        preds = {'bev_embed': bev_embed, 'occ': occ_outs, 'flow': flow_outs}
        cls = getattr(occ_outs, '_occ_cls', None)
        if cls is not None:     # the fused heads kernel decoded the classes in the same pass: an explicit entry
            preds['occ_cls'] = cls
        return preds

    def loss(self, voxel_semantics, voxel_flow, mask_camera, preds_dicts, gt_bboxes_ignore=None,
             img_metas=None):
        loss_occ, loss_flow = self.loss_single(voxel_semantics, voxel_flow, mask_camera,
                                               preds_dicts['occ'].float(),
                                               preds_dicts['flow'].float())
        return dict(loss_occ=loss_occ, loss_flow=loss_flow)

    def loss_single(self, voxel_semantics, voxel_flow, mask_camera, occ, flow):
        voxel_semantics = voxel_semantics.long()
        if self.use_mask:
            # the reference's masked branch only produces loss_occ (loss_flow is left undefined
            # there, bevformer_occ_head.py:183-188); the flow term here follows the unmasked branch
            mask_camera = mask_camera.reshape(-1)
            loss_occ = self.loss_occ(occ.reshape(-1, self.num_classes), voxel_semantics.reshape(-1),
                                     mask_camera, avg_factor=mask_camera.sum())
        else:
            loss_occ = self.loss_occ(occ.reshape(-1, self.num_classes), voxel_semantics.reshape(-1))
        loss_flow = self.loss_flow(flow.reshape(-1, 2), voxel_flow.reshape(-1, 2))
        return loss_occ, loss_flow

    def get_occ(self, preds_dicts, img_metas=None, rescale=False):
        """-> (class index per voxel (B, W, H, Z) int64, flow (B, W, H, Z, 2))."""
        occ = preds_dicts['occ']
        # 'occ_cls' (argmax of the logits, first index on ties, 0 for a row with a NaN — what softmax(-1).argmax(-1)
        # yields) is written by the fused heads kernel in the pass that writes `occ`; it is only valid while `occ` is
        # unmodified: callers that edit the logits must drop the key
        cls = preds_dicts.get('occ_cls')
        if cls is not None and cls.shape == occ.shape[:-1] and cls.device == occ.device:
            return cls, preds_dicts['flow']
        occ_score = occ.float().softmax(-1).argmax(-1)
        return occ_score, preds_dicts['flow']
