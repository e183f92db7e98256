# BEVFormer-occ, base geometry: 6 cameras 928x1600 (900x1600 padded to /32) -> 200x200x16 voxels,
# 17 semantic classes + 2-D flow.  Same model as the reference's
# projects/configs/bevformer/bevformer_base_occ.py (which this loader also reads unchanged — see
# tests/test_config_registry.py); written out here so bench.py / smoke() do not depend on the
# reference tree being present.  Model-only: data pipeline / schedule keys are omitted.
plugin = True
plugin_dir = 'projects/mmdet3d_plugin/'

pc_range = [-40, -40, -1.0, 40, 40, 5.4]
dim = 256
levels = 4
classes = 17
bev_h, bev_w, voxel_z = 200, 200, 16

model = dict(
    type='BEVFormerOcc',
    use_grid_mask=True,
    video_test_mode=True,
    img_backbone=dict(type='ResNet', depth=50, num_stages=4, out_indices=(1, 2, 3), frozen_stages=1,
                      norm_cfg=dict(type='BN2d', requires_grad=True), norm_eval=True,
                      style='pytorch', with_cp=False),
    img_neck=dict(type='FPN', in_channels=[512, 1024, 2048], out_channels=dim, start_level=0,
                  add_extra_convs='on_output', num_outs=levels, relu_before_extra_convs=True),
    pts_bbox_head=dict(
        type='BEVFormerOccHead', pc_range=pc_range, bev_h=bev_h, bev_w=bev_w, num_classes=classes,
        in_channels=dim, with_box_refine=True, as_two_stage=False, use_mask=False,
        loss_occ=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0),
        loss_flow=dict(type='L1Loss', loss_weight=0.25),
        positional_encoding=dict(type='LearnedPositionalEncoding', num_feats=dim // 2,
                                 row_num_embed=bev_h, col_num_embed=bev_w),
        transformer=dict(
            type='TransformerOcc', pillar_h=voxel_z, num_classes=classes, use_3d=True,
            use_conv=False, norm_cfg=dict(type='BN'), norm_cfg_3d=dict(type='BN3d'),
            rotate_prev_bev=True, use_shift=True, use_can_bus=True, embed_dims=dim,
            encoder=dict(
                type='BEVFormerEncoder', num_layers=4, pc_range=pc_range, num_points_in_pillar=8,
                return_intermediate=False,
                transformerlayers=dict(
                    type='BEVFormerLayer',
                    attn_cfgs=[
                        dict(type='TemporalSelfAttention', embed_dims=dim, num_levels=1),
                        dict(type='SpatialCrossAttention', pc_range=pc_range, embed_dims=dim,
                             deformable_attention=dict(type='MSDeformableAttention3D',
                                                       embed_dims=dim, num_points=8,
                                                       num_levels=levels)),
                    ],
                    feedforward_channels=dim * 2, ffn_dropout=0.1,
                    operation_order=('self_attn', 'norm', 'cross_attn', 'norm', 'ffn', 'norm'))))))

# synthetic-input geometry used by bench.py / tests (nuScenes-nominal rig, occnet_amd/synthetic.py)
input_geometry = dict(num_cams=6, img_h=928, img_w=1600)
