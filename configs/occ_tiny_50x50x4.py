# Plumbing-size config (BASELINE.json configs[0]): 1 camera 256x256, 50x50x4 voxels.  Every module is
# size-parametric, so this is the base model with smaller numbers; no image backbone (features in).
_base_ = ['./occ_base_200x200x16.py']
bev_h, bev_w, voxel_z = 50, 50, 4
model = dict(
    img_backbone=None, img_neck=None,
    pts_bbox_head=dict(
        bev_h=bev_h, bev_w=bev_w,
        positional_encoding=dict(row_num_embed=bev_h, col_num_embed=bev_w),
        transformer=dict(pillar_h=voxel_z, num_cams=1, rotate_center=[25, 25],
                         encoder=dict(
                             num_layers=2, num_points_in_pillar=4,
                             transformerlayers=dict(attn_cfgs=[      # lists replace the base's: one camera
                                 dict(type='TemporalSelfAttention', embed_dims=256, num_levels=1),
                                 dict(type='SpatialCrossAttention', pc_range=[-40, -40, -1.0, 40, 40, 5.4],
                                      embed_dims=256, num_cams=1,
                                      deformable_attention=dict(type='MSDeformableAttention3D', embed_dims=256,
                                                                num_points=8, num_levels=4)),
                             ])))))
input_geometry = dict(num_cams=1, img_h=256, img_w=256, feat_shapes=((32, 32), (16, 16), (8, 8), (4, 4)))
