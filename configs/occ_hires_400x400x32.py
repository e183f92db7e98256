# High-resolution grid (BASELINE.json configs[4]): 400x400x32 voxels (0.2 m), 160 000 BEV queries,
# 256 // 32 = 8 decoder input channels.  Same cameras and feature maps as the base config.
_base_ = ['./occ_base_200x200x16.py']
bev_h, bev_w, voxel_z = 400, 400, 32
model = dict(
    pts_bbox_head=dict(
        bev_h=bev_h, bev_w=bev_w,
        positional_encoding=dict(row_num_embed=bev_h, col_num_embed=bev_w),
        transformer=dict(pillar_h=voxel_z, rotate_center=[200, 200])))
