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
                         encoder=dict(num_layers=2, num_points_in_pillar=4))))
input_geometry = dict(num_cams=1, img_h=256, img_w=256)
