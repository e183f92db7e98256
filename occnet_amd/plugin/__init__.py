"""Host-side mirror of the reference's `projects.mmdet3d_plugin` package for the forward hot path.

Importing this package registers every name the occ configs resolve by string.
Unlike the reference's plugin import it has no side effects beyond registration: no extension is
JIT-compiled and no dataset / metric code is pulled in.
"""
from . import bricks  # noqa: F401  (norm / conv / activation / FFN / positional encoding / losses)
from .backbone import FPN, ResNet  # noqa: F401
from .bevformer_occ import BEVFormerOcc  # noqa: F401
from .bevformer_occ_head import BEVFormerOccHead  # noqa: F401
from .config import Config, ConfigDict, import_plugin  # noqa: F401
from .detection_names import (CustomMSDeformableAttention, DetectionTransformerDecoder,  # noqa: F401
                              LearnedPositionalEncoding3D, PerceptionTransformer)
from .encoder import BEVFormerEncoder, BEVFormerLayer, MyCustomBaseTransformerLayer  # noqa: F401
from .functions import (MultiScaleDeformableAttnFunction_fp16,  # noqa: F401
                        MultiScaleDeformableAttnFunction_fp32)
from .registry import *  # noqa: F401,F403
from .registry import (BBOX_ASSIGNERS, DATASETS, MATCH_COST, OPTIMIZERS, PIPELINES, RUNNERS,
                       SAMPLER, register_parse_only)
from .spatial_cross_attention import MSDeformableAttention3D, SpatialCrossAttention  # noqa: F401
from .temporal_self_attention import TemporalSelfAttention  # noqa: F401
from .transformer_occ import TransformerOcc  # noqa: F401

# names that only have to parse (data side / optimisation / detection leftovers in train_cfg)
register_parse_only(DATASETS, ['NuSceneOcc', 'ConcatDataset'])
register_parse_only(PIPELINES, ['LoadMultiViewImageFromFiles', 'LoadOccGTFromFile',
                                'PhotoMetricDistortionMultiViewImage', 'NormalizeMultiviewImage',
                                'PadMultiViewImage', 'DefaultFormatBundle3D', 'CustomCollect3D',
                                'MultiScaleFlipAug3D', 'LoadAnnotations3D', 'ObjectRangeFilter',
                                'ObjectNameFilter', 'RandomScaleImageMultiViewImage'])
register_parse_only(SAMPLER, ['DistributedGroupSampler', 'DistributedSampler'])
register_parse_only(BBOX_ASSIGNERS, ['HungarianAssigner3D'])
register_parse_only(MATCH_COST, ['FocalLossCost', 'BBox3DL1Cost', 'IoUCost'])
register_parse_only(OPTIMIZERS, ['AdamW', 'AdamW2'])
register_parse_only(RUNNERS, ['EpochBasedRunner', 'EpochBasedRunner_video'])
