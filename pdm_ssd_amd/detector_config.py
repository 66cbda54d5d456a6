"""The PDM-SSD model configuration the benchmark and tests build (key names as an OpenPCDet YAML would hold them; the
reference's own YAML files are git-ignored and absent).  Backbone = upstream pointrcnn.yaml's
PointNet2MSG; point head = its PointHeadBox settings; neck / heat-map head = this repo's spec."""
from types import SimpleNamespace

from . import synthetic
from .config import cfg_from_dict
from .pointnet2_backbone import POINTRCNN_MSG_CFG

CLASS_NAMES = ['Car', 'Pedestrian', 'Cyclist']
VOXEL_SIZE = [0.05, 0.05, 0.1]
GRID_SIZE = [1408, 1600, 40]

PDM_SSD_CFG = {
    'NAME': 'PDMSSD',
    'BACKBONE_3D': dict(POINTRCNN_MSG_CFG),
    'MAP_TO_BEV': {'NAME': 'PDMNeck', 'SOURCE_LAYER': 2, 'FEATURE_DIM': 128, 'DILATION': [7, 7, 1], 'SH_DEGREE': 2,
                   'BEV_STRIDE': 8, 'HEIGHT_BINS': 1, 'INPUT_CHANNELS': 256, 'NORMALIZE': True},
    'DENSE_HEAD': {'NAME': 'PDMHeatmapHead', 'CLASS_AGNOSTIC': False, 'SHARED_CONV_CHANNEL': 64, 'NUM_CONTEXT_CONV': 1, 'CONTEXT_CONV': 'separable',
                   'TARGET_ASSIGNER_CONFIG': {'FEATURE_MAP_STRIDE': 8, 'GAUSSIAN_OVERLAP': 0.1, 'MIN_RADIUS': 2},
                   'LOSS_CONFIG': {'LOSS_WEIGHTS': {'cls_weight': 1.0}}},
    'POINT_HEAD': {'NAME': 'PointHeadBox', 'CLS_FC': [256, 256], 'REG_FC': [256, 256], 'CLASS_AGNOSTIC': False,
                   'USE_POINT_FEATURES_BEFORE_FUSION': False,
                   'TARGET_CONFIG': {'GT_EXTRA_WIDTH': [0.2, 0.2, 0.2], 'BOX_CODER': 'PointResidualCoder',
                                     'BOX_CODER_CONFIG': {'use_mean_size': True,
                                                          'mean_size': [[3.9, 1.6, 1.56], [0.8, 0.6, 1.73], [1.76, 0.6, 1.73]]}},
                   'LOSS_CONFIG': {'LOSS_REG': 'WeightedSmoothL1Loss',
                                   'LOSS_WEIGHTS': {'point_cls_weight': 1.0, 'point_box_weight': 1.0,
                                                    'code_weights': [1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0]}}},
    'POST_PROCESSING': {'RECALL_THRESH_LIST': [0.3, 0.5, 0.7], 'SCORE_THRESH': 0.1, 'OUTPUT_RAW_SCORE': False,
                        'NMS_CONFIG': {'MULTI_CLASSES_NMS': False, 'NMS_TYPE': 'nms_gpu', 'NMS_THRESH': 0.1,
                                       'NMS_PRE_MAXSIZE': 4096, 'NMS_POST_MAXSIZE': 500}},
}


def synthetic_dataset(num_point_features=4):
    """The attributes Detector3DTemplate.build_networks reads from a dataset (detector3d_template.py:36-43)."""
    return SimpleNamespace(class_names=CLASS_NAMES, grid_size=GRID_SIZE, voxel_size=VOXEL_SIZE,
                           point_cloud_range=list(synthetic.KITTI_RANGE),
                           point_feature_encoder=SimpleNamespace(num_point_features=num_point_features))


def build_pdm_ssd(model_cfg=None, num_point_features=4):
    from .detectors import build_network
    cfg = cfg_from_dict(PDM_SSD_CFG if model_cfg is None else model_cfg)
    return build_network(cfg, num_class=len(CLASS_NAMES), dataset=synthetic_dataset(num_point_features))
