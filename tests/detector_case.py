"""The small full-detector training case shared by tests/test_cpu_autograd.py (CPU: the checker checked) and
tests/test_configs_gpu.py (GPU: BASELINE configs[3] parity): PDM-SSD with a three-level PointNet2MSG, the PDM neck,
both heads, 2 x 2048 lidar-like points with a few labelled boxes that hold points."""
import numpy as np
import torch

from pdm_ssd_amd import synthetic
from pdm_ssd_amd.detector_config import PDM_SSD_CFG, build_pdm_ssd

SMALL = dict(PDM_SSD_CFG)
SMALL['BACKBONE_3D'] = {'NAME': 'PointNet2MSG',
                        'SA_CONFIG': {'NPOINTS': [512, 128, 32], 'RADIUS': [[0.5, 1.0], [1.0, 2.0], [2.0, 4.0]],
                                      'NSAMPLE': [[16, 32], [16, 32], [16, 32]],
                                      'MLPS': [[[16, 16, 32], [32, 32, 64]], [[64, 64, 128], [64, 96, 128]],
                                               [[128, 196, 256], [128, 196, 256]]]},
                        'FP_MLPS': [[128, 128], [256, 256], [512, 512]]}
SMALL['MAP_TO_BEV'] = dict(PDM_SSD_CFG['MAP_TO_BEV'], FEATURE_DIM=32, DILATION=[5, 5, 1])


def scene_boxes(B, M, seed):
    rng = np.random.default_rng(seed)
    gt = np.zeros((B, M, 8), dtype=np.float32)
    sizes = np.array([[3.9, 1.6, 1.56], [0.8, 0.6, 1.73], [1.76, 0.6, 1.73]], dtype=np.float32)
    for b in range(B):
        k = M - b
        cls = rng.integers(1, 4, k)
        gt[b, :k, 0] = rng.uniform(5, 60, k); gt[b, :k, 1] = rng.uniform(-30, 30, k); gt[b, :k, 2] = rng.uniform(-1.5, -0.5, k)
        gt[b, :k, 3:6] = sizes[cls - 1] * rng.uniform(0.9, 1.1, (k, 3))
        gt[b, :k, 6] = rng.uniform(-np.pi, np.pi, k)
        gt[b, :k, 7] = cls
    return gt


def build_case(B=2, N=2048, seed=1):
    """-> (model on the CPU in train mode, clouds (B, N, 4) numpy, gt_boxes (B, 6, 8) numpy)"""
    torch.manual_seed(seed)
    model = build_pdm_ssd(SMALL).train()
    with torch.no_grad():
        model.map_to_bev_module.coef.weight.normal_(0.0, 0.02)   # zero-initialised by the module: give the SH path a gradient
    cl = synthetic.lidar_like_clouds(B, N, 5)
    gt = scene_boxes(B, 6, 3)
    # a cluster of points on the first box of every sample, so the point head has foreground targets
    cl[:, :200, :3] = gt[:, :1, :3] + np.random.default_rng(0).normal(0, 0.5, (B, 200, 3)).astype(np.float32)
    return model, cl, gt


def grad_errors(got, want):
    """relative L2 error per parameter, measured against the larger of the parameter's own gradient norm and 1e-3 of
    the largest gradient norm of the model (a gradient that is ~0 in exact arithmetic has no relative error)."""
    floor = 1e-3 * max(float(w.norm()) for w in want.values())
    return {k: float((got[k].float().cpu() - w).norm()) / max(float(w.norm()), floor) for k, w in want.items()}
