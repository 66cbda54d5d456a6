"""pdm_ssd_amd — MI355X-native (gfx950) implementation of PDM-SSD's point-cloud hot path.

Public surface mirrors the reference's pcdet modules for this path:
  pdm_ssd_amd.pointnet2_batch.pointnet2_utils    <-> pcdet/ops/pointnet2/pointnet2_batch/pointnet2_utils.py
  pdm_ssd_amd.pointnet2_batch.pointnet2_modules  <-> pcdet/ops/pointnet2/pointnet2_batch/pointnet2_modules.py
  pdm_ssd_amd.pointnet2_backbone.PointNet2MSG    <-> pcdet/models/backbones_3d/pointnet2_backbone.py:9-94
  pdm_ssd_amd.pdm_neck.PDMNeck                   <-> the MAP_TO_BEV slot (no reference source)
All operators run hand-written HIP kernels through libpdmssd_hip.so (include/pdmssd_hip.h).
"""
__version__ = "0.1.0"
