"""Instance-aware set abstraction: an SA layer whose down-sampling keeps the points
with the highest predicted foreground score instead of running FPS.

The PDM-SSD / IA-SSD lineage samples its deeper layers this way; that code is absent from the reference snapshot
(SURVEY F1), so the layer is build-defined on top of the snapshot's PointnetSAModuleMSG
(/root/reference/pcdet/ops/pointnet2/pointnet2_batch/pointnet2_modules.py:58-99, whose constructor and forward it
extends): `sample_type` 'D-FPS' = the reference's FPS; 'cls_aware' / 'ctr_aware' = the npoint input points with the
highest sigmoid(max_class logit) (`pdm_topk_sampling`: ties by lower index).  A `confidence_mlp` adds the per-point
class head whose logits the NEXT layer samples by; 'ctr_aware' differs from 'cls_aware' only in how that head is
supervised (centre-ness weighted targets), not in the forward pass.

Why it matters on this hardware: FPS is a latency-bound dependency chain (4095 serial iterations for 16384 -> 4096,
3.5 ms at bs=32); the score ranking is one radix-select + LDS sort per cloud (tens of microseconds).
"""
from typing import List, Optional

import torch
import torch.nn as nn

from .pointnet2_batch import pointnet2_utils
from .pointnet2_batch.pointnet2_modules import PointnetSAModuleMSG

SAMPLE_TYPES = ('D-FPS', 'cls_aware', 'ctr_aware')


class PointnetSAModuleMSG_WithSampling(PointnetSAModuleMSG):
    def __init__(self, *, npoint: int, sample_type: str = 'D-FPS', radii: List[float], nsamples: List[int],
                 mlps: List[List[int]], use_xyz: bool = True, pool_method: str = 'max_pool',
                 confidence_mlp: Optional[List[int]] = None, num_class: int = 3):
        assert sample_type in SAMPLE_TYPES, sample_type
        super().__init__(npoint=npoint, radii=radii, nsamples=nsamples, mlps=mlps, use_xyz=use_xyz, pool_method=pool_method)
        self.sample_type = sample_type
        self.confidence_layers = None
        if confidence_mlp:
            cin = sum(spec[-1] for spec in mlps)
            layers = []
            for cout in confidence_mlp:
                layers += [nn.Conv1d(cin, cout, kernel_size=1, bias=False), nn.BatchNorm1d(cout), nn.ReLU()]
                cin = cout
            layers.append(nn.Conv1d(cin, num_class, kernel_size=1, bias=True))
            self.confidence_layers = nn.Sequential(*layers)

    @torch.no_grad()
    def sample_indices(self, xyz: torch.Tensor, cls_features: Optional[torch.Tensor] = None) -> torch.Tensor:
        """int32 (B, npoint): FPS indices, or the npoint highest-scoring points (cls_features (B,N,num_class) logits)."""
        if self.sample_type == 'D-FPS':
            return pointnet2_utils.farthest_point_sample(xyz.contiguous(), self.npoint)
        assert cls_features is not None, f"sample_type {self.sample_type} needs the previous layer's class logits"
        score = torch.sigmoid(cls_features.max(dim=-1)[0])
        return pointnet2_utils.topk_sample(score, self.npoint)

    def forward(self, xyz: torch.Tensor, features: torch.Tensor = None, cls_features: torch.Tensor = None,
                new_xyz=None, idx_list=None):
        """xyz (B,N,3), features (B,C,N), cls_features (B,N,num_class) | None ->
        new_xyz (B,npoint,3), new_features (B,sum_k mlps[k][-1],npoint), cls_preds (B,npoint,num_class) | None."""
        if new_xyz is None:
            new_xyz = self.sample_from_idx(xyz, self.sample_indices(xyz, cls_features))
        new_xyz, new_features = super().forward(xyz, features, new_xyz=new_xyz, idx_list=idx_list)
        cls_preds = None
        if self.confidence_layers is not None:
            cls_preds = self.confidence_layers(new_features.contiguous()).transpose(1, 2).contiguous()
        return new_xyz, new_features, cls_preds
