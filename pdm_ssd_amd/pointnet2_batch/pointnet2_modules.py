"""Set-abstraction / feature-propagation modules over the HIP operators.

Constructor signatures, attribute names and state_dict keys (`mlps.{s}.{0,3,6}.weight`, BatchNorm at
`{1,4,7}`; `mlp.{0,3}.weight`) match
/root/reference/pcdet/ops/pointnet2/pointnet2_batch/pointnet2_modules.py so upstream checkpoints
load; forward semantics follow the cited lines.
"""
from typing import List, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import pointnet2_utils
from .. import fused

import os
from ..fused_bn import TrainSequential
# The autograd path feeds the shared MLPs (Conv2d 1x1 + BatchNorm2d + ReLU, torch/MIOpen) a channels-last tensor:
# MIOpen's NHWC batch-norm and implicit-GEMM kernels need no layout transposes (measured bs=32 train step 67.9 -> 49.7 ms).
CHANNELS_LAST_TRAINING = os.environ.get("PDM_CHANNELS_LAST", "1") == "1"
PRE_MIN_CIN = 16   # hoist the first layer's feature block when it is at least one 16-deep K block wide


def _shared_mlp(spec: List[int]) -> nn.Sequential:
    """Conv2d(1x1, no bias) -> BatchNorm2d -> ReLU per stage (ref :91-97, :132-139)."""
    layers = []
    for cin, cout in zip(spec[:-1], spec[1:]):
        layers += [nn.Conv2d(cin, cout, kernel_size=1, bias=False), nn.BatchNorm2d(cout), nn.ReLU()]
    return TrainSequential(*layers)


class _PointnetSAModuleBase(nn.Module):
    """ref :9-55."""

    def __init__(self):
        super().__init__()
        self.npoint = None
        self.groupers = None
        self.mlps = None
        self.pool_method = 'max_pool'

    def sample(self, xyz: torch.Tensor) -> Optional[torch.Tensor]:
        """FPS + gather of the sampled coordinates (ref :30-35) -> (B, npoint, 3)."""
        if self.npoint is None:
            return None
        return self.sample_from_idx(xyz, pointnet2_utils.farthest_point_sample(xyz, self.npoint))

    @staticmethod
    def sample_from_idx(xyz: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
        """The gather half of sample(): FPS indices (B,npoint) -> (B,npoint,3)."""
        xyz_flipped = xyz.transpose(1, 2).contiguous()
        return pointnet2_utils.gather_operation(xyz_flipped, idx).transpose(1, 2).contiguous()

    def _fused_packs(self, xyz, features):
        """Packed weights per scale when the fused fp32-MFMA inference path applies, else None:
        eval mode, no autograd, max-pool, QueryAndGroup(use_xyz) groupers, nsample % 16 == 0."""
        if self.training or torch.is_grad_enabled() or not getattr(self, 'use_fused', True):
            return None
        if self.pool_method != 'max_pool' or self.npoint is None or not xyz.is_cuda or xyz.dtype != torch.float32:
            return None
        if features is not None and features.dtype != torch.float32:
            return None
        cin = 0 if features is None else features.shape[1]
        packs = []
        for i, (grouper, mlp) in enumerate(zip(self.groupers, self.mlps)):
            if not isinstance(grouper, pointnet2_utils.QueryAndGroup) or not grouper.use_xyz or grouper.nsample % 16:
                return None
            # reference channel order [xyz(3), features(cin)] -> kernel order [features, xyz]
            perm = list(range(3, 3 + cin)) + [0, 1, 2]
            pk = fused.cached_pack(self, i, mlp, xyz.device, perm)
            if pk is None or pk.cin != cin + 3:
                return None
            packs.append(pk)
        return packs

    @torch.no_grad()
    def query(self, xyz, new_xyz):
        """Neighbour indices of every scale, [(B,M,nsample) int32]: coordinate-only, so a caller may compute them
        ahead of time (pdm_ssd_amd/pipeline.py) and pass them to forward(..., idx_list=)."""
        xyz, new_xyz = xyz.contiguous(), new_xyz.contiguous()
        with pointnet2_utils.shared_search_grids():   # one grid of `xyz` for every radius
            idxs = [pointnet2_utils.ball_query(g.radius, g.nsample, xyz, new_xyz) for g in self.groupers]
        return self._maybe_pack_all(idxs, xyz.shape[1])

    def _maybe_pack_all(self, idxs, n):
        """_maybe_pack of every scale; the two scales of an MSG level share one count -> scan -> fill sequence."""
        if (len(idxs) == 2 and getattr(self, 'use_pack', True) and all(i.shape[2] in (16, 32) for i in idxs) and not self.training
                and not torch.is_grad_enabled() and getattr(self, 'use_fused', True) and getattr(self, 'use_pair', True)):
            return fused.sa_pack_pair(idxs[0], idxs[1], n)
        return [self._maybe_pack(i, n) for i in idxs]

    def _maybe_pack(self, idx, n):
        """Compact the neighbour list for the fused kernels (fused.sa_pack) when the module takes its fused inference
        path: (pack, meta) instead of the (B,M,nsample) tensor.  `use_pack = False` keeps the dense list."""
        if (getattr(self, 'use_pack', True) and idx.shape[2] in (16, 32) and not self.training
                and not torch.is_grad_enabled() and getattr(self, 'use_fused', True)):
            return fused.sa_pack(idx, n)
        return idx

    def _forward_fused(self, packs, xyz, features, new_xyz, idx_list=None):
        B, M = new_xyz.shape[0], new_xyz.shape[1]
        feat_pm = None if features is None else features.transpose(1, 2).contiguous()  # (B,N,C) point-major
        xyz = xyz.contiguous()
        new_xyz = new_xyz.contiguous()
        ctot = sum(pk.cout for pk in packs)
        out_pm = torch.empty((B, M, ctot), dtype=torch.float32, device=xyz.device)
        coff = 0
        cin = 0 if feat_pm is None else feat_pm.shape[2]
        pre = None
        if cin >= PRE_MIN_CIN and getattr(self, 'use_pre', True):
            # hoist W1[:, features] out of the (centre, neighbour) loop: one projection of the N source points
            # serves every scale (reference channel order [xyz(3), features(cin)], pointnet2_utils.py:254)
            pre = fused.cached_pre_packs(self, 'pre', list(self.mlps), xyz.device, range(3, 3 + cin), range(3))
        if pre is not None:
            prepack, packs = pre
            z = torch.empty((B, xyz.shape[1], prepack.width), dtype=torch.float32, device=xyz.device)
            fused.rows_forward(prepack, feat_pm, z, relu_last=False)
        idxs = list(idx_list) if idx_list is not None else \
            self._maybe_pack_all([pointnet2_utils.ball_query(g.radius, g.nsample, xyz, new_xyz) for g in self.groupers], xyz.shape[1])
        if len(packs) == 2 and all(isinstance(q, tuple) for q in idxs) and getattr(self, 'use_pair', True):
            # both scales over their compacted lists in ONE launch: a deep level holds a few hundred row tiles per scale,
            # which two launches run one after the other on a mostly idle chip (bit-identical: csrc/fused_mlp.hip)
            coffs = [0, packs[0].cout]
            fused.sa_level_forward_packed(list(packs), xyz, new_xyz, feat_pm, None if pre is None else z,
                                          [0, 0] if pre is None else [prepack.offsets[0], prepack.offsets[1]], idxs,
                                          [g.nsample for g in self.groupers], out_pm, coffs)
            return new_xyz, out_pm.transpose(1, 2)
        for i, (grouper, pk) in enumerate(zip(self.groupers, packs)):
            idx = idxs[i]
            if isinstance(idx, tuple):   # compacted list: padding copies of the first hit are not computed
                fused.sa_scale_forward_packed(pk, xyz, new_xyz, feat_pm, None if pre is None else z,
                                              0 if pre is None else prepack.offsets[i], idx, grouper.nsample, out_pm, coff)
            elif pre is not None:
                fused.sa_scale_forward_pre(pk, xyz, new_xyz, z, prepack.offsets[i], idx, out_pm, coff)
            else:
                fused.sa_scale_forward(pk, xyz, new_xyz, feat_pm, idx, out_pm, coff)
            coff += pk.cout
        return new_xyz, out_pm.transpose(1, 2)  # logical (B, C, M) over point-major storage

    def forward(self, xyz: torch.Tensor, features: torch.Tensor = None, new_xyz=None,
                idx_list=None) -> Tuple[torch.Tensor, torch.Tensor]:
        """xyz (B,N,3), features (B,C,N) -> new_xyz (B,npoint,3), new_features (B, sum_k mlps[k][-1], npoint).

        In eval mode without autograd each scale runs as one fused HIP kernel (gather + MLP on fp32 MFMA
        + max-pool); the returned features are then a (B,C,M) VIEW of point-major (B,M,C) storage.
        """
        with pointnet2_utils.shared_search_grids():   # the scales search the same `xyz`, which nothing here rewrites
            return self._forward(xyz, features, new_xyz, idx_list)

    def _forward(self, xyz, features, new_xyz, idx_list):
        if new_xyz is None:
            new_xyz = self.sample(xyz)
        packs = self._fused_packs(xyz, features)
        if packs is not None and all(pk.cout % 4 == 0 for pk in packs):
            return self._forward_fused(packs, xyz, features, new_xyz, idx_list)
        from .. import fused_bn
        rows_path = (CHANNELS_LAST_TRAINING and fused_bn.ENABLED and fused_bn.ROWS_GEMM and xyz.is_cuda
                     and torch.is_autocast_enabled('cuda') and torch.get_autocast_dtype('cuda') == torch.bfloat16)
        if features is not None and not rows_path:
            features = features.contiguous()
        pooled = []
        for grouper, mlp in zip(self.groupers, self.mlps):
            if CHANNELS_LAST_TRAINING and isinstance(grouper, pointnet2_utils.QueryAndGroup):
                grouper.channels_last = True   # the grouped tensor is born NHWC (bf16 under autocast): no copies
                # 16-byte rows for the bf16 MFMA layers (zero channels up to a multiple of 8) — only for the consumer that
                # knows about the padding: TrainSequential's training stack.  In eval mode, under no_grad, with a plain
                # nn.Sequential or avg_pool the torch layers see exactly 3 + C channels.
                grouper.pad_to_8 = bool(rows_path and self.training and torch.is_grad_enabled()
                                        and isinstance(mlp, TrainSequential) and self.pool_method == 'max_pool')
            x = grouper(xyz, new_xyz, features)
            if CHANNELS_LAST_TRAINING:
                x = x.contiguous(memory_format=torch.channels_last)
            if self.pool_method == 'max_pool' and isinstance(mlp, TrainSequential):
                x = mlp.forward_max_pooled(x)   # last BatchNorm + ReLU and the max over nsample as one operator when training
            elif self.pool_method == 'max_pool':
                x = F.max_pool2d(mlp(x), kernel_size=[1, x.size(3)])
            elif self.pool_method == 'avg_pool':
                x = mlp(x)  # (B, mlp[-1], npoint, nsample)
                x = F.avg_pool2d(x, kernel_size=[1, x.size(3)])
            else:
                raise NotImplementedError
            pooled.append(x.squeeze(-1))
        if rows_path and all(p.dim() == 3 and p.transpose(1, 2).is_contiguous() for p in pooled):
            # every scale's pooled output is a (B, C, M) view of point-major (B, M, C) storage: concatenate THERE, so the result
            # is again such a view — the next level's grouping and the FP modules read rows without a transposing copy
            return new_xyz, torch.cat([p.transpose(1, 2) for p in pooled], dim=2).transpose(1, 2)
        return new_xyz, torch.cat(pooled, dim=1)


class PointnetSAModuleMSG(_PointnetSAModuleBase):
    """Multi-scale grouping SA layer (ref :58-99).

    Like the reference (:86-88) the constructor adds 3 to `mlps[i][0]` IN PLACE when use_xyz — callers
    that reuse a spec list rely on that side effect.  `bn` is accepted and ignored, as upstream.
    """

    def __init__(self, *, npoint: int, radii: List[float], nsamples: List[int], mlps: List[List[int]],
                 bn: bool = True, use_xyz: bool = True, pool_method='max_pool'):
        super().__init__()
        assert len(radii) == len(nsamples) == len(mlps)
        self.npoint = npoint
        self.groupers = nn.ModuleList()
        self.mlps = nn.ModuleList()
        for radius, nsample, mlp_spec in zip(radii, nsamples, mlps):
            self.groupers.append(
                pointnet2_utils.QueryAndGroup(radius, nsample, use_xyz=use_xyz)
                if npoint is not None else pointnet2_utils.GroupAll(use_xyz))
            if use_xyz:
                mlp_spec[0] += 3
            self.mlps.append(_shared_mlp(mlp_spec))
        self.pool_method = pool_method


class PointnetSAModule(PointnetSAModuleMSG):
    """Single-scale SA layer (ref :102-119)."""

    def __init__(self, *, mlp: List[int], npoint: int = None, radius: float = None, nsample: int = None,
                 bn: bool = True, use_xyz: bool = True, pool_method='max_pool'):
        super().__init__(mlps=[mlp], npoint=npoint, radii=[radius], nsamples=[nsample], bn=bn,
                         use_xyz=use_xyz, pool_method=pool_method)


class PointnetFPModule(nn.Module):
    """Feature propagation (ref :122-170)."""

    def __init__(self, *, mlp: List[int], bn: bool = True):
        super().__init__()
        self.mlp = _shared_mlp(mlp)

    @staticmethod
    @torch.no_grad()
    def interpolation(unknown, known):
        """(idx, weight) of ref :153-156: coordinate-only, may be computed ahead and passed as forward(..., interp=)."""
        return pointnet2_utils.three_nn_weights(unknown, known)

    def forward(self, unknown: torch.Tensor, known: torch.Tensor, unknow_feats: torch.Tensor,
                known_feats: torch.Tensor, interp=None, defer=None) -> torch.Tensor:
        """unknown (B,n,3), known (B,m,3), unknow_feats (B,C1,n), known_feats (B,C2,m) -> (B, mlp[-1], n).

        In eval mode without autograd, interpolation + concat + MLP run as one fused HIP kernel and the
        result is a (B,C,n) VIEW of point-major (B,n,C) storage.
        """
        if known is not None:
            fused_ok = not (self.training or torch.is_grad_enabled()) and getattr(self, 'use_fused', True) \
                and known_feats.is_cuda and known_feats.dtype == torch.float32
            if fused_ok and interp is not None:
                idx, weight = interp
            elif fused_ok or (unknown.is_cuda and not (unknown.requires_grad or known.requires_grad)):
                # ref :153-156 in two launches (coordinates carry no gradient in this pipeline: in training too, instead of
                # sqrt / add / reciprocal / sum / div as five torch kernels per module)
                idx, weight = pointnet2_utils.three_nn_weights(unknown, known)
            else:
                dist, idx = pointnet2_utils.three_nn(unknown.contiguous(), known.contiguous())
                dist_recip = 1.0 / (dist + 1e-8)  # ref :154
                weight = dist_recip / torch.sum(dist_recip, dim=2, keepdim=True)
            if fused_ok:
                pk = fused.cached_pack(self, 0, self.mlp, known_feats.device)
                cs = 0 if unknow_feats is None else unknow_feats.shape[1]
                if pk is not None and pk.cin == known_feats.shape[1] + cs and pk.cout % 4 == 0:
                    known_pm = known_feats.transpose(1, 2).contiguous()
                    skip_pm = None if unknow_feats is None else unknow_feats.float().transpose(1, 2).contiguous()
                    out_pm = torch.empty((unknown.shape[0], unknown.shape[1], pk.cout), dtype=torch.float32,
                                         device=known_feats.device)
                    ck = known_feats.shape[1]
                    pre = None
                    if getattr(self, 'use_pre', True) and ck >= PRE_MIN_CIN:
                        # interpolation is linear: apply W1[:, known] to the m known points, interpolate after
                        # (channel order of ref :165 is [interpolated(ck), skip(cs)])
                        pre = fused.cached_pre_packs(self, 'pre', [self.mlp], known_feats.device, range(ck),
                                                     range(ck, ck + cs), min_in=1)
                    if pre is not None:
                        prepack, (pk1,) = pre
                        z = torch.empty((known_pm.shape[0], known_pm.shape[1], prepack.width), dtype=torch.float32,
                                        device=known_feats.device)
                        fused.rows_forward(prepack, known_pm, z, relu_last=False)
                        if defer is not None:
                            # the caller runs this module's last step itself (with the point head: fused.fp_head_forward) or
                            # calls materialize(); the returned view is of rows not yet written
                            defer.append(fused.DeferredFP(pk1, z, skip_pm, idx.contiguous(), weight.contiguous(), out_pm))
                        else:
                            fused.fp_forward_pre(pk1, z, skip_pm, idx, weight.contiguous(), out_pm)
                    else:
                        fused.fp_forward(pk, known_pm, skip_pm, idx, weight.contiguous(), out_pm)
                    return out_pm.transpose(1, 2)
            from .. import fused_bn
            rows_path = (self.training and torch.is_grad_enabled() and CHANNELS_LAST_TRAINING and fused_bn.ENABLED and fused_bn.ROWS_GEMM
                         and known_feats.is_cuda and isinstance(self.mlp, TrainSequential)
                         and torch.is_autocast_enabled('cuda') and torch.get_autocast_dtype('cuda') == torch.bfloat16
                         and known.shape[1] <= 16384 and unknown.shape[1] <= 65535)
            if rows_path:
                # interpolation + concat written once as the bf16 rows the MFMA layers read (zero channels up to a multiple
                # of 8); the features arrive and leave as (B, C, n) VIEWS of point-major storage: nothing is transposed
                x = pointnet2_utils.interp_concat_rows(known_feats.transpose(1, 2), None if unknow_feats is None else unknow_feats.transpose(1, 2),
                                                       idx.contiguous(), weight.contiguous())
                return self.mlp(x).squeeze(-1)
            interpolated = pointnet2_utils.three_interpolate(known_feats.contiguous(), idx, weight)
        else:
            interpolated = known_feats.expand(*known_feats.size()[0:2], unknown.size(1))
        x = interpolated if unknow_feats is None else torch.cat([interpolated, unknow_feats], dim=1)
        return self.mlp(x.unsqueeze(-1)).squeeze(-1)
