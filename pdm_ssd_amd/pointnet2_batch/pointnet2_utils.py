"""Operator API of the PointNet++ batch ops on MI355X.

Mirrors the public names, argument order, shapes, dtypes and zero-fill behaviour of
/root/reference/pcdet/ops/pointnet2/pointnet2_batch/pointnet2_utils.py (cited per class), so the
OpenPCDet module stack can import this file in its place.  Every operator dispatches to a
hand-written HIP kernel in libpdmssd_hip.so; there is no PyTorch or CPU fallback.

Additions over the reference: `QueryAndGroup(fused=True)` (default) runs ball query, both gathers,
the centre subtraction and the concat as one native call writing (B, 3+C, M, ns) once; the
autograd result is identical.  Floating inputs under autocast are computed in fp32 (coordinates
and distances never leave fp32, so indices do not depend on the training dtype).
"""
from typing import Optional, Tuple

import torch
import torch.nn as nn
from torch.autograd import Function

from . import pointnet2_batch_hip as pointnet2


def shared_search_grids(cross_stream=False):
    """`with shared_search_grids():` — the ball queries / three_nn calls inside the block that search the same point
    set share one search grid (pointnet2_batch_hip.GRID_CACHE).  The caller promises not to rewrite those point sets
    inside the block; every grid is dropped when the outermost block ends.  cross_stream: also across streams the caller
    has ordered behind the building one (see _GridCache.scope)."""
    return pointnet2.GRID_CACHE.scope(cross_stream)

_FP32_FWD = dict(device_type="cuda", cast_inputs=torch.float32)


def _new(ref: torch.Tensor, shape, dtype, fill=None) -> torch.Tensor:
    t = torch.empty(shape, dtype=dtype, device=ref.device)
    if fill is not None:
        t.fill_(fill)
    return t


class FarthestPointSampling(Function):
    """ref pointnet2_utils.py:10-33 — xyz (B,N,3) -> int32 (B,npoint); no gradient."""

    @staticmethod
    def forward(ctx, xyz: torch.Tensor, npoint: int) -> torch.Tensor:
        assert xyz.is_contiguous()
        B, N, _ = xyz.size()
        xyz = xyz.float()
        idx = _new(xyz, (B, npoint), torch.int32)
        temp = _new(xyz, (B, N), torch.float32, fill=1e10)  # ref :26
        pointnet2.farthest_point_sampling_wrapper(B, N, npoint, xyz, temp, idx)
        ctx.mark_non_differentiable(idx)
        return idx

    @staticmethod
    def backward(ctx, grad=None):
        return None, None


farthest_point_sample = furthest_point_sample = FarthestPointSampling.apply


def fps_segments(jobs, npoint):
    """Resumable FPS, several batches side by side in ONE launch (pdm_furthest_point_sampling_jobs): jobs = list of
    (xyz (B,N,3), temp (B,N), idx (B,npoint), j0, j1); job q computes samples [j0, j1) of its batch in place, continuing
    from the state an earlier segment left in temp / idx (temp = 1e10 everywhere before the first segment, j0 = 1).
    The segments of one batch, run in order, give exactly farthest_point_sample(xyz, npoint)."""
    import ctypes
    from .. import _native
    assert 1 <= len(jobs) <= 4
    B, N, _ = jobs[0][0].shape
    for xyz, temp, idx, j0, j1 in jobs:
        assert xyz.is_contiguous() and temp.is_contiguous() and idx.is_contiguous()
        assert xyz.shape == (B, N, 3) and temp.shape == (B, N) and idx.shape == (B, npoint)
        assert xyz.dtype == torch.float32 and temp.dtype == torch.float32 and idx.dtype == torch.int32
    n = len(jobs)
    P = ctypes.c_void_p * n
    I = ctypes.c_int * n
    ws_bytes = _native.lib().pdm_furthest_point_sampling_ws_bytes(B, N)   # 0 unless N > 16384 (cooperating workgroups)
    ws = [torch.empty((max(ws_bytes, 8),), dtype=torch.uint8, device=jobs[0][0].device) for _ in jobs] if ws_bytes else []
    _native.call("pdm_furthest_point_sampling_jobs", torch.cuda.current_stream(jobs[0][0].device).cuda_stream, n, B, N,
                 npoint, ctypes.cast(P(*[j[0].data_ptr() for j in jobs]), ctypes.c_void_p),
                 ctypes.cast(P(*[j[1].data_ptr() for j in jobs]), ctypes.c_void_p),
                 ctypes.cast(P(*[j[2].data_ptr() for j in jobs]), ctypes.c_void_p),
                 ctypes.cast(I(*[j[3] for j in jobs]), ctypes.c_void_p), ctypes.cast(I(*[j[4] for j in jobs]), ctypes.c_void_p),
                 ctypes.cast(P(*[w.data_ptr() for w in ws]), ctypes.c_void_p) if ws else None, ws_bytes)
    for w in ws:   # cooperating workgroups wait for each other with bounded spins: a give-up raises (deferred check)
        _native.fps_watch(w, B, N)
    return ws      # during graph capture the caller keeps these and checks them with _native.fps_check_workspace


def topk_sample(scores: torch.Tensor, npoint: int) -> torch.Tensor:
    """Score-ranked sampling (the instance-aware alternative to FPS): scores (B,N) f32 ->
    int32 (B,npoint), the indices of the npoint highest scores in descending order.  Total order (pdm_topk_sampling):
    ties by lower index, -0.0 < +0.0, NaN ranks first as in torch.topk; npoint <= N and <= 16384.  No gradient."""
    from .. import _native
    assert scores.dim() == 2 and scores.is_cuda
    scores = scores.detach().float().contiguous()
    B, N = scores.shape
    idx = torch.empty((B, npoint), dtype=torch.int32, device=scores.device)
    _native.call("pdm_topk_sampling", torch.cuda.current_stream(scores.device).cuda_stream, B, N, int(npoint),
                 scores.data_ptr(), idx.data_ptr())
    return idx


class GatherOperation(Function):
    """ref pointnet2_utils.py:39-70 — features (B,C,N), idx (B,npoint) -> (B,C,npoint)."""

    @staticmethod
    @torch.amp.custom_fwd(**_FP32_FWD)
    def forward(ctx, features: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
        assert features.is_contiguous()
        assert idx.is_contiguous()
        B, npoint = idx.size()
        _, C, N = features.size()
        out = _new(features, (B, C, npoint), torch.float32)
        pointnet2.gather_points_wrapper(B, C, N, npoint, features, idx, out)
        ctx.for_backwards = (idx, C, N)
        return out

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, grad_out):
        idx, C, N = ctx.for_backwards
        B, npoint = idx.size()
        grad_features = _new(grad_out, (B, C, N), torch.float32, fill=0)  # ref :67
        pointnet2.gather_points_grad_wrapper(B, C, N, npoint, grad_out.float().contiguous(), idx,
                                             grad_features)
        return grad_features, None


gather_operation = GatherOperation.apply


class ThreeNN(Function):
    """ref pointnet2_utils.py:76-102 — three nearest known points of every unknown point.

    Returns (dist (B,n,3) = sqrt of the squared distances, idx (B,n,3) int32); no gradient.
    """

    @staticmethod
    def forward(ctx, unknown: torch.Tensor, known: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        assert unknown.is_contiguous()
        assert known.is_contiguous()
        unknown, known = unknown.float(), known.float()
        B, N, _ = unknown.size()
        m = known.size(1)
        dist2 = _new(unknown, (B, N, 3), torch.float32)
        idx = _new(unknown, (B, N, 3), torch.int32)
        pointnet2.three_nn_wrapper(B, N, m, unknown, known, dist2, idx)
        dist = torch.sqrt(dist2)  # ref :98
        ctx.mark_non_differentiable(dist, idx)
        return dist, idx

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None


three_nn = ThreeNN.apply


def three_nn_weights(unknown: torch.Tensor, known: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """(idx (B,n,3) int32, weight (B,n,3)) = three_nn followed by the inverse-distance weighting of
    ref pointnet2_modules.py:154-156 (dist_recip = 1/(dist+1e-8), normalised), in two launches; no gradient."""
    unknown, known = unknown.float().contiguous(), known.float().contiguous()
    B, N, _ = unknown.size()
    dist2 = _new(unknown, (B, N, 3), torch.float32)
    idx = _new(unknown, (B, N, 3), torch.int32)
    pointnet2.three_nn_wrapper(B, N, known.size(1), unknown, known, dist2, idx)
    weight = torch.empty_like(dist2)
    pointnet2.three_nn_weights_wrapper(B * N, dist2, None, weight)
    return idx, weight


class ThreeInterpolate(Function):
    """ref pointnet2_utils.py:108-150 — features (B,C,M), idx/weight (B,n,3) -> (B,C,n)."""

    @staticmethod
    @torch.amp.custom_fwd(**_FP32_FWD)
    def forward(ctx, features: torch.Tensor, idx: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
        assert features.is_contiguous()
        assert idx.is_contiguous()
        assert weight.is_contiguous()
        B, c, m = features.size()
        n = idx.size(1)
        ctx.three_interpolate_for_backward = (idx, weight, m)
        out = _new(features, (B, c, n), torch.float32)
        pointnet2.three_interpolate_wrapper(B, c, m, n, features, idx, weight, out)
        return out

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, grad_out: torch.Tensor):
        idx, weight, m = ctx.three_interpolate_for_backward
        B, c, n = grad_out.size()
        grad_features = _new(grad_out, (B, c, m), torch.float32, fill=0)  # ref :146
        pointnet2.three_interpolate_grad_wrapper(B, c, n, m, grad_out.float().contiguous(), idx, weight,
                                                 grad_features)
        return grad_features, None, None


three_interpolate = ThreeInterpolate.apply


class InterpConcatRows(Function):
    """The FP module's MLP input for the training path, in ONE kernel: cat([three_interpolate(known_feats, idx, weight),
    unknow_feats], dim=1) (ref pointnet2_modules.py:158-165) written as bf16 rows (B, n, ld), ld = C2 + C1 rounded up to a
    multiple of 8 with the extra channels zero — what the bf16 MFMA layers read (fused_bn.rows_linear).  Each element is the
    fp32 value of the reference expression rounded to nearest even, i.e. autocast's cast of the concatenated tensor.
    known_rows (B, m, C2), skip_rows (B, n, C1) | None: point-major rows, fp32 or bf16.  Returns the logical
    (B, ld, n, 1) tensor (a channels-last view of the rows).  Gradients: known_rows (through an inverted index, no atomics),
    skip_rows (a column block of the incoming gradient)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda")
    def forward(ctx, known_rows, skip_rows, idx, weight):
        from .. import _native
        B, m, C2 = known_rows.shape
        n = idx.shape[1]
        C1 = 0 if skip_rows is None else skip_rows.shape[2]
        ld = (C2 + C1 + 7) // 8 * 8
        known_rows = known_rows.contiguous()
        skip = None if skip_rows is None else skip_rows.contiguous()
        assert known_rows.dtype in (torch.float32, torch.bfloat16) and (skip is None or skip.dtype in (torch.float32, torch.bfloat16))
        out = torch.empty((B, n, ld), dtype=torch.bfloat16, device=known_rows.device)
        _native.call("pdm_interp_concat_rows", torch.cuda.current_stream(out.device).cuda_stream, B, n, m, C2, C1, ld,
                     known_rows.data_ptr(), 1 if known_rows.dtype == torch.bfloat16 else 0,
                     0 if skip is None else skip.data_ptr(), 0 if skip is None or skip.dtype != torch.bfloat16 else 1,
                     idx.data_ptr(), weight.data_ptr(), out.data_ptr())
        ctx.geom = (B, n, m, C2, C1, ld, known_rows.dtype, None if skip is None else skip.dtype)
        ctx.save_for_backward(idx, weight)
        return out.view(B, n, 1, ld).permute(0, 3, 1, 2)

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g):
        from .. import _native
        idx, weight = ctx.saved_tensors
        B, n, m, C2, C1, ld, kdtype, sdtype = ctx.geom
        rows = g.permute(0, 2, 3, 1).reshape(B, n, ld)
        if rows.dtype != torch.bfloat16 or not rows.is_contiguous():
            rows = rows.to(torch.bfloat16).contiguous()
        dknown = dskip = None
        if ctx.needs_input_grad[0]:
            ob = kdtype == torch.bfloat16        # written in the known rows' type (the kernel rounds as the cast would)
            dknown = torch.empty((B, m, C2), dtype=torch.bfloat16 if ob else torch.float32, device=g.device)
            nbytes = _native.lib().pdm_three_interpolate_grad_ws_bytes(B, n, m)
            ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=g.device)
            _native.call("pdm_interp_concat_rows_grad_out", torch.cuda.current_stream(g.device).cuda_stream, B, n, m, C2, ld, rows.data_ptr(),
                         idx.data_ptr(), weight.data_ptr(), dknown.data_ptr(), 1 if ob else 0, ws.data_ptr(), nbytes)
            if not ob and kdtype != torch.float32:
                dknown = dknown.to(kdtype)
        if C1 and ctx.needs_input_grad[1]:
            dskip = rows[:, :, C2:C2 + C1]
            if sdtype != torch.bfloat16:
                dskip = dskip.to(sdtype)
        return dknown, dskip, None, None


interp_concat_rows = InterpConcatRows.apply


class GroupingOperation(Function):
    """ref pointnet2_utils.py:156-194 — features (B,C,N), idx (B,npoint,nsample) -> (B,C,npoint,nsample)."""

    @staticmethod
    @torch.amp.custom_fwd(**_FP32_FWD)
    def forward(ctx, features: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
        assert features.is_contiguous()
        assert idx.is_contiguous()
        B, nfeatures, nsample = idx.size()
        _, C, N = features.size()
        out = _new(features, (B, C, nfeatures, nsample), torch.float32)
        pointnet2.group_points_wrapper(B, C, N, nfeatures, nsample, features, idx, out)
        ctx.for_backwards = (idx, N)
        return out

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, grad_out: torch.Tensor):
        idx, N = ctx.for_backwards
        B, C, npoint, nsample = grad_out.size()
        grad_features = _new(grad_out, (B, C, N), torch.float32, fill=0)  # ref :190
        pointnet2.group_points_grad_wrapper(B, C, N, npoint, nsample, grad_out.float().contiguous(), idx,
                                            grad_features)
        return grad_features, None


grouping_operation = GroupingOperation.apply


class BallQuery(Function):
    """ref pointnet2_utils.py:200-225 — (radius, nsample, xyz (B,N,3), new_xyz (B,M,3)) -> int32 (B,M,nsample).

    Rows of empty balls are all zero (the caller-side zero fill of ref :218); no gradient.
    """

    @staticmethod
    def forward(ctx, radius: float, nsample: int, xyz: torch.Tensor, new_xyz: torch.Tensor) -> torch.Tensor:
        assert new_xyz.is_contiguous()
        assert xyz.is_contiguous()
        xyz, new_xyz = xyz.float(), new_xyz.float()
        B, N, _ = xyz.size()
        npoint = new_xyz.size(1)
        # the reference zero-fills idx here (ref :218) because its kernel leaves the rows of empty balls untouched; this
        # library's kernels write those all-zero rows themselves (pointnet2_batch_hip.BALL_QUERY_DEFINES_EVERY_ROW), so
        # the fill pass — a launch per call — is not needed for the same result
        idx = _new(xyz, (B, npoint, nsample), torch.int32, fill=None if pointnet2.BALL_QUERY_DEFINES_EVERY_ROW else 0)
        pointnet2.ball_query_wrapper(B, N, npoint, radius, nsample, new_xyz, xyz, idx)
        ctx.mark_non_differentiable(idx)
        return idx

    @staticmethod
    def backward(ctx, a=None):
        return None, None, None, None


ball_query = BallQuery.apply


class _FusedQueryAndGroup(Function):
    """One native call for ball query + grouped xyz (centred) + grouped features + concat.

    Forward value == cat([grouping(xyz^T, idx) - new_xyz^T[..., None], grouping(features, idx)], 1)
    (ref :249-257).  Gradient flows to `features` only, exactly as in the unfused graph where xyz
    reaches the output through non-differentiated leaf coordinates.
    """

    @staticmethod
    @torch.amp.custom_fwd(**_FP32_FWD)
    def forward(ctx, radius, nsample, xyz, new_xyz, features):
        B, N, _ = xyz.size()
        M = new_xyz.size(1)
        C = 0 if features is None else features.size(1)
        idx = _new(xyz, (B, M, nsample), torch.int32)
        out = _new(xyz, (B, 3 + C, M, nsample), torch.float32)
        pointnet2.query_and_group_wrapper(B, N, M, C, radius, nsample, xyz, new_xyz, features, idx, out)
        ctx.for_backwards = (idx, N, C)
        return out

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, grad_out):
        idx, N, C = ctx.for_backwards
        if C == 0 or not ctx.needs_input_grad[4]:
            return None, None, None, None, None
        B, _, M, ns = grad_out.size()
        g = grad_out[:, 3:].float().contiguous()
        grad_features = _new(grad_out, (B, C, N), torch.float32, fill=0)
        pointnet2.group_points_grad_wrapper(B, C, N, M, ns, g, idx, grad_features)
        return None, None, None, None, grad_features


class _FusedQueryAndGroupCL(Function):
    """_FusedQueryAndGroup with the result in channels-last memory, bf16 when `out_bf16` (autocast): the logical
    (B, 3+C, M, ns) tensor the reference returns, stored as (B, M, ns, ld) — what the rows kernels of the training path
    consume (csrc/train_gemm.hip), without the layout copy and the dtype copy over the largest tensor of the step.  Values:
    the fp32 result, rounded to nearest even when bf16 (= what the autocast cast produces).  Gradient to `features` only.
    pad_to_8: ld = 3 + C rounded up to a multiple of 8 (16-byte bf16 rows) and the returned tensor HAS ld channels, the
    extra ones zero — by contract with fused_bn.rows_linear, whose first layer meets them with zero weight columns."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda")      # (no blanket fp32 cast: bf16 source features are read as they are, see below)
    def forward(ctx, radius, nsample, xyz, new_xyz, features, out_bf16, pad_to_8=False):
        from .. import _native
        xyz, new_xyz = xyz.float(), new_xyz.float()
        B, N, _ = xyz.size()
        M = new_xyz.size(1)
        C = 0 if features is None else features.size(1)
        ld = (3 + C + 7) // 8 * 8 if pad_to_8 else 3 + C
        with torch.autocast("cuda", enabled=False):
            idx = ball_query(radius, nsample, xyz, new_xyz)
        # bf16 features (the pooled output of the level before, under autocast) go into the bf16 result exactly: no fp32 copy
        # of them in front of the kernel (six casts of the levels' feature tensors per training step)
        feat_bf16 = features is not None and features.dtype == torch.bfloat16 and out_bf16 and ld % 8 == 0
        if features is not None and not feat_bf16:
            features = features.float()
        feat_pm = None if features is None else features.transpose(1, 2).contiguous()
        out = torch.empty((B, M, nsample, ld), dtype=torch.bfloat16 if out_bf16 else torch.float32, device=xyz.device)
        _native.call("pdm_group_concat_cl_ld_f", torch.cuda.current_stream(xyz.device).cuda_stream, B, N, M, C, nsample,
                     xyz.data_ptr(), new_xyz.data_ptr(), 0 if feat_pm is None else feat_pm.data_ptr(), 1 if feat_bf16 else 0, idx.data_ptr(),
                     out.data_ptr(), 1 if out_bf16 else 0, ld)
        ctx.for_backwards = (idx, N, C, ld)
        return out.permute(0, 3, 1, 2)

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, grad_out):
        idx, N, C, ld = ctx.for_backwards
        if C == 0 or not ctx.needs_input_grad[4]:
            return None, None, None, None, None, None, None
        from .. import _native
        B, _, M, ns = grad_out.size()
        if N > 16384 or grad_out.dtype not in (torch.float32, torch.bfloat16):
            g = grad_out[:, 3:3 + C].float().contiguous()
            grad_features = _new(grad_out, (B, C, N), torch.float32, fill=0)
            pointnet2.group_points_grad_wrapper(B, C, N, M, ns, g, idx, grad_features)
            return None, None, None, None, grad_features, None, None
        g = grad_out.permute(0, 2, 3, 1)          # (B, M, ns, ld): a view when the gradient arrives channels-last
        if not g.is_contiguous():
            g = g.contiguous()
        grad_pm = torch.empty((B, N, C), dtype=torch.float32, device=grad_out.device)
        nbytes = _native.lib().pdm_group_concat_cl_grad_ws_bytes(B, N, M, ns)
        ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=grad_out.device)
        _native.call("pdm_group_concat_cl_grad_ld", torch.cuda.current_stream(grad_out.device).cuda_stream, B, N, M, C, ns,
                     g.data_ptr(), 1 if g.dtype == torch.bfloat16 else 0, ld, idx.data_ptr(), grad_pm.data_ptr(), ws.data_ptr(), nbytes)
        # (B, C, N) view of point-major storage: the source features of the training path are themselves such views
        return None, None, None, None, grad_pm.transpose(1, 2), None, None


class QueryAndGroup(nn.Module):
    """ref pointnet2_utils.py:231-264."""

    def __init__(self, radius: float, nsample: int, use_xyz: bool = True, fused: bool = True):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz
        self.fused = fused
        self.channels_last = False   # True: the result is stored (B, M, ns, 3+C), bf16 under autocast (training path)
        self.pad_to_8 = False        # with channels_last: the tensor gets round8(3+C) channels, the extra ones zero

    def forward(self, xyz: torch.Tensor, new_xyz: torch.Tensor, features: Optional[torch.Tensor] = None):
        """xyz (B,N,3), new_xyz (B,npoint,3), features (B,C,N) -> (B, 3+C, npoint, nsample)."""
        if features is None:
            assert self.use_xyz, "Cannot have not features and not use xyz as a feature!"
        rows_feat = features is not None and features.dim() == 3 and features.transpose(1, 2).is_contiguous()
        if self.fused and self.use_xyz and xyz.is_contiguous() and new_xyz.is_contiguous() and self.channels_last and rows_feat:
            # features arrive as a (B, C, N) view of point-major storage: exactly what the channels-last kernel reads
            bf16 = torch.is_autocast_enabled('cuda') and torch.get_autocast_dtype('cuda') == torch.bfloat16
            return _FusedQueryAndGroupCL.apply(self.radius, self.nsample, xyz, new_xyz, features, bf16, bool(self.pad_to_8 and bf16))
        if self.fused and self.use_xyz and xyz.is_contiguous() and new_xyz.is_contiguous() and (
                features is None or features.is_contiguous()):
            if self.channels_last:   # set by the SA module on its training path
                bf16 = torch.is_autocast_enabled('cuda') and torch.get_autocast_dtype('cuda') == torch.bfloat16
                return _FusedQueryAndGroupCL.apply(self.radius, self.nsample, xyz, new_xyz, features, bf16, bool(self.pad_to_8 and bf16))
            return _FusedQueryAndGroup.apply(self.radius, self.nsample, xyz, new_xyz, features)
        idx = ball_query(self.radius, self.nsample, xyz, new_xyz)
        grouped_xyz = grouping_operation(xyz.transpose(1, 2).contiguous(), idx)
        grouped_xyz -= new_xyz.transpose(1, 2).unsqueeze(-1)
        if features is None:
            return grouped_xyz
        grouped_features = grouping_operation(features, idx)
        return torch.cat([grouped_xyz, grouped_features], dim=1) if self.use_xyz else grouped_features


class GroupAll(nn.Module):
    """ref pointnet2_utils.py:267-290 — (B, 3+C, 1, N), no centring."""

    def __init__(self, use_xyz: bool = True):
        super().__init__()
        self.use_xyz = use_xyz

    def forward(self, xyz: torch.Tensor, new_xyz: torch.Tensor, features: Optional[torch.Tensor] = None):
        grouped_xyz = xyz.transpose(1, 2).unsqueeze(2)
        if features is None:
            return grouped_xyz
        grouped_features = features.unsqueeze(2)
        return torch.cat([grouped_xyz, grouped_features], dim=1) if self.use_xyz else grouped_features
