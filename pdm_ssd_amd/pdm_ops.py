"""Operator layer of the PDM neck: autograd wrapper over pdm_scatter_bev / pdm_scatter_bev_grad.

No reference counterpart exists; the arithmetic is the written spec in DESIGN.md
("PDM spec") whose normative CPU statement is oracle/pdm_oracle.c.
"""
from typing import Sequence, Tuple

import numpy as np
import torch
from torch.autograd import Function

from . import _native


class BevGrid:
    """fp32 grid parameters shared bit-for-bit by host, kernels and oracle."""

    def __init__(self, point_cloud_range: Sequence[float], cell_size: Sequence[float]):
        r = np.asarray(point_cloud_range, dtype=np.float64)
        cs = np.asarray(cell_size, dtype=np.float64)
        dims = np.round((r[3:6] - r[0:3]) / cs).astype(np.int64)
        self.origin = r[0:3].astype(np.float32)
        self.cell = cs.astype(np.float32)
        self.inv_cell = (np.float32(1.0) / self.cell).astype(np.float32)
        self.W, self.H, self.D = int(dims[0]), int(dims[1]), int(dims[2])

    def floats(self):
        return [float(v) for v in (*self.origin, *self.cell, *self.inv_cell)]


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def _chk(name, t, shape=None):
    if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise ValueError(f"{name} must be a contiguous fp32 GPU tensor")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError(f"{name} has shape {tuple(t.shape)}, expected {tuple(shape)}")


class PDMScatter(Function):
    """(xyz (B,P,3), feat (B,P,C), sh (B,P,(L+1)^2), inv2s2 (B,P)) -> (grid, wsum).

    layout 1: grid (B,H,W,C*D) — channels-last storage; `.permute(0,3,1,2)` is the (B,C*D,H,W) BEV map.
    layout 0: grid (B,C*D,H,W) contiguous.  wsum (B,H,W,D).  Gradients: feat, sh, inv2s2.
    """

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, xyz, feat, sh, inv2s2, grid_spec: BevGrid, kernel: Tuple[int, int, int], degree: int,
                layout: int = 1):
        B, P, _ = xyz.shape
        C = feat.shape[2]
        nsh = (degree + 1) ** 2
        _chk("xyz", xyz, (B, P, 3)); _chk("feat", feat, (B, P, C)); _chk("sh", sh, (B, P, nsh))
        _chk("inv2s2", inv2s2, (B, P))
        g = grid_spec
        shape = (B, g.H, g.W, C * g.D) if layout == 1 else (B, C * g.D, g.H, g.W)
        grid = torch.zeros(shape, dtype=torch.float32, device=xyz.device)
        wsum = torch.zeros((B, g.H, g.W, g.D), dtype=torch.float32, device=xyz.device)
        _native.call("pdm_scatter_bev", _stream(xyz), B, P, C, degree, xyz.data_ptr(), feat.data_ptr(),
                     sh.data_ptr(), inv2s2.data_ptr(), *g.floats(), g.W, g.H, g.D, *kernel, layout,
                     grid.data_ptr(), wsum.data_ptr())
        ctx.save_for_backward(xyz, feat, sh, inv2s2)
        ctx.spec = (g, tuple(kernel), degree, layout)
        return grid, wsum

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, dgrid, dwsum):
        xyz, feat, sh, inv2s2 = ctx.saved_tensors
        g, kernel, degree, layout = ctx.spec
        B, P, _ = xyz.shape
        C = feat.shape[2]
        dgrid = dgrid.float().contiguous()
        dwsum_ptr = 0
        if dwsum is not None:
            dwsum = dwsum.float().contiguous()
            dwsum_ptr = dwsum.data_ptr()
        dfeat = torch.empty_like(feat)
        dsh = torch.empty_like(sh)
        dinv = torch.empty_like(inv2s2)
        _native.call("pdm_scatter_bev_grad", _stream(xyz), B, P, C, degree, xyz.data_ptr(), feat.data_ptr(),
                     sh.data_ptr(), inv2s2.data_ptr(), *g.floats(), g.W, g.H, g.D, *kernel, layout,
                     dgrid.data_ptr(), dwsum_ptr, dfeat.data_ptr(), dsh.data_ptr(), dinv.data_ptr())
        return None, dfeat, dsh, dinv, None, None, None, None


def pdm_scatter(xyz, feat, sh, inv2s2, grid_spec, kernel, degree, layout=1):
    return PDMScatter.apply(xyz, feat, sh, inv2s2, grid_spec, tuple(kernel), degree, layout)


def bev_normalize_(grid, wsum, C, grid_spec, layout=1, eps=1e-6):
    """In-place grid /= wsum where |wsum| > eps (inference path; use torch ops when grads are needed)."""
    g = grid_spec
    _chk("grid", grid); _chk("wsum", wsum)
    B = grid.shape[0]
    _native.call("pdm_bev_normalize", _stream(grid), B, C, g.W, g.H, g.D, layout, float(eps), grid.data_ptr(),
                 wsum.data_ptr())
    return grid


class BevNormalize(Function):
    """grid / wsum where |wsum| > eps (channels-last layout), IN PLACE on `grid`, differentiable w.r.t. both:
    one kernel forward, one kernel backward (the torch expression costs six passes over the 0.6 GB grid)."""

    @staticmethod
    def forward(ctx, grid, wsum, C, grid_spec, eps):
        g = grid_spec
        grid = grid.contiguous()
        _native.call("pdm_bev_normalize", _stream(grid), grid.shape[0], C, g.W, g.H, g.D, 1, float(eps), grid.data_ptr(),
                     wsum.data_ptr())
        ctx.mark_dirty(grid)
        ctx.save_for_backward(grid, wsum)
        ctx.spec = (C, g, float(eps))
        return grid

    @staticmethod
    def backward(ctx, dy):
        y, wsum = ctx.saved_tensors
        C, g, eps = ctx.spec
        dy = dy.float().contiguous()
        dx = torch.empty_like(y)
        dw = torch.empty_like(wsum)
        _native.call("pdm_bev_normalize_grad", _stream(y), y.shape[0], C, g.W, g.H, g.D, eps, y.data_ptr(), wsum.data_ptr(),
                     dy.data_ptr(), dx.data_ptr(), dw.data_ptr())
        return dx, dw, None, None, None


def bev_normalize(grid, wsum, C, grid_spec, eps=1e-6):
    """Differentiable normalisation of a channels-last PDM grid (consumes `grid`)."""
    return BevNormalize.apply(grid, wsum, C, grid_spec, eps)


def pdm_gather(xyz, feat, sh, inv2s2, grid_spec, kernel, degree, normalize=True, eps=1e-6):
    """Inference form of pdm_scatter (+ normalise): no autograd, no atomics, bitwise reproducible.
    Returns (grid (B,H,W,C*D) channels-last storage, wsum (B,H,W,D)); every cell is written."""
    B, P, _ = xyz.shape
    C = feat.shape[2]
    nsh = (degree + 1) ** 2
    xyz, feat, sh, inv2s2 = (t.float().contiguous() for t in (xyz, feat, sh, inv2s2))
    _chk("xyz", xyz, (B, P, 3)); _chk("feat", feat, (B, P, C)); _chk("sh", sh, (B, P, nsh)); _chk("inv2s2", inv2s2, (B, P))
    g = grid_spec
    grid = torch.empty((B, g.H, g.W, C * g.D), dtype=torch.float32, device=xyz.device)
    wsum = torch.empty((B, g.H, g.W, g.D), dtype=torch.float32, device=xyz.device)
    nbytes = _native.lib().pdm_gather_bev_workspace_bytes(B, P, g.W, g.H, kernel[0], kernel[1])
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=xyz.device)
    _native.call("pdm_gather_bev", _stream(xyz), B, P, C, degree, xyz.data_ptr(), feat.data_ptr(), sh.data_ptr(),
                 inv2s2.data_ptr(), *g.floats(), g.W, g.H, g.D, *kernel, 1 if normalize else 0, float(eps),
                 grid.data_ptr(), wsum.data_ptr(), ws.data_ptr(), nbytes)
    return grid, wsum


class PDMGatherNormalized(Function):
    """Training form of the neck's grid with the GATHER kernel in the forward pass: (xyz, feat, sh, inv2s2) -> (grid / wsum, wsum),
    channels-last.  Same sum as PDMScatter + BevNormalize (ascending point order per cell instead of the atomics' arrival order),
    but the grid is written once — no 0.6 GB zero fill, no float atomics, no separate normalisation pass (1.0 -> 0.2 ms at
    bs = 32).  Backward: pdm_bev_normalize_grad for dL/dwsum only, then pdm_scatter_bev_grad_normalized, which divides by
    wsum on its own reads of the incoming gradient — no dL/dgrid tensor (0.58 GB at bs = 32) is written or read."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, xyz, feat, sh, inv2s2, grid_spec: BevGrid, kernel: Tuple[int, int, int], degree: int, eps: float):
        grid, wsum = pdm_gather(xyz, feat, sh, inv2s2, grid_spec, kernel, degree, normalize=True, eps=eps)
        ctx.save_for_backward(xyz.contiguous(), feat.contiguous(), sh.contiguous(), inv2s2.contiguous(), grid, wsum)
        ctx.spec = (grid_spec, tuple(kernel), degree, float(eps))
        ctx.mark_non_differentiable(wsum)
        ctx.set_materialize_grads(False)     # no zero tensor for wsum's gradient
        return grid, wsum

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, dy, _dwsum=None):
        if dy is None:
            return (None,) * 8
        xyz, feat, sh, inv2s2, y, wsum = ctx.saved_tensors
        g, kernel, degree, eps = ctx.spec
        B, P, _ = xyz.shape
        C = feat.shape[2]
        dy = dy.float().contiguous()
        dw = torch.empty_like(wsum)
        s = _stream(y)
        _native.call("pdm_bev_normalize_grad", s, B, C, g.W, g.H, g.D, eps, y.data_ptr(), wsum.data_ptr(), dy.data_ptr(),
                     0, dw.data_ptr())
        dfeat = torch.empty_like(feat)
        dsh = torch.empty_like(sh)
        dinv = torch.empty_like(inv2s2)
        _native.call("pdm_scatter_bev_grad_normalized", s, B, P, C, degree, xyz.data_ptr(), feat.data_ptr(), sh.data_ptr(),
                     inv2s2.data_ptr(), *g.floats(), g.W, g.H, g.D, *kernel, 1, dy.data_ptr(), wsum.data_ptr(), eps, dw.data_ptr(),
                     dfeat.data_ptr(), dsh.data_ptr(), dinv.data_ptr())
        return None, dfeat, dsh, dinv, None, None, None, None


def pdm_gather_normalized(xyz, feat, sh, inv2s2, grid_spec, kernel, degree, eps=1e-6):
    return PDMGatherNormalized.apply(xyz, feat, sh, inv2s2, grid_spec, tuple(kernel), degree, eps)


def gather_supported(C, D):
    """D == 1, C <= 256: register accumulators.  Otherwise the LDS accumulator tile (64*D cells x C channels)
    must fit 64 KB."""
    if D == 1 and C <= 256:
        return True
    ncell = 64 * D
    return (ncell * C + ncell + ncell * 16 + 16 * C + 16 * 20) * 4 + 16 * 12 <= 64 * 1024       # PG_CHUNK = 16 points staged per pass
