"""Gaussian heat-map targets for the BEV head, on the device.

gaussian_radius: the three-root rule of /root/reference/pcdet/models/model_utils/centernet_utils.py:9-35.
draw_gaussians: what the reference does box by box on the host (gaussian2D + draw_gaussian_to_heatmap, :38-70, called
from dense_heads/center_head.py:100-160) — a (2r+1)^2 window of exp(-(dx^2+dy^2)/(2 sigma^2)), sigma = (2r+1)/6,
entries below eps * max zeroed, max-merged into the map — as ONE batched scatter-max over all boxes.
"""
import torch


def gaussian_radius(height, width, min_overlap=0.5):
    a1 = 1
    b1 = (height + width)
    c1 = width * height * (1 - min_overlap) / (1 + min_overlap)
    r1 = (b1 + (b1 ** 2 - 4 * a1 * c1).sqrt()) / 2
    a2 = 4
    b2 = 2 * (height + width)
    c2 = (1 - min_overlap) * width * height
    r2 = (b2 + (b2 ** 2 - 4 * a2 * c2).sqrt()) / 2
    a3 = 4 * min_overlap
    b3 = -2 * min_overlap * (height + width)
    c3 = (min_overlap - 1) * width * height
    r3 = (b3 + (b3 ** 2 - 4 * a3 * c3).sqrt()) / 2
    return torch.min(torch.min(r1, r2), r3)


def draw_gaussians(heatmap, cls_idx, centers_int, radius, valid, max_radius=8):
    """heatmap (B, C, H, W) zero-initialised, updated in place; per box: batch index / class channel in `cls_idx`
    (B, M, 2) long, integer centre cell (B, M, 2) = (x, y), integer radius (B, M), valid (B, M) bool.
    The batched window is (2 max_radius + 1)^2 cells: a box of radius <= max_radius gets exactly the reference's window;
    a larger one keeps the reference's sigma = (2 r + 1) / 6 (from the UNCLAMPED radius) and loses only the part of its
    window beyond max_radius cells — pass max_radius >= the largest radius (PDMHeatmapHead derives it from MAX_RADIUS
    in the target config) for exact targets."""
    B, C, H, W = heatmap.shape
    dev = heatmap.device
    k = torch.arange(-max_radius, max_radius + 1, device=dev)
    dy, dx = torch.meshgrid(k, k, indexing="ij")                                     # (K, K)
    r_true = radius.clamp(min=0).float()[..., None, None]                            # (B, M, 1, 1)
    r = r_true.clamp(max=max_radius)
    sigma = (2 * r_true + 1) / 6
    g = torch.exp(-(dx * dx + dy * dy).float() / (2 * sigma * sigma))                # (B, M, K, K)
    inside = (dx.abs() <= r) & (dy.abs() <= r)
    g = torch.where(g < torch.finfo(torch.float32).eps, torch.zeros_like(g), g)       # h[h < eps * h.max()] = 0, max = 1
    x = centers_int[..., 0, None, None] + dx
    y = centers_int[..., 1, None, None] + dy
    ok = inside & valid[..., None, None] & (x >= 0) & (x < W) & (y >= 0) & (y < H)
    flat = ((cls_idx[..., 0, None, None] * C + cls_idx[..., 1, None, None]) * H + y) * W + x
    # entries outside the window / map / padding boxes carry the value 0 (a no-op under max on a non-negative map); they
    # are sent to addresses of their own — all of them on element 0 made the scatter's atomics queue up (1.2 ms)
    spread = torch.arange(flat.numel(), device=dev).view_as(flat) % heatmap.numel()
    flat = torch.where(ok, flat, spread)
    vals = torch.where(ok, g, torch.zeros_like(g))
    heatmap.view(-1).scatter_reduce_(0, flat.reshape(-1), vals.reshape(-1), reduce="amax", include_self=True)
    return heatmap
