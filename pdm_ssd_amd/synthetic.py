"""Synthetic KITTI-range point clouds (SURVEY.md section 8 D1 / BASELINE.md 2.2).

uniform:    x~U[0,70.4), y~U[-40,40), z~U[-3,1), intensity~U[0,1), rng = default_rng(1234 + sample_index)
lidar_like: 64 elevation rings, range ~ Exp clipped to 70 m, ground plane near z = -1.7 m — dense near
            the sensor, sparse far away, so small-radius balls are not all empty as in `uniform`.
Both return float32 arrays of shape (B, N, 4) = [x, y, z, intensity].
"""
import numpy as np

KITTI_RANGE = (0.0, -40.0, -3.0, 70.4, 40.0, 1.0)


def uniform_clouds(B, N, seed0=1234):
    out = np.empty((B, N, 4), dtype=np.float32)
    for s in range(B):
        rng = np.random.default_rng(seed0 + s)
        out[s, :, 0] = rng.uniform(0.0, 70.4, N)
        out[s, :, 1] = rng.uniform(-40.0, 40.0, N)
        out[s, :, 2] = rng.uniform(-3.0, 1.0, N)
        out[s, :, 3] = rng.uniform(0.0, 1.0, N)
    return out


def lidar_like_clouds(B, N, seed0=1234):
    out = np.empty((B, N, 4), dtype=np.float32)
    for s in range(B):
        rng = np.random.default_rng(seed0 + s)
        pts = np.empty((0, 3))
        while pts.shape[0] < N:
            k = 4 * N
            ring = rng.integers(0, 64, k)
            elev = np.deg2rad(-24.8 + ring * (26.8 / 63.0))       # HDL-64E: +2 .. -24.8 degrees
            azim = rng.uniform(-np.pi / 2, np.pi / 2, k)           # forward half (KITTI camera FOV crop)
            rho = np.minimum(rng.exponential(18.0, k) + 2.0, 70.0)
            # rays below the horizon stop at the ground plane (sensor 1.73 m above ground)
            ground = np.where(elev < -0.01, 1.73 / np.maximum(-np.sin(elev), 1e-3), np.inf)
            rho = np.minimum(rho, ground)
            x = rho * np.cos(elev) * np.cos(azim)
            y = rho * np.cos(elev) * np.sin(azim)
            z = rho * np.sin(elev) + rng.normal(0.0, 0.02, k)
            p = np.stack([x, y, z], 1)
            ok = (p[:, 0] >= 0) & (p[:, 0] < 70.4) & (p[:, 1] >= -40) & (p[:, 1] < 40) & (p[:, 2] >= -3) & (p[:, 2] < 1)
            pts = np.concatenate([pts, p[ok]], 0)
        out[s, :, :3] = pts[:N].astype(np.float32)
        out[s, :, 3] = rng.uniform(0.0, 1.0, N).astype(np.float32)
    return out


def to_batch_points(clouds):
    """(B,N,4) -> OpenPCDet `points` (B*N, 5) = [batch_idx, x, y, z, intensity] (dataset.py:241-244)."""
    B, N, _ = clouds.shape
    bidx = np.repeat(np.arange(B, dtype=np.float32), N)[:, None]
    return np.concatenate([bidx, clouds.reshape(B * N, 4)], 1).astype(np.float32)
