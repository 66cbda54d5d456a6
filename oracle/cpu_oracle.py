"""numpy/ctypes front end of the CPU oracle (oracle/pointnet2_oracle.c, oracle/pdm_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under pdm_ssd_amd/ may import this module.

Function names and argument order follow the reference's Python operator API
(/root/reference/pcdet/ops/pointnet2/pointnet2_batch/pointnet2_utils.py) so the parity tests read
like calls into the reference.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libpdmssd_oracle.so")
_lib = None

DIST_PINNED, DIST_NONE, DIST_HIPCC_DEFAULT = 0, 1, 2

_f32p = ctypes.POINTER(ctypes.c_float)
_i32p = ctypes.POINTER(ctypes.c_int)


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    srcs = [os.path.join(_HERE, f) for f in ("pointnet2_oracle.c", "pointnet2_stack_oracle.c", "vector_pool_oracle.c", "iou3d_oracle.c", "pdm_oracle.c", "Makefile")]
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "-B", "libpdmssd_oracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_f32p)


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(_i32p)


def set_dist_mode(mode):
    lib().oracle_set_dist_mode(int(mode))


def set_threads(t):
    lib().oracle_set_threads(int(t))


def max_threads():
    return int(lib().oracle_max_threads())


def opt_n_threads(n):
    return int(lib().oracle_opt_n_threads(int(n)))


def furthest_point_sample(xyz, npoint, block_size=0, return_temp=False):
    """xyz (B,N,3) -> idx (B,npoint) int32.  pointnet2_utils.py:12-29."""
    xyz, px = _f(xyz)
    B, N, _ = xyz.shape
    temp = np.full((B, N), 1e10, dtype=np.float32)
    idx = np.zeros((B, npoint), dtype=np.int32)
    rc = lib().oracle_furthest_point_sampling(B, N, npoint, px, temp.ctypes.data_as(_f32p),
                                              idx.ctypes.data_as(_i32p), int(block_size))
    assert rc == 0
    return (idx, temp) if return_temp else idx


def gather_operation(features, idx):
    """features (B,C,N), idx (B,M) -> (B,C,M).  pointnet2_utils.py:42-60."""
    features, pf = _f(features)
    idx, pi = _i(idx)
    B, C, N = features.shape
    M = idx.shape[1]
    out = np.empty((B, C, M), dtype=np.float32)
    lib().oracle_gather_points(B, C, N, M, pf, pi, out.ctypes.data_as(_f32p))
    return out


def gather_operation_grad(grad_out, idx, N):
    grad_out, pg = _f(grad_out)
    idx, pi = _i(idx)
    B, C, M = grad_out.shape
    g = np.zeros((B, C, N), dtype=np.float32)
    lib().oracle_gather_points_grad(B, C, N, M, pg, pi, g.ctypes.data_as(_f32p))
    return g


def ball_query(radius, nsample, xyz, new_xyz):
    """-> idx (B,M,nsample) int32, zero rows for empty balls.  pointnet2_utils.py:203-221."""
    xyz, px = _f(xyz)
    new_xyz, pn = _f(new_xyz)
    B, N, _ = xyz.shape
    M = new_xyz.shape[1]
    idx = np.zeros((B, M, nsample), dtype=np.int32)
    lib().oracle_ball_query(B, N, M, ctypes.c_float(radius), int(nsample), pn, px,
                            idx.ctypes.data_as(_i32p))
    return idx


def grouping_operation(features, idx):
    """features (B,C,N), idx (B,M,ns) -> (B,C,M,ns).  pointnet2_utils.py:159-177."""
    features, pf = _f(features)
    idx, pi = _i(idx)
    B, C, N = features.shape
    _, M, ns = idx.shape
    out = np.empty((B, C, M, ns), dtype=np.float32)
    lib().oracle_group_points(B, C, N, M, ns, pf, pi, out.ctypes.data_as(_f32p))
    return out


def grouping_operation_grad(grad_out, idx, N):
    grad_out, pg = _f(grad_out)
    idx, pi = _i(idx)
    B, C, M, ns = grad_out.shape
    g = np.zeros((B, C, N), dtype=np.float32)
    lib().oracle_group_points_grad(B, C, N, M, ns, pg, pi, g.ctypes.data_as(_f32p))
    return g


def three_nn(unknown, known):
    """-> (dist (B,n,3) = sqrt(dist2), idx (B,n,3)).  pointnet2_utils.py:79-98."""
    unknown, pu = _f(unknown)
    known, pk = _f(known)
    B, n, _ = unknown.shape
    m = known.shape[1]
    dist2 = np.empty((B, n, 3), dtype=np.float32)
    idx = np.empty((B, n, 3), dtype=np.int32)
    lib().oracle_three_nn(B, n, m, pu, pk, dist2.ctypes.data_as(_f32p), idx.ctypes.data_as(_i32p))
    return np.sqrt(dist2), idx


def three_nn_dist2(unknown, known):
    unknown, pu = _f(unknown)
    known, pk = _f(known)
    B, n, _ = unknown.shape
    m = known.shape[1]
    dist2 = np.empty((B, n, 3), dtype=np.float32)
    idx = np.empty((B, n, 3), dtype=np.int32)
    lib().oracle_three_nn(B, n, m, pu, pk, dist2.ctypes.data_as(_f32p), idx.ctypes.data_as(_i32p))
    return dist2, idx


def three_interpolate(features, idx, weight):
    """features (B,C,m), idx/weight (B,n,3) -> (B,C,n).  pointnet2_utils.py:111-131."""
    features, pf = _f(features)
    idx, pi = _i(idx)
    weight, pw = _f(weight)
    B, C, m = features.shape
    n = idx.shape[1]
    out = np.empty((B, C, n), dtype=np.float32)
    lib().oracle_three_interpolate(B, C, m, n, pf, pi, pw, out.ctypes.data_as(_f32p))
    return out


def three_interpolate_grad(grad_out, idx, weight, m):
    grad_out, pg = _f(grad_out)
    idx, pi = _i(idx)
    weight, pw = _f(weight)
    B, C, n = grad_out.shape
    g = np.zeros((B, C, m), dtype=np.float32)
    lib().oracle_three_interpolate_grad(B, C, n, m, pg, pi, pw, g.ctypes.data_as(_f32p))
    return g


def query_and_group(radius, nsample, xyz, new_xyz, features=None):
    """QueryAndGroup.forward with use_xyz=True (pointnet2_utils.py:241-264) -> (out (B,3+C,M,ns), idx)."""
    xyz, px = _f(xyz)
    new_xyz, pn = _f(new_xyz)
    B, N, _ = xyz.shape
    M = new_xyz.shape[1]
    if features is not None:
        features, pf = _f(features)
        C = features.shape[1]
    else:
        pf, C = None, 0
    idx = np.zeros((B, M, nsample), dtype=np.int32)
    out = np.empty((B, 3 + C, M, nsample), dtype=np.float32)
    lib().oracle_query_and_group(B, N, M, C, ctypes.c_float(radius), int(nsample), px, pn, pf,
                                 idx.ctypes.data_as(_i32p), out.ctypes.data_as(_f32p))
    return out, idx


# ----------------------------------------------------------------------------- PDM (build-defined spec)

def pdm_grid_params(point_cloud_range, cell_size):
    """fp32 origin / cell / inv_cell triple and integer dims (W,H,D) shared by oracle and product."""
    r = np.asarray(point_cloud_range, dtype=np.float64)
    cs = np.asarray(cell_size, dtype=np.float64)
    dims = np.round((r[3:6] - r[0:3]) / cs).astype(np.int64)
    origin = r[0:3].astype(np.float32)
    cell = cs.astype(np.float32)
    inv_cell = (np.float32(1.0) / cell).astype(np.float32)
    return origin, cell, inv_cell, (int(dims[0]), int(dims[1]), int(dims[2]))


def pdm_scatter(xyz, feat, sh, inv2s2, origin, cell, inv_cell, dims, kernel, degree, layout=1):
    """-> (grid, wsum).  grid is (B,H,W,C*D) for layout 1 or (B,C*D,H,W) for layout 0; wsum (B,H,W,D)."""
    xyz, px = _f(xyz)
    feat, pf = _f(feat)
    sh, ps = _f(sh)
    inv2s2, pv = _f(inv2s2)
    origin, po = _f(origin)
    cell, pc = _f(cell)
    inv_cell, pic = _f(inv_cell)
    B, P, _ = xyz.shape
    C = feat.shape[2]
    W, H, D = dims
    shape = (B, H, W, C * D) if layout == 1 else (B, C * D, H, W)
    grid = np.zeros(shape, dtype=np.float32)
    wsum = np.zeros((B, H, W, D), dtype=np.float32)
    rc = lib().oracle_pdm_scatter(B, P, C, int(degree), px, pf, ps, pv, po, pc, pic, W, H, D,
                                  int(kernel[0]), int(kernel[1]), int(kernel[2]), int(layout),
                                  grid.ctypes.data_as(_f32p), wsum.ctypes.data_as(_f32p))
    assert rc == 0, rc
    return grid, wsum


def pdm_normalize(grid, wsum, C, dims, layout=1, eps=1e-6):
    W, H, D = dims
    B = grid.shape[0]
    grid = np.ascontiguousarray(grid, dtype=np.float32).copy()
    wsum, pw = _f(wsum)
    lib().oracle_pdm_normalize(B, C, W, H, D, int(layout), ctypes.c_float(eps),
                               grid.ctypes.data_as(_f32p), pw)
    return grid


def pdm_scatter_grad(xyz, feat, sh, inv2s2, origin, cell, inv_cell, dims, kernel, degree, dgrid,
                     dwsum=None, layout=1):
    xyz, px = _f(xyz)
    feat, pf = _f(feat)
    sh, ps = _f(sh)
    inv2s2, pv = _f(inv2s2)
    origin, po = _f(origin)
    cell, pc = _f(cell)
    inv_cell, pic = _f(inv_cell)
    dgrid, pg = _f(dgrid)
    if dwsum is not None:
        dwsum, pw = _f(dwsum)
    else:
        pw = None
    B, P, _ = xyz.shape
    C = feat.shape[2]
    W, H, D = dims
    dfeat = np.zeros_like(feat)
    dsh = np.zeros_like(sh)
    dinv = np.zeros((B, P), dtype=np.float32)
    rc = lib().oracle_pdm_scatter_grad(B, P, C, int(degree), px, pf, ps, pv, po, pc, pic, W, H, D,
                                       int(kernel[0]), int(kernel[1]), int(kernel[2]), int(layout),
                                       pg, pw, dfeat.ctypes.data_as(_f32p),
                                       dsh.ctypes.data_as(_f32p), dinv.ctypes.data_as(_f32p))
    assert rc == 0, rc
    return dfeat, dsh, dinv


# ---- pointnet2_stack (ragged batches): oracle/pointnet2_stack_oracle.c ---------------------------------------

def stack_ball_query(radius, nsample, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt):
    """pointnet2_stack/pointnet2_utils.py:11-42 -> (idx (M,nsample) int32 local to the sample, empty_ball_mask (M,))."""
    xyz, px = _f(xyz); new_xyz, pn = _f(new_xyz)
    xc, pxc = _i(xyz_batch_cnt); nc, pnc = _i(new_xyz_batch_cnt)
    M = new_xyz.shape[0]
    idx = np.zeros((M, nsample), dtype=np.int32)
    lib().oracle_stack_ball_query(len(xc), M, ctypes.c_float(radius), int(nsample), pn, pnc, px, pxc,
                                  idx.ctypes.data_as(_i32p))
    empty = idx[:, 0] == -1
    idx[empty] = 0
    return idx, empty


def stack_grouping_operation(features, features_batch_cnt, idx, idx_batch_cnt):
    """(N,C), (M,nsample) -> (M,C,nsample).  pointnet2_stack/pointnet2_utils.py:55-86."""
    features, pf = _f(features); idx, pi = _i(idx)
    fc, pfc = _i(features_batch_cnt); ic, pic = _i(idx_batch_cnt)
    M, ns = idx.shape
    C = features.shape[1]
    out = np.empty((M, C, ns), dtype=np.float32)
    lib().oracle_stack_group_points(len(ic), M, C, ns, pf, pfc, pi, pic, out.ctypes.data_as(_f32p))
    return out


def stack_grouping_operation_grad(grad_out, idx, idx_batch_cnt, features_batch_cnt, N):
    grad_out, pg = _f(grad_out); idx, pi = _i(idx)
    fc, pfc = _i(features_batch_cnt); ic, pic = _i(idx_batch_cnt)
    M, C, ns = grad_out.shape
    g = np.zeros((N, C), dtype=np.float32)
    lib().oracle_stack_group_points_grad(len(ic), M, C, N, ns, pg, pi, pic, pfc, g.ctypes.data_as(_f32p))
    return g


def stack_three_nn(unknown, unknown_batch_cnt, known, known_batch_cnt):
    """-> (dist (N,3) = sqrt(dist2), idx (N,3) int32 GLOBAL).  pointnet2_stack/pointnet2_utils.py:230-254."""
    unknown, pu = _f(unknown); known, pk = _f(known)
    uc, puc = _i(unknown_batch_cnt); kc, pkc = _i(known_batch_cnt)
    N = unknown.shape[0]
    d2 = np.zeros((N, 3), dtype=np.float32)
    idx = np.zeros((N, 3), dtype=np.int32)
    lib().oracle_stack_three_nn(len(uc), N, pu, puc, pk, pkc, d2.ctypes.data_as(_f32p), idx.ctypes.data_as(_i32p))
    with np.errstate(invalid="ignore"):
        return np.sqrt(d2), idx


def stack_three_interpolate(features, idx, weight):
    features, pf = _f(features); idx, pi = _i(idx); weight, pw = _f(weight)
    N, C = idx.shape[0], features.shape[1]
    out = np.empty((N, C), dtype=np.float32)
    lib().oracle_stack_three_interpolate(N, C, pf, pi, pw, out.ctypes.data_as(_f32p))
    return out


def stack_three_interpolate_grad(grad_out, idx, weight, M):
    grad_out, pg = _f(grad_out); idx, pi = _i(idx); weight, pw = _f(weight)
    N, C = grad_out.shape
    g = np.zeros((M, C), dtype=np.float32)
    lib().oracle_stack_three_interpolate_grad(N, C, pg, pi, pw, g.ctypes.data_as(_f32p))
    return g


def stack_furthest_point_sample(xyz, xyz_batch_cnt, npoint):
    """-> idx (sum npoint,) int32 GLOBAL.  npoint: int or per-sample list.  pointnet2_stack/pointnet2_utils.py:193-218."""
    xyz, px = _f(xyz)
    xc, pxc = _i(xyz_batch_cnt)
    B = len(xc)
    if np.isscalar(npoint):
        npoint = [int(npoint)] * B
    mc, pmc = _i(npoint)
    temp = np.full((xyz.shape[0],), 1e10, dtype=np.float32)
    out = np.zeros((int(mc.sum()),), dtype=np.int32)
    rc = lib().oracle_stack_furthest_point_sampling(B, px, temp.ctypes.data_as(_f32p), pxc, out.ctypes.data_as(_i32p), pmc)
    assert rc == 0
    return out


# ---- voxel query and vector pool (N3 remainder): oracle/vector_pool_oracle.c ------------------------------------

def stack_voxel_query(max_range, radius, nsample, xyz, new_xyz, new_coords, point_indices):
    """pointnet2_stack/voxel_query_utils.py:10-47 -> (idx (M,nsample) int32 GLOBAL, empty_ball_mask (M,))."""
    xyz, px = _f(xyz); new_xyz, pn = _f(new_xyz)
    new_coords, pc = _i(new_coords); point_indices, pp = _i(point_indices)
    M = new_coords.shape[0]
    _, Z, Y, X = point_indices.shape
    idx = np.zeros((M, nsample), dtype=np.int32)
    zr, yr, xr = max_range
    lib().oracle_stack_voxel_query(M, Z, Y, X, int(nsample), ctypes.c_float(radius), int(zr), int(yr), int(xr), pn, px, pc, pp,
                                   idx.ctypes.data_as(_i32p))
    empty = idx[:, 0] == -1
    idx[empty] = 0
    return idx, empty


def stack_query_local_neighbor_idxs(support_xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, avg_length, max_neighbour_distance,
                                    nsample, neighbor_type):
    """One call of query_stacked_local_neighbor_idxs_wrapper_stack on a stack of avg_length * M slots
    -> (stack (avg_length*M,) int32, start_len (M,2) int32, total)."""
    support_xyz, ps = _f(support_xyz); new_xyz, pn = _f(new_xyz)
    xc, pxc = _i(xyz_batch_cnt); nc, pnc = _i(new_xyz_batch_cnt)
    M = new_xyz.shape[0]
    stack = np.zeros((int(avg_length) * M,), dtype=np.int32)
    start_len = np.zeros((M, 2), dtype=np.int32)
    cumsum = np.zeros((1,), dtype=np.int32)
    total = lib().oracle_stack_query_local_neighbor_idxs(ps, pxc, pn, pnc, stack.ctypes.data_as(_i32p), start_len.ctypes.data_as(_i32p),
                                                         cumsum.ctypes.data_as(_i32p), int(avg_length),
                                                         ctypes.c_float(max_neighbour_distance), len(xc), M, int(nsample),
                                                         int(neighbor_type))
    return stack, start_len, int(total)


def stack_three_nn_by_local_idxs(support_xyz, new_xyz_grid_centers, stack_neighbor_idxs, start_len):
    """One call of query_three_nn_by_stacked_local_idxs_wrapper_stack -> (dist2 (M,G,3), idx (M,G,3) int32, -1 = none)."""
    support_xyz, ps = _f(support_xyz); centers, pc = _f(new_xyz_grid_centers)
    stack, pst = _i(stack_neighbor_idxs); start_len, psl = _i(start_len)
    M, G = centers.shape[0], centers.shape[1]
    d2 = np.zeros(centers.shape, dtype=np.float32)
    idx = np.full(centers.shape, -1, dtype=np.int32)
    lib().oracle_stack_three_nn_by_local_idxs(ps, pc, idx.ctypes.data_as(_i32p), d2.ctypes.data_as(_f32p), pst, psl, M, G)
    return d2, idx


def stack_three_nn_for_vector_pool_by_two_step(support_xyz, xyz_batch_cnt, new_xyz, new_xyz_grid_centers, new_xyz_batch_cnt,
                                               max_neighbour_distance, nsample, neighbor_type, avg_length_of_neighbor_idxs,
                                               num_total_grids, neighbor_distance_multiplier):
    """ThreeNNForVectorPoolByTwoStep.forward (pointnet2_utils.py:306-352), retry loop included
    -> (dist (M,G,3) = sqrt(dist2), idx (M,G,3) int32 GLOBAL or -1, avg_length)."""
    M = np.asarray(new_xyz).shape[0]
    avg = int(avg_length_of_neighbor_idxs)
    while True:
        cap = avg * M
        stack, start_len, total = stack_query_local_neighbor_idxs(
            support_xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, avg,
            float(max_neighbour_distance) * float(neighbor_distance_multiplier), nsample, neighbor_type)
        avg = total // M + int(total % M > 0)
        if total <= cap:
            break
    d2, idx = stack_three_nn_by_local_idxs(support_xyz, new_xyz_grid_centers, stack[:total], start_len)
    assert idx.shape[1] == num_total_grids
    return np.sqrt(d2), idx, avg


def stack_vector_pool_once(support_xyz, xyz_batch_cnt, support_features, new_xyz, new_xyz_batch_cnt, num_grid, max_neighbour_distance,
                           num_c_out, use_xyz, num_max_sum_points, nsample, neighbor_type, pooling_type):
    """One call of vector_pool_wrapper: -> (SUMS new_features (M,c_out), new_local_xyz (M,3G), point_cnt_of_grid (M,G),
    grouped_idxs (num_max_sum_points,3), number of entries wanted)."""
    support_xyz, ps = _f(support_xyz); feats, pf = _f(support_features); new_xyz, pn = _f(new_xyz)
    xc, pxc = _i(xyz_batch_cnt); nc, pnc = _i(new_xyz_batch_cnt)
    gx, gy, gz = (int(v) for v in num_grid)
    G = gx * gy * gz
    M, c_in = new_xyz.shape[0], feats.shape[1]
    nf = np.zeros((M, num_c_out), dtype=np.float32)
    nl = np.zeros((M, 3 * G), dtype=np.float32)
    pc = np.zeros((M, G), dtype=np.int32)
    grouped = np.zeros((int(num_max_sum_points), 3), dtype=np.int32)
    total = lib().oracle_stack_vector_pool(ps, pf, pxc, pn, nf.ctypes.data_as(_f32p), nl.ctypes.data_as(_f32p), pnc,
                                           pc.ctypes.data_as(_i32p), grouped.ctypes.data_as(_i32p), gx, gy, gz,
                                           ctypes.c_float(max_neighbour_distance), len(xc), M, c_in, int(num_c_out),
                                           int(bool(use_xyz)), int(num_max_sum_points), int(nsample), int(neighbor_type),
                                           int(pooling_type))
    return nf, nl, pc, grouped, int(total)


def stack_vector_pool(support_xyz, xyz_batch_cnt, support_features, new_xyz, new_xyz_batch_cnt, num_grid, max_neighbour_distance,
                      num_c_out_each_grid, use_xyz, num_mean_points_per_grid=100, nsample=-1, neighbor_type=0, pooling_type=0):
    """VectorPoolWithVoxelQuery.forward (pointnet2_utils.py:360-428), retry loop included
    -> dict(new_features (M, G*ceg), new_local_xyz (M, 3G), num_mean_points_per_grid, point_cnt_of_grid (M,G), grouped_idxs (T,3))."""
    G = int(num_grid[0]) * int(num_grid[1]) * int(num_grid[2])
    c_out = num_c_out_each_grid * G
    M = np.asarray(new_xyz).shape[0]
    mean = int(num_mean_points_per_grid)
    while True:
        cap = mean * M
        nf, nl, pc, grouped, total = stack_vector_pool_once(support_xyz, xyz_batch_cnt, support_features, new_xyz, new_xyz_batch_cnt,
                                                            num_grid, max_neighbour_distance, c_out, use_xyz, cap, nsample,
                                                            neighbor_type, pooling_type)
        mean = total // M + int(total % M > 0)
        if total <= cap:
            break
    norm = np.maximum(pc[:, :, None].astype(np.float32), np.float32(1e-6))
    nf = (nf.reshape(M, G, num_c_out_each_grid) / norm).reshape(M, c_out)
    if use_xyz:
        nl = (nl.reshape(M, G, 3) / norm).reshape(M, 3 * G)
    return {'new_features': nf, 'new_local_xyz': nl, 'num_mean_points_per_grid': mean, 'point_cnt_of_grid': pc,
            'grouped_idxs': grouped[:total]}


def stack_vector_pool_grad(grad_new_features, point_cnt_of_grid, grouped_idxs, N, num_c_in):
    g, pg = _f(grad_new_features); pc, ppc = _i(point_cnt_of_grid); gi, pgi = _i(grouped_idxs)
    M, c_out = g.shape
    out = np.zeros((N, num_c_in), dtype=np.float32)
    lib().oracle_stack_vector_pool_grad(pg, ppc, pgi, out.ctypes.data_as(_f32p), int(N), M, c_out, int(num_c_in), pc.shape[1],
                                        gi.shape[0])
    return out


# ---- input path (N1): the build-defined draw of pdm_sample_points, in numpy ------------------------------------

def _fmix32(h):
    h = np.asarray(h, dtype=np.uint32).copy()
    with np.errstate(over="ignore"):
        h ^= h >> np.uint32(16); h *= np.uint32(0x85ebca6b); h ^= h >> np.uint32(13); h *= np.uint32(0xc2b2ae35)
        h ^= h >> np.uint32(16)
    return h


def _ip_key(seed, cloud, stream, i):
    with np.errstate(over="ignore"):
        base = _fmix32(np.uint32(seed) ^ (np.uint32(cloud) * np.uint32(0x9E3779B1)) ^ (np.uint32(stream) * np.uint32(0x7F4A7C15)))
        return _fmix32(base + np.asarray(i, dtype=np.uint32) * np.uint32(0x9E3779B9))


def sample_points_choice(points, num_points, seed, cloud):
    """Raw-row index per output row for ONE cloud (points (N,C)), following data_processor.py:189-210 with the draw
    defined in pdm_ssd_amd/csrc/input_path.hip: k random members = the k smallest (key_1, i); shuffle = ascending
    (key_2 or key_3 for the extra copy, i, copy)."""
    pts = np.ascontiguousarray(points, dtype=np.float32)
    N, P = pts.shape[0], int(num_points)
    x, y, z = pts[:, 0], pts[:, 1], pts[:, 2]
    depth = np.sqrt(((x * x).astype(np.float32) + (y * y).astype(np.float32)).astype(np.float32) + (z * z).astype(np.float32))
    near = depth.astype(np.float32) < np.float32(40.0)
    idx = np.arange(N, dtype=np.int64)
    k1 = _ip_key(seed, cloud, 1, idx)

    def draw(members, k):
        order = np.lexsort((members, k1[members]))          # by key, then index
        return members[order[:k]]

    if P < N:
        far = idx[~near]
        chosen = np.concatenate([far, draw(idx[near], P - len(far))]) if P > len(far) else draw(idx, P)
        copy = np.zeros(len(chosen), dtype=np.int64)
    else:
        extra = draw(idx, P - N)
        chosen = np.concatenate([idx, extra])
        copy = np.concatenate([np.zeros(N, dtype=np.int64), np.ones(len(extra), dtype=np.int64)])
    k2 = np.where(copy == 1, _ip_key(seed, cloud, 3, chosen), _ip_key(seed, cloud, 2, chosen))
    order = np.lexsort((copy, chosen, k2))
    return chosen[order].astype(np.int32)


def sample_points_batch(clouds, num_points, seed):
    """list of (N_i, C) arrays -> (points (B*num_points, 1+C) rows [cloud, x, y, z, ...], choice (B*num_points,))."""
    rows, choices = [], []
    for b, c in enumerate(clouds):
        ch = sample_points_choice(c, num_points, seed, b)
        sel = np.asarray(c, dtype=np.float32)[ch]
        rows.append(np.concatenate([np.full((len(ch), 1), b, dtype=np.float32), sel], axis=1))
        choices.append(ch)
    return np.concatenate(rows), np.concatenate(choices)


# ---- rotated-box IoU / NMS (N2): oracle/iou3d_oracle.c ---------------------------------------------------------

def boxes_overlap_bev(boxes_a, boxes_b, iou=False):
    a, pa = _f(boxes_a); b, pb = _f(boxes_b)
    out = np.zeros((a.shape[0], b.shape[0]), dtype=np.float32)
    lib().oracle_boxes_pairwise_bev(1 if iou else 0, a.shape[0], pa, b.shape[0], pb, out.ctypes.data_as(_f32p))
    return out


def boxes_iou_bev(boxes_a, boxes_b):
    return boxes_overlap_bev(boxes_a, boxes_b, iou=True)


def boxes_aligned_overlap_bev(boxes_a, boxes_b):
    a, pa = _f(boxes_a); b, pb = _f(boxes_b)
    out = np.zeros((a.shape[0],), dtype=np.float32)
    lib().oracle_boxes_aligned_overlap_bev(a.shape[0], pa, pb, out.ctypes.data_as(_f32p))
    return out


def nms(sorted_boxes, thresh, normal=False):
    """Greedy NMS over boxes already in descending score order -> kept positions (int64)."""
    b, pb = _f(sorted_boxes)
    keep = np.zeros((b.shape[0],), dtype=np.int64)
    k = lib().oracle_nms(1 if normal else 0, b.shape[0], pb, ctypes.c_float(thresh), keep.ctypes.data_as(ctypes.POINTER(ctypes.c_longlong)))
    return keep[:k]


def points_in_boxes(points, boxes):
    """points (B,M,3), boxes (B,T,7) -> (B,M) int32 first containing box or -1 (roiaware_pool3d_kernel.cu:313-336)."""
    pts, pp = _f(points); bx, pb = _f(boxes)
    B, M, _ = pts.shape
    out = np.zeros((B, M), dtype=np.int32)
    lib().oracle_points_in_boxes(B, bx.shape[1], M, pb, pp, out.ctypes.data_as(_i32p))
    return out


def topk_key(scores):
    """Sort key of pdm_topk_sampling (csrc/topk_sampling.hip: smaller key = higher rank): order-preserving integer image
    of the float, inverted; NaN of either sign ranks first."""
    bits = np.ascontiguousarray(scores, dtype=np.float32).view(np.uint32)
    mono = np.where(bits & np.uint32(0x80000000), ~bits, bits | np.uint32(0x80000000))
    key = ~mono
    return np.where((bits & np.uint32(0x7fffffff)) > np.uint32(0x7f800000), np.uint32(0), key).astype(np.uint32)


def topk_sampling(scores, k):
    """scores (B,N) -> int32 (B,k): indices of the k highest scores, descending; ties by lower index (stable sort)."""
    scores = np.asarray(scores, dtype=np.float32)
    key = topk_key(scores)
    return np.argsort(key, axis=1, kind='stable')[:, :k].astype(np.int32)
