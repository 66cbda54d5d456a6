"""Known answers for oracle/vector_pool_oracle.c (voxel query + vector-pool family).

The reference holds no tests or vectors for these operators, so the oracle is pinned here by answers worked out by
hand from the kernels' text (voxel_query_gpu.cu, vector_pool_gpu.cu) and by agreement with the ball-query and
three-nn oracles where the semantics coincide.  CPU only.
"""
import numpy as np

from oracle import cpu_oracle as o


def test_voxel_query_known_answers():
    # one sample, volume Z,Y,X = 1,3,3; voxel (0,y,x) holds point 3*y + x, except voxel (0,1,1) which is empty (-1)
    pts = np.array([[x + 0.5, y + 0.5, 0.5] for y in range(3) for x in range(3)], np.float32)
    vox = np.arange(9, dtype=np.int32).reshape(1, 1, 3, 3)
    vox[0, 0, 1, 1] = -1
    centre = np.array([[1.5, 1.5, 0.5], [0.5, 0.5, 0.5], [40.0, 40.0, 0.5]], np.float32)
    coords = np.array([[0, 0, 1, 1], [0, 0, 0, 0], [0, 0, 2, 2]], np.int32)
    # radius 1.0 keeps the 4-neighbourhood (d2 = 1 <= 1), drops the diagonals (d2 = 2); window order y outer, x inner
    idx, empty = o.stack_voxel_query((0, 1, 1), 1.0, 6, pts, centre, coords, vox)
    np.testing.assert_array_equal(idx[0], [1, 3, 5, 7, 1, 1])           # four hits, then the FIRST hit as padding
    np.testing.assert_array_equal(idx[1], [0, 1, 3, 0, 0, 0])           # corner: the window is clipped to the volume
    assert empty.tolist() == [False, False, True] and (idx[2] == 0).all()
    # nsample smaller than the hits: the first nsample in window order
    idx, _ = o.stack_voxel_query((0, 1, 1), 1.0, 2, pts, centre, coords, vox)
    np.testing.assert_array_equal(idx[0], [1, 3])
    # x_range 0: only the centre's own column of voxels
    idx, _ = o.stack_voxel_query((0, 1, 0), 1.0, 3, pts, centre, coords, vox)
    np.testing.assert_array_equal(idx[0], [1, 7, 1])


def test_voxel_query_with_a_full_window_is_the_ball_query():
    """One point per voxel, numbered in voxel order: a window over the whole volume visits the points in index order,
    which is the stacked ball query (global indices here, no ties at the radius)."""
    rng = np.random.default_rng(0)
    Z, Y, X = 2, 5, 6
    B = 2
    grid = np.stack(np.meshgrid(np.arange(Z), np.arange(Y), np.arange(X), indexing="ij"), -1).reshape(-1, 3)
    pts, vox = [], np.full((B, Z, Y, X), -1, np.int32)
    for b in range(B):
        keep = rng.random(len(grid)) < 0.8
        for z, y, x in grid[keep]:
            vox[b, z, y, x] = len(pts)
            pts.append([x + rng.random(), y + rng.random(), z + rng.random()])
    pts = np.array(pts, np.float32)
    counts = [int((vox[b] >= 0).sum()) for b in range(B)]
    new, coords, mc = [], [], []
    for b in range(B):
        sel = rng.choice(np.flatnonzero(vox[b].ravel() >= 0), 7, replace=False)
        for s in sel:
            z, y, x = np.unravel_index(s, (Z, Y, X))
            new.append(pts[vox[b, z, y, x]] + 0.1)
            coords.append([b, z, y, x])
        mc.append(7)
    new, coords = np.array(new, np.float32), np.array(coords, np.int32)
    idx, empty = o.stack_voxel_query((Z, Y, X), 1.7, 8, pts, new, coords, vox)
    ridx, rempty = o.stack_ball_query(1.7, 8, pts, counts, new, mc)
    starts = np.concatenate([[0], np.cumsum(counts)])[:-1]
    ridx = ridx + np.repeat(starts, 7)[:, None].astype(np.int32)
    ridx[rempty] = 0
    np.testing.assert_array_equal(idx, ridx)
    np.testing.assert_array_equal(empty, rempty)


def test_local_neighbor_lists_known_answers():
    # sample 0: 5 points on the x axis at 0,1,2,3,4; sample 1: 3 points at 10,11,12 (global indices 5,6,7)
    xyz = np.array([[0, 0, 0], [1, 0, 0], [2, 0, 0], [3, 0, 0], [4, 0, 0], [10, 1, 1], [11, 0, 0], [12, 0, 0]], np.float32)
    new = np.array([[2, 0, 0], [0, 0, 0], [11, 0, 0]], np.float32)
    stack, sl, total = o.stack_query_local_neighbor_idxs(xyz, [5, 3], new, [2, 1], 4, 1.0, -1, 1)
    # ball, distance 1 (d2 <= r2 keeps the boundary): {1,2,3}, {0,1}, {6,7} — point 5 is at d2 = 3
    assert total == 7
    np.testing.assert_array_equal(sl, [[0, 3], [3, 2], [5, 2]])
    np.testing.assert_array_equal(stack[:7], [1, 2, 3, 0, 1, 6, 7])
    # cube of half-width 1 also takes point 5 (|l| <= 1 on every axis)
    stack, sl, total = o.stack_query_local_neighbor_idxs(xyz, [5, 3], new, [2, 1], 4, 1.0, -1, 0)
    np.testing.assert_array_equal(sl, [[0, 3], [3, 2], [5, 3]])
    np.testing.assert_array_equal(stack[:8], [1, 2, 3, 0, 1, 5, 6, 7])
    # nsample 2: the first two by index
    stack, sl, total = o.stack_query_local_neighbor_idxs(xyz, [5, 3], new, [2, 1], 4, 1.0, 2, 0)
    np.testing.assert_array_equal(sl, [[0, 2], [2, 2], [4, 2]])
    np.testing.assert_array_equal(stack[:6], [1, 2, 0, 1, 5, 6])
    # a stack of 1 * 3 slots: the total still counts everything, writes stop at the capacity (centre 1 is cut, 2 dropped)
    stack, sl, total = o.stack_query_local_neighbor_idxs(xyz, [5, 3], new, [2, 1], 1, 1.0, 2, 0)
    assert total == 6 and stack.shape == (3,)
    np.testing.assert_array_equal(stack, [1, 2, 0])


def test_local_neighbor_lists_stop_at_1000():
    xyz = np.zeros((1500, 3), np.float32)
    xyz[:, 0] = np.arange(1500) * 1e-4
    _, sl, total = o.stack_query_local_neighbor_idxs(xyz, [1500], xyz[:2], [2], 1000, 5.0, -1, 1)
    assert total == 2000 and sl[:, 1].tolist() == [1000, 1000]


def test_three_nn_by_two_step_matches_the_stack_three_nn():
    """With a radius that takes in the whole sample the stacked list is the sample itself in index order, so the
    three-nn over it is the plain stacked three_nn of the cell centres."""
    rng = np.random.default_rng(3)
    counts, mc, G = [300, 200], [11, 6], 8
    xyz = rng.uniform(-2, 2, (500, 3)).astype(np.float32)
    new = rng.uniform(-2, 2, (17, 3)).astype(np.float32)
    centres = (new[:, None, :] + rng.uniform(-0.5, 0.5, (17, G, 3))).astype(np.float32)
    dist, idx, avg = o.stack_three_nn_for_vector_pool_by_two_step(xyz, counts, new, centres, mc, 5.0, -1, 1, 7, G, 2.0)
    rdist, ridx = o.stack_three_nn(centres.reshape(-1, 3), [11 * G, 6 * G], xyz, counts)
    np.testing.assert_array_equal(idx.reshape(-1, 3), ridx)
    np.testing.assert_array_equal(dist.reshape(-1, 3), rdist)
    assert avg == int(np.ceil((11 * 300 + 6 * 200) / 17))
    # the starting guess for the stack size does not change the answer
    d2, i2, a2 = o.stack_three_nn_for_vector_pool_by_two_step(xyz, counts, new, centres, mc, 5.0, -1, 1, 1000, G, 2.0)
    np.testing.assert_array_equal(i2, idx)
    assert a2 == avg


def test_three_nn_by_two_step_short_and_empty_lists():
    xyz = np.array([[0, 0, 0], [1, 0, 0], [50, 0, 0]], np.float32)
    new = np.array([[0, 0, 0], [50, 0, 0], [100, 0, 0]], np.float32)
    centres = new[:, None, :].copy()
    dist, idx, _ = o.stack_three_nn_for_vector_pool_by_two_step(xyz, [3], new, centres, [3], 1.0, -1, 1, 2, 1, 1.5)
    np.testing.assert_array_equal(idx[0, 0], [0, 1, 0])        # two neighbours: the best repeats in the third slot
    np.testing.assert_array_equal(idx[1, 0], [2, 2, 2])        # one neighbour: it fills all three
    np.testing.assert_array_equal(idx[2, 0], [-1, -1, -1])     # none
    assert np.isinf(dist[2, 0]).all() and dist[0, 0].tolist() == [0.0, 1.0, 0.0]


def test_vector_pool_known_answers():
    # one centre at the origin, cube of half-width 1 split 2 x 1 x 1 along x: cell 0 = x in [-1,0), cell 1 = x in [0,1]
    xyz = np.array([[-0.5, 0, 0], [0.5, 0, 0], [0.25, 0.5, 0], [3, 0, 0], [-1.0, 0, 0], [1.0, 0, 0]], np.float32)
    feat = np.arange(24, dtype=np.float32).reshape(6, 4)       # 4 input channels folded onto 2 per cell
    new = np.zeros((1, 3), np.float32)
    r = o.stack_vector_pool(xyz, [6], feat, new, [1], (2, 1, 1), 1.0, 2, True)
    # cell 0 <- points 0, 4; cell 1 <- points 1, 2, 5 (x = 1.0: floor(2/1) = 2 -> linear index clamped to the last cell)
    np.testing.assert_array_equal(r['point_cnt_of_grid'], [[2, 3]])
    fold = feat[:, :2] + feat[:, 2:]
    want = np.concatenate([(fold[0] + fold[4]) / 2, (fold[1] + fold[2] + fold[5]) / 3])
    np.testing.assert_allclose(r['new_features'][0], want, rtol=0, atol=1e-6)
    np.testing.assert_allclose(r['new_local_xyz'][0], [-0.75, 0, 0, (0.5 + 0.25 + 1) / 3, 0.5 / 3, 0], atol=1e-6)
    np.testing.assert_array_equal(r['grouped_idxs'], [[0, 0, 0], [1, 0, 1], [2, 0, 1], [4, 0, 0], [5, 0, 1]])
    assert r['num_mean_points_per_grid'] == 5
    # nsample 3: only the first three neighbours by index are pooled
    r3 = o.stack_vector_pool(xyz, [6], feat, new, [1], (2, 1, 1), 1.0, 2, True, nsample=3)
    np.testing.assert_array_equal(r3['point_cnt_of_grid'], [[1, 2]])
    np.testing.assert_array_equal(r3['grouped_idxs'][:, 0], [0, 1, 2])
    # pooling_type 1: the first point of each cell, and (with '=') the LAST input channel of each residue
    r1 = o.stack_vector_pool(xyz, [6], feat, new, [1], (2, 1, 1), 1.0, 2, True, pooling_type=1)
    np.testing.assert_array_equal(r1['point_cnt_of_grid'], [[1, 1]])
    np.testing.assert_array_equal(r1['new_features'][0], [feat[0, 2], feat[0, 3], feat[1, 2], feat[1, 3]])
    np.testing.assert_array_equal(r1['grouped_idxs'], [[0, 0, 0], [1, 0, 1]])
    # ball instead of cube drops nothing here except by radius: a corner point would go
    rb = o.stack_vector_pool(np.array([[0.8, 0.8, 0]], np.float32), [1], np.ones((1, 2), np.float32), new, [1], (1, 1, 1), 1.0, 2,
                             False, neighbor_type=1)
    assert rb['point_cnt_of_grid'].sum() == 0 and (rb['new_features'] == 0).all()


def test_vector_pool_retry_guess_does_not_change_the_answer_and_grad_is_the_adjoint():
    rng = np.random.default_rng(5)
    counts, mc = [400, 250], [30, 20]
    xyz = rng.uniform(-2, 2, (650, 3)).astype(np.float32)
    feat = rng.normal(size=(650, 8)).astype(np.float32)
    new = rng.uniform(-2, 2, (50, 3)).astype(np.float32)
    a = o.stack_vector_pool(xyz, counts, feat, new, mc, (3, 3, 3), 1.2, 4, True, num_mean_points_per_grid=1)
    b = o.stack_vector_pool(xyz, counts, feat, new, mc, (3, 3, 3), 1.2, 4, True, num_mean_points_per_grid=500)
    for k in ('new_features', 'new_local_xyz', 'point_cnt_of_grid', 'grouped_idxs'):
        np.testing.assert_array_equal(a[k], b[k])
    assert a['num_mean_points_per_grid'] == b['num_mean_points_per_grid']
    assert a['point_cnt_of_grid'].sum() == len(a['grouped_idxs'])
    # every grouped entry's support point belongs to the centre's sample
    k, pt = a['grouped_idxs'][:, 0], a['grouped_idxs'][:, 1]
    assert ((k < 400) == (pt < 30)).all()
    # average pooling is linear in the features: <g, F x> == <F^T g, x>
    g = rng.normal(size=a['new_features'].shape).astype(np.float32)
    gx = o.stack_vector_pool_grad(g, a['point_cnt_of_grid'], a['grouped_idxs'], 650, 8)
    lhs = float((g.astype(np.float64) * a['new_features']).sum())
    rhs = float((gx.astype(np.float64) * feat).sum())
    assert abs(lhs - rhs) < 1e-3 * max(1.0, abs(lhs))
