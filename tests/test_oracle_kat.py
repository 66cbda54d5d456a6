"""Hand-derived known answers for the CPU oracle (the reference ships no tests: SURVEY.md F2).
Every expectation below is worked out from the cited reference lines, not from running code."""
import numpy as np

from oracle import cpu_oracle as o


def test_opt_n_threads_rule():
    # cuda_utils.h:10-14: min(2^floor(log2 n), 1024), >= 1
    assert [o.opt_n_threads(n) for n in (1, 2, 3, 63, 64, 1000, 1024, 1025, 16384, 65536)] == \
        [1, 2, 2, 32, 64, 512, 1024, 1024, 1024, 1024]


def test_fps_unit_cube_order():
    # 8 corners of the unit cube in index order (z fastest).  idx[0] = 0 (sampling_gpu.cu:118-120); the
    # farthest from corner 0 is corner 7 (d=3); then all of {1..6} are at min-distance... 1, 2, 4 have
    # d(0)=1,d(7)=2 -> 1; 3, 5, 6 have d(0)=2,d(7)=1 -> 1: six-way tie at 1.  Block size is 8, every
    # thread holds one point, and the tree keeps the LEFT operand on ties (:93-98):
    # level half=4: slots (0,4)(1,5)(2,6)(3,7) -> values {0:0,1:1,...}: slot0 = max(v0=0, v4=1) -> 4;
    #   slot1 = (v1=1, v5=1) tie -> 1; slot2 = (v2, v6) tie -> 2; slot3 = (v3=1, v7=0) -> 3
    # level half=2: slot0 = (4 | 2) tie -> 4; slot1 = (1 | 3) tie -> 1;  level half=1: (4 | 1) tie -> 4.
    cube = np.array([[[x, y, z] for x in (0, 1) for y in (0, 1) for z in (0, 1)]], dtype=np.float32)
    idx = o.furthest_point_sample(cube, 3)
    assert idx.tolist() == [[0, 7, 4]]


def test_fps_tie_break_is_tree_order_not_smallest_thread():
    # N = 4 -> block size 4.  Points 1, 2 and 3 are all at distance 1 from point 0 (ties on every slot).
    # Tree: half=2: slot0 = (v0=0, v2=1) -> 2; slot1 = (v1=1, v3=1) tie -> 1.  half=1: (2 | 1) tie -> LEFT = 2.
    # "smallest index among the maxima" would say 1; the reference's reduction says 2.
    pts = np.array([[[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]]], dtype=np.float32)
    idx = o.furthest_point_sample(pts, 2)
    assert idx.tolist() == [[0, 2]]
    # priority among equal maxima = smallest bit-reversed thread id: for 8 threads 0,4,2,6,1,5,3,7
    ring = np.zeros((1, 9, 3), dtype=np.float32)  # N = 9 -> block size 8; thread 0 owns points 0 and 8
    ang = np.arange(1, 8) * 0.7
    ring[0, 1:8, 0], ring[0, 1:8, 1] = np.cos(ang), np.sin(ang)  # points 1..7 on the unit circle around point 0
    ring[0, 8] = [0, 0, 0.5]
    # computed fp32 distances are not all bit-equal; make them so: use axis-aligned unit offsets instead
    ring[0, 1:8] = [[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1], [1, 0, 0]]
    idx = o.furthest_point_sample(ring, 2)
    assert idx.tolist() == [[0, 4]]  # all of 1..7 tie at d=1; bit-reversal order prefers thread 4


def test_fps_duplicated_cloud_picks_follow_block_1024():
    # N = 2048, second half duplicates the first (data_processor.py:206-210 pads like this).  Thread t owns
    # points t and t+1024 = the same coordinates; strict '>' (:143-144) keeps the first (lower) index.
    rng = np.random.default_rng(1)
    base = rng.uniform(-5, 5, (1, 1024, 3)).astype(np.float32)
    xyz = np.concatenate([base, base], 1)
    idx = o.furthest_point_sample(xyz, 64)
    assert (idx < 1024).all()
    assert np.array_equal(idx, o.furthest_point_sample(base, 64))


def test_fps_m_zero_and_temp():
    xyz = np.zeros((1, 5, 3), dtype=np.float32)
    idx, temp = o.furthest_point_sample(xyz, 3, return_temp=True)
    assert idx.tolist() == [[0, 0, 0]]  # every distance is 0: argmax stays at k = 0 (best=-1, first point wins)
    assert (temp == 0).all()


def test_ball_query_hits_padding_and_empty():
    # ball_query_gpu.cu:35-50.  Points on a line at x = 0,1,2,...,9; radius 1.5 (r^2 = 2.25, strict '<').
    xyz = np.zeros((1, 10, 3), dtype=np.float32); xyz[0, :, 0] = np.arange(10)
    new_xyz = np.array([[[0, 0, 0], [5, 0, 0], [100, 0, 0], [2.5, 0, 0]]], dtype=np.float32)
    idx = o.ball_query(1.5, 4, xyz, new_xyz)
    assert idx[0, 0].tolist() == [0, 1, 0, 0]      # 2 hits, rest padded with the first hit
    assert idx[0, 1].tolist() == [4, 5, 6, 4]      # 3 hits in index order, padded
    assert idx[0, 2].tolist() == [0, 0, 0, 0]      # empty ball: caller's zeros untouched
    assert idx[0, 3].tolist() == [2, 3, 2, 2]      # |2.5-1|=1.5 -> d2 = 2.25 is NOT < 2.25
    idx = o.ball_query(1.5, 2, xyz, new_xyz)
    assert idx[0, 1].tolist() == [4, 5]            # stops after nsample hits (:48)


def test_three_nn_ties_and_short_known_set():
    # interpolate_gpu.cu:37-56: strict '<' cascade in ascending k -> equal distances keep index order
    known = np.zeros((1, 5, 3), dtype=np.float32); known[0, :, 0] = [2, -2, 1, -1, 3]
    unknown = np.zeros((1, 1, 3), dtype=np.float32)
    d, i = o.three_nn(unknown, known)
    assert i[0, 0].tolist() == [2, 3, 0] and d[0, 0].tolist() == [1, 1, 2]
    d, i = o.three_nn(unknown, known[:, :2])
    assert i[0, 0].tolist() == [0, 1, 0] and d[0, 0, 0] == 2 and d[0, 0, 1] == 2 and np.isinf(d[0, 0, 2])


def test_gather_group_interpolate_small():
    feat = np.arange(12, dtype=np.float32).reshape(1, 2, 6)
    assert o.gather_operation(feat, np.array([[5, 0, 5]], np.int32)).tolist() == [[[5, 0, 5], [11, 6, 11]]]
    idx = np.array([[[0, 1], [5, 5]]], np.int32)
    g = o.grouping_operation(feat, idx)
    assert g.shape == (1, 2, 2, 2) and g[0, 1].tolist() == [[6, 7], [11, 11]]
    gg = o.grouping_operation_grad(np.ones_like(g), idx, 6)
    assert gg[0, 0].tolist() == [1, 1, 0, 0, 0, 2]   # duplicates accumulate (group_points_gpu.cu:30)
    w = np.array([[[0.5, 0.25, 0.25]]], np.float32)
    it = o.three_interpolate(feat, np.array([[[0, 2, 4]]], np.int32), w)
    assert it[0, :, 0].tolist() == [0 * .5 + 2 * .25 + 4 * .25, 6 * .5 + 8 * .25 + 10 * .25]


def test_query_and_group_channel_order_and_centring():
    # pointnet2_utils.py:250-257: xyz channels first, centred; features after
    xyz = np.array([[[0, 0, 0], [1, 2, 3], [9, 9, 9]]], np.float32)
    new_xyz = np.array([[[1, 1, 1]]], np.float32)
    feat = np.array([[[10, 20, 30]]], np.float32)
    out, idx = o.query_and_group(4.0, 2, xyz, new_xyz, feat)
    assert idx.tolist() == [[[0, 1]]]
    assert out[0, :, 0, :].tolist() == [[-1, 0], [-1, 1], [-1, 2], [10, 20]]


def test_distance_modes_differ_only_in_last_bits():
    rng = np.random.default_rng(3)
    xyz = rng.uniform(0, 70, (1, 2048, 3)).astype(np.float32)
    new_xyz = np.ascontiguousarray(xyz[:, :256])
    ref = o.ball_query(2.0, 32, xyz, new_xyz)
    flips = {}
    for mode in (o.DIST_NONE, o.DIST_HIPCC_DEFAULT):
        o.set_dist_mode(mode)
        try:
            flips[mode] = int((o.ball_query(2.0, 32, xyz, new_xyz) != ref).any(-1).sum())
        finally:
            o.set_dist_mode(o.DIST_PINNED)
    # rows that flip are rare (a point within 1 ulp of the sphere); the count is reported in DESIGN.md
    assert all(v <= 8 for v in flips.values()), flips


# ---- pointnet2_stack oracle (ragged batches) -----------------------------------------------------------------

def _ragged(rng, counts):
    return [rng.uniform(0, 10, (n, 3)).astype(np.float32) for n in counts]


def test_stack_oracle_equals_batch_oracle_on_equal_counts():
    """Same arithmetic as the batch operators: on an equal-count batch the stacked results are the batch results
    with the documented index conventions (ball query local, three_nn / FPS global)."""
    rng = np.random.default_rng(3)
    B, N, M = 3, 700, 60
    xyz = rng.uniform(0, 10, (B, N, 3)).astype(np.float32)
    flat, cnt = xyz.reshape(-1, 3), [N] * B
    fi = o.furthest_point_sample(xyz, M, block_size=1024)          # the stack kernel always runs 1024 threads
    np.testing.assert_array_equal(o.stack_furthest_point_sample(flat, cnt, M).reshape(B, M), fi + np.arange(B)[:, None] * N)
    new = np.stack([xyz[b, fi[b]] for b in range(B)])
    sidx, empty = o.stack_ball_query(1.5, 16, flat, cnt, new.reshape(-1, 3), [M] * B)
    np.testing.assert_array_equal(sidx.reshape(B, M, 16), o.ball_query(1.5, 16, xyz, new))
    assert not empty.any()
    d, i = o.three_nn(xyz, new)
    sd, si = o.stack_three_nn(flat, cnt, new.reshape(-1, 3), [M] * B)
    np.testing.assert_array_equal(si.reshape(B, N, 3), i + np.arange(B)[:, None, None] * M)
    np.testing.assert_array_equal(sd.reshape(B, N, 3), d)


def test_stack_ball_query_known_answers():
    # sample 0: 4 points on a line; sample 1: 2 points; centres: one per sample + one empty ball
    xyz = np.array([[0, 0, 0], [1, 0, 0], [2, 0, 0], [3, 0, 0], [10, 0, 0], [10.5, 0, 0]], np.float32)
    new = np.array([[1.1, 0, 0], [50, 0, 0], [10.2, 0, 0]], np.float32)
    idx, empty = o.stack_ball_query(1.0, 4, xyz, [4, 2], new, [2, 1])
    np.testing.assert_array_equal(idx[0], [1, 2, 1, 1])      # |1.1-1|, |1.1-2| < 1 (strict), padded with the first hit
    np.testing.assert_array_equal(idx[1], [0, 0, 0, 0])      # empty ball: row zeroed, mask set
    np.testing.assert_array_equal(idx[2], [0, 1, 0, 0])      # indices are LOCAL to sample 1
    np.testing.assert_array_equal(empty, [False, True, False])
    # a centre past the counted ones belongs to the last sample (the kernels' linear scan)
    idx2, _ = o.stack_ball_query(1.0, 2, xyz, [4, 2], np.array([[10.1, 0, 0]] * 4, np.float32), [1, 1])
    np.testing.assert_array_equal(idx2[2], [0, 1])


def test_stack_three_nn_short_sample_and_interpolate():
    known = np.array([[0, 0, 0], [1, 0, 0], [5, 0, 0], [6, 0, 0], [7, 0, 0], [8, 0, 0]], np.float32)
    unknown = np.array([[0.4, 0, 0], [6.1, 0, 0]], np.float32)
    dist, idx = o.stack_three_nn(unknown, [1, 1], known, [2, 4])
    np.testing.assert_array_equal(idx[0], [0, 1, 0])          # only two candidates: third slot = 0 + start, dist inf
    assert np.isinf(dist[0, 2]) and np.allclose(dist[0, :2], [0.4, 0.6])
    np.testing.assert_array_equal(idx[1], [3, 4, 2])          # GLOBAL indices (start of sample 1 = 2)
    feats = np.arange(12, dtype=np.float32).reshape(6, 2)
    w = np.array([[0.5, 0.5, 0.0], [0.2, 0.3, 0.5]], np.float32)
    out = o.stack_three_interpolate(feats, idx, w)
    np.testing.assert_allclose(out[1], 0.2 * feats[3] + 0.3 * feats[4] + 0.5 * feats[2], rtol=1e-6)
    g = o.stack_three_interpolate_grad(np.ones((2, 2), np.float32), idx, w, 6)
    np.testing.assert_allclose(g[:, 0], [0.5, 0.5, 0.5, 0.2, 0.3, 0.0], rtol=1e-6)


def test_stack_group_and_fps_ragged():
    rng = np.random.default_rng(5)
    pts = _ragged(rng, [5, 1300, 40])
    flat = np.concatenate(pts)
    out = o.stack_furthest_point_sample(flat, [5, 1300, 40], [3, 64, 0])
    assert out.shape == (67,) and out[0] == 0 and out[3] == 5          # first pick of every sample = its first point
    assert (out[:3] < 5).all() and ((out[3:] >= 5) & (out[3:] < 1305)).all()
    for b, (s, m) in enumerate([(0, 3), (5, 64)]):                      # each sample alone gives the same picks
        alone = o.stack_furthest_point_sample(pts[b], [len(pts[b])], [m])
        np.testing.assert_array_equal(out[[0, 3][b]:[3, 67][b]], alone + s)
    feats = rng.standard_normal((1345, 3)).astype(np.float32)
    idx = np.array([[0, 4], [1299, 7], [39, 0]], np.int32)
    grouped = o.stack_grouping_operation(feats, [5, 1300, 40], idx, [1, 1, 1])
    np.testing.assert_array_equal(grouped[1, :, 0], feats[5 + 1299])
    np.testing.assert_array_equal(grouped[2, :, 1], feats[1305])
    g = o.stack_grouping_operation_grad(np.ones_like(grouped), idx, [1, 1, 1], [5, 1300, 40], 1345)
    assert g.sum() == grouped.size and g[1305, 0] == 1 and g[0, 0] == 1


# ---- rotated-box IoU / NMS oracle (closed-form answers) --------------------------------------------------------

def test_iou3d_oracle_closed_form_areas():
    b = lambda x, y, dx, dy, h: [x, y, 0.0, dx, dy, 1.0, h]
    boxes = np.array([b(0, 0, 4, 2, 0.0), b(0, 0, 4, 2, np.pi / 2), b(1, 0, 4, 2, 0.0), b(10, 10, 1, 1, 0.3),
                      b(0, 0, 2, 2, 0.0), b(0, 0, 2, 2, np.pi / 4), b(0.5, 0.5, 1, 1, np.pi)], np.float32)
    ov = o.boxes_overlap_bev(boxes, boxes)
    np.testing.assert_allclose(np.diag(ov), boxes[:, 3] * boxes[:, 4], rtol=1e-5)        # a box with itself
    np.testing.assert_allclose(ov[0, 1], 4.0, rtol=1e-5)                                   # 4x2 against its 90 degree turn: 2x2
    np.testing.assert_allclose(ov[0, 2], 6.0, rtol=1e-5)                                   # shifted by 1 along x: 3x2
    assert ov[0, 3] == 0.0 and ov[3, 0] == 0.0                                             # disjoint
    np.testing.assert_allclose(ov[4, 5], 8.0 * (np.sqrt(2.0) - 1.0), rtol=1e-5)            # square vs 45 degree square: octagon
    np.testing.assert_allclose(ov[4, 6], 1.0, rtol=1e-5)                                   # contained unit square, heading pi
    np.testing.assert_allclose(ov, ov.T, rtol=1e-5, atol=1e-6)
    iou = o.boxes_iou_bev(boxes, boxes)
    np.testing.assert_allclose(iou[0, 2], 6.0 / 10.0, rtol=1e-5)
    np.testing.assert_allclose(o.boxes_aligned_overlap_bev(boxes, boxes[::-1].copy()), [ov[i, 6 - i] for i in range(7)], rtol=1e-6)


def test_iou3d_oracle_nms_known_answer():
    b = lambda x, y, h=0.0: [x, y, 0.0, 2.0, 2.0, 1.0, h]
    # in score order: A; B overlaps A by 1x2 (IoU 1/3); C overlaps A by 1.8x2 (IoU 0.818); D far away; E overlaps B strongly
    boxes = np.array([b(0, 0), b(1.0, 0), b(0.2, 0), b(10, 0), b(1.1, 0, 0.05)], np.float32)
    np.testing.assert_array_equal(o.nms(boxes, 0.5), [0, 1, 3])          # C removed by A, E removed by B
    # at 0.3 B (IoU 1/3 with A) goes too, so nothing suppresses E any more (its IoU with A is about 0.29): E is kept
    np.testing.assert_array_equal(o.nms(boxes, 0.3), [0, 3, 4])
    np.testing.assert_array_equal(o.nms(boxes, 0.9), [0, 1, 2, 3, 4])
    np.testing.assert_array_equal(o.nms(boxes, 0.5, normal=True), [0, 1, 3])
    assert len(o.nms(boxes[:0], 0.5)) == 0


def test_topk_sampling_kat():
    """Hand-checkable order of pdm_topk_sampling's spec: NaN (either sign) first, then +inf, descending scores with equal
    values by lower index, +0.0 above -0.0, -inf last."""
    import numpy as np
    from oracle import cpu_oracle as oracle
    s = np.array([[0.5, np.nan, 0.5, -0.0, 0.0, np.inf, -np.inf, 2.0, -np.nan]], dtype=np.float32)
    assert oracle.topk_sampling(s, 9).tolist() == [[1, 8, 5, 7, 0, 2, 4, 3, 6]]
    assert oracle.topk_sampling(s, 3).tolist() == [[1, 8, 5]]
    t = np.zeros((2, 6), dtype=np.float32)
    assert oracle.topk_sampling(t, 4).tolist() == [[0, 1, 2, 3]] * 2
