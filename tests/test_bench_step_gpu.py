"""The object bench.py TIMES — bench.Bench: four distinct batches rotated through static input buffers by
pdm_copy_many, PipelinedHotPath with BOTH heads (level-1 FPS as resumable segments of batches i+1 .. i+S on a side
stream, the rest of batch i+1's coordinate chain on another, the neck + heat-map head beside the FP layers, the point
head behind them), the whole step captured into a hipGraph and replayed — must return, for EVERY batch of the
rotation, bit for bit what the plain serial detector loop returns for that batch (the reference's loop:
/root/reference/pcdet/models/detectors/point_rcnn.py:9-11), and the serial loop must agree with the CPU statement of
the step (oracle operators + the detector's torch layers on the CPU) to 1e-4.

The kernels are the same on both sides, so a difference here is a stale hand-over buffer, a missing stream edge or a
workspace shared by two concurrent branches — the class of error no per-operator test can see.
"""
import numpy as np
import pytest
import torch

import bench
from pdm_ssd_amd import synthetic

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def detector(dev):
    return bench.build_detector(dev)


def run_and_compare(b, runs):
    """`runs` steps of the timed object; after each, every output against the serial detector on the consumed batch."""
    seen = set()
    with torch.no_grad():
        for r in range(runs):
            b.run()
            torch.cuda.synchronize()
            got = [t.clone() for t in b.outputs()]
            src = b.consumed_batch()
            seen.add(src.data_ptr())
            assert torch.equal(b.inputs[0], src), "stage 0 does not hold the batch the rotation says it holds"
            want = b.step_serial(src)
            torch.cuda.synchronize()
            for name, g, w in zip(b.OUTPUTS, got, want):
                assert g.data_ptr() != w.data_ptr()
                assert torch.isfinite(g).all(), f"run {r}: {name} holds non-finite values"
                assert torch.equal(g, w), (f"run {r} ({b.mode}, depth {b.depth}): {name} differs from the serial detector in "
                                           f"{int((g != w).sum())} of {g.numel()} elements")
    return seen


@pytest.mark.parametrize("B,N,kind", [(4, 16384, "uniform"), (4, 16384, "lidar"), (2, 65536, "lidar"), (8, 16384, "uniform")])
def test_graph_replayed_step_equals_serial_detector(detector, dev, B, N, kind):
    """bench.py's default object (depth 4, hipGraph) at the bench's point counts: 7 steps, i.e. the four-batch rotation
    wraps and every pipeline stage has been refilled by the hand-over at least once."""
    b = bench.Bench(detector, B, N, kind, 4, dev, seed0=31 + B, graph=True)
    assert b.mode == "hipGraph" and b.depth == 4
    seen = run_and_compare(b, 7)
    assert len(seen) == bench.Bench.NBATCH          # every distinct batch went through stage 0
    outs = [t.clone() for t in b.outputs()]
    b.run()                                          # different batch -> different results (the outputs are not stuck)
    torch.cuda.synchronize()
    assert not torch.equal(outs[1], b.outputs()[1]) and not torch.equal(outs[0], b.outputs()[0])
    if N > 16384:
        b.pipe.check_sampling()
    assert b.verify(2)["vs_serial"] == "bit-equal"


@pytest.mark.parametrize("depth,graph", [(4, False), (3, True), (2, True), (2, False), (1, True)])
def test_other_depths_and_eager_launch_equal_serial_detector(detector, dev, depth, graph):
    b = bench.Bench(detector, 2, 16384, "lidar", depth, dev, seed0=77, graph=graph)
    assert b.depth == depth and b.mode == ("hipGraph" if graph else "eager")
    run_and_compare(b, 6)


def test_verify_detects_a_stale_stage(detector, dev):
    """The check itself: corrupt one hand-over buffer of the pipeline behind its back (what a missing copy would leave)
    and verify() must fail."""
    b = bench.Bench(detector, 2, 16384, "uniform", 4, dev, seed0=5, graph=True)
    b.verify(1)
    b.pipe.cur['sampled_xyz'][0].add_(0.25)          # level-1 sampled set of the batch about to be processed
    with pytest.raises(AssertionError):
        b.verify(1)


def test_serial_detector_matches_cpu_statement(detector, dev):
    """The serial side of the comparison above against the CPU statement of the same step: oracle C operators for the
    backbone and the neck (oracle/cpu_backbone.py), the detector's own torch layers on the CPU for the two heads."""
    import copy

    from oracle import cpu_backbone
    B, N = 2, 16384
    b = bench.Bench(detector, B, N, "lidar", 4, dev, seed0=900, graph=True)
    b.run()
    torch.cuda.synchronize()
    pts = b.consumed_batch()
    sf, pf, boxes, hm = [t.clone() for t in b.outputs()]
    clouds = np.ascontiguousarray(pts[:, 1:5].cpu().numpy().reshape(B, N, 4))
    m = copy.deepcopy(detector).cpu().eval()
    with torch.no_grad():
        ref = cpu_backbone.backbone_forward(m.backbone_3d, clouds)
        ref_sf = cpu_backbone.neck_forward(m.map_to_bev_module, ref['sa_xyz'], ref['sa_features'])
        dd = m.dense_head({'spatial_features': torch.from_numpy(ref_sf)})
        pd = m.point_head({'batch_size': B, 'point_features': torch.from_numpy(ref['point_features']),
                           'point_coords': pts[:, :4].cpu()})
    tol = 1e-4
    np.testing.assert_allclose(pf.cpu().numpy(), ref['point_features'], rtol=tol, atol=tol * max(1.0, float(np.abs(ref['point_features']).max())))
    np.testing.assert_allclose(sf.cpu().numpy(), ref_sf, rtol=tol, atol=tol * float(np.abs(ref_sf).max()))
    want_hm = dd['bev_heatmap'].numpy()
    np.testing.assert_allclose(hm.cpu().numpy(), want_hm, rtol=tol, atol=tol * max(1.0, float(np.abs(want_hm).max())))
    want_boxes = pd['batch_box_preds'].numpy()
    got_boxes = boxes.cpu().numpy()
    # the decoded class picks the mean-size row: compare the boxes of points whose top-2 class logits are not within 1e-3
    top2 = torch.topk(pd['batch_cls_preds'], 2, dim=1).values
    clear = ((top2[:, 0] - top2[:, 1]) > 1e-3).numpy()
    assert clear.mean() > 0.9
    np.testing.assert_allclose(got_boxes[clear], want_boxes[clear], rtol=tol, atol=tol * max(1.0, float(np.abs(want_boxes).max())))
