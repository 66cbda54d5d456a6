"""Rotated NMS timing: 4096 candidate boxes (pre_maxsize of the reference's configs), device-side mask reduction."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from pdm_ssd_amd.iou3d_nms import iou3d_nms_utils as iu
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
for n, spread in ((4096, 40.0), (4096, 10.0), (512, 20.0)):
    b = np.concatenate([rng.uniform(-spread, spread, (n, 2)), rng.uniform(-1, 1, (n, 1)), rng.uniform(1.5, 4.5, (n, 1)),
                        rng.uniform(1.0, 2.2, (n, 1)), rng.uniform(1, 2, (n, 1)), rng.uniform(-3.14, 3.14, (n, 1))], 1).astype(np.float32)
    boxes, scores = torch.from_numpy(b).to(dev), torch.rand(n, device=dev)
    sel, _ = iu.nms_gpu(boxes, scores, 0.1); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        sel, _ = iu.nms_gpu(boxes, scores, 0.1)
    e1.record(); torch.cuda.synchronize()
    e2, e3 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e2.record()
    for _ in range(10):
        iu.boxes_iou_bev(boxes, boxes)
    e3.record(); torch.cuda.synchronize()
    print(f"n={n} spread={spread}: nms_gpu {e0.elapsed_time(e1)/10:.3f} ms ({len(sel)} kept); boxes_iou_bev n x n {e2.elapsed_time(e3)/10:.3f} ms")
