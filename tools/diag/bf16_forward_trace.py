"""Where does the bf16-autocast GPU forward of the detector's training step leave the bf16-emulating CPU graph
(oracle/cpu_detector.py)?  Relative L2 distance of the tensors the step exposes, GPU bf16 vs emulation, emulation vs fp32
graph and GPU fp32 vs fp32 graph, plus the same with the fused BatchNorm / split-K layers switched off (PDM_FUSED_BN=0)."""
import copy, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from detector_case import build_case
from oracle import cpu_detector, cpu_oracle
from pdm_ssd_amd import detectors, synthetic, fused_bn
cpu_oracle.build()
dev = torch.device("cuda:0")
model, clouds, gt = build_case()
ref32 = cpu_detector.detector_train_step(model, clouds, gt, bf16=False)
emu = cpu_detector.detector_train_step(model, clouds, gt, bf16=True)

def gpu(autocast):
    m = copy.deepcopy(model).to(dev).train()
    batch = {'batch_size': clouds.shape[0], 'points': torch.from_numpy(synthetic.to_batch_points(clouds)).to(dev), 'gt_boxes': torch.from_numpy(gt).to(dev)}
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
        for mod in m.module_list:
            batch = mod(batch)
    fr = m.point_head.forward_ret_dict
    return {'sa_features': [None if t is None else t.detach().float().cpu() for t in batch['sa_features']],
            'point_features': batch['point_features'].detach().float().cpu(), 'spatial_features': batch['spatial_features'].detach().float().cpu(),
            'point_cls_preds': fr['point_cls_preds'].detach().float().cpu(), 'point_box_preds': fr['point_box_preds'].detach().float().cpu(),
            'hm_logits': m.dense_head.forward_ret_dict['hm_logits'].detach().float().cpu()}

def rel(a, b):
    return float((a - b).norm() / b.norm().clamp(min=1e-30))

def table(tag, g16, g32):
    print(f"--- {tag}: tensor | GPU bf16 vs emulation | GPU bf16 vs fp32 graph | emulation vs fp32 graph | GPU fp32 vs fp32 graph")
    for k in ('sa1', 'sa2', 'sa3', 'point_features', 'spatial_features', 'point_cls_preds', 'point_box_preds', 'hm_logits'):
        if k.startswith('sa'):
            i = int(k[2])
            a, e, r, b = g16['sa_features'][i], emu['sa_features'][i], ref32['sa_features'][i], g32['sa_features'][i]
        else:
            a, e, r, b = g16[k], emu[k], ref32[k], g32[k]
        print(f"{k:18s} {rel(a, e):.4f} {rel(a, r):.4f} {rel(e, r):.4f} {rel(b, r):.2e}")

g32 = gpu(False)
table("fused BatchNorm + split-K layers (default)", gpu(True), g32)
fused_bn.ENABLED = False
table("PDM_FUSED_BN=0 (torch / MIOpen layers)", gpu(True), g32)
fused_bn.ENABLED = True
from pdm_ssd_amd.pointnet2_batch import pointnet2_modules
pointnet2_modules.CHANNELS_LAST_TRAINING = False
table("channels-last grouping off (reference-layout grouped tensors, torch casts)", gpu(True), g32)
