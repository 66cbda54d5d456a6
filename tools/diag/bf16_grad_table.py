"""Per-parameter gradient error of the bf16-autocast training step against the fp32 CPU autograd graph
(the comparison tests/test_configs_gpu.py asserts on), printed as a table."""
import copy, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_configs_gpu as t
from oracle import cpu_autograd, cpu_oracle
from pdm_ssd_amd import synthetic
from pdm_ssd_amd.pointnet2_backbone import PointNet2MSG
cpu_oracle.build()
dev = torch.device("cuda:0")
torch.manual_seed(11)
bb = PointNet2MSG(t.TRAIN_CFG, input_channels=4).train()
neck = t.make_neck(t.TRAIN_NECK, seed=12).train()
B, N = int(sys.argv[1]) if len(sys.argv) > 1 else 2, int(sys.argv[2]) if len(sys.argv) > 2 else 2048
clouds = synthetic.lidar_like_clouds(B, N, 31)
bb_c, neck_c = copy.deepcopy(bb), copy.deepcopy(neck)
out = cpu_autograd.train_forward(bb_c, neck_c, clouds)
loss = t._train_loss(out['point_features'], out['spatial_features']); loss.backward()
want = {f"backbone.{k}": p.grad.clone() for k, p in bb_c.named_parameters()}
want.update({f"neck.{k}": p.grad.clone() for k, p in neck_c.named_parameters()})
ref = {'bb': bb, 'neck': neck, 'clouds': clouds}
t.TRAIN_B = B
for ac in (False, True):
    l, grads, bd = t._gpu_train_step(ref, dev, ac)
    err = t._grad_errors(grads, want)
    print("autocast", ac, "loss", l, "ref", float(loss))
    for k in want:
        cos = float((grads[k] * want[k]).sum() / (grads[k].norm() * want[k].norm() + 1e-30))
        if ac: print(f"  {k:50s} |g|={float(want[k].norm()):.3e} relL2={err[k]:.4f} cos={cos:.5f}")
    fg = torch.cat([grads[k].reshape(-1) for k in want]); fw = torch.cat([want[k].reshape(-1) for k in want])
    print("  whole vector relL2", float((fg - fw).norm() / fw.norm()), "worst", max(err.values()))
