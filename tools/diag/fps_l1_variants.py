import sys; sys.path.insert(0,'/root/repo')
import numpy as np, torch
from pdm_ssd_amd import _native, synthetic
from pdm_ssd_amd.pointnet2_batch import pointnet2_utils as pu
dev=torch.device('cuda:0'); l=_native.lib()
for kind in ('uniform','lidar'):
    gen = synthetic.uniform_clouds if kind=='uniform' else synthetic.lidar_like_clouds
    xyz=torch.from_numpy(np.ascontiguousarray(gen(32,16384,5)[:,:,:3])).to(dev)
    ref=None
    for v in (0,3,1,2):
        l.pdm_tune_fps_variant(v)
        idx=pu.furthest_point_sample(xyz,4096); torch.cuda.synchronize()
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): idx=pu.furthest_point_sample(xyz,4096)
        e1.record(); torch.cuda.synchronize()
        if ref is None: ref=idx.clone()
        print(kind,"variant",v,round(e0.elapsed_time(e1)/3,3),"ms same=",torch.equal(idx,ref))
l.pdm_tune_fps_variant(0)
