import sys; sys.path.insert(0,'/root/repo')
import torch
from pdm_ssd_amd import _native
from pdm_ssd_amd.pointnet2_batch import pointnet2_utils as pu
dev=torch.device('cuda:0'); l=_native.lib()
for N,m in [(8192,2048),(4096,1024),(2048,512),(1024,256)]:
    xyz=torch.rand(32,N,3,device=dev)*50
    ref=None
    for v in (0,8,5):
        l.pdm_tune_fps_variant(v)
        idx=pu.furthest_point_sample(xyz,m); torch.cuda.synchronize()
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): idx=pu.furthest_point_sample(xyz,m)
        e1.record(); torch.cuda.synchronize()
        if ref is None: ref=idx.clone()
        print(N,m,"variant",v,round(e0.elapsed_time(e1)/5*1e3,1),"us",round(e0.elapsed_time(e1)/5*1e3/(m-1),3),"us/iter same=",torch.equal(idx,ref))
l.pdm_tune_fps_variant(0)
