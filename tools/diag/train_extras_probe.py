import sys, time, copy, torch
sys.path.insert(0,'/root/repo')
import bench
from types import SimpleNamespace
dev=torch.device('cuda:0')
model=bench.build_detector(dev)
_,pts=bench.make_batch(32,16384,'uniform',1234,dev)
args=SimpleNamespace(steps=8,warmup=3,serial=False,train_profile=None)
# A: fresh model
r=bench.train_bench(args, copy.deepcopy(model), pts, 32,16384,0,1,0,dev)
print('fresh', r['ms_per_step'])
# B: after an inference Bench existed
b=bench.Bench(model,32,16384,'uniform',4,dev,seed0=1234)
t=b.timed(10,3); print('infer ms', t/10*1e3)
r=bench.train_bench(args, copy.deepcopy(model), pts, 32,16384,0,1,0,dev)
print('with live Bench', r['ms_per_step'])
del b; torch.cuda.empty_cache()
r=bench.train_bench(args, copy.deepcopy(model), pts, 32,16384,0,1,0,dev)
print('after del Bench', r['ms_per_step'])
# host-only time of a step: launch without sync, measure enqueue time
