import os, sys, time
sys.path.insert(0, os.getcwd())
os.environ["CML_SHARD_FORCE_DIST"]="1"
import torch
from libcml_amd import shard
torch.cuda.set_device(0)
g = shard.Group(device=torch.device("cuda",0))
for _ in range(5): g.barrier()
ts=[]
for _ in range(20):
    t0=time.perf_counter(); g.barrier(); ts.append(time.perf_counter()-t0)
print("nccl barrier (all_reduce + cuda sync): median %.1f us min %.1f" % (1e6*sorted(ts)[10], 1e6*min(ts)))
ts=[]
for _ in range(20):
    t0=time.perf_counter(); v=g.max(1.0); ts.append(time.perf_counter()-t0)
print("nccl max(): median %.1f us" % (1e6*sorted(ts)[10]))
import torch.distributed as dist
pg = dist.new_group(backend="gloo")
for _ in range(5): dist.barrier(group=pg)
ts=[]
for _ in range(20):
    t0=time.perf_counter(); dist.barrier(group=pg); ts.append(time.perf_counter()-t0)
print("gloo barrier: median %.1f us" % (1e6*sorted(ts)[10]))
g.close()
