"""Run on the GPU box with CMLHOST_TIMING=sum: the host mirror's laps (libcml_amd/host/HostLap.h) over the 48-frame sequence shard, second pass
(the first pass warms allocations, pools and code objects and is forgotten).  The table is printed when the process ends."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("CMLHOST_TIMING", "sum")
import numpy as np
from libcml_amd import device, sequence, host
seq = sequence.make_sequence(n_frames=48, seed=0x5EED)
for rep in range(2):
    ctx = device.Ctx(max_frames=8, max_points=8192, max_residuals=8192 * 8)
    pipe = sequence.DirectPipeline(ctx, seq.K, seq.w, seq.h, seq.levels)
    if rep == 1:
        host.lib().cmlhost_laps_reset()
    pipe.run(seq)
    if rep == 1:
        for k, v in pipe.timing_summary().items():
            print("%-28s calls %3d  mean %.3f ms  median %.3f  max %.3f" % (k, v["calls"], v["mean_ms"], v["median_ms"], v["max_ms"]))
    pipe.close(); ctx.close()
