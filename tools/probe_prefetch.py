"""development: a sequence shard with and without the next frame's pyramid handed to the image worker (cmlhip_pyramid_build_async)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from libcml_amd import device, sequence
seq = sequence.make_sequence(n_frames=48, seed=0x5EED)
for rep in range(4):
    ctx = device.Ctx(max_frames=8, max_points=8192, max_residuals=8192 * 8)
    pipe = sequence.DirectPipeline(ctx, seq.K, seq.w, seq.h, seq.levels)
    pipe.prefetch = rep % 2 == 1
    t0 = time.perf_counter(); pipe.run(seq); dt = time.perf_counter() - t0
    v = pipe.lib_times["pyramid_build"]
    lib = sum(sum(x) for k, x in pipe.lib_times.items())
    print("pass", rep, "prefetch", pipe.prefetch, "total %.1f ms" % (1e3 * dt), "library %.1f ms" % (1e3 * lib), "pyramid lib calls (us):", [int(1e6 * x) for x in v[:6]],
          "sum %.2f ms" % (1e3 * sum(v)), "track mean %.3f trace mean %.3f" % (1e3 * np.mean(pipe.lib_times["trackWithMotionModel"]), 1e3 * np.mean(pipe.lib_times["traceNewCoarse"])))
    pipe.close(); ctx.close()
