"""development: seconds inside each library call of a keyframe (the sequence driver's lib_times keep one entry per call)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from libcml_amd import device, sequence
seq = sequence.make_sequence(n_frames=48, seed=0x5EED)
for rep in range(2):
    ctx = device.Ctx(max_frames=8, max_points=8192, max_residuals=8192 * 8)
    pipe = sequence.DirectPipeline(ctx, seq.K, seq.w, seq.h, seq.levels)
    calls = {}
    orig = pipe._c
    def _c(stage, fn, *a, **kw):
        t0 = time.perf_counter(); r = orig(stage, fn, *a, **kw); dt = time.perf_counter() - t0
        calls.setdefault(stage + ":" + getattr(fn, "__name__", str(fn)), []).append(dt)
        return r
    pipe._c = _c
    pipe.run(seq)
    if rep == 1:
        for k, v in sorted(calls.items()):
            print("%-52s calls %3d  median %.3f ms  mean %.3f ms" % (k, len(v), 1e3 * np.median(v), 1e3 * np.mean(v)))
    pipe.close(); ctx.close()
