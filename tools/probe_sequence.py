"""Run on the GPU box: the direct pipeline (libcml_amd/sequence.py) over a synthetic sequence, stage timings and drift against the truth."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from libcml_amd import device, sequence

n = int(sys.argv[1]) if len(sys.argv) > 1 else 48
t0 = time.time()
seq = sequence.make_sequence(n_frames=n)
print("sequence made in %.1f s, keyframes %s" % (time.time() - t0, seq.keyframes))
ctx = device.Ctx(max_frames=8, max_points=8192, max_residuals=8192 * 8)
pipe = sequence.DirectPipeline(ctx, seq.K, seq.w, seq.h, seq.levels)
t0 = time.time()
kfset = set(seq.keyframes)
pipe.bootstrap(seq.gray[0], seq.R_true[0], seq.t_true[0], seq.boot_px, seq.boot_idepth)
for k in range(1, n):
    ok = pipe.keyframe(seq.gray[k]) if k in kfset else pipe.non_keyframe(seq.gray[k])
    R, t = pipe.history[-1]
    c = -R.T @ t; ct = -seq.R_true[k].T @ seq.t_true[k]
    ang = np.degrees(np.arccos(np.clip((np.trace(R @ seq.R_true[k].T) - 1) / 2, -1, 1)))
    cnt = pipe.ba.counts()
    if k in kfset or not ok:
        _, alive, _ = pipe.ba.points()
        print("frame %2d %s ok=%d  |dc| %.4f m  rot %.3f deg  window %d  active points %d  residuals %d  a,b %.3f %.2f (true %.3f %.2f)" % (
            k, "KF" if k in kfset else "  ", ok, np.linalg.norm(c - ct), ang, len(pipe.kfs), int(alive.sum()), cnt["residuals"], pipe.last_exposure[0], pipe.last_exposure[1], *seq.aff_true[k]))
print("total %.2f s for %d frames" % (time.time() - t0, n))
print(json.dumps(pipe.stats))
for k, v in pipe.timing_summary().items():
    print("%-28s calls %3d  mean %.3f ms  median %.3f  max %.3f" % (k, v["calls"], v["mean_ms"], v["median_ms"], v["max_ms"]))
pipe.close(); ctx.close()
