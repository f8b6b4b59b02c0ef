"""Run on the GPU box: where DSOBundleAdjustment::run spends its time at sequence sizes (CMLHOST_TIMING / CMLHIP_TIMING laps of the last keyframes)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from libcml_amd import device, sequence
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seq = sequence.make_sequence(n_frames=n)
ctx = device.Ctx(max_frames=8, max_points=8192, max_residuals=8192 * 8)
pipe = sequence.DirectPipeline(ctx, seq.K, seq.w, seq.h, seq.levels)
kfset = set(seq.keyframes)
pipe.bootstrap(seq.gray[0], seq.R_true[0], seq.t_true[0], seq.boot_px, seq.boot_idepth)
last2 = sorted(kfset)[-2:]
for k in range(1, n):
    if k in last2:
        os.environ["CMLHOST_TIMING"] = "1"; os.environ["CMLHIP_TIMING"] = "1"
        c = pipe.ba.counts(); print("---- keyframe at frame %d (window before: %d frames)" % (k, c["frames"]), file=sys.stderr)
    (pipe.keyframe if k in kfset else pipe.non_keyframe)(seq.gray[k])
    if k in last2:
        c = pipe.ba.counts(); print("     run() took %.0f us; window %d frames, %d points, %d residuals in the lists" % (1e6 * pipe.times["run"][-1], c["frames"], c["points"], c["residuals"]), file=sys.stderr)
    os.environ.pop("CMLHOST_TIMING", None); os.environ.pop("CMLHIP_TIMING", None)
for k, v in pipe.timing_summary().items():
    print("%-28s calls %3d  mean %.3f ms  median %.3f  max %.3f" % (k, v["calls"], v["mean_ms"], v["median_ms"], v["max_ms"]))
