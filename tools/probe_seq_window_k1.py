"""Run on the GPU box: the residual kernel's dispatch duration on a SEQUENCE window (the device window the last keyframe of a 48-frame shard left), cold
(first iterations after an idle gap) and warm (after 300 back-to-back iterations) — beside bench.py's figure for the synthetic config-B window."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from libcml_amd import device, sequence
seq = sequence.make_sequence(n_frames=48)
ctx = device.Ctx(max_frames=8, max_points=8192, max_residuals=8192 * 8)
pipe = sequence.DirectPipeline(ctx, seq.K, seq.w, seq.h, seq.levels)
kfset = set(seq.keyframes)
pipe.bootstrap(seq.gray[0], seq.R_true[0], seq.t_true[0], seq.boot_px, seq.boot_idepth)
lastkf = max(kfset)
for k in range(1, lastkf):
    (pipe.keyframe if k in kfset else pipe.non_keyframe)(seq.gray[k])
# the last keyframe by hand up to run(): the window stays committed on the device
ba = pipe.ba
iid, Rn, tn, a, b, ok = pipe.track(seq.gray[lastkf])
pipe.trace(iid, Rn, tn, a, b, traced_fid=pipe.n_fid)
ba.flag_frames_for_marginalization_v(pipe._immature_counts()); ba.add_frame(iid, Rn, tn, a, b, 1.0)
assert ba.run(), ba.last_error()
N, P, R = ctx.refresh_window_size()
print("window N=%d P=%d R=%d" % (N, P, R))
assert ba.begin_resident(), ba.last_error()
def k1(n, stride=1):
    ctx.profile_stride(stride); ctx.profile_select(1); ctx.profile_enable(n)
    for _ in range(n):
        ctx.ba_iteration_async(1e-5)
    ctx.sync()
    lin_ms, _ss, _e, ns = ctx.profile_read()
    return 1e3 * lin_ms, ns
time.sleep(0.05)
print("cold: K1 %.2f us over %d dispatches (4 iterations after 50 ms of idle)" % k1(4))
time.sleep(0.05)
print("cold again: K1 %.2f us over %d" % k1(4))
for _ in range(300):
    ctx.ba_iteration_async(1e-5)
ctx.sync()
print("warm: K1 %.2f us over %d dispatches (behind 300 back-to-back iterations)" % k1(50))
st = ctx.ba_states()
print("states: IN %d OOB %d OUTLIER %d good %d" % ((st["state"] == 0).sum(), (st["state"] == 1).sum(), (st["state"] == 2).sum(), st["good"].sum()))
