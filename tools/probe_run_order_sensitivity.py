"""development: what the ORDER of the oracle's own fp32 sums is worth on the run() windows of one soak sequence.  The checker's noise ensemble perturbs the
inverse depths by 1e-7 / 1e-6; the device differs from the oracle by the order of every sum of the window at once.  Here the oracle runs again on the same inputs
with its point and residual lists permuted (its AccumulatorApprox / Accumulator sums then run in another order, nothing else changes): the distance of those runs from the
unpermuted one, per iteration energy and pose, beside the device's.
Usage: python tools/probe_run_order_sensitivity.py <sequence index> [max window size to report]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from libcml_amd import device, sequence
from tests import sequence_check as SC

s_ = int(sys.argv[1]) if len(sys.argv) > 1 else 198
nmax = int(sys.argv[2]) if len(sys.argv) > 2 else 2
seq = sequence.make_sequence(n_frames=28, seed=0x5EED + 101 * s_, shard=s_)
ctx = device.Ctx(max_frames=8, max_points=8192, max_residuals=8192 * 8)


class Chk(SC.SequenceChecker):
    def on_run(self, info):
        fr, pt, rs = info["before"]
        if len(fr) <= nmax:
            HM, bM = info["prior"]
            I = SC.inputs_from_export(fr, pt, rs, info["grads0"], self.K, self.w, self.h)
            o = SC.oracle_run(I, HM, bM)
            eo = np.asarray(o["log"]["energy"])
            ed = np.asarray(info["energies"])[-info["iterations"]:] if info["iterations"] else np.zeros(0)
            n = min(len(ed), len(eo) - 1)
            print("run N=%d R=%d: oracle energies %s" % (I.N, I.R, ["%.6g" % e for e in eo]))
            print("   device  energies %s  worst relative distance %.2e" % (["%.6g" % e for e in ed], float(np.abs(ed[:n] / eo[1:1 + n] - 1).max()) if n else 0.0))
            for trial in range(6):
                I2 = SC.inputs_from_export(fr, pt, rs, info["grads0"], self.K, self.w, self.h)
                rng = np.random.default_rng(7000 + trial)
                pp = rng.permutation(I2.P); inv = np.empty(I2.P, np.int64); inv[pp] = np.arange(I2.P)      # new point k = old point pp[k]
                I2.points = np.ascontiguousarray(I2.points[pp])
                res = I2.residuals.copy(); res["point"] = inv[res["point"]]
                I2.residuals = np.ascontiguousarray(res[rng.permutation(I2.R)])
                m = SC.oracle_run(I2, HM, bM)
                em = np.asarray(m["log"]["energy"])
                k = min(len(em), len(eo))
                dR = max(float(np.abs(o["poses"][j][0] - m["poses"][j][0]).max()) for j in range(I.N)); dt = max(float(np.abs(o["poses"][j][1] - m["poses"][j][1]).max()) for j in range(I.N))
                print("   oracle, point / residual lists permuted #%d: energies %s  worst relative distance from the unpermuted oracle %.2e  |dR| %.2e |dt| %.2e" % (
                    trial, ["%.6g" % e for e in em], float(np.abs(em[1:k] / eo[1:k] - 1).max()) if k > 1 else 0.0, dR, dt))
        return super().on_run(info)


chk = Chk(ctx, seq.K, seq.w, seq.h, seq.levels, strict=False)
pipe = sequence.DirectPipeline(ctx, seq.K, seq.w, seq.h, seq.levels, observer=chk)
pipe.run(seq)
print("failures", [f[:160] for f in chk.report["failures"]])
pipe.close(); ctx.close()
