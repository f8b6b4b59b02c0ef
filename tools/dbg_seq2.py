import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from libcml_amd import device, sequence
from tests import sequence_check as SC
seq = sequence.make_sequence(n_frames=48, seed=0x5EED + (0xC0FFEE & 0xff))
ctx = device.Ctx(max_frames=8, max_points=8192, max_residuals=8192 * 8)
chk = SC.SequenceChecker(ctx, seq.K, seq.w, seq.h, seq.levels, strict=False)
orig = chk.on_run
def on_run(info):
    fr, pt, rs = info["before"]; HM, bM = info["prior"]
    I = SC.inputs_from_export(fr, pt, rs, info["grads0"], chk.K, chk.w, chk.h)
    o = SC.oracle_run(I, HM, bM)
    print("N=%d R=%d iterations dev %d oracle %d" % (I.N, I.R, info["iterations"], o["iterations"]))
    print(" dev energies   ", np.array2string(np.asarray(info["energies"])[-info["iterations"]:], precision=3))
    print(" oracle energies", np.array2string(np.asarray(o["log"]["energy"]), precision=3))
    x0 = o["log"]["x"][0]
    print(" oracle |x0| per frame:", [float(np.abs(x0[4+8*k:12+8*k]).max()) for k in range(I.N)])
    for trial in range(4):
        I2 = SC.inputs_from_export(fr, pt, rs, info["grads0"], chk.K, chk.w, chk.h)
        rng = np.random.default_rng(trial)
        I2.points["idepth"] *= (1 + 1e-7 * rng.standard_normal(I2.P))
        o2 = SC.oracle_run(I2, HM, bM)
        dR = max(np.abs(o["poses"][k][0] - o2["poses"][k][0]).max() for k in range(I.N))
        print("   oracle with inverse depths perturbed by 1e-7 (relative): energies", np.array2string(np.asarray(o2["log"]["energy"][1:5]), precision=1), "pose dR %.2e flips %d" % (dR, int((o["good"] != o2["good"]).sum())))
    orig(info)
chk.on_run = on_run
pipe = sequence.DirectPipeline(ctx, seq.K, seq.w, seq.h, seq.levels, observer=chk)
pipe.run(seq, n_frames=int(sys.argv[1]) if len(sys.argv) > 1 else 6)
print(chk.report["failures"])
