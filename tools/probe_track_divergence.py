"""development: for one soak sequence, the tracked frames whose result left the fixed bars — the oracle's trial log of the winning hypothesis beside
the device's accept sequence for the same hypothesis, and the margin of the accept test (E_new / n_new against E / n, TR.cpp:163) at the first
trial where they part"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from libcml_amd import abi, device, sequence
from tests import sequence_check as SC, oracle_lib as O, trk_opt_setup as TO
s_ = int(sys.argv[1]) if len(sys.argv) > 1 else 55
seq = sequence.make_sequence(n_frames=28, seed=0x5EED + 101 * s_, shard=s_)
ctx = device.Ctx(max_frames=8, max_points=8192, max_residuals=8192 * 8)


class Chk(SC.SequenceChecker):
    def on_track(self, info):
        n0 = self.report.get("track_yardstick_used", 0)
        super().on_track(info)
        if self.report.get("track_yardstick_used", 0) == n0:
            return
        P = TO.Problem()
        _g, grads = O.build_pyramid(info["gray"], self.levels)
        P.levels = self.levels; P.imgs = [np.ascontiguousarray(g, np.float32) for g in grads]; P.uvic = self.ref
        P.ref_exp = info["ref_exp"]; P.init_exp = info["init_exp"]; P.prm = abi.default_tracker_params()

        class _W: pass
        P.W = _W(); P.W.K = self.K
        r = info["result"]; w = max(int(r["winner"]), 0)
        R0, t0 = info["hyps"][w]
        q = TO.orc_problem(P)
        out = O.OrcTrkResult(); log = (O.OrcTrkStep * 512)()
        T_ = O.se3_from_Rt(R0, t0); a, b = C.c_double(P.init_exp[0]), C.c_double(P.init_exp[1])
        O.lib().orc_tracker_optimize(C.byref(q), C.byref(T_), C.byref(a), C.byref(b), C.byref(out), log, 512)
        dres = self.ctx.tracker_optimize_batch(info["image_id"], self.levels, self.K, info["ref_exp"], info["init_exp"], P.prm, [(R0, t0)])[0]
        n = min(out.n_steps, dres.n_steps)
        print("frame with a result outside the bars: winner hypothesis", w, "oracle trials", out.n_steps, "device trials", dres.n_steps)
        for i in range(n):
            if log[i].accept != dres.step_accept[i] or log[i].level != dres.step_level[i]:
                en, eo = log[i].E_new / max(log[i].n_new, 1), log[i].E_old / max(log[i].n_old, 1)
                print("   first trial that differs: #%d level %d: oracle accept %d, device accept %d; oracle's test E_new/n_new = %.9g against E/n = %.9g: margin %.2e relative" % (
                    i, log[i].level, log[i].accept, dres.step_accept[i], en, eo, abs(en / eo - 1)))
                break
        else:
            print("   the common trials agree; the runs differ in length only")


chk = Chk(ctx, seq.K, seq.w, seq.h, seq.levels, strict=False)
pipe = sequence.DirectPipeline(ctx, seq.K, seq.w, seq.h, seq.levels, observer=chk)
pipe.run(seq)
print("failures", chk.report["failures"])
pipe.close(); ctx.close()
