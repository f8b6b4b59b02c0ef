"""development: one soak sequence (index on the command line) under the checker, with the tracker's early exit on / off and the level split at 512 / 1024"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libcml_amd import device, sequence
from tests import sequence_check as SC
s_ = int(sys.argv[1]) if len(sys.argv) > 1 else 15
early = int(sys.argv[2]) if len(sys.argv) > 2 else 1
prefetch = int(sys.argv[3]) if len(sys.argv) > 3 else 1
seq = sequence.make_sequence(n_frames=28, seed=0x5EED + 101 * s_, shard=s_)
ctx = device.Ctx(max_frames=8, max_points=8192, max_residuals=8192 * 8)
chk = SC.SequenceChecker(ctx, seq.K, seq.w, seq.h, seq.levels, strict=False)
pipe = sequence.DirectPipeline(ctx, seq.K, seq.w, seq.h, seq.levels, observer=chk)
pipe.trk.set_param("batchedEarlyExit", early)
pipe.prefetch = bool(prefetch)
st = pipe.run(seq)
rep = chk.report
print("seq", s_, "early", early, "prefetch", prefetch, "split", os.environ.get("CMLHIP_TRACKER_SPLIT"), "failures", rep["failures"], "worst track", {k: "%.2e" % v for k, v in rep["worst"].items() if k.startswith("track")}, "flips", rep["flips"])
pipe.close(); ctx.close()
for d in rep.get("tracker_winner_detail", []):
    print("   winner/tries flip:", d)
for y in rep.get("run_yardstick", []):
    print("   run yardstick: N", y["N"], "R", y["R"], "iterations", y.get("iterations"), "accepted by", y.get("accepted_by"), "spread", {k: "%.2e" % v for k, v in y.get("ensemble_spread", {}).items()}, "device vs oracle", {k: ("%.2e" % v if isinstance(v, float) else v) for k, v in y["device_vs_oracle"].items()})
    print("      energies device", ["%.6g" % x for x in y.get("energies_device", [])], "oracle", ["%.6g" % x for x in y.get("energies_oracle", [])])
    print("      device vs nearest member", {k: ("%.2e" % v if isinstance(v, float) else v) for k, v in y["device_vs_nearest_member"].items()})
    for name, dd in y["members_vs_oracle"].items():
        print("      member vs oracle  %-34s" % name, {k: ("%.2e" % v if isinstance(v, float) else v) for k, v in dd.items()})
