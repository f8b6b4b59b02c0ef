import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from libcml_amd import device, sequence
from tests import sequence_check as SC
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seq = sequence.make_sequence(n_frames=n)
ctx = device.Ctx(max_frames=8, max_points=8192, max_residuals=8192 * 8)
chk = SC.SequenceChecker(ctx, seq.K, seq.w, seq.h, seq.levels, strict=False); chk.debug = True
pipe = sequence.DirectPipeline(ctx, seq.K, seq.w, seq.h, seq.levels, observer=chk)
stats = pipe.run(seq)
rep = chk.report
print(json.dumps({"stats": stats, "stages": rep["stages"], "worst": rep["worst"], "flips": rep["flips"], "counts": {k: v for k, v in rep.items() if isinstance(v, int)}}, indent=1))
print("failures:", len(rep["failures"]))
for f in rep["failures"][:30]: print("  ", f)
print("oracle seconds:", {k: round(v, 2) for k, v in chk.oracle_seconds.items()})
