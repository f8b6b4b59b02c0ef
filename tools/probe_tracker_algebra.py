"""Where the single-wave algebra of k_tracker_optimize goes (build with CML_HIPCC_EXTRA=-DTO_PROFILE: the three in-kernel clocks land in
unused slots of the result): pivoted 8x8 LDL^T, lane 0's pose + evaluation constants, sums -> system + accept."""
import sys
sys.path.insert(0, ".")
import numpy as np
from libcml_amd import device
from tests import trk_opt_setup as TS
P = TS.make_problem("B")
ctx = device.Ctx(max_frames=8)
ctx.pyramid_build(501, P.W.gray[P.s.new], P.levels)
for l in range(P.levels): ctx.tracker_set_reference(l, P.uvic[l])
hyps = [TS.perturbed(P, (0.004, -0.003, 0.002), (0.03, -0.02, 0.025))]
for _ in range(3): res = ctx.tracker_optimize_batch(501, P.levels, P.W.K, P.ref_exp, P.init_exp, P.prm, hyps)
r = res[0]
print("trials", r.n_steps, "eval", r.eval_us, "algebra", r.algebra_us, "ldlt", r.pass_rmse[7], "lane0 pose+prepare", r.pass_rmse[6], "finish+accept", r.pass_rmse[5],
      "| pose split: increment %.1f  SE3::exp %.1f  product+store %.1f  prepare %.1f" % (r.relAff[0], r.relAff[1], r.flow[0], r.flow[1]),
      "| eval split: points+mfma %.1f  tiles+barriers %.1f  tile sum %.1f  exchange %.1f" % tuple(r.covariance[k] for k in range(4)))
