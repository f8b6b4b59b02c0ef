"""CPU check of the replay plumbing of tests/resident_check.py (the checker of the timed kernels): an oracle window advanced by
oracle primitives, then one of its passes replayed by ResidentReplay from a snapshot of the state, must reproduce that pass bit for bit."""
import ctypes as C

import numpy as np

from tests import ba_ref_run, ba_setup as S
from tests import oracle_lib as O
from tests import resident_check as RC


def test_replay_reproduces_an_oracle_pass():
    I = S.make_inputs("small")
    ob = S.OracleBA(I)
    ob.linearize(); ob.apply(1)
    pre = ob.states()
    # step the points a little and move the pairs, as an iteration would
    idepth = np.array([ob.w.contents.points[i].idepth for i in range(I.P)]) * (1 + 1e-3 * np.sin(np.arange(I.P)))
    for i in range(I.P):
        ob.w.contents.points[i].idepth = idepth[i]
    pairs = I.pairs.copy()
    pairs["t"] += 1e-4
    ob.set_pairs(pairs)
    th = np.array([ob.w.contents.frame_energy_th[k] for k in range(I.N)], np.float32)
    ob.linearize()
    rj = ob.rJ(0)
    new_state = ob.states()["new_state"].copy()
    ob.apply(1)
    post = ob.states()
    jp = ob.view("JpJdF", 8 * I.R, np.float32).reshape(-1, 8).copy()
    rp = RC.ResidentReplay(I.prm, I.frames_dev, [I.grads[k][0] for k in range(I.N)], I.points, I.residuals)
    o = rp.replay(pre, pairs, th, idepth)
    assert np.array_equal(o["new_state"], new_state) and np.array_equal(o["state"], post["state"]) and np.array_equal(o["good"], post["good"])
    for k in ("energy", "new_energy", "new_energy_wo"):
        assert np.array_equal(o[k].view(np.uint32), post[k].view(np.uint32)), k
    g = post["good"] == 1
    assert g.sum() > 50
    assert np.array_equal(o["jpjdf"][g].view(np.uint32), jp[g].view(np.uint32))
    assert np.array_equal(o["efsj"][g].view(np.uint32), rj[g].view(np.uint32))
    # and a perturbed threshold changes the classification: the replay is not vacuous
    o2 = rp.replay(pre, pairs, th * 0.01, idepth)
    assert (o2["new_state"] != new_state).sum() > 0
    rp.close()


def test_bench_compact_line_stays_under_the_driver_limit():
    """round 4's full result object (27.5 KB as one line: the driver could not parse it) through bench.compact_line: < 3 KB, contract keys kept"""
    import json
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    out = json.load(open(os.path.join(root, "profiles", "round4_bench_driver_cmd.json")))
    assert len(json.dumps(out)) > 20000
    line = json.dumps(bench.compact_line(out, "bench_detail.json"), separators=(",", ":"))
    assert len(line) < bench.LINE_LIMIT
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline", "parity_checked", "parity_ok", "configs", "sequence", "tracker", "solve"):
        assert k in d, k
    assert d["value"] == out["value"] and d["roofline"]["frac"] == out["roofline"]["frac"]
    assert set(d["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"}
    small = json.dumps(bench.compact_line(out, "bench_detail.json", contract_only=True), separators=(",", ":"))
    assert len(small) < len(line) and "roofline" in json.loads(small)
