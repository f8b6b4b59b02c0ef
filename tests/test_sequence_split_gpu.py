"""The sequence shard with the mapper beside the tracker (libcml_amd/sequence.SplitPipeline): two contexts, two streams, Hybrid::directMappingLoop on a
host thread of its own (slam/modslam/Hybrid.cpp:103-106, direct/Mapping.cpp:3-41) while the SLAM thread keeps tracking against the previous keyframe.
(1) the split schedule, run inline on ONE thread, under the lock-step checker: every stage replayed by the oracle from the product's state, as
tests/test_sequence_gpu.py does for the single-context shard (one checker per context: reference lists + tracking | everything else);
(2) the same schedule with the mapper on its own thread: every tracked pose and every hand-over (reference-list points, optimised pose, energies of
run(), outlier list, window poses) identical IN EVERY BIT to the inline run — concurrency changes when things happen, not what is computed."""
import numpy as np
import pytest

from libcml_amd import device, sequence
from tests import sequence_check as SC

pytestmark = pytest.mark.gpu


def _run(seq, threaded, checked=False):
    ct = device.Ctx(max_frames=8, max_points=8192, max_residuals=8192 * 8)
    cm = device.Ctx(max_frames=8, max_points=8192, max_residuals=8192 * 8)
    chk_t = chk_m = None
    if checked:
        chk_t = SC.SequenceChecker(ct, seq.K, seq.w, seq.h, seq.levels, strict=True)
        chk_m = SC.SequenceChecker(cm, seq.K, seq.w, seq.h, seq.levels, strict=True)
    pipe = sequence.SplitPipeline(ct, cm, seq.K, seq.w, seq.h, seq.levels, threaded=threaded, front_observer=chk_t, observer=chk_m)
    try:
        stats = pipe.run(seq)
        return stats, list(pipe.front.results), list(pipe.kf_log), (chk_t.report if checked else None, chk_m.report if checked else None)
    finally:
        pipe.close(); ct.close(); cm.close()


def _bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint8)


def test_split_shard_inline_against_oracle_and_threaded_bit_identical():
    seq = sequence.make_sequence(n_frames=40, seed=0x5EED)
    stats_i, trk_i, kf_i, (rep_t, rep_m) = _run(seq, threaded=False, checked=True)
    assert not rep_t["failures"] and not rep_m["failures"], (rep_t["failures"], rep_m["failures"])
    assert stats_i["frames"] == 40 and stats_i["keyframes"] == len(seq.keyframes) and stats_i["tracking_lost"] == 0
    assert rep_t["stages"]["track"] == 39 and rep_t["stages"]["coarse"] == len(seq.keyframes) - 1
    assert rep_m["stages"]["run"] == len(seq.keyframes) - 1 and rep_m["stages"]["trace"] == 39
    assert stats_i["max_window"] == 7 and stats_i["marginalized_frames"] >= 2
    assert rep_m.get("run_yardstick_used", 0) <= 1 and rep_t.get("track_yardstick_used", 0) <= 1
    # the tracked frame beside a keyframe's mapping really ran against the PREVIOUS reference (lag 1): the schedule is what the test claims
    stats_t, trk_t, kf_t, _ = _run(seq, threaded=True)
    assert stats_t["frames"] == 40 and stats_t["keyframes"] == stats_i["keyframes"]
    assert len(trk_t) == len(trk_i) == 39 and len(kf_t) == len(kf_i)
    for a, b in zip(trk_i, trk_t):
        assert a[0] == b[0] and a[5] == b[5] and a[6] == b[6]
        for x, y in zip(a[1:5], b[1:5]):
            assert np.array_equal(_bits(np.asarray(x, np.float64)), _bits(np.asarray(y, np.float64))), ("tracked frame", a[0])
    for (ka, ha), (kb, hb) in zip(kf_i, kf_t):
        assert ka == kb and ha.keys() == hb.keys()
        for key in ha:
            if key == "window":
                assert len(ha[key]) == len(hb[key])
                for pa, pb in zip(ha[key], hb[key]):
                    assert np.array_equal(_bits(pa[0]), _bits(pb[0])) and np.array_equal(_bits(pa[1]), _bits(pb[1])) and pa[2:] == pb[2:]
            elif key in ("ab", "iterations"):
                assert ha[key] == hb[key], (ka, key)
            else:
                assert np.array_equal(_bits(np.asarray(ha[key])), _bits(np.asarray(hb[key]))), (ka, key)
