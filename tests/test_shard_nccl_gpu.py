"""The process-group path of the shard runner on REAL hardware with backend "nccl" (= RCCL): bench.py launched through
torch.distributed.run as the driver launches it for N > 1, here with one rank (the GPU box has one GPU) and the group forced on, so
that init_process_group("nccl"), the all-reduce barrier and the MAX / SUM reductions of the timing contract execute on the device.
(The N > 1 logic itself — shard assignment, max-over-ranks — is covered by the gloo world-2 test tests/test_shard_cpu.py.)"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_through_torchrun_with_rccl_group():
    env = dict(os.environ, CML_SHARD_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "8", "--warmup", "2", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["steps"] == 8 and d["value"] > 1e6 and d["scaling"] == "weak"
    assert d["roofline"]["achieved"] > 0


def test_bench_with_two_ranks_on_the_one_gpu(tmp_path):
    """Config D's command — `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N …` — with N = 2 REAL ranks doing device work.  The GPU box has one GPU
    and RCCL refuses two ranks on one device, so both ranks run on device 0 (CML_BENCH_SHARE_DEVICE=1) and the group is gloo (CML_SHARD_BACKEND=gloo): what runs is
    this file's N > 1 logic — one window and one 48-frame sequence shard per rank (its own seed), barriers on both sides of every timed region, MAX over ranks of
    the time, SUM of the units, rank 0 alone printing the one line — not the RCCL transport (covered with one rank above)."""
    env = dict(os.environ, CML_BENCH_SHARE_DEVICE="1", CML_SHARD_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    detail = str(tmp_path / "detail.json")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29537", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "3", "--detail", detail]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines                                   # rank 0 alone prints
    assert len(lines[0]) < 4096
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 10 and d["scaling"] == "weak" and d["config"]["shards"] == 2
    # value = the units of BOTH ranks / the slower rank's time
    assert abs(d["value"] - 2 * 14000 * 1e3 / d["ms_per_step"]) <= 1e-6 * d["value"]
    assert d["parity_checked"] is True and d["parity_ok"] is True
    ss = d["sequence_shards"]
    assert ss["shards"] == 2 and ss["frames_per_s_total"] > 0 and ss["tracking_lost_total"] == 0
    assert "cpu_baseline" not in d                                    # reported at N = 1 only
