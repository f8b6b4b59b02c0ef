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
