"""world_size-2 gloo test of the shard runner used by bench.py for N > 1 (no GPU, no data-path collective)."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import os, sys, time, json
    sys.path.insert(0, %r)
    from libcml_amd import shard
    g = shard.Group(backend="gloo")
    assert g.world == 2
    mine = shard.shards_for_rank(5, g.rank, g.world)
    units = 1000.0 * len(mine)
    def run():
        time.sleep(0.05 * (g.rank + 1))          # rank 1 is slower: the timed region must report the MAX
    dt = shard.timed_region(g, lambda: None, run)
    total = g.sum(units)
    ids = g.sum(float(sum(mine)))
    if g.rank == 0:
        print(json.dumps({"dt": dt, "total": total, "ids": ids, "mine": mine}))
    g.close()
''') % ROOT


def test_two_rank_gloo_barrier_and_reductions(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29533", str(script)],
                         capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    r = json.loads(line)
    assert r["total"] == 5000.0 and r["ids"] == 10.0 and r["mine"] == [0, 2, 4]
    assert r["dt"] >= 0.095, "the region time must be the max over ranks"


def test_shard_assignment_is_a_partition():
    from libcml_amd import shard
    for world in (1, 2, 4, 8):
        got = sorted(s for r in range(world) for s in shard.shards_for_rank(8, r, world))
        assert got == list(range(8))
        assert all(len(shard.shards_for_rank(8, r, world)) == 8 // world for r in range(world))


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` started WITHOUT a torchrun environment (as the driver starts N = 1) must start 2 ranks itself and report
    the size of the group that formed.  --launch-only keeps the device out of it: gloo, no GPU here."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-only"], capture_output=True, text=True,
                         env=env, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout                       # rank 0 only
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["requested_gpus"] == 2 and r["launch_only"] is True and r["backend"] == "gloo"
