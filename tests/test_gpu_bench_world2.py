"""GPU: bench.py's N>1 path end to end with two ranks sharing the one GPU of the test box (gloo rendezvous on 127.0.0.1, since RCCL
needs one device per rank): proof shards signed per rank and all-gathered, replica verification, the sharded MSM (2^20 and the 2^24
strong-scaling entry; term sharding and window sharding) with its all-gather of Jacobian partials as one stream-ordered chain,
max-over-ranks timing, one JSON line from rank 0 -- and every sharding gives the point of the single-rank run, which bench.py itself
checks against the reference."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ["--steps", "1", "--warmup", "1", "--batch", "512", "--no-cpu-baseline", "--no-dropin", "--no-distinct"]


def _run(world, sharding=None, port=29517, self_launch=False):
    env = dict(os.environ, S2K_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    if sharding:
        env["S2K_MSM_SHARDING"] = sharding
    if world == 1:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + COMMON
    elif self_launch:                                           # the way the driver invokes a 1-GPU bench, with --gpus N: bench.py starts its own ranks
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world)] + COMMON
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port", str(port),
               os.path.join(ROOT, "bench.py"), "--gpus", str(world)] + COMMON
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                  # only rank 0 prints
    return json.loads(lines[0])


def test_gpus_flag_is_checked():
    """--gpus N with RCCL needs N devices, and a launcher that started another number of ranks is an error, not a silent 1-GPU run"""
    env = dict(os.environ); [env.pop(k, None) for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "S2K_DIST_BACKEND")]
    import torch
    if torch.cuda.device_count() < 2:
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + COMMON, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode != 0 and "--gpus 2" in out.stderr and not [l for l in out.stdout.splitlines() if l.startswith("{")]
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + COMMON, cwd=ROOT, env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", S2K_DIST_BACKEND="gloo"),
                         capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "must agree" in out.stderr


def test_bench_two_ranks_one_gpu():
    j1 = _run(1)
    assert j1["msm"]["verified"] and j1["msm"]["strong_2p24"]["verified"]          # the single-rank points are the reference's
    for k, (sharding, port) in enumerate((("terms", 29517), ("windows", 29519))):
        j = _run(2, sharding, port, self_launch=(k == 0))       # once without any launcher, once under torch.distributed.run
        assert j["collective"]["world_size"] == 2 and j["collective"]["rank_devices"] == [0, 0] and j["collective"]["backend"] == "gloo"
        assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["value"] > 0 and j["config"]["batch_per_gpu"] == 512
        m = j["msm"]
        assert m["terms"] == 1 << 20 and m["terms_per_rank"] == 1 << 19 and m["verified"] and m["frac"] > 0
        assert m["strong_2p24"]["terms"] == 1 << 24 and m["strong_2p24"]["terms_per_rank"] == 1 << 23 and m["strong_2p24"]["verified"]
        assert ("windows" in m["sharding"]) == (sharding == "windows")
        assert m["result_x"] == j1["msm"]["result_x"] and m["strong_2p24"]["result_x"] == j1["msm"]["strong_2p24"]["result_x"]
