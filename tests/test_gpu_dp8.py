"""Eight ranks of `bench.py --gpus 8` on the ONE GPU of the test box (SWIFTLLM_SPAWN_DEVICES=0,0,0,0,0,0,0,0): the launch
plumbing the driver's 8-GPU scaling run will exercise — self-spawn, rank environment, disjoint core sets, the gloo control
group's barriers, max-over-ranks timing, one JSON line from rank 0 — executed end to end before that run, so that its
first execution measures scaling instead of debugging launch code. Plumbing only: eight replicas sharing
one GPU say nothing about scaling, and no scaling claim is made from this test."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_eight_rank_bench_plumbing_on_one_gpu():
    env = dict(os.environ)
    for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "WORLD_SIZE", "RANK", "LOCAL_RANK",
              "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["SWIFTLLM_SPAWN_DEVICES"] = ",".join(["0"] * 8)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--model", "tiny", "--batch", "4",
           "--prompt-len", "64", "--kv-blocks", "256", "--steps", "6", "--warmup", "2", "--no-extras", "--no-cpu-baseline",
           "--no-reference"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]            # rank 0 owns stdout: exactly one JSON line
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["steps"] == 6 and out["warmup"] == 2 and out["scaling"] == "weak"
    assert out["config"]["global_batch"] == 8 * 4
    assert "dp8" in out["config"]["parallelism"] and "no collective" in out["config"]["parallelism"]
    # value = the units ALL ranks processed / the slowest rank's time
    assert out["value"] == pytest.approx(8 * 4 * 6 / (out["ms_per_step"] * 6 * 1e-3), rel=1e-3)
    aff = out["config"]["cpu_affinity"]
    allowed = len(os.sched_getaffinity(0))
    if allowed >= 8 and aff.get("cores"):               # rank 0's share of the allowed cores: eight disjoint sets fit
        assert 1 <= aff["cores"] <= allowed // 8 + 1, (aff, allowed)
