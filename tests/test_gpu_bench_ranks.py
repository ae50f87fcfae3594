"""The N > 1 step of bench.py run end to end on the one-GPU test box.

* two ranks, started by bench.py itself from a plain `python bench.py --gpus 2` (no launcher):
  both share cuda:0, so the exchange is the Python rehearsal over gloo (RCCL refuses two ranks on
  one device); the numbers mean nothing, the parity fields do. The run carries the second leg with
  1000 queries per step (the batch of the 8-GPU benchmark configuration) through the same exchange.
* one rank with the exchange forced on: the step is ONE trag_shard_search_device call and the
  all-gather is RCCL inside the library."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _last_json(out):
    return json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])


@pytest.mark.gpu
def test_two_ranks_share_one_gpu_self_launched():
    env = dict(os.environ, TRAG_BENCH_ONE_GPU="1", TRAG_BENCH_DEBUG="1", TRAG_BENCH_WATCHDOG_S="120")
    for name in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(name, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2",
           "--rows", "300001", "--batch", "3"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    r = _last_json(out)
    assert r["n_gpus"] == 2 and r["config"]["rows_per_gpu"] == 150001
    assert r["parity"] == {"uncertified_queries": 0, "sorted": True, "device_merge_equals_host_merge": True}
    assert "cpu_baseline" not in r and "REHEARSAL" in r["data"]
    leg = r["batch_1000"]
    assert leg["batch"] == 1000 and leg["unfinished_queries"] == 0 and leg["sorted"]
    assert leg["all_ranks_hold_the_same_lists"] is True


@pytest.mark.gpu
def test_one_rank_step_is_one_c_call_over_rccl():
    env = dict(os.environ, TRAG_BENCH_FORCE_EXCHANGE="1", TRAG_BENCH_DEBUG="1", TRAG_BENCH_WATCHDOG_S="120")
    for name in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "TRAG_BENCH_ONE_GPU"):
        env.pop(name, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "8", "--warmup", "2",
           "--rows", "400000", "--no-cpu-baseline", "--no-pmc"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    r = _last_json(out)
    assert "RCCL all-gather" in r["config"]["exchange"]
    assert r["parity"]["uncertified_queries"] == 0 and r["parity"]["device_merge_equals_host_merge"] is True
    leg = r["batch_1000"]
    assert leg["unfinished_queries"] == 0 and leg["sorted"] and leg["batch"] == 1000
