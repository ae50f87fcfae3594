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
    p = r["parity"]
    assert p["uncertified_queries"] == 0 and p["sorted"] is True and p["device_merge_equals_host_merge"] is True
    # the sharded result against ONE store holding the whole corpus (exact scan), and the oracle leg
    assert p["sharded_equals_single_store"] is True
    chk = p["single_store_check"]
    assert chk["whole_corpus_rows"] == 300001 and chk["headline_identical"] and chk["batch_1000_identical"]
    assert chk["headline_queries_compared"] == 8 and chk["batch_1000_queries_compared"] == 32
    assert chk["unfinished_skipped"] == 0
    po = p["parity_vs_oracle"]
    assert po["rows"] == 300001 and po["ranks_identical"] and po["score_bits_identical"]
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
    # what RCCL itself reports: the communicator spans as many ranks as the run has, on this rank's device
    assert r["config"]["rccl_version"] >= 20000 and r["config"]["rccl_comm_world"] == r["n_gpus"] == 1
    (mine,) = r["config"]["ranks"]
    assert {k: mine[k] for k in ("comm_world", "comm_rank", "comm_device", "rank", "local_device")} == \
        {"comm_world": 1, "comm_rank": 0, "comm_device": 0, "rank": 0, "local_device": 0}
    assert mine["pid"] > 0 and mine["host"] and mine["gpu"]        # which process on which GPU ran the rank
    assert r["parity"]["uncertified_queries"] == 0 and r["parity"]["device_merge_equals_host_merge"] is True
    assert r["parity"]["sharded_equals_single_store"] is True
    assert r["parity"]["single_store_check"]["batch_1000_identical"] is True
    assert "parity_vs_oracle" not in r["parity"]      # (--no-cpu-baseline: the oracle leg is left out)
    leg = r["batch_1000"]
    assert leg["unfinished_queries"] == 0 and leg["sorted"] and leg["batch"] == 1000


def test_self_launch_is_refused_under_a_profiler():
    """CPU: `bench.py --gpus N` inside a profiler session must not start the ranks itself (the
    profiler's library has initialised the GPU in the process); it says what to do and exits non-zero."""
    env = dict(os.environ, ROCPROFILER_REGISTER_FORCE_LOAD="1")
    for name in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(name, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True,
                         text=True, timeout=120, env=env, cwd=ROOT)
    assert out.returncode != 0 and "under a profiler" in out.stderr
