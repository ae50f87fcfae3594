"""The N > 1 step of bench.py — local scan, pack, ONE all-gather, device merge — run end to end
with two ranks. The box has one GPU, so both ranks share cuda:0 and the gather goes through gloo
(staged through pinned host memory by ShardExchange); the numbers of such a run mean nothing,
the parity fields do."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_two_ranks_share_one_gpu():
    env = dict(os.environ, TRAG_BENCH_ONE_GPU="1", TRAG_BENCH_DEBUG="1", TRAG_BENCH_WATCHDOG_S="90")
    port = 29600 + os.getpid() % 300
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "6", "--warmup", "2", "--rows", "300001", "--batch", "3"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=170, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    r = json.loads(line)
    assert r["n_gpus"] == 2 and r["config"]["rows_per_gpu"] == 150001
    assert r["parity"] == {"uncertified_queries": 0, "sorted": True, "device_merge_equals_host_merge": True}
    assert "cpu_baseline" not in r and "REHEARSAL" in r["data"]
