#!/bin/bash
# GPU-box helper: `bash scripts/gpu_check.sh <pytest target or ""> <bench case> ...`
# Runs the given GPU tests (if any), then the named bench cases, each under its own time limit;
# one summary line per case, the JSON lines under gpurun_out/.
set -e
T="$1"; shift || true
if [ -n "$T" ]; then
  timeout -k 10 900 python -u -m pytest $T -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/gc_tests.log 2>&1 || { tail -40 gpurun_out/gc_tests.log; exit 1; }
  tail -2 gpurun_out/gc_tests.log
fi
B="timeout -k 10 300 python bench.py --no-cpu-baseline"
for c in "$@"; do
  case $c in
    10M) A="";; 10Mm) A="--mirror";;
    1M) A="--rows 1000000 --steps 200";; 1Mm) A="--rows 1000000 --steps 200 --mirror";;
    125) A="--rows 1250000 --steps 400 --warmup 20";; 125m) A="--rows 1250000 --steps 400 --warmup 20 --mirror";;
    c3) A="--rows 1000000 --dim 768 --batch 256";; c3m) A="--rows 1000000 --dim 768 --batch 256 --mirror";;
    c3f) A="--rows 1000000 --dim 768 --batch 256 --mode 3";;
    c4) A="--rows 1250000 --batch 1000";; c4m) A="--rows 1250000 --batch 1000 --mirror";;
    c5) A="--rows 2000000 --dim 4096 --hybrid";; c5m) A="--rows 2000000 --dim 4096 --hybrid --mirror";;
    d4k) A="--rows 2000000 --dim 4096";; d4km) A="--rows 2000000 --dim 4096 --mirror";;
    d768) A="--rows 2000000 --dim 768";; d1536) A="--rows 1000000 --dim 1536";; d128) A="--rows 8000000 --dim 128";;
    1Meu) A="--rows 1000000 --steps 200 --metric euclidean";; 1Mdot) A="--rows 1000000 --steps 200 --metric dot";;
    c1) A="--rows 10000 --host --steps 2000 --warmup 50";;
    100k) A="--rows 100000 --steps 500 --warmup 20";; 10k) A="--rows 10000 --steps 1000 --warmup 20";;
    b64m) A="--batch 64 --mirror";; b128m) A="--batch 128 --mirror";; b8m) A="--batch 8 --mirror";;
    b32) A="--batch 32";; b32m) A="--batch 32 --mirror";;
    b256) A="--batch 256";; b256m) A="--batch 256 --mirror";;
    b4) A="--batch 4";; b4m) A="--batch 4 --mirror";; b2m) A="--batch 2 --mirror";;
    *) echo "unknown case $c"; exit 2;;
  esac
  $B $A > gpurun_out/gc_$c.json 2> gpurun_out/gc_$c.err || { tail -5 gpurun_out/gc_$c.err; exit 1; }
  python - $c <<'PY'
import json,sys
r=json.load(open(f"gpurun_out/gc_{sys.argv[1]}.json"))
p=r["parity"]
print("%-6s q/s %10.1f  ms/step %7.3f  scan_ms %7.3f  %s frac %.3f  ok=%s  %s" % (sys.argv[1], r["value"], r["ms_per_step"], r["roofline"]["kernel_avg_ms"], r["roofline"]["bound"], r["roofline"]["frac"], p["uncertified_queries"]==0 and p["sorted"] and p.get("exact_scan_identical",True), r["config"]["path_used"][:34]))
PY
done
