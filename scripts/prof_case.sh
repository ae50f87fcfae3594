#!/bin/bash
# GPU-box helper: kernel-trace statistics of one bench case. `bash scripts/prof_case.sh <name> <bench args...>`
# leaves gpurun_out/prof_<name>_kernel_stats.csv
set -e
N="$1"; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$N
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$N -o p -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-pmc "$@" > $GRAFT_REPO_ROOT/gpurun_out/prof_$N.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof_$N.err
F=$(find /tmp/prof_$N -name "*kernel_stats.csv" | head -1)
cp "$F" $GRAFT_REPO_ROOT/gpurun_out/prof_${N}_kernel_stats.csv
python3 - "$F" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:9]:
    print("%-90s calls %6s avg_us %10.1f pct %5s" % (r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
