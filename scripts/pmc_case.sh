#!/bin/bash
# GPU-box helper: one counter pass over a bench case. `bash scripts/pmc_case.sh <name> "<counters>" <bench args...>`
# prints per-kernel averages and leaves gpurun_out/pmc_<name>.csv (kernel, counter, mean value over launches).
set -e
N="$1"; C="$2"; shift; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_$N
timeout -s KILL 200 rocprofv3 --pmc $C --output-format csv -d /tmp/pmc_$N -o p -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-pmc --steps 3 --warmup 1 "$@" > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/pmc_$N.err
python3 - /tmp/pmc_$N $GRAFT_REPO_ROOT/gpurun_out/pmc_$N.csv <<'PY'
import csv,glob,sys,collections
acc=collections.defaultdict(list)
for f in glob.glob(sys.argv[1]+"/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[(r["Kernel_Name"].split("(")[0][:70], r["Counter_Name"])].append(float(r["Counter_Value"]))
out=open(sys.argv[2],"w",newline=""); wr=csv.writer(out); wr.writerow(["kernel","counter","mean","launches"])
tot=collections.defaultdict(float)
for (k,c),v in acc.items(): tot[k]+=sum(v) if c in ("SQ_WAVE_CYCLES","SQ_BUSY_CYCLES","FETCH_SIZE","GRBM_GUI_ACTIVE") else 0
top=sorted(tot,key=tot.get,reverse=True)[:4]
for (k,c),v in sorted(acc.items()):
    wr.writerow([k,c,"%.1f"%(sum(v)/len(v)),len(v)])
    if k in top: print("%-70s %-28s %16.1f  x%d"%(k,c,sum(v)/len(v),len(v)))
PY
