#!/bin/bash
# GPU-box helper: average duration of the kernels whose name contains <pattern>, for several library
# variants (results may be wrong on purpose: timing only).  scripts/ablate.sh "<bench args>" <pattern> <variant> ...
ARGS="$1"; PAT="$2"; shift 2
cd /tmp && export TMPDIR=/tmp
for V in "$@"; do
  if [ "$V" = base ]; then unset TRAG_LIB; else export TRAG_LIB=$GRAFT_REPO_ROOT/trueno-rag_amd/lib_var/$V/libtrag.so; fi
  rm -rf /tmp/abl_$V
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abl_$V -o p -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-pmc --steps 10 $ARGS > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/abl_$V.err
  F=$(find /tmp/abl_$V -name "*kernel_stats.csv" | head -1)
  python3 - "$F" "$PAT" "$V" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Name"]:
        print("%-8s %-70s calls %5s avg_us %9.1f" % (sys.argv[3], r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3))
PY
done
