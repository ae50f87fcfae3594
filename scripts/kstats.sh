#!/bin/bash
# GPU-box helper: per-kernel average durations of one bench case for several library variants:
#   scripts/kstats.sh "<bench args>" <variant> ...      ("base" = the in-tree library)
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
ARGS="$1"; shift
for V in "$@"; do
  if [ "$V" = base ]; then unset TRAG_LIB; else export TRAG_LIB=$GRAFT_REPO_ROOT/trueno-rag_amd/lib_var/$V/libtrag.so; fi
  rm -rf /tmp/ks_$V
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$V -o ks -- python3 bench.py --no-cpu-baseline --no-pmc --steps 30 $ARGS > gpurun_out/ks_$V.json 2> gpurun_out/ks_$V.err || { echo "$V FAILED"; tail -3 gpurun_out/ks_$V.err; exit 1; }
  F=$(find /tmp/ks_$V -name '*kernel_stats.csv' | head -1)
  cp "$F" gpurun_out/ks_${V}_kernel_stats.csv
  echo "== $V"
  python3 - $F <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if int(r["Calls"])>=10 and "synth" not in r["Name"]:
        print("  %-78s calls %4s avg %9.1f us" % (r["Name"][:78], r["Calls"], float(r["AverageNs"])/1000))
PY
done
