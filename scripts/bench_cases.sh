#!/bin/bash
# Runs a list of bench cases and prints one line per case: name, ms/step, dominant kernel ms.
# usage: scripts/bench_cases.sh "<name>|<bench args>" ...
for C in "$@"; do
  NAME="${C%%|*}"; ARGS="${C#*|}"
  timeout -k 10 150 python bench.py --no-cpu-baseline --no-pmc --steps 10 $ARGS > gpurun_out/x.json 2> gpurun_out/x.err || { echo "$NAME FAILED"; tail -3 gpurun_out/x.err; exit 1; }
  python3 -c "
import json; r=json.load(open('gpurun_out/x.json')); print('$NAME', round(r['ms_per_step'],3), round(r['roofline']['kernel_avg_ms'],3), r.get('parity'))"
done
