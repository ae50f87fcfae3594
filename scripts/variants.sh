#!/bin/bash
# GPU-box helper: the same bench case against several library variants (scripts/build_variant.sh):
#   scripts/variants.sh "<bench args>" <variant> ...      ("base" = the in-tree library)
ARGS="$1"; shift
for V in "$@"; do
  if [ "$V" = base ]; then unset TRAG_LIB; else export TRAG_LIB=$GRAFT_REPO_ROOT/trueno-rag_amd/lib_var/$V/libtrag.so; fi
  timeout -k 10 150 python bench.py --no-cpu-baseline --no-pmc --steps 30 $ARGS > gpurun_out/v_$V.json 2> gpurun_out/v_$V.err || { echo "$V FAILED"; tail -3 gpurun_out/v_$V.err; exit 1; }
  python3 -c "
import json; r=json.load(open('gpurun_out/v_$V.json')); m=r.get('with_bf16_mirror') or {}; print('%-8s ms/step %.3f scan %.3f | mirror %.3f scan %.3f' % ('$V', r['ms_per_step'], r['roofline']['kernel_avg_ms'], m.get('ms_per_step',0), m.get('scan_kernel_avg_ms',0)), r['parity']['uncertified_queries'])"
done
