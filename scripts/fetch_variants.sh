#!/bin/bash
# GPU-box helper: FETCH_SIZE (bytes = KiB x 2048 on gfx950) and duration of one kernel for several variants:
#   scripts/fetch_variants.sh "<bench args>" <kernel pattern> <variant> ...
ARGS="$1"; PAT="$2"; shift 2
for V in "$@"; do
  if [ "$V" = base ]; then unset TRAG_LIB; else export TRAG_LIB=$GRAFT_REPO_ROOT/trueno-rag_amd/lib_var/$V/libtrag.so; fi
  bash $GRAFT_REPO_ROOT/scripts/pmc_case.sh f_$V "FETCH_SIZE GRBM_GUI_ACTIVE" $ARGS > /dev/null || { echo "$V FAILED"; exit 1; }
  python3 - $GRAFT_REPO_ROOT/gpurun_out/pmc_f_$V.csv "$PAT" "$V" <<'PY'
import csv,sys
d={}
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["kernel"]: d[r["counter"]]=float(r["mean"])
if d: print("%-8s fetch %.3f GB  gui_cycles %.0f" % (sys.argv[3], d["FETCH_SIZE"]*2048/1e9, d["GRBM_GUI_ACTIVE"]/8))
PY
done
