#!/bin/bash
# GPU-box helper: the matrix-pipe counter pass of one bench case for several library variants:
#   scripts/pmc_variants.sh "<bench args>" <kernel pattern> <variant> ...
# prints, per variant, the kernel's busy share of the matrix pipe and the clock the counters imply.
ARGS="$1"; PAT="$2"; shift 2
C="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
for V in "$@"; do
  if [ "$V" = base ]; then unset TRAG_LIB; else export TRAG_LIB=$GRAFT_REPO_ROOT/trueno-rag_amd/lib_var/$V/libtrag.so; fi
  bash $GRAFT_REPO_ROOT/scripts/pmc_case.sh v_$V "$C" $ARGS > /dev/null || { echo "$V FAILED"; exit 1; }
  python3 - $GRAFT_REPO_ROOT/gpurun_out/pmc_v_$V.csv "$PAT" "$V" <<'PY'
import csv,sys
d={}
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["kernel"]: d[r["counter"]]=float(r["mean"])
if d:
    gui=d["GRBM_GUI_ACTIVE"]/8
    print("%-8s mfma_busy %.3f  wait_any/wave %.3f  active_inst/wave %.3f  gui_cycles %.0f" % (sys.argv[3], d["SQ_VALU_MFMA_BUSY_CYCLES"]/(gui*1024), d["SQ_WAIT_ANY"]/d["SQ_WAVE_CYCLES"], d["SQ_ACTIVE_INST_ANY"]/d["SQ_WAVE_CYCLES"], gui))
PY
done
