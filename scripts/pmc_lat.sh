#!/bin/bash
# GPU-box helper: memory-latency counters of one kernel for several library variants:
#   scripts/pmc_lat.sh "<bench args>" <kernel pattern> <variant> ...
ARGS="$1"; PAT="$2"; shift 2
C="SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"
for V in "$@"; do
  if [ "$V" = base ]; then unset TRAG_LIB; else export TRAG_LIB=$GRAFT_REPO_ROOT/trueno-rag_amd/lib_var/$V/libtrag.so; fi
  bash $GRAFT_REPO_ROOT/scripts/pmc_case.sh l_$V "$C" $ARGS > /dev/null || { echo "$V FAILED"; tail -3 $GRAFT_REPO_ROOT/gpurun_out/pmc_l_$V.err; exit 1; }
  python3 - $GRAFT_REPO_ROOT/gpurun_out/pmc_l_$V.csv "$PAT" "$V" <<'PY'
import csv,sys
d={}
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["kernel"]: d[r["counter"]]=float(r["mean"])
if d:
    print("%-8s vmem_rd %.0f  vmem latency %.0f cyc  lds insts %.0f  lds latency %.0f cyc  active_vmem/wave %.3f active_lds/wave %.3f" % (
        sys.argv[3], d["SQ_INSTS_VMEM_RD"], d["SQ_INST_LEVEL_VMEM"]/max(d["SQ_INSTS_VMEM_RD"],1), d["SQ_INSTS_LDS"],
        d["SQ_INST_LEVEL_LDS"]/max(d["SQ_INSTS_LDS"],1), d["SQ_ACTIVE_INST_VMEM"]/d["SQ_WAVE_CYCLES"], d["SQ_ACTIVE_INST_LDS"]/d["SQ_WAVE_CYCLES"]))
PY
done
