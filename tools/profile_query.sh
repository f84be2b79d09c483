#!/bin/bash
# PMC passes over the closest-hit query kernel at a given rays-per-lane setting.
#   tools/profile_query.sh <tag> <rays per lane> [grid per CU]
set -u
TAG=$1; export PT_QUERY_RAYS=$2; export PT_QUERY_GRID=${3:-0}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {
    local name=$1; shift
    timeout -s KILL 120 rocprofv3 "$@" --output-format csv -d "$OUT/$name" -- python3 "$ROOT/tools/primary_hits_run.py" cornell_2 4096 \
        > "$OUT/$name.log" 2>&1
    local rc=$?
    echo "$name rc=$rc"
    return $rc
}
run stats --kernel-trace --stats || exit 1
run pmc_sq1 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY || exit 1
run pmc_sq2 --pmc SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_WAIT_ANY SQ_INST_CYCLES_SALU SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE || exit 1
run pmc_sq3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_TRANS SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM_WR SQ_WAVE_DEP_WAIT SQ_IFETCH || echo "pmc_sq3 failed (counter names)"
python3 "$ROOT/tools/pmc_summary.py" "$OUT" primary_hits
