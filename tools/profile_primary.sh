#!/bin/bash
# rocprofv3 passes over the closest-hit kernel on its own (tools/primary_hits_run.py):
# one kernel-trace/stats run and separate PMC passes.  Output: gpurun_out/prof_<tag>/.
#   tools/profile_primary.sh <tag> [scene [side]]
set -u
TAG=${1:-primary}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp

PROG_ARGS="$*"
run() {  # name, rocprofv3 options...
    local name=$1; shift
    timeout -s KILL 120 rocprofv3 "$@" --output-format csv -d "$OUT/$name" -- python3 "$ROOT/tools/primary_hits_run.py" $PROG_ARGS \
        > "$OUT/$name.log" 2>&1
    local rc=$?
    echo "$name rc=$rc"
    return $rc
}

run stats --kernel-trace --stats || exit 1
run pmc_sq1 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY || exit 1
run pmc_sq2 --pmc SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_WAIT_ANY SQ_INST_CYCLES_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE || exit 1
cat "$OUT/stats.log" | tail -3
