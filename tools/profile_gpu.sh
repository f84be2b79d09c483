#!/bin/bash
# Profiles bench.py on the GPU box with rocprofv3: one kernel-trace/stats run and
# separate PMC passes (counters are never combined with a trace).  Output goes to
# gpurun_out/prof_<tag>/; copy the summaries worth keeping into profiles/.
#   tools/profile_gpu.sh <tag> [bench.py arguments]
set -u
TAG=${1:-run}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --cpu-spp 0 $*"

run() {  # name, rocprofv3 options...
    local name=$1; shift
    timeout -s KILL 240 rocprofv3 "$@" --output-format csv -d "$OUT/$name" -- python3 "$ROOT/bench.py" $ARGS \
        > "$OUT/$name.log" 2>&1
    local rc=$?
    echo "$name rc=$rc"
    return $rc
}

run stats --kernel-trace --stats || exit 1
run pmc_sq1 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY || exit 1
run pmc_sq2 --pmc SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_WAIT_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE || exit 1
run pmc_fetch --pmc FETCH_SIZE || exit 1
run pmc_write --pmc WRITE_SIZE || exit 1
find "$OUT" -name "*.csv" | head -40
