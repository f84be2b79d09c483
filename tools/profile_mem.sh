#!/bin/bash
# Memory-pipeline PMC passes (vector L1 / texture addresser / L2) for bench.py,
# each in a run of its own.  Output: gpurun_out/prof_<tag>/mem_*.
#   tools/profile_mem.sh <tag> [bench.py arguments]
set -u
TAG=${1:-run}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --cpu-spp 0 $*"
run() {
    local name=$1; shift
    timeout -s KILL 240 rocprofv3 "$@" --output-format csv -d "$OUT/$name" -- python3 "$ROOT/bench.py" $ARGS \
        > "$OUT/$name.log" 2>&1
    local rc=$?
    echo "$name rc=$rc"
    return $rc
}
run mem_tcp1 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE || exit 1
run mem_tcp2 --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TOTAL_READ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum TA_DATA_STALLED_BY_TC_CYCLES_sum || exit 1
