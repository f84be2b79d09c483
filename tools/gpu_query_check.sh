#!/bin/bash
# Parity of the closest-hit query kernels at every rays-per-lane setting, then their timings.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
K='primary_hits or intersect or known_answers or extreme or soak_of_random_scenes'
for n in auto 4 8 2 1; do
    if [ $n = auto ]; then unset PT_QUERY_RAYS; else export PT_QUERY_RAYS=$n; fi
    timeout -k 10 420 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 180 --timeout-method thread -k "$K" \
        > gpurun_out/query_parity_$n.log 2>&1
    rc=$?
    tail -2 gpurun_out/query_parity_$n.log
    [ $rc -ne 0 ] && { echo "parity failed at PT_QUERY_RAYS=$n (rc $rc)"; tail -30 gpurun_out/query_parity_$n.log; exit 1; }
done
unset PT_QUERY_RAYS
timeout -k 10 300 python -u tools/exp_query.py 4096 > gpurun_out/exp_query.log 2>&1 || { tail -20 gpurun_out/exp_query.log; exit 1; }
cat gpurun_out/exp_query.log
