#!/bin/bash
# A/B of two builds of the library on one bench configuration:
#   tools/ab.sh "<bench.py arguments>" <lib A> <lib B> ...
# prints ms per step and the render kernel's average duration for each.
args=$1; shift
for lib in "$@"; do
    PT_AMD_LIB=$PWD/$lib timeout -k 10 180 python bench.py $args --cpu-spp 0 > gpurun_out/ab_one.json 2> gpurun_out/ab.err || exit 1
    python - "$lib" "$args" <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab_one.json"))
print("%-46s %-48s %8.3f ms/step  kernel %s ms" % (sys.argv[1], sys.argv[2], d["ms_per_step"], d["roofline"].get("kernel_ms")), flush=True)
PY
done
