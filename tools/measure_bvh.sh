#!/bin/bash
# The hierarchy configurations of tools/measure_configs.sh only (quick loop while tuning
# the hierarchy kernel).
set -e
run() { name=$1; shift; timeout -k 10 400 python bench.py "$@" --cpu-spp 0 > gpurun_out/m_$name.json 2> gpurun_out/m_$name.err; python - "$name" <<'PY'
import json, sys
d = json.load(open("gpurun_out/m_%s.json" % sys.argv[1]))
o = d.get("other_numeric") or {}
print("%-14s %9.0f Msamples/s %9.2f ms/step seg/sample %.3f | other %s %.0f Msamples/s %.2f ms" % (
    sys.argv[1], d["value"], d["ms_per_step"], d["segments_per_sample"], o.get("numeric"), o.get("value", 0), o.get("kernel_ms", 0)), flush=True)
PY
}
run hf64 --scene heightfield --size 800 800 --spp 64 --steps 5 --warmup 2
run hf16 --scene heightfield --size 800 800 --spp 16 --steps 5 --warmup 2
run hf11 --scene heightfield:11 --size 800 800 --spp 64 --steps 5 --warmup 1
run hf100 --scene heightfield:100 --size 800 800 --spp 64 --steps 5 --warmup 1
