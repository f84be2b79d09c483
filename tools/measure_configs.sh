#!/bin/bash
# The configurations of the design notes' table of measurements, one bench.py run
# each (on the GPU box): Msamples/s, ms per step, roofline fraction, both
# numeric policies.  Results as JSON in gpurun_out/m_<name>.json.
set -e
run() { name=$1; shift; timeout -k 10 400 python bench.py "$@" --cpu-spp 0 > gpurun_out/m_$name.json 2> gpurun_out/m_$name.err; python - "$name" <<'PY'
import json, sys
d = json.load(open("gpurun_out/m_%s.json" % sys.argv[1]))
o = d.get("other_numeric") or {}
print("%-14s %9.0f Msamples/s %9.2f ms/step seg/sample %.3f frac %.3f | other %s %.0f Msamples/s %.2f ms frac %s" % (
    sys.argv[1], d["value"], d["ms_per_step"], d["segments_per_sample"], d["roofline"]["frac"], o.get("numeric"), o.get("value", 0), o.get("kernel_ms", 0), o.get("roofline_frac")), flush=True)
PY
}
run cornell2 --steps 5 --warmup 1
run cornell2_wf --steps 3 --warmup 1 --layout wavefront
run cornell --scene cornell --size 800 800 --spp 256 --steps 5 --warmup 1
run orange --scene orange_plane --size 1600 1600 --spp 128 --steps 5 --warmup 1
run hf64 --scene heightfield --size 800 800 --spp 64 --steps 5 --warmup 2
run hf16 --scene heightfield --size 800 800 --spp 16 --steps 5 --warmup 2
run hf16_wf --scene heightfield --size 800 800 --spp 16 --steps 3 --warmup 1 --layout wavefront
run hf11_scan --scene heightfield:11 --size 800 800 --spp 64 --steps 5 --warmup 1 --accel none
run hf11 --scene heightfield:11 --size 800 800 --spp 64 --steps 5 --warmup 1
run glass800 --scene glass_stress --size 800 800 --spp 256 --steps 5 --warmup 1
run glass4k --scene glass_stress --size 3840 2160 --spp 4096 --steps 2 --warmup 1
