"""Writes profiles/<tag>.txt from the rocprofv3 output of tools/profile_primary.sh (k_primary_hits
over the camera rays of a 4096 x 4096 image): per kernel the kernel-trace statistics, the PMC
means per launch, and what follows from them - instructions per wave and per ray-surface test,
SIMD cycles per vector instruction, the fraction of the FP32 vector peak.  Appends the output of
tools/micro/scan_body (the scan in steady state) when given.

    python tools/make_primary_summary.py gpurun_out/prof_<tag> <tag> [scan_body.log]
"""
import collections
import csv
import glob
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PEAK = 157.3e12
RAYS = 4096 * 4096
FLOPS = RAYS * (5 * 63 + 3 * 18)          # cornell_2: 5 triangles, 3 spheres

src, tag = sys.argv[1], sys.argv[2]
lines = ["rocprofv3 summary '%s' (tools/profile_primary.sh: python3 tools/primary_hits_run.py cornell_2 4096," % tag,
         "k_primary_hits over the 16 777 216 camera rays of a 4096 x 4096 image of cornell_2 (5 triangles + 3 spheres),",
         "3 launches per numeric policy, each with the rays per lane the library chooses for a query of this size (N rays per",
         "lane: records broadcast from LDS into vector registers))", ""]


def newest(pattern):
    by_dir = {}
    for f in glob.glob(pattern):
        d = os.path.dirname(os.path.dirname(f))
        if d not in by_dir or os.path.getmtime(f) > os.path.getmtime(by_dir[d]):
            by_dir[d] = f
    return sorted(by_dir.values())


stats = {}
lines.append("== kernel-trace --stats ==")
for f in newest(src + "/stats/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        lines.append("%-72s calls=%-4s avg_ns=%-12s min_ns=%-10s max_ns=%s" % (
            r["Name"][:72], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"]))
        if "primary_hits" in r["Name"]:
            stats[r["Name"]] = float(r["AverageNs"])
for f in newest(src + "/stats/*/*_kernel_trace.csv"):
    seen = set()
    for r in csv.DictReader(open(f)):
        if "primary_hits" in r["Kernel_Name"] and r["Kernel_Name"] not in seen:
            seen.add(r["Kernel_Name"])
            lines.append("dispatch %s: VGPR_Count=%s (rocprofv3 halves the allocation for wave64) SGPR_Count=%s "
                         "LDS_Block_Size=%s Scratch_Size=%s Grid=%s Workgroup=%s" % (
                             r["Kernel_Name"][:60], r["VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"],
                             r["Scratch_Size"], r["Grid_Size_X"], r["Workgroup_Size_X"]))

agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in newest(src + "/pmc_*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "primary_hits" in r["Kernel_Name"]:
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name in sorted(agg):
    m = {k: sum(v) / len(v) for k, v in agg[name].items()}
    policy = "FAST" if re.search(r"<true", name) else "EXACT"
    per_lane = re.search(r"multi<\w+, (\d+)", name)
    n = int(per_lane.group(1)) if per_lane else 1
    lines += ["", "== PMC means per launch, %s: %s (separate passes) ==" % (policy, name[:70])]
    for k in sorted(m):
        lines.append("  %-26s %.5g" % (k, m[k]))
    waves = m.get("SQ_WAVES")
    if waves and "SQ_INSTS_VALU" in m and "GRBM_GUI_ACTIVE" in m:
        cycles = m["GRBM_GUI_ACTIVE"] / 8
        valu = m["SQ_INSTS_VALU"] / waves / n
        ms = stats.get(name, 0.0) / 1e6
        lines.append("  derived: %d ray(s) per lane; per wave and ray (= per 64 rays x 8 surfaces): %.1f vector, %.1f scalar "
                     "instructions" % (n, valu, m.get("SQ_INSTS_SALU", 0) / waves / n))
        lines.append("           %.1f vector instructions per ray-surface test; lane utilisation %.3f" % (
            valu / 8, m["SQ_THREAD_CYCLES_VALU"] / 64 / m["SQ_ACTIVE_INST_VALU"] if "SQ_THREAD_CYCLES_VALU" in m else float("nan")))
        lines.append("           %.0f shader cycles per launch (GRBM_GUI_ACTIVE / 8): %.2f SIMD-cycles per vector instruction "
                     "over 1024 SIMDs" % (cycles, cycles * 1024 / m["SQ_INSTS_VALU"]))
        if ms:
            lines.append("           algorithmic %.3e flop / %.4f ms = %.1f TFLOP/s = %.3f of 157.3" % (
                FLOPS, ms, FLOPS / ms / 1e9, FLOPS / (ms * 1e-3) / PEAK))
if len(sys.argv) > 3 and os.path.exists(sys.argv[3]):
    lines += ["", "== the scan in steady state (tools/micro/scan_body: nothing but the scan in the loop, rays with origins of",
              "their own; N = 0: one ray per lane, records through the scalar cache, as the render kernels scan) =="]
    lines += [l.rstrip() for l in open(sys.argv[3]) if l.strip()]
out = os.path.join(ROOT, "profiles", tag + ".txt")
with open(out, "w") as f:
    f.write("\n".join(lines) + "\n")
print("wrote", out)
