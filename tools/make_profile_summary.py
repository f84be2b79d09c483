"""Writes profiles/<tag>.txt from a rocprofv3 output directory made by
tools/profile_gpu.sh: the --kernel-trace --stats table, per-dispatch register
use, PMC means for the render kernel, and HBM traffic per launch corrected as
the gfx950 notes prescribe (FETCH_SIZE counts half of a read stream's bytes:
doubled; WRITE_SIZE taken as is; both in KiB).  With --traffic it also writes
profiles/pmc_traffic.json, which bench.py reports as roofline.traffic.

   python tools/make_profile_summary.py gpurun_out/prof_<tag> <tag> [kernel-substring] [--traffic]
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def newest(pattern):
    """One file per pass directory: gpurun merges every run's output into the same local
    directories, so an earlier run's files may still lie beside the latest one's."""
    by_dir = {}
    for f in glob.glob(pattern):
        d = os.path.dirname(os.path.dirname(f))
        if d not in by_dir or os.path.getmtime(f) > os.path.getmtime(by_dir[d]):
            by_dir[d] = f
    return sorted(by_dir.values())


src, tag = sys.argv[1], sys.argv[2]
rest = [a for a in sys.argv[3:] if not a.startswith("--")]
kern = rest[0] if rest else "render_mega"
lines = ["rocprofv3 summary '%s' (tools/profile_gpu.sh: python3 bench.py --steps 2 --warmup 1 ...)" % tag, ""]

lines.append("== kernel-trace --stats (all kernels) ==")
for f in newest(src + "/stats/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        lines.append("%-70s calls=%-5s avg_ns=%-14s min_ns=%-12s max_ns=%-12s pct=%s" % (
            r["Name"][:70], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"], r["Percentage"]))
avg_ns = None
for f in newest(src + "/stats/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if kern in r["Name"]:
            avg_ns = float(r["AverageNs"])
lines.append("")
lines.append("== dispatch of *%s* ==" % kern)
for f in newest(src + "/stats/*/*_kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            lines.append("VGPR_Count=%s Accum_VGPR_Count=%s SGPR_Count=%s LDS_Block_Size=%s Scratch_Size=%s "
                         "Workgroup=%s Grid=%s" % (r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"],
                                                   r["LDS_Block_Size"], r["Scratch_Size"], r["Workgroup_Size_X"],
                                                   r["Grid_Size_X"]))
            break
agg = collections.defaultdict(list)
for f in newest(src + "/pmc_*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
lines.append("")
lines.append("== PMC means per launch of *%s* (separate passes) ==" % kern)
mean = {k: sum(v) / len(v) for k, v in agg.items()}
for k in sorted(mean):
    lines.append("%-28s %.6g   (n=%d)" % (k, mean[k], len(agg[k])))
if "FETCH_SIZE" in mean and "WRITE_SIZE" in mean:
    hbm = (2 * mean["FETCH_SIZE"] + mean["WRITE_SIZE"]) * 1024
    lines.append("")
    lines.append("HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 = %.4g" % hbm)
    if avg_ns:
        lines.append("  over the kernel's %.3f ms: %.2f GB/s of 8000 GB/s" % (avg_ns / 1e6, hbm / avg_ns))
    if "--traffic" in sys.argv:
        sys.path.insert(0, ROOT)
        from bench import kernel_source_hash
        # kernel_ms and the hash of the device sources tie the counters to the kernel they were
        # collected on: bench.py reports them only for that kernel
        rec = {"hbm_bytes_per_launch": hbm, "fetch_size_kib": mean["FETCH_SIZE"],
               "write_size_kib": mean["WRITE_SIZE"], "kernel": kern, "source": tag + ".txt",
               "kernel_ms": avg_ns / 1e6 if avg_ns else None, "source_sha1": kernel_source_hash()}
        if "SQ_INSTS_VALU" in mean and "GRBM_GUI_ACTIVE" in mean:
            rec["valu_insts_per_simd_cycle"] = mean["SQ_INSTS_VALU"] / (1024 * mean["GRBM_GUI_ACTIVE"] / 8)
        if "SQ_THREAD_CYCLES_VALU" in mean and "SQ_ACTIVE_INST_VALU" in mean:
            rec["valu_lane_utilisation"] = mean["SQ_THREAD_CYCLES_VALU"] / (64 * mean["SQ_ACTIVE_INST_VALU"])
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w") as f:
            json.dump(rec, f)
if "SQ_INSTS_VALU" in mean and "SQ_THREAD_CYCLES_VALU" in mean and "SQ_ACTIVE_INST_VALU" in mean:
    lines.append("VALU lane utilisation = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU) = %.3f" % (
        mean["SQ_THREAD_CYCLES_VALU"] / (64 * mean["SQ_ACTIVE_INST_VALU"])))
if "GRBM_GUI_ACTIVE" in mean and avg_ns:
    lines.append("clock = GRBM_GUI_ACTIVE / 8 / kernel time = %.2f GHz" % (mean["GRBM_GUI_ACTIVE"] / 8 / avg_ns))
if "SQ_INSTS_VALU" in mean and avg_ns and "GRBM_GUI_ACTIVE" in mean:
    cyc = mean["GRBM_GUI_ACTIVE"] / 8
    lines.append("VALU instructions per SIMD-cycle = SQ_INSTS_VALU / (1024 SIMDs x cycles) = %.3f" % (
        mean["SQ_INSTS_VALU"] / (1024 * cyc)))
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
out = os.path.join(ROOT, "profiles", tag + ".txt")
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
