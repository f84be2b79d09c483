"""Summarises rocprofv3 CSV output of tools/profile_gpu.sh for one kernel.
   python tools/pmc_summary.py gpurun_out/prof_<tag> [kernel-substring]"""
import collections
import csv
import glob
import sys

root = sys.argv[1]
kern = sys.argv[2] if len(sys.argv) > 2 else "render_mega"
for f in sorted(glob.glob(root + "/stats/*/*_kernel_stats.csv")):
    for r in csv.DictReader(open(f)):
        if kern in r["Name"]:
            print("kernel_stats: calls=%s avg_ns=%s min_ns=%s max_ns=%s  %s" % (
                r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"], r["Name"][:60]))
agg = collections.defaultdict(list)
meta = {}
for f in sorted(glob.glob(root + "/pmc_*/*/*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta = {k: r[k] for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size",
                                      "Scratch_Size", "Grid_Size", "Workgroup_Size")}
print("dispatch:", meta)
for k in sorted(agg):
    v = agg[k]
    print("%-28s n=%d mean=%.6g" % (k, len(v), sum(v) / len(v)))
