"""Turns one tools/profile_round.sh output directory (gpurun_out/<tag>/) into the small files
kept under profiles/:

  profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary of `bench.py`
  profiles/<tag>_pmc_summary.json   per-kernel averages of every counter pass
  profiles/<tag>_bench.json         the bench line of the same tree
  profiles/filip_sim_traffic.json   HBM bytes per launch of the dominant kernel (read by bench.py)

HBM traffic follows the gfx950 rules: FETCH_SIZE and WRITE_SIZE come from separate passes,
are in KiB, and FETCH_SIZE is doubled (it tallies 128-B requests of wide coalesced reads at
64 B); WRITE_SIZE is taken as is.

usage: python3 tools/summarise_profiles.py <tag>
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r1"
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)


def short(name):
    name = name.split("(")[0]
    return name.replace("void ", "").strip()


stats = sorted(glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv")), key=os.path.getmtime)[-1:]
if stats:
    rows = list(csv.DictReader(open(stats[0])))
    with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in rows:
            w.writerow([short(r["Name"])[:80], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"],
                        r["StdDev"]])

pmc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(src, "pmc_*", "*", "*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if k.startswith("at::") or k.startswith("__amd"):
            continue
        pmc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        pmc[k]["_vgpr"] = [float(r["VGPR_Count"])]
        pmc[k]["_lds_bytes"] = [float(r["LDS_Block_Size"])]
        pmc[k]["_grid"] = [float(r["Grid_Size"])]
        pmc[k]["_wg"] = [float(r["Workgroup_Size"])]
summary = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in pmc.items()}
for k, d in summary.items():
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        d["hbm_bytes_per_launch"] = (2.0 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024.0
    if "TCC_HIT_sum" in d and d["TCC_HIT_sum"] + d.get("TCC_MISS_sum", 0) > 0:
        d["l2_hit_rate"] = d["TCC_HIT_sum"] / (d["TCC_HIT_sum"] + d["TCC_MISS_sum"])
with open(os.path.join(dst, f"{tag}_pmc_summary.json"), "w") as f:
    json.dump(summary, f, indent=1, sort_keys=True)

bench = os.path.join(src, "bench.json")
if os.path.exists(bench):
    shutil.copy(bench, os.path.join(dst, f"{tag}_bench.json"))

for k, d in summary.items():
    if k.startswith("filip_sim") and "hbm_bytes_per_launch" in d:
        with open(os.path.join(dst, "filip_sim_traffic.json"), "w") as f:
            json.dump({"kernel": k, "source": f"profiles/{tag}_pmc_summary.json", "fetch_size_kib": d["FETCH_SIZE"],
                       "write_size_kib": d["WRITE_SIZE"], "hbm_bytes_per_launch": d["hbm_bytes_per_launch"],
                       "rule": "2*FETCH_SIZE + WRITE_SIZE, KiB -> bytes (gfx950: FETCH_SIZE counts 128-B requests as 64 B)"}, f, indent=1)
        print(k, "HBM bytes per launch: %.1f MB" % (d["hbm_bytes_per_launch"] / 1e6))
print("wrote", sorted(os.listdir(dst)))
