#!/usr/bin/env python3
"""Turns the rocprofv3 output of scripts/collect_profiles.sh into the committed summaries under profiles/:
  profiles/<tag>_kernel_stats.csv      rocprofv3 --kernel-trace --stats of `bench.py --steps 20 --warmup 3`
  profiles/<tag>_pmc_per_kernel.json   FETCH_SIZE / WRITE_SIZE / TCC counters, mean per launch
  profiles/pmc_update_kernel.json      HBM traffic per launch of the update kernel (read by bench.py -> roofline.traffic)
HBM correction (MI355X_MICROARCH.md, "HBM"): FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of
the bytes of a wide (16 B/lane) coalesced stream, so reads are doubled; WRITE_SIZE is exact for 16-B stores and float
atomics.  The update kernel's reads are float4 streams (points, normals) -> doubled."""
import collections
import csv
import glob
import json
import os
import shutil
import sys


def newest(paths):
    """the most recently written of several runs' files (a directory merged from two collections holds both)"""
    return sorted(paths, key=os.path.getmtime)

src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof = os.path.join(root, "profiles")
os.makedirs(prof, exist_ok=True)

stats = newest(glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True))[-1]
shutil.copy(stats, os.path.join(prof, f"{tag}_kernel_stats.csv"))

per_kernel = collections.defaultdict(dict)
for group in ("fetch", "write", "tcc", "rdreq"):
    found = newest(glob.glob(os.path.join(src, group, "**", "*counter_collection.csv"), recursive=True))
    if not found:
        continue
    f = found[-1]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("chadhip::", "").replace("void ", "")
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for name, counters in acc.items():
        for c, vals in counters.items():
            per_kernel[name][c] = sum(vals) / len(vals)
            per_kernel[name]["launches_sampled"] = len(vals)
json.dump(per_kernel, open(os.path.join(prof, f"{tag}_pmc_per_kernel.json"), "w"), indent=1, sort_keys=True)

durations = {}
for r in csv.DictReader(open(stats)):
    durations[r["Name"].split("(")[0].replace("chadhip::", "").replace("void ", "")] = float(r["AverageNs"])
upd = [k for k in per_kernel if k.startswith("k_update_tiled<false")][0]
u = per_kernel[upd]
traffic = (2.0 * u["FETCH_SIZE"] + u["WRITE_SIZE"]) * 1024.0
# exact read bytes from the L2's memory-side read requests by size (32 / 64 / 128 bytes), where that pass was taken
read_exact = None
if "TCC_EA0_RDREQ_128B_sum" in u:
    read_exact = 32.0 * u["TCC_EA0_RDREQ_32B_sum"] + 64.0 * u["TCC_EA0_RDREQ_64B_sum"] + 128.0 * u["TCC_EA0_RDREQ_128B_sum"]
    traffic = read_exact + u["WRITE_SIZE"] * 1024.0
try:
    import subprocess
    commit = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None
except Exception:
    commit = None
# the calibration of scripts/calib_fetch.hip (256 MiB read once per kernel): what FETCH_SIZE and the request counters report of it
calib = {}
for group, label in (("calib_fetch", "FETCH_SIZE"), ("calib_rdreq", "rdreq")):
    found = newest(glob.glob(os.path.join(src, group, "**", "*counter_collection.csv"), recursive=True))
    if not found:
        continue
    acc = collections.defaultdict(dict)
    for r in csv.DictReader(open(found[-1])):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[name][r["Counter_Name"]] = float(r["Counter_Value"])
    for name, c in acc.items():
        if not name.startswith("k_"):
            continue
        if label == "FETCH_SIZE":
            calib.setdefault(name, {})["FETCH_SIZE_over_bytes"] = round(c.get("FETCH_SIZE", 0.0) * 1024.0 / (256 << 20), 4)
        else:
            b = 32.0 * c.get("TCC_EA0_RDREQ_32B_sum", 0) + 64.0 * c.get("TCC_EA0_RDREQ_64B_sum", 0) + 128.0 * c.get("TCC_EA0_RDREQ_128B_sum", 0)
            calib.setdefault(name, {})["request_bytes_over_bytes"] = round(b / (256 << 20), 4)
json.dump({
    "kernel": upd, "tag": tag, "measured_on_commit": commit, "calibration_256MiB_read_once": calib,
    "read_bytes_from_request_sizes": read_exact,
    "command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE | TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum (separate passes) --kernel-trace -- python3 bench.py --steps 6 --warmup 2 --cpu-seconds 0",
    "fetch_size_kib_raw": u["FETCH_SIZE"], "write_size_kib": u["WRITE_SIZE"],
    "correction": "reads = 32 x RDREQ_32B + 64 x RDREQ_64B + 128 x RDREQ_128B where that pass was taken (exact: see calibration), else FETCH_SIZE doubled "
                  "(gfx950 FETCH_SIZE counts 128-B requests as 64 B: right for coalesced 16 B/lane streams, an overestimate for this kernel's lane-strided loads); writes as reported",
    "hbm_bytes_per_launch": round(traffic),
    "atomic_requests_per_launch": u.get("TCC_EA0_ATOMIC_sum"),
    "l2_hit_rate": u["TCC_HIT_sum"] / (u["TCC_HIT_sum"] + u["TCC_MISS_sum"]),
    "kernel_trace_avg_us": durations.get(upd, 0) / 1e3,
}, open(os.path.join(prof, "pmc_update_kernel.json"), "w"), indent=1)
print(open(os.path.join(prof, "pmc_update_kernel.json")).read())
for k, v in sorted(durations.items(), key=lambda kv: -kv[1])[:8]:
    print(f"{k:32s} {v / 1e3:8.2f} us")

# cfg5 (scripts/outdoor_bench.py): the same two summaries, if collect_profiles.sh has taken them
if glob.glob(os.path.join(src, "cfg5_trace", "**", "*kernel_stats.csv"), recursive=True):
    stats5 = newest(glob.glob(os.path.join(src, "cfg5_trace", "**", "*kernel_stats.csv"), recursive=True))[-1]
    shutil.copy(stats5, os.path.join(prof, f"{tag}_cfg5_kernel_stats.csv"))
    per5 = collections.defaultdict(dict)
    for group in ("cfg5_fetch", "cfg5_write", "cfg5_tcc", "cfg5_rdreq"):
        found = newest(glob.glob(os.path.join(src, group, "**", "*counter_collection.csv"), recursive=True))
        if not found:
            continue
        f = found[-1]
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("chadhip::", "").replace("void ", "")
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for name, counters in acc.items():
            for c, vals in counters.items():
                per5[name][c] = sum(vals) / len(vals)
                per5[name]["launches_sampled"] = len(vals)
    json.dump(per5, open(os.path.join(prof, f"{tag}_cfg5_pmc_per_kernel.json"), "w"), indent=1, sort_keys=True)
    print("cfg5:")
    for r in csv.DictReader(open(stats5)):
        print(f'{r["Name"].split("(")[0][:60]:60s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"]) / 1e3:8.2f} us  {r["Percentage"]:>6s}%')
    u5 = [k for k in per5 if k.startswith("k_update_tiled<false")]
    if u5:
        c5 = per5[u5[0]]
        if "TCC_EA0_RDREQ_128B_sum" in c5:   # bytes read, from the request sizes (the same arithmetic as cfg2's traffic figure)
            c5["read_bytes_from_request_sizes"] = 32 * c5.get("TCC_EA0_RDREQ_32B_sum", 0) + 64 * c5.get("TCC_EA0_RDREQ_64B_sum", 0) + 128 * c5.get("TCC_EA0_RDREQ_128B_sum", 0)
            if "WRITE_SIZE" in c5:
                c5["hbm_bytes_per_launch"] = c5["read_bytes_from_request_sizes"] + c5["WRITE_SIZE"] * 1024   # (WRITE_SIZE in KiB)
        json.dump(per5, open(os.path.join(prof, f"{tag}_cfg5_pmc_per_kernel.json"), "w"), indent=1, sort_keys=True)
        print("cfg5 update kernel counters per launch:", json.dumps(c5))

# the same trajectory kept as its DAG only (scripts/cfg5_dagonly.py under the kernel trace)
dag = newest(glob.glob(os.path.join(src, "cfg5dag_trace", "**", "*kernel_stats.csv"), recursive=True))
if dag:
    shutil.copy(dag[-1], os.path.join(prof, f"{tag}_cfg5_dagonly_kernel_stats.csv"))
xchg = newest(glob.glob(os.path.join(src, "cfg5xchg_trace", "**", "*kernel_stats.csv"), recursive=True))
if xchg:
    shutil.copy(xchg[-1], os.path.join(prof, f"{tag}_cfg5_exchange_kernel_stats.csv"))
