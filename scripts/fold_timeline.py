#!/usr/bin/env python3
"""Timeline of CHAD_FLAG_DAG_ONLY roll-overs from a rocprofv3 --kernel-trace of scripts/cfg5_dagonly.py: per roll-over the span of the
closing chain (k_table_keys .. k_dag_small_levels), its kernels' own time, the gaps between them, and the frames' update kernels
that started meanwhile.   usage: fold_timeline.py <kernel_trace.csv> [--detail N]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
detail = int(sys.argv[sys.argv.index("--detail") + 1]) if "--detail" in sys.argv else -1
def nm(r): return r["Kernel_Name"].split("(")[0].replace("chadhip::", "").replace("void ", "")
for r in rows: r["s"], r["e"], r["n"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), nm(r)
rows.sort(key=lambda r: r["s"])
starts = [i for i, r in enumerate(rows) if r["n"].startswith("k_table_keys")]
ends = [i for i, r in enumerate(rows) if r["n"].startswith("k_dag_small_levels")]
upd = [r for r in rows if r["n"].startswith("k_update")]
print(f"{len(starts)} roll-overs, {len(upd)} frames; trace span {(rows[-1]['e'] - rows[0]['s']) / 1e6:.2f} ms")
prev_end = None
for k, i0 in enumerate(starts):
    i1 = next((j for j in ends if j > i0), None)
    if i1 is None: break
    q = rows[i0]["Queue_Id"]
    chain = [r for r in rows[i0:i1 + 1] if r["Queue_Id"] == q and (r["n"].startswith("k_dag") or r["n"].startswith("k_table") or r["n"].startswith("k_sorted") or "IndexPayload" in r["n"] or r["n"].startswith("k_radix_h") or r["n"].startswith("k_radix_b") or "rocclr" in r["n"])]
    t0, t1 = rows[i0]["s"], rows[i1]["e"]
    busy = sum(r["e"] - r["s"] for r in chain)
    frames = sum(1 for r in upd if t0 <= r["s"] <= t1)
    since = f", {(t0 - prev_end) / 1e3:7.0f} us after the previous chain" if prev_end else ""
    print(f"roll-over {k}: chain {(t1 - t0) / 1e3:7.0f} us, {len(chain)} launches busy {busy / 1e3:7.0f} us, {frames} frame updates started meanwhile{since}")
    prev_end = t1
    if k == detail:
        last = None
        for r in rows[i0:i1 + 1]:
            if r["Queue_Id"] != q: continue
            gap = (r["s"] - last) / 1e3 if last else 0.0
            print(f"   {(r['s'] - t0) / 1e3:8.1f} us  dur {(r['e'] - r['s']) / 1e3:7.1f}  gap {gap:7.1f}  {r['n'][:40]}")
            last = r["e"]
# the frames' own pace: time between successive update kernels' starts
gaps = sorted((b["s"] - a["s"]) / 1e3 for a, b in zip(upd, upd[1:]))
if gaps: print(f"update-to-update: median {gaps[len(gaps) // 2]:.0f} us, p90 {gaps[len(gaps) * 9 // 10]:.0f} us, max {gaps[-1]:.0f} us")
