"""Per-launch timeline of the LAST step of a rocprofv3 --kernel-trace csv: start offset, duration, queue, kernel (what overlaps what).
   python tools/timeline.py <kernel_trace.csv> [anchor substring = first kernel of a step]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
anchor = sys.argv[2] if len(sys.argv) > 2 else "layer1_fused"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
# the last step = from a little before the last-but-one group of anchors to the last anchor group
starts = [i for k, i in enumerate(idx) if k == 0 or int(rows[i]["Start_Timestamp"]) - int(rows[idx[k - 1]]["Start_Timestamp"]) > 1_000_000]
a, b = (starts[-2], starts[-1]) if len(starts) > 1 else (starts[-1], len(rows))
a = max(0, a - 6)
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f us  +%8.1f us  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"][:90]))
