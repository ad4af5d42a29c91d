"""Timeline of the LAST front-end step in a rocprofv3 kernel-trace CSV: merged busy intervals, idle gaps > 15 us,
and what ran on either side of each gap (two streams may overlap, so busy time is the union of intervals)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "yolo_preproc" in r["Kernel_Name"]]
lo = idx[-2]; hi = idx[-1]          # one full step: from the second-last preproc to the last
seg = rows[lo:hi]
t0 = int(seg[0]["Start_Timestamp"])
cur_end = t0; busy = 0; last = seg[0]
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s > cur_end:
        if s - cur_end > 15000:
            print(f"{(cur_end - t0) / 1e3:9.1f} us  idle {(s - cur_end) / 1e3:7.1f} us   after {last['Kernel_Name'][:50]:50s} before {r['Kernel_Name'][:50]}")
        busy += 0
        cur_start = s
    if e > cur_end:
        busy += e - max(s, cur_end)
        cur_end = e; last = r
print(f"step span {(int(rows[hi]['Start_Timestamp']) - t0) / 1e3:.1f} us, GPU busy (union) {busy / 1e3:.1f} us, {len(seg)} dispatches")
