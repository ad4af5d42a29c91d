"""Print the dispatches of the LAST repetition found in a rocprofv3 kernel-trace CSV, in start order, with gaps."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = sys.argv[2] if len(sys.argv) > 2 else "yolo_preproc"
idx = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
seg = rows[idx[-1]:]
t0 = int(seg[0]["Start_Timestamp"]); prev_end = t0
tot = 0
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    g = f"{r['Grid_Size_X']}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']}/{r['Workgroup_Size_X']}" if "Grid_Size_X" in r else ""
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:6.1f}  {g:22s} {r['Kernel_Name'][:100]}")
    prev_end = max(prev_end, e); tot += e - s
print(f"span {(prev_end - t0) / 1e3:.1f} us, busy {tot / 1e3:.1f} us, {len(seg)} dispatches")
