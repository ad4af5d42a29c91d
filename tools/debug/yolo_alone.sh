#!/bin/bash
# YOLO chain alone (B = 8, no SuperPoint next to it): per-launch durations in chain order from a rocprofv3 kernel trace.
# usage: tools/debug/yolo_alone.sh [output file under gpurun_out/, default yolo_alone.txt]
OUT=${1:-yolo_alone.txt}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_yolo
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_yolo -o y -- python3 tools/debug/yolo_trace.py > gpurun_out/prof_yolo.log 2>&1
OUT=$OUT python3 - <<'PY'
import csv, glob, os
out = os.path.join('gpurun_out', os.environ.get('OUT', 'yolo_alone.txt'))
rows = list(csv.DictReader(open(glob.glob('gpurun_out/prof_yolo/*kernel_trace.csv')[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'yolo_preproc' in r['Kernel_Name']]
lo = idx[-1]
seg = rows[lo:]
t0 = int(seg[0]['Start_Timestamp'])
tot = 0
with open(out, 'w') as f:
    prev_end = None
    for r in seg:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        nm = r['Kernel_Name'].replace('void dani::', '').replace('dani::', '')[:78]
        wg = int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z']) // max(int(r['Workgroup_Size_X']), 1)
        gap = (s - prev_end) / 1e3 if prev_end else 0.0
        f.write('%8.1f dur %7.1f gap %5.1f wgs %6d %s\n' % ((s - t0) / 1e3, (e - s) / 1e3, gap, wg, nm))
        prev_end = e; tot += e - s
    f.write('chain span %.1f us, sum of kernel durations %.1f us, %d launches\n' % ((prev_end - t0) / 1e3, tot / 1e3, len(seg)))
print(open(out).read()[-400:])
PY
rm -rf gpurun_out/prof_yolo
