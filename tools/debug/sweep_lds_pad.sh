#!/bin/bash
# Sweep of the SuperPoint conv LDS reservation (slot for the concurrent YOLO chain): prints value / phase times per setting.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
: > gpurun_out/sweep_pad.txt
for cfg in "0 0" "24 0x02" "24 0x0E" "24 0x3E" "24 0xFFE" "30 0x0E" "30 0x3E" "30 0xFFE" "40 0xFFE"; do
  set -- $cfg
  DANI_SP_LDS_PAD_KB=$1 DANI_SP_LDS_PAD_MASK=$2 timeout -k 10 200 python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-side-modes > gpurun_out/sw.json 2> gpurun_out/sw.err || { echo "pad $1 mask $2 FAILED" >> gpurun_out/sweep_pad.txt; tail -3 gpurun_out/sw.err >> gpurun_out/sweep_pad.txt; continue; }
  python3 - "$1" "$2" >> gpurun_out/sweep_pad.txt <<'PY'
import json, sys
d = json.load(open('gpurun_out/sw.json'))
s = d['stages_ms_per_step']
print('pad', sys.argv[1], 'mask', sys.argv[2], 'fps', d['value'], 'median', d['step_ms']['median'], 'phase1', s['phase1_enqueue_sp_network_then_wait_for_yolo_boxes'],
      'lg', s['phase3_lightglue_enqueue_and_final_sync'], 'spconv', s['sp_conv_stack_device_ms_sum_of_12_layers'])
PY
done
cat gpurun_out/sweep_pad.txt
