#!/bin/bash
# Sweep: how many of YOLO's first stages are enqueued ahead of SuperPoint's MFMA layers (DANI_YOLO_HEAD; 0 = SuperPoint first).
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
: > gpurun_out/sweep_head.txt
for h in 0 1 2 3 4 5 0 3; do
  DANI_YOLO_HEAD=$h timeout -k 10 200 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-side-modes > gpurun_out/sw.json 2> gpurun_out/sw.err || { echo "head $h FAILED" >> gpurun_out/sweep_head.txt; tail -3 gpurun_out/sw.err >> gpurun_out/sweep_head.txt; continue; }
  python3 - "$h" >> gpurun_out/sweep_head.txt <<'PY'
import json, sys
d = json.load(open('gpurun_out/sw.json'))
s = d['stages_ms_per_step']
print('head', sys.argv[1], 'fps', d['value'], 'median', d['step_ms']['median'], 'phase1', s['phase1_enqueue_sp_network_then_wait_for_yolo_boxes'],
      'phase2', s['phase2_sp_candidates_sync_quadtree_descriptors_enqueue'], 'lg', s['phase3_lightglue_enqueue_and_final_sync'], 'spconv', s['sp_conv_stack_device_ms_sum_of_12_layers'])
PY
done
cat gpurun_out/sweep_head.txt
