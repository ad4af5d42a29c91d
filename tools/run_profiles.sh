#!/bin/bash
# Collect the rocprofv3 evidence behind bench.py's roofline object (run ON the GPU box, from the repo root):
#   1. --kernel-trace --stats of the default bench command           -> gpurun_out/prof_stats/
#   2. three separate --pmc passes (kernel-trace only, no other trace domains: the pool refuses those combinations)
#      pmc1 SQ/GRBM cycle counters, pmc2 FETCH_SIZE, pmc3 WRITE_SIZE  -> gpurun_out/pmc{1,2,3}/
# then `python tools/summarize_pmc.py <tag> <kernel-substring>` (here or back home) writes profiles/.
set -e -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_stats gpurun_out/pmc1 gpurun_out/pmc2 gpurun_out/pmc3
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -o s -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-side-modes > gpurun_out/prof_stats.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc1 -o p -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-side-modes > gpurun_out/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc2 -o p -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-side-modes > gpurun_out/pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc3 -o p -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-side-modes > gpurun_out/pmc3.log 2>&1
ls gpurun_out/prof_stats gpurun_out/pmc1 gpurun_out/pmc2 gpurun_out/pmc3
