set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
(cat /sys/fs/cgroup/cpu.max; nproc; cat /proc/self/cgroup | head -3; python -c "import os;print(len(os.sched_getaffinity(0)))") > gpurun_out/cpuinfo.txt 2>&1
rm -rf gpurun_out/prof_c4 gpurun_out/prof_c3
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c4 -o s -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side-modes > gpurun_out/prof_c4.log 2>&1 && \
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c3 -o s -- python3 bench.py --config 3 --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/prof_c3.log 2>&1 && \
python3 - <<'PY' > gpurun_out/cpub.json 2> gpurun_out/cpub.err
import sys, json; sys.path.insert(0, '.')
import bench
from dani_slam_amd import synth
g, b = synth.bench_frames(range(8))
print(json.dumps(bench.cpu_baseline(g, b, synth.bench_weights())))
PY
ls gpurun_out/prof_c4 gpurun_out/prof_c3; find gpurun_out/prof_c4 gpurun_out/prof_c3 -name "*kernel_trace.csv" -size +20M -delete; tail -2 gpurun_out/prof_c3.log
