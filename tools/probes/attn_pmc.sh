#!/bin/bash
# PMC passes over tools/probes/attn_bench (run ON the GPU box from the repo root); kernel-trace only, one group per pass.
set -e -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/attnpmc
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/attnpmc/a -o p -- tools/probes/attn_bench 1025 > gpurun_out/attnpmc_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/attnpmc/b -o p -- tools/probes/attn_bench 1025 > gpurun_out/attnpmc_b.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES SQ_INSTS_MFMA --output-format csv -d gpurun_out/attnpmc/c -o p -- tools/probes/attn_bench 1025 > gpurun_out/attnpmc_c.log 2>&1
ls gpurun_out/attnpmc
