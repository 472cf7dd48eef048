#!/bin/bash
# PMC passes over the GEMM micro-benchmark (run on the GPU box). usage: tools/pmc_gemm.sh <outdir> <shape...>
set -e
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/$OUT
export ITERS=3
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE --output-format csv -d $R/$OUT/p1 -- python3 $R/tools/bench_gemm.py "$@" > $R/$OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL --output-format csv -d $R/$OUT/p2 -- python3 $R/tools/bench_gemm.py "$@" > $R/$OUT/p2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_HIT_sum --output-format csv -d $R/$OUT/p3 -- python3 $R/tools/bench_gemm.py "$@" > $R/$OUT/p3.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_MISS_sum TCP_TCC_READ_REQ_sum --output-format csv -d $R/$OUT/p4 -- python3 $R/tools/bench_gemm.py "$@" > $R/$OUT/p4.log 2>&1
echo done
