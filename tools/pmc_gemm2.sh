#!/bin/bash
# extra PMC passes on one GEMM shape (run on the GPU box)
set -e
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/$OUT
export ITERS=2
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $R/$OUT/q1 -- python3 $R/tools/bench_gemm.py "$@" > $R/$OUT/q1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_IFETCH --output-format csv -d $R/$OUT/q2 -- python3 $R/tools/bench_gemm.py "$@" > $R/$OUT/q2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC --output-format csv -d $R/$OUT/q3 -- python3 $R/tools/bench_gemm.py "$@" > $R/$OUT/q3.log 2>&1
python3 $R/tools/pmc_summary.py $R/$OUT gemm > $R/$OUT/summary.txt
rm -rf $R/$OUT/q1 $R/$OUT/q2 $R/$OUT/q3
cat $R/$OUT/summary.txt
