#!/bin/bash
# PMC passes over the attention A/B micro-benchmark (run on the GPU box). usage: tools/pmc_attn.sh <outdir> <batch:seq:heads>
set -e
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/$OUT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $R/$OUT/p1 -- python3 $R/tools/bench_attn_ab.py "$@" > $R/$OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $R/$OUT/p2 -- python3 $R/tools/bench_attn_ab.py "$@" > $R/$OUT/p2.log 2>&1
python3 $R/tools/pmc_summary.py $R/$OUT attn_spatial > $R/$OUT/summary.txt 2>&1
echo done
