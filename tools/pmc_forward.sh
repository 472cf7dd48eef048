#!/bin/bash
# PMC passes over two UNet steps (run on the GPU box): HBM bytes and MFMA busy per kernel.  usage: pmc_forward.sh <outdir>
set -e
OUT=$1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/$OUT
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$OUT/fetch -- python3 $R/tools/one_forward.py > $R/$OUT/fetch.log 2>&1
echo fetch done
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$OUT/write -- python3 $R/tools/one_forward.py > $R/$OUT/write.log 2>&1
echo write done
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $R/$OUT/mfma -- python3 $R/tools/one_forward.py > $R/$OUT/mfma.log 2>&1
echo mfma done
# keep only what fits the 64 MiB merge limit: the summariser runs on the box, raw CSVs are dropped afterwards
