#!/bin/bash
# The round's committed profile set (run on the GPU box): usage: tools/profile_round.sh <tag> <commit>
#   gpurun_out/<tag>_b_serial_stats.csv / _c_2streams_stats.csv : rocprofv3 --kernel-trace --stats of one / two micro-batches
#                                                                   of two videos (25 / 50 UNet forwards at batch 2)
#   gpurun_out/<tag>_a_pmc.txt + <tag>_pmc_traffic.json          : the three --pmc passes folded per kernel
#   gpurun_out/<tag>_clock_in_kernel.json                         : shader clock held inside the K loops (experiments build stamps)
#   gpurun_out/<tag>_bench.json                                   : the default bench line (per_template FLOPs)
#   gpurun_out/<tag>_per_template.txt                             : fraction of peak per kernel template
set -eu
TAG=${1:?usage: profile_round.sh <tag> <commit>}; COMMIT=${2:?usage: profile_round.sh <tag> <commit>}
R=${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT is unset: run this on the GPU box (gpurun exports it)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_ps -- python3 $R/bench.py --concurrent 1 --steps 2 --warmup 0 --no-cpu-baseline --no-roofline --no-decode --no-batch1-leg > $R/gpurun_out/${TAG}_b_serial.json 2> $R/gpurun_out/${TAG}_b_serial.err
cp $(find $R/gpurun_out/${TAG}_ps -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${TAG}_b_serial_stats.csv
rm -rf "$R/gpurun_out/${TAG}_ps"
echo serial done
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_p2 -- python3 $R/bench.py --steps 4 --warmup 0 --no-cpu-baseline --no-roofline --no-decode --no-batch1-leg > $R/gpurun_out/${TAG}_c_2streams.json 2> $R/gpurun_out/${TAG}_c_2streams.err
cp $(find $R/gpurun_out/${TAG}_p2 -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${TAG}_c_2streams_stats.csv
rm -rf "$R/gpurun_out/${TAG}_p2"
echo two-streams done
cd $R
bash tools/pmc_forward.sh gpurun_out/${TAG}_pmc
python3 tools/pmc_forward_summary.py gpurun_out/${TAG}_pmc gpurun_out/${TAG}_pmc_traffic.json $COMMIT > gpurun_out/${TAG}_a_pmc.txt
rm -rf "gpurun_out/${TAG}_pmc/fetch" "gpurun_out/${TAG}_pmc/write" "gpurun_out/${TAG}_pmc/mfma"
echo pmc done
python3 tools/clock_in_kernel.py gpurun_out/${TAG}_clock_in_kernel.json "$COMMIT" > gpurun_out/${TAG}_clock_in_kernel.txt 2>&1
echo clock done
python3 bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python3 tools/per_template_summary.py gpurun_out/${TAG}_b_serial_stats.csv gpurun_out/${TAG}_bench.json 25 > gpurun_out/${TAG}_per_template.txt
cat gpurun_out/${TAG}_per_template.txt
