#!/bin/bash
# kernel resource usage of one HIP source: name, VGPRs, spilled VGPRs, scratch bytes per lane
# usage: tools/kres.sh <file.hip> [extra hipcc flags]
src=$1; shift
cd "$(dirname "$0")/../video-diffusion-pipeline-parallel_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include "$@" -Rpass-analysis=kernel-resource-usage -c "$src" -o /dev/null 2>&1 |
  awk '/Function Name:/ {name=$0; sub(/.*Function Name: /,"",name); sub(/ \[.*/,"",name)}
       / VGPRs:/ {v=$0; sub(/.* VGPRs: /,"",v); sub(/ \[.*/,"",v)}
       /VGPRs Spill:/ {s=$0; sub(/.*Spill: /,"",s); sub(/ \[.*/,"",s)}
       /ScratchSize/ {c=$0; sub(/.*: /,"",c); sub(/ \[.*/,"",c)}
       /LDS Size/ {printf "%-110s vgpr %4s spill %4s scratch %5s\n", name, v, s, c}'
