#!/bin/bash
# Development aid: registers / scratch / occupancy of every patch_kernel instantiation (extra flags pass through).
hipcc --offload-arch=gfx950 -std=c++20 -O3 -fno-slp-vectorize -munsafe-fp-atomics -c -Rpass-analysis=kernel-resource-usage \
  "$@" -o /tmp/rpsf_resusage.o "$(dirname "$0")/../regularizepsf_amd/csrc/rpsf.hip" 2>&1 |
  awk '/Function Name/ {name=$0; sub(/.*Name: /,"",name); sub(/ \[.*/,"",name); keep = name ~ /patch_kernel/}
       keep && /remark: +(VGPRs:|ScratchSize|Occupancy|VGPRs Spill)/ {v=$0; sub(/.*remark: +/,"",v); sub(/ \[-R.*/,"",v); line = line " | " v}
       keep && /LDS Size/ {n=name; sub(/.*CfgI/,"Cfg<",n); sub(/EEEv.*/,">",n); gsub(/ELi/,",",n); gsub(/Li/,"",n); print n line; line=""}'
