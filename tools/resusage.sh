#!/bin/bash
# usage: tools/resusage.sh file.hip [filter]  -- one line per kernel: name, SGPR, VGPR, AGPR, scratch, occupancy, LDS
cd "$(dirname "$0")/../pda_amd/csrc"
/opt/rocm/bin/hipcc -Rpass-analysis=kernel-resource-usage --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -ffp-contract=off $EXTRA -c "$1" -o /tmp/_res.o 2>&1 |
  awk '/Function Name/{n=$NF} /TotalSGPRs/{s=$(NF-1)} / VGPRs:/{v=$(NF-1)} /AGPRs:/{a=$(NF-1)} /ScratchSize/{sc=$(NF-1)} /Occupancy/{o=$(NF-1)} /VGPRs Spill/{sp=$(NF-1)} /LDS Size/{print n, "sgpr="s, "vgpr="v, "agpr="a, "scratch="sc, "spill="sp, "occ="o, "lds="$(NF-1)}' | c++filt | grep "${2:-.}"
