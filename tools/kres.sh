#!/bin/bash
# kernel resource usage of one translation unit as the compiler reports it: name, VGPRs, SGPR / VGPR spills, scratch, occupancy, LDS
# usage: tools/kres.sh rc_correct.hip [filter-regex] [extra hipcc flags...]
cd "$(dirname "$0")/../rcorrector_amd/csrc" || exit 1
f=$1; shift
pat=${1:-.}; shift
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Rpass-analysis=kernel-resource-usage "$@" -c "$f" -o /dev/null 2>&1 |
  sed 's/ \[-Rpass-analysis=kernel-resource-usage\]//' |
  awk '/Function Name:/ {name=$NF} / VGPRs: / {v=$NF} /SGPRs Spill:/ {ss=$NF} /VGPRs Spill:/ {vs=$NF} /ScratchSize/ {sc=$NF} /Occupancy/ {oc=$NF} /LDS Size/ {print name, "vgpr="v, "sgpr_spill="ss, "vgpr_spill="vs, "scratch="sc, "occ="oc, "lds="$NF}' |
  grep -E "$pat" | while read n rest; do echo "$(echo $n | c++filt | cut -c1-70) $rest"; done
