#!/bin/bash
# usage: tools/resource_usage.sh file.hip [-DMACRO=VALUE …] — kernel | SGPR | VGPR | scratch B/lane | waves/SIMD | LDS B
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c -ffp-contract=off -Rpass-analysis=kernel-resource-usage "${@:2}" "$1" -o /dev/null 2>&1 \
 | grep -E "Function Name|TotalSGPRs|VGPRs:|ScratchSize|Occupancy|LDS Size" \
 | sed -E 's/.*remark: //; s/ \[-Rpass.*//' \
 | awk '/Function Name/{if(l)print l; l=$3; next}{n=$NF; l=l" "n}END{print l}' \
 | while read name rest; do echo "$(echo $name | c++filt | sed -E 's/\(.*//; s/^void //') $rest"; done
