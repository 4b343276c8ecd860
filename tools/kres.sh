#!/bin/bash
# kernel resource usage of one .hip file: name, VGPRs, SGPRs, occupancy, LDS
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -c -ffp-contract=off --offload-arch=gfx950 -I"$(dirname "$0")/../include" "$1" -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 \
 | grep -E "Function Name|VGPRs:|TotalSGPRs:|Occupancy|LDS Size|ScratchSize" | sed -E 's/^.*remark: +//; s/ \[-Rpass.*//; s/^.*Function Name: /@/' | tr '\n' ' ' | tr '@' '\n' | sed -E 's/ +/ /g'
echo
