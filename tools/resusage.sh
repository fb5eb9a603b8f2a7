#!/bin/bash
# usage: tools/resusage.sh lavender_amd/csrc/foo.hip [extra hipcc flags]  -> one line per kernel: VGPR / AGPR / spills / LDS / occupancy
f=$1; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -Wno-unused-value "$@" -Rpass-analysis=kernel-resource-usage -c "$f" -o /tmp/ru/$(basename "$f").o 2>&1 \
 | grep -E "remark:" | sed -E 's/^.*remark: //; s/ \[-Rpass.*$//' \
 | awk '/Function Name/{if(n)print n" | "s; n=$3; s=""} /VGPRs:|AGPRs:|ScratchSize|Occupancy|LDS Size|SGPRs:|Spill/{s=s" "$0";"} END{print n" | "s}' | sed 's/  */ /g'
