#!/bin/bash
# usage: tools/kernel_dev/build.sh <file.hip>  -> prints VGPR / scratch / occupancy of every kernel in it
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc $KDEV_FLAGS -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -Rpass-analysis=kernel-resource-usage -c "$1" -o /tmp/kdev.o 2>&1 | grep -E "error|Function Name|VGPRs:|AGPRs|ScratchSize|Spill|Occupancy" | sed 's/.*remark: *//; s/\[-Rpass.*//'
