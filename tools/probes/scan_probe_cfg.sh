#!/bin/bash
# thread / lane-split configurations of the combination kernel (run on the GPU box): NV NT LPC triples
cd $(dirname $0)
for cfg in ${CFGS:-"18 1024 8" "18 512 4" "18 512 8" "18 1024 4" "18 256 2" "32 1024 8" "32 1024 4" "32 512 4"}; do
  set -- $cfg
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -DNVP=$1 -DRTOC_SCAN_PROBE=0 -DRTOC_SCAN_FORCE_NT=$2 -DRTOC_SCAN_FORCE_LPC=$3 scan_probe.hip -o scan_probe_cfg.bin 2>/dev/null && echo "NT=$2 LPC=$3: $(./scan_probe_cfg.bin | head -1)"
done
