#!/bin/bash
# builds and runs the combination-level probe for a few part masks (run on the GPU box)
cd $(dirname $0)
for nv in ${NVS:-18 32}; do for pr in ${PROBES:-0 1 3}; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -DNVP=$nv -DRTOC_SCAN_PROBE=$pr scan_probe.hip -o scan_probe_${nv}_${pr}.bin 2>/dev/null && ./scan_probe_${nv}_${pr}.bin
done; done
