#!/usr/bin/env bash
# round-4 probe 1: dependent-kernel overlap (device-side completion counters) + cost of integer-limb statistics atomics
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r04p1; mkdir -p $O
hipcc --offload-arch=gfx950 -O2 tools/probes/pdl_probe.hip -o /tmp/pdl_probe || exit 1
timeout 120 /tmp/pdl_probe 512 > $O/pdl_512.txt 2>&1; echo "rc=$?" >> $O/pdl_512.txt
timeout 120 /tmp/pdl_probe 256 > $O/pdl_256.txt 2>&1; echo "rc=$?" >> $O/pdl_256.txt
timeout 120 env GPU_MAX_HW_QUEUES=8 /tmp/pdl_probe 512 > $O/pdl_512_q8.txt 2>&1; echo "rc=$?" >> $O/pdl_512_q8.txt
cat $O/pdl_512.txt $O/pdl_256.txt $O/pdl_512_q8.txt
