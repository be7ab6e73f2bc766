#!/bin/bash
# host-pointer path: transparent huge pages of the box, then config 2 through the plain API
cat /sys/kernel/mm/transparent_hugepage/enabled /sys/kernel/mm/transparent_hugepage/defrag 2>/dev/null || true
timeout -k 10 300 python tools/perf_host_pointers.py 2>&1 | grep -v amdgpu.ids
