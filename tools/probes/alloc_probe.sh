#!/bin/bash
# gpurun -- 'bash tools/probes/alloc_probe.sh'  -> gpurun_out/alloc_probe.log
mkdir -p gpurun_out
hipcc -O2 -o /tmp/alloc_probe tools/probes/alloc_probe.cpp || exit 1
{
for mode in s s p p; do echo "== mode $mode"; ( time /tmp/alloc_probe $mode ) 2>&1 | grep -v "^$\|user\|sys"; done
echo "== with explicit free"; ( time /tmp/alloc_probe s free ) 2>&1 | grep -v "^$\|user\|sys"
echo "== after 3 s pause"; sleep 3; ( time /tmp/alloc_probe s ) 2>&1 | grep -v "^$\|user\|sys"
} > gpurun_out/alloc_probe.log 2>&1
cat gpurun_out/alloc_probe.log
