#!/bin/bash
mkdir -p gpurun_out
timeout 12 python tools/r02p.py > gpurun_out/r02p_worklist.txt 2>&1
echo "exit $?"; grep -v WARNING gpurun_out/r02p_worklist.txt | tail -5
