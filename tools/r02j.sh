#!/bin/bash
# last GPU minutes of round 2: second mismapper pass, workgroups in flight x memo size, one 30 M sample
mkdir -p gpurun_out
timeout 200 python tools/r02j.py 30000000 1024:21 4096:21 8192:20 > gpurun_out/r02j_heavy_ab.txt 2> gpurun_out/r02j_heavy_ab.err
echo "exit $?"; cat gpurun_out/r02j_heavy_ab.txt; tail -3 gpurun_out/r02j_heavy_ab.err
