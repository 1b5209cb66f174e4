#!/bin/bash
# round 2, last GPU seconds: step budget of the first mismapper pass (thread per read) at 10 M fragments
mkdir -p gpurun_out
timeout 100 python tools/r02j.py 10000000 4096:21:65536 4096:21:8192 4096:21:2048 4096:21:512 > gpurun_out/r02l_first_pass_ab.txt 2> gpurun_out/r02l_first_pass_ab.err
echo "exit $?"; cat gpurun_out/r02l_first_pass_ab.txt; tail -2 gpurun_out/r02l_first_pass_ab.err
