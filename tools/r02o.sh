#!/bin/bash
# round 2, the remaining GPU seconds: where the output side of the 10 M step (1.07 s of 2.19 s) goes
mkdir -p gpurun_out
timeout 36 python bench.py --fragments 10000000 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r02o_bench.json 2> gpurun_out/r02o_bench.err
echo "exit $?"; grep "step done" gpurun_out/r02o_bench.err | cut -c1-900
