#!/bin/bash
# round 2, the very last GPU seconds: the 10 M end-to-end line with the final code (step budget 2048, queue + 4096 workgroups in the second pass)
mkdir -p gpurun_out
timeout 62 python bench.py --fragments 10000000 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r02n_bench.json 2> gpurun_out/r02n_bench.err
echo "exit $?"; cut -c1-400 gpurun_out/r02n_bench.json; grep "bench " gpurun_out/r02n_bench.err | tail -4
