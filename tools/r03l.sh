#!/bin/bash
# round 3, tenth GPU session: bench.py exactly as the driver runs it (the 100 M sample in its child, 20 + 5 steps, the leg with ordinary pairs, the reference on its bounded sample);
# the mismapper stress (config 3) at the sizes whose read lists fit 32-bit offsets
mkdir -p gpurun_out
T=r03l
( time timeout 1700 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${T}_bench_default.json 2> gpurun_out/${T}_bench_default.err ) 2> gpurun_out/${T}_bench_default.time; echo "bench exit $?"; cat gpurun_out/${T}_bench_default.time | tr '\n' ' '; echo
cut -c1-1500 gpurun_out/${T}_bench_default.json; grep "bench " gpurun_out/${T}_bench_default.err | grep -v "step done" | cut -c1-300 | tail -12
for fragments in 3000000 5000000; do
  timeout 300 python bench.py --fragments $fragments --stress --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${T}_stress_$fragments.json 2> gpurun_out/${T}_stress_$fragments.err; echo "stress $fragments exit $?"; cut -c1-400 gpurun_out/${T}_stress_$fragments.json; grep "step done\|ERROR" gpurun_out/${T}_stress_$fragments.err | tail -2 | cut -c1-400
done
