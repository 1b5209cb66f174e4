#!/bin/bash
# round 3, third GPU session: what the slow reads of the second mismapper pass consist of at 100 M (calls, iterations, seeds, seeds walked), with and without the memo of the seeds
mkdir -p gpurun_out
T=r03d
export ARRIBA_BENCH_VERBOSE=1 ARRIBA_MISMAPPER_TIMES=1 ARRIBA_BENCH_REUSE=1
run() { # name, fragments, directory, extra environment...
  local name=$1 fragments=$2 dir=$3; shift 3
  env "$@" timeout 300 python bench.py --fragments $fragments --steps 1 --warmup 0 --no-cpu-baseline --keep $dir > gpurun_out/${T}_$name.json 2> gpurun_out/${T}_$name.err
  echo "$name exit $?"; grep "mismapper_heavy_kernel\]" gpurun_out/${T}_$name.err | tail -25 | cut -c1-700; grep "step done" gpurun_out/${T}_$name.err | tail -1 | cut -c1-420
  cp $dir/fusions.rank0.tsv /tmp/${T}_$name.tsv
}
D=/dev/shm/r03d_100m; mkdir -p $D
run 100m_seeds_again 100000000 $D ARRIBA_MISMAPPER_SEEDS_ONCE=0
run 100m_seeds_once 100000000 $D A=1
cmp /tmp/${T}_100m_seeds_again.tsv /tmp/${T}_100m_seeds_once.tsv && echo "100 M: fusions.tsv identical"
rm -rf $D
