#!/bin/bash
# round 3, second GPU session: the second pass of filter_mismappers by seeds (mismapper_core.hpp: align_extend_seed) against the schedule of round 2, on ONE 100 M sample and ONE 10 M
# sample (--keep / ARRIBA_BENCH_REUSE), with the time every read of the second pass took (ARRIBA_MISMAPPER_TIMES); the output files of the two schedules must be identical
mkdir -p gpurun_out
T=r03c
export ARRIBA_BENCH_VERBOSE=1 ARRIBA_MISMAPPER_TIMES=1 ARRIBA_BENCH_REUSE=1
run() { # name, fragments, directory, extra environment...
  local name=$1 fragments=$2 dir=$3; shift 3
  env "$@" timeout 300 python bench.py --fragments $fragments --steps 1 --warmup 1 --no-cpu-baseline --keep $dir > gpurun_out/${T}_$name.json 2> gpurun_out/${T}_$name.err
  echo "$name exit $?"; grep "mismapper_heavy_kernel\]" gpurun_out/${T}_$name.err | tail -1 | cut -c1-600; grep "step done" gpurun_out/${T}_$name.err | tail -1 | cut -c1-420; grep "^\[writer\]" gpurun_out/${T}_$name.err | tail -12 | cut -c1-200
  cp $dir/fusions.rank0.tsv /tmp/${T}_$name.tsv
}
D=/dev/shm/r03c_10m; mkdir -p $D
run 10m_by_call 10000000 $D ARRIBA_MISMAPPER_BY_ITERATION=0
run 10m_by_seed 10000000 $D A=1
cmp /tmp/${T}_10m_by_call.tsv /tmp/${T}_10m_by_seed.tsv && echo "10 M: fusions.tsv identical"
run 10m_by_seed_steps512 10000000 $D ARRIBA_FIRST_PASS_STEPS=512
run 10m_by_seed_steps1024 10000000 $D ARRIBA_FIRST_PASS_STEPS=1024
cmp /tmp/${T}_10m_by_call.tsv /tmp/${T}_10m_by_seed_steps512.tsv && echo "10 M, 512 steps: fusions.tsv identical"
rm -rf $D
D=/dev/shm/r03c_100m; mkdir -p $D
run 100m_by_seed 100000000 $D ARRIBA_WRITER_PROFILE=1
run 100m_by_call 100000000 $D ARRIBA_MISMAPPER_BY_ITERATION=0
cmp /tmp/${T}_100m_by_call.tsv /tmp/${T}_100m_by_seed.tsv && echo "100 M: fusions.tsv identical"
run 100m_by_seed_8192wg 100000000 $D ARRIBA_HEAVY_WORKGROUPS=8192 ARRIBA_MEMO_SLOTS_LOG2=19
run 100m_by_seed_steps512 100000000 $D ARRIBA_FIRST_PASS_STEPS=512
cmp /tmp/${T}_100m_by_call.tsv /tmp/${T}_100m_by_seed_steps512.tsv && echo "100 M, 512 steps: fusions.tsv identical"
rm -rf $D
