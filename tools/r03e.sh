#!/bin/bash
# round 3, fourth GPU session: filter_mismappers' second pass as one sweep over the read positions (mismapper_core.hpp: align_by_sweep) against the schedule of round 2;
# the step timed through arriba_workflow_sample (libarriba_workflow.so); first-pass budgets; threads of the writer
mkdir -p gpurun_out
T=r03e
export ARRIBA_BENCH_VERBOSE=1 ARRIBA_BENCH_REUSE=1
run() { # name, fragments, directory, extra environment...
  local name=$1 fragments=$2 dir=$3; shift 3
  env "$@" timeout 300 python bench.py --fragments $fragments --steps 1 --warmup 1 --no-cpu-baseline --keep $dir > gpurun_out/${T}_$name.json 2> gpurun_out/${T}_$name.err
  echo "$name exit $?"; grep "mismapper_heavy_kernel\]" gpurun_out/${T}_$name.err | tail -25 | cut -c1-600; grep "step done" gpurun_out/${T}_$name.err | tail -1 | cut -c1-420; grep "^\[writer\] rows\|^\[writer\] thread" gpurun_out/${T}_$name.err | tail -2
  cp $dir/fusions.rank0.tsv /tmp/${T}_$name.tsv
}
D=/dev/shm/r03e_10m; mkdir -p $D
run 10m_round2 10000000 $D ARRIBA_MISMAPPER_SWEEP=0
run 10m_sweep 10000000 $D ARRIBA_MISMAPPER_TIMES=1
cmp /tmp/${T}_10m_round2.tsv /tmp/${T}_10m_sweep.tsv && echo "10 M: fusions.tsv identical"
run 10m_sweep_steps256 10000000 $D ARRIBA_FIRST_PASS_STEPS=256
run 10m_sweep_steps512 10000000 $D ARRIBA_FIRST_PASS_STEPS=512
cmp /tmp/${T}_10m_round2.tsv /tmp/${T}_10m_sweep_steps256.tsv && echo "10 M, 256 steps: fusions.tsv identical"
env timeout 200 python bench.py --fragments 10000000 --steps 1 --warmup 1 --no-cpu-baseline --python-stages --keep $D > gpurun_out/${T}_10m_python_stages.json 2> gpurun_out/${T}_10m_python_stages.err; grep "step done" gpurun_out/${T}_10m_python_stages.err | tail -1 | cut -c1-420
rm -rf $D
D=/dev/shm/r03e_100m; mkdir -p $D
run 100m_sweep 100000000 $D ARRIBA_MISMAPPER_TIMES=1 ARRIBA_WRITER_PROFILE=1
run 100m_round2 100000000 $D ARRIBA_MISMAPPER_SWEEP=0
cmp /tmp/${T}_100m_round2.tsv /tmp/${T}_100m_sweep.tsv && echo "100 M: fusions.tsv identical"
run 100m_sweep_steps256 100000000 $D ARRIBA_FIRST_PASS_STEPS=256
run 100m_sweep_steps512_8192wg 100000000 $D ARRIBA_FIRST_PASS_STEPS=512 ARRIBA_HEAVY_WORKGROUPS=8192 ARRIBA_MEMO_SLOTS_LOG2=19
cmp /tmp/${T}_100m_round2.tsv /tmp/${T}_100m_sweep_steps256.tsv && echo "100 M, 256 steps: fusions.tsv identical"
run 100m_writer32 100000000 $D ARRIBA_WRITER_THREADS=32 ARRIBA_WRITER_PROFILE=1
run 100m_writer64 100000000 $D ARRIBA_WRITER_THREADS=64 ARRIBA_WRITER_PROFILE=1
rm -rf $D
