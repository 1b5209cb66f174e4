#!/bin/bash
# round 3, first GPU session: HEAD of round 2 measured again (the verdict's items 1-2): the 100 M sample (3 steps, stage by stage), the 10 M sample under rocprofv3
# (kernel trace) and the two PMC passes, the mismapper stress (config 3) at 10 M, the task list of the second mismapper pass on / off
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash tools/r03b.sh'
mkdir -p gpurun_out
T=r03b
export ARRIBA_BENCH_VERBOSE=1
timeout 400 python bench.py --fragments 100000000 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${T}_bench100m.json 2> gpurun_out/${T}_bench100m.err; echo "100m exit $?"; cut -c1-300 gpurun_out/${T}_bench100m.json; grep "step done" gpurun_out/${T}_bench100m.err | cut -c1-700
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_${T} -o bench10m -- python $GRAFT_REPO_ROOT/bench.py --fragments 10000000 --no-cpu-baseline --steps 3 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/${T}_bench10m_traced.json 2> $GRAFT_REPO_ROOT/gpurun_out/${T}_bench10m_traced.err; echo "traced exit $?"
cd $GRAFT_REPO_ROOT
find gpurun_out/prof_${T} -name '*.db' | head -1 | xargs -I{} python tools/rocprof_summary.py {} "${T}: rocprofv3 --kernel-trace --stats -- python bench.py --fragments 10000000 --no-cpu-baseline --steps 3 --warmup 1" > gpurun_out/${T}_kernel_stats.txt 2>&1
rm -rf gpurun_out/prof_${T}
head -40 gpurun_out/${T}_kernel_stats.txt | cut -c1-200
cut -c1-300 gpurun_out/${T}_bench10m_traced.json; grep "step done" gpurun_out/${T}_bench10m_traced.err | tail -1 | cut -c1-700
cd /tmp
for COUNTER in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $COUNTER --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_${T}_$COUNTER -o pmc -- python $GRAFT_REPO_ROOT/bench.py --fragments 10000000 --no-cpu-baseline --steps 1 --warmup 0 > $GRAFT_REPO_ROOT/gpurun_out/${T}_pmc_$COUNTER.log 2>&1
done
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py gpurun_out/${T}_pmc_kernels.json gpurun_out/pmc_${T}_FETCH_SIZE gpurun_out/pmc_${T}_WRITE_SIZE > gpurun_out/${T}_pmc_summary.txt 2>&1
rm -rf gpurun_out/pmc_${T}_FETCH_SIZE gpurun_out/pmc_${T}_WRITE_SIZE
head -30 gpurun_out/${T}_pmc_summary.txt | cut -c1-200
timeout 200 python bench.py --fragments 10000000 --stress --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${T}_bench10m_stress.json 2> gpurun_out/${T}_bench10m_stress.err; echo "stress exit $?"; cut -c1-300 gpurun_out/${T}_bench10m_stress.json; grep "step done" gpurun_out/${T}_bench10m_stress.err | tail -1 | cut -c1-700
ARRIBA_MISMAPPER_WORKLIST=0 timeout 120 python bench.py --fragments 10000000 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${T}_bench10m_recursion.json 2> gpurun_out/${T}_bench10m_recursion.err; echo "recursion exit $?"; grep "step done" gpurun_out/${T}_bench10m_recursion.err | tail -1 | cut -c1-700
nproc > gpurun_out/${T}_box.txt; free -g >> gpurun_out/${T}_box.txt; lscpu | head -20 >> gpurun_out/${T}_box.txt; numactl -H >> gpurun_out/${T}_box.txt 2>&1
