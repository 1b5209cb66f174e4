#!/bin/bash
# round 3, the evidence of the round's last state: rocprofv3 --kernel-trace --stats of bench.py at 10 M fragments and of one 100 M step, the two PMC passes (FETCH_SIZE and
# WRITE_SIZE apart) at 10 M, bench.py as the driver runs it, the whole GPU tier
#   /usr/local/graft/bin/gpurun --timeout 2100 -- 'bash tools/r03s.sh'
mkdir -p gpurun_out
T=r03s
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${T}_10m -o bench10m -- python $R/bench.py --fragments 10000000 --no-cpu-baseline --no-normal-pairs --steps 3 --warmup 2 > $R/gpurun_out/${T}_bench10m_traced.json 2> $R/gpurun_out/${T}_bench10m_traced.err; echo "traced 10m exit $?"
cd $R
find gpurun_out/prof_${T}_10m -name '*.db' | head -1 | xargs -I{} python tools/rocprof_summary.py {} "${T}: rocprofv3 --kernel-trace --stats -- python bench.py --fragments 10000000 --no-cpu-baseline --no-normal-pairs --steps 3 --warmup 2" > gpurun_out/${T}_kernel_stats_10m.txt 2>&1
rm -rf gpurun_out/prof_${T}_10m
head -12 gpurun_out/${T}_kernel_stats_10m.txt | cut -c1-200
cd /tmp
for COUNTER in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $COUNTER --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${T}_$COUNTER -o pmc -- python $R/bench.py --fragments 10000000 --no-cpu-baseline --no-normal-pairs --steps 1 --warmup 0 > $R/gpurun_out/${T}_pmc_$COUNTER.log 2>&1; echo "pmc $COUNTER exit $?"
done
cd $R
python tools/pmc_summary.py gpurun_out/${T}_pmc_kernels.json gpurun_out/pmc_${T}_FETCH_SIZE gpurun_out/pmc_${T}_WRITE_SIZE > gpurun_out/${T}_pmc_summary.txt 2>&1
rm -rf gpurun_out/pmc_${T}_FETCH_SIZE gpurun_out/pmc_${T}_WRITE_SIZE
head -16 gpurun_out/${T}_pmc_summary.txt | cut -c1-200
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${T}_100m -o bench100m -- python $R/bench.py --fragments 100000000 --no-cpu-baseline --no-normal-pairs --steps 1 --warmup 1 > $R/gpurun_out/${T}_bench100m_traced.json 2> $R/gpurun_out/${T}_bench100m_traced.err; echo "traced 100m exit $?"
cd $R
find gpurun_out/prof_${T}_100m -name '*.db' | head -1 | xargs -I{} python tools/rocprof_summary.py {} "${T}: rocprofv3 --kernel-trace --stats -- python bench.py --fragments 100000000 --no-cpu-baseline --no-normal-pairs --steps 1 --warmup 1" > gpurun_out/${T}_kernel_stats_100m.txt 2>&1
rm -rf gpurun_out/prof_${T}_100m
head -12 gpurun_out/${T}_kernel_stats_100m.txt | cut -c1-200
( time timeout 1700 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${T}_bench_as_the_driver_runs_it.json 2> gpurun_out/${T}_bench_as_the_driver_runs_it.err ) 2> gpurun_out/${T}_bench_as_the_driver_runs_it.time; echo "bench exit $?"; cat gpurun_out/${T}_bench_as_the_driver_runs_it.time | tr '\n' ' '; echo; cut -c1-400 gpurun_out/${T}_bench_as_the_driver_runs_it.json
timeout 1100 python -m pytest tests -x -q -m gpu > gpurun_out/${T}_pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -4 gpurun_out/${T}_pytest_gpu.log | cut -c1-300
