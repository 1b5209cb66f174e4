#!/bin/bash
# round 3, at the round's last commit that changes device code: kernel trace and the two PMC passes of bench.py at 10 M fragments again (bgzf_crc_kernel staged in LDS and
# run_name_order_kernel came after r03s), then the whole GPU tier
mkdir -p gpurun_out
T=r03w
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${T}_10m -o bench10m -- python $R/bench.py --fragments 10000000 --no-cpu-baseline --no-normal-pairs --steps 3 --warmup 2 > $R/gpurun_out/${T}_bench10m_traced.json 2> $R/gpurun_out/${T}_bench10m_traced.err; echo "traced 10m exit $?"
cd $R
find gpurun_out/prof_${T}_10m -name '*.db' | head -1 | xargs -I{} python tools/rocprof_summary.py {} "${T}: rocprofv3 --kernel-trace --stats -- python bench.py --fragments 10000000 --no-cpu-baseline --no-normal-pairs --steps 3 --warmup 2" > gpurun_out/${T}_kernel_stats_10m.txt 2>&1
rm -rf gpurun_out/prof_${T}_10m
head -8 gpurun_out/${T}_kernel_stats_10m.txt | cut -c1-200
cd /tmp
for COUNTER in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $COUNTER --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${T}_$COUNTER -o pmc -- python $R/bench.py --fragments 10000000 --no-cpu-baseline --no-normal-pairs --steps 1 --warmup 0 > $R/gpurun_out/${T}_pmc_$COUNTER.log 2>&1; echo "pmc $COUNTER exit $?"
done
cd $R
python tools/pmc_summary.py gpurun_out/${T}_pmc_kernels.json gpurun_out/pmc_${T}_FETCH_SIZE gpurun_out/pmc_${T}_WRITE_SIZE > gpurun_out/${T}_pmc_summary.txt 2>&1
rm -rf gpurun_out/pmc_${T}_FETCH_SIZE gpurun_out/pmc_${T}_WRITE_SIZE
head -8 gpurun_out/${T}_pmc_summary.txt | cut -c1-200
timeout 1100 python -m pytest tests -x -q -m gpu > gpurun_out/${T}_pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -4 gpurun_out/${T}_pytest_gpu.log | cut -c1-300
