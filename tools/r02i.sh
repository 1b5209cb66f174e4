# r02i: last check of the round on hardware: ingest tests, the fallback path of the default bench (100 M child -> config 2), kernel trace at 10 M
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 200 python -m pytest tests -m gpu -x -q -k "device_ingest_builds or workflow_from_the_bam" > gpurun_out/r02i_pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/r02i_pytest.log
tail -3 gpurun_out/r02i_pytest.log
ARRIBA_BENCH_TIME_BUDGET=50 timeout 330 python bench.py --steps 2 --warmup 0 --no-cpu-baseline > gpurun_out/r02i_bench_default.json 2> gpurun_out/r02i_bench_default.err; echo "bench exit $?" >> gpurun_out/r02i_bench_default.err
cut -c1-500 gpurun_out/r02i_bench_default.json; grep "bench " gpurun_out/r02i_bench_default.err | tail -12
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r02i -o bench -- python $R/bench.py --fragments 10000000 --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r02i_bench_traced.json 2> $R/gpurun_out/r02i_bench_traced.err
cd $R
find gpurun_out/prof_r02i -name '*.db' | head -1 | xargs -I{} python tools/rocprof_summary.py {} "r02i: rocprofv3 --kernel-trace --stats -- python bench.py --fragments 10000000 --steps 2 --warmup 1 --no-cpu-baseline" > gpurun_out/r02i_kernel_stats.txt 2>&1
rm -rf gpurun_out/prof_r02i
head -10 gpurun_out/r02i_kernel_stats.txt | cut -c1-150; cut -c1-200 gpurun_out/r02i_bench_traced.json
