set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
nproc > gpurun_out/nproc.txt; lscpu | head -20 >> gpurun_out/nproc.txt
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
timeout 900 python bench.py > gpurun_out/bench_1.json 2> gpurun_out/bench_1.err; echo "bench exit $?" >> gpurun_out/bench_1.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r1b -o bench10m -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/bench_prof.json 2> $GRAFT_REPO_ROOT/gpurun_out/bench_prof.err
cd $GRAFT_REPO_ROOT
find gpurun_out/prof_r1b -name '*.db' | head -1 | xargs -I{} python tools/rocprof_summary.py {} "round 1b: bench.py 10M fragments, steps 3 warmup 1" > gpurun_out/prof_r1b_summary.txt 2>&1
find gpurun_out/prof_r1b -name '*.db' -delete
tail -5 gpurun_out/pytest_gpu.log; cat gpurun_out/bench_1.json; head -12 gpurun_out/prof_r1b_summary.txt
