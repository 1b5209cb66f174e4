# r02f: second pass of the mismapper search with shared seeds: 10 M and 30 M
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 240 python bench.py --fragments 10000000 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r02f_bench10m.json 2> gpurun_out/r02f_bench10m.err; echo "bench exit $?" >> gpurun_out/r02f_bench10m.err
cut -c1-300 gpurun_out/r02f_bench10m.json; grep "bench " gpurun_out/r02f_bench10m.err | tail -3
ARRIBA_BENCH_VERBOSE=1 timeout 200 python bench.py --fragments 30000000 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r02f_bench30m.json 2> gpurun_out/r02f_bench30m.err; echo "bench exit $?" >> gpurun_out/r02f_bench30m.err
cut -c1-300 gpurun_out/r02f_bench30m.json; grep "bench " gpurun_out/r02f_bench30m.err | tail -8
