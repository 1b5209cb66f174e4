# r02h: memo of failed nested calls in the second pass of the mismapper search: 10 M and 30 M
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 120 python bench.py --fragments 10000000 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r02h_bench10m.json 2> gpurun_out/r02h_bench10m.err; echo "bench exit $?" >> gpurun_out/r02h_bench10m.err
cut -c1-300 gpurun_out/r02h_bench10m.json; grep "bench " gpurun_out/r02h_bench10m.err | tail -3
ARRIBA_BENCH_VERBOSE=1 timeout 150 python bench.py --fragments 30000000 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r02h_bench30m.json 2> gpurun_out/r02h_bench30m.err; echo "bench exit $?" >> gpurun_out/r02h_bench30m.err
cut -c1-300 gpurun_out/r02h_bench30m.json; grep "bench " gpurun_out/r02h_bench30m.err | tail -6
