# r02e: mismapper extension loop; 10 M, 30 M, 100 M (stage progress on stderr)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 300 python bench.py --fragments 10000000 --steps 2 --warmup 1 > gpurun_out/r02e_bench10m.json 2> gpurun_out/r02e_bench10m.err; echo "bench exit $?" >> gpurun_out/r02e_bench10m.err
cut -c1-400 gpurun_out/r02e_bench10m.json; grep "bench " gpurun_out/r02e_bench10m.err | tail -4
timeout 300 python bench.py --fragments 30000000 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r02e_bench30m.json 2> gpurun_out/r02e_bench30m.err; echo "bench exit $?" >> gpurun_out/r02e_bench30m.err
cut -c1-400 gpurun_out/r02e_bench30m.json; grep "bench " gpurun_out/r02e_bench30m.err | tail -30
timeout 330 python bench.py --fragments 100000000 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r02e_bench100m.json 2> gpurun_out/r02e_bench100m.err; echo "bench exit $?" >> gpurun_out/r02e_bench100m.err
cut -c1-400 gpurun_out/r02e_bench100m.json; grep -E "bench |File|line " gpurun_out/r02e_bench100m.err | tail -40
