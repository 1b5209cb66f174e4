# r02d: the reworked mismapper kernels: end-to-end at 10 M, A/B against the wavefront-per-read pass alone, then 100 M
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 400 python bench.py --fragments 10000000 --steps 2 --warmup 1 > gpurun_out/r02d_bench10m.json 2> gpurun_out/r02d_bench10m.err; echo "bench exit $?" >> gpurun_out/r02d_bench10m.err
cut -c1-1800 gpurun_out/r02d_bench10m.json; grep "bench " gpurun_out/r02d_bench10m.err | tail -8
ARRIBA_MISMAPPER_FIRST_PASS=0 timeout 300 python bench.py --fragments 10000000 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r02d_bench10m_wave_only.json 2> gpurun_out/r02d_bench10m_wave_only.err
grep "bench " gpurun_out/r02d_bench10m_wave_only.err | tail -4
timeout 480 python bench.py --fragments 100000000 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r02d_bench100m.json 2> gpurun_out/r02d_bench100m.err; echo "bench exit $?" >> gpurun_out/r02d_bench100m.err
cut -c1-1800 gpurun_out/r02d_bench100m.json; grep "bench " gpurun_out/r02d_bench100m.err | tail -8
