# r02c: end-to-end bench at 10 M and 100 M after the writer / candidate compaction / mismapper changes, then the whole GPU tier
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python bench.py --fragments 10000000 --steps 2 --warmup 1 > gpurun_out/r02c_bench10m.json 2> gpurun_out/r02c_bench10m.err; echo "bench exit $?" >> gpurun_out/r02c_bench10m.err
cut -c1-2500 gpurun_out/r02c_bench10m.json; grep -v "stop codon" gpurun_out/r02c_bench10m.err | tail -3
timeout 900 python bench.py --fragments 100000000 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r02c_bench100m.json 2> gpurun_out/r02c_bench100m.err; echo "bench exit $?" >> gpurun_out/r02c_bench100m.err
cut -c1-2500 gpurun_out/r02c_bench100m.json; grep -v "stop codon" gpurun_out/r02c_bench100m.err | tail -3
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02c_pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/r02c_pytest_gpu.log
tail -8 gpurun_out/r02c_pytest_gpu.log
