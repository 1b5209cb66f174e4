# r02b: first GPU run of the device ingest: its GPU tests, then the end-to-end bench at 10 M and 100 M fragments
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q -k "device_ingest" > gpurun_out/r02b_pytest_ingest.log 2>&1; echo "pytest exit $?" >> gpurun_out/r02b_pytest_ingest.log
tail -5 gpurun_out/r02b_pytest_ingest.log
timeout 900 python bench.py --fragments 10000000 --steps 2 --warmup 1 > gpurun_out/r02b_bench10m.json 2> gpurun_out/r02b_bench10m.err; echo "bench exit $?" >> gpurun_out/r02b_bench10m.err
cut -c1-3000 gpurun_out/r02b_bench10m.json; tail -5 gpurun_out/r02b_bench10m.err
timeout 1200 python bench.py --fragments 100000000 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r02b_bench100m.json 2> gpurun_out/r02b_bench100m.err; echo "bench exit $?" >> gpurun_out/r02b_bench100m.err
cut -c1-3000 gpurun_out/r02b_bench100m.json; tail -5 gpurun_out/r02b_bench100m.err
