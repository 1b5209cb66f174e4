# r02g FRAGMENTS: profiles of the end-to-end bench (rocprofv3 kernel trace + the two PMC passes), the bench line, then GPU tests
FR=${1:-10000000}
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
# the 100 M sample (BASELINE.json's metric), one step
timeout 260 python bench.py --fragments 100000000 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r02g_bench100m.json 2> gpurun_out/r02g_bench100m.err; echo "bench exit $?" >> gpurun_out/r02g_bench100m.err
cut -c1-400 gpurun_out/r02g_bench100m.json; grep "bench " gpurun_out/r02g_bench100m.err | tail -6
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r02g -o bench -- python $R/bench.py --fragments $FR --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r02g_bench_traced.json 2> $R/gpurun_out/r02g_bench_traced.err
cd $R
find gpurun_out/prof_r02g -name '*.db' | head -1 | xargs -I{} python tools/rocprof_summary.py {} "r02g: rocprofv3 --kernel-trace --stats -- python bench.py --fragments $FR --steps 2 --warmup 1 --no-cpu-baseline" > gpurun_out/r02g_kernel_stats.txt 2>&1
rm -rf gpurun_out/prof_r02g
head -12 gpurun_out/r02g_kernel_stats.txt | cut -c1-160
cd /tmp
for COUNTER in FETCH_SIZE WRITE_SIZE; do
  timeout 240 rocprofv3 --pmc $COUNTER --kernel-trace --output-format csv -d $R/gpurun_out/pmc_r02g_$COUNTER -o pmc -- python $R/bench.py --fragments $FR --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/r02g_pmc_$COUNTER.log 2>&1
done
cd $R
python tools/pmc_summary.py gpurun_out/r02g_pmc_kernels.json gpurun_out/pmc_r02g_FETCH_SIZE gpurun_out/pmc_r02g_WRITE_SIZE > gpurun_out/r02g_pmc_summary.txt 2>&1
python - <<PY
import json
kernels = json.load(open("gpurun_out/r02g_pmc_kernels.json"))
line = json.loads(open("gpurun_out/r02g_bench_traced.json").read().strip().splitlines()[-1])
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bench.py --fragments $FR --steps 1 --warmup 0; KB summed over the dispatches of a kernel", "fragments": line["config"]["fragments_per_gpu"], "kernels": kernels}, open("gpurun_out/r02g_pmc.json", "w"), indent=1, sort_keys=True)
PY
rm -rf gpurun_out/pmc_r02g_FETCH_SIZE gpurun_out/pmc_r02g_WRITE_SIZE
head -8 gpurun_out/r02g_pmc_summary.txt | cut -c1-200
cp gpurun_out/r02g_pmc.json profiles/pmc_latest.json
timeout 300 python bench.py --fragments $FR > gpurun_out/r02g_bench.json 2> gpurun_out/r02g_bench.err; echo "bench exit $?" >> gpurun_out/r02g_bench.err
cut -c1-600 gpurun_out/r02g_bench.json; grep "bench " gpurun_out/r02g_bench.err | tail -3
timeout 900 python -m pytest tests -m gpu -x -q $PYTEST_SELECT > gpurun_out/r02g_pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/r02g_pytest_gpu.log
tail -4 gpurun_out/r02g_pytest_gpu.log
