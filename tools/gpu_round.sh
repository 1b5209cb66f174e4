# tools/gpu_round.sh TAG [pmc] -- one GPU-box round: GPU parity tests, bench line, rocprofv3 kernel trace, optional PMC passes.
TAG=${1:-r01x}
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export ARRIBA_BENCH_CACHE=/tmp/arriba_bench_cache   # the bench invocations below share one ingested batch
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/${TAG}_pytest_gpu.log
# SKIP_PLAIN_BENCH=1: the traced run below also prints the bench line (GPU minutes are scarce: every bench invocation generates and ingests 10 M fragments first)
# BENCH_FLAGS, e.g. BENCH_FLAGS=--workflow: one untimed pass of the whole workflow (candidate-level stages, output files) is added to the bench line
if [ -z "$SKIP_PLAIN_BENCH" ]; then timeout 900 python bench.py $BENCH_FLAGS > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench exit $?" >> gpurun_out/${TAG}_bench.err; fi
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG} -o bench10m -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_bench_prof.json 2> $GRAFT_REPO_ROOT/gpurun_out/${TAG}_bench_prof.err
cd $GRAFT_REPO_ROOT
if [ -n "$SKIP_PLAIN_BENCH" ]; then cp gpurun_out/${TAG}_bench_prof.json gpurun_out/${TAG}_bench.json; fi
find gpurun_out/prof_${TAG} -name '*.db' | head -1 | xargs -I{} python tools/rocprof_summary.py {} "${TAG}: rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --steps 3 --warmup 1 (10 M fragments)" > gpurun_out/${TAG}_kernel_stats.txt 2>&1
rm -rf gpurun_out/prof_${TAG}
if [ "$2" = "pmc" ]; then
  cd /tmp
  for COUNTER in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $COUNTER --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_$COUNTER -o pmc -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 1 --warmup 0 > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc_$COUNTER.log 2>&1
  done
  cd $GRAFT_REPO_ROOT
  python tools/pmc_summary.py gpurun_out/${TAG}_pmc_kernels.json gpurun_out/pmc_${TAG}_FETCH_SIZE gpurun_out/pmc_${TAG}_WRITE_SIZE > gpurun_out/${TAG}_pmc_summary.txt 2>&1
  python - <<PY
import json
kernels = json.load(open("gpurun_out/${TAG}_pmc_kernels.json"))
line = json.loads(open("gpurun_out/${TAG}_bench.json").read().strip().splitlines()[-1])
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bench.py --steps 1 --warmup 0; KB summed over the dispatches of a kernel", "fragments": line["config"]["fragments_per_gpu"], "kernels": kernels}, open("gpurun_out/${TAG}_pmc.json", "w"), indent=1, sort_keys=True)
PY
  rm -rf gpurun_out/pmc_${TAG}_FETCH_SIZE gpurun_out/pmc_${TAG}_WRITE_SIZE
fi
tail -5 gpurun_out/${TAG}_pytest_gpu.log; cat gpurun_out/${TAG}_bench.json; head -14 gpurun_out/${TAG}_kernel_stats.txt
if [ "$3" = "probe" ]; then timeout 600 python tools/stage_probe.py 3000000 > gpurun_out/${TAG}_probe.json 2> gpurun_out/${TAG}_probe.err; cat gpurun_out/${TAG}_probe.json; fi
if [ "$4" = "dryrun2" ]; then
  ARRIBA_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --fragments 2000000 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_bench_2ranks_dryrun.json 2> gpurun_out/${TAG}_bench_2ranks_dryrun.err; tail -1 gpurun_out/${TAG}_bench_2ranks_dryrun.json | cut -c1-1500; tail -5 gpurun_out/${TAG}_bench_2ranks_dryrun.err
fi
if [ "$5" = "diag" ]; then
  # where the wave cycles of every kernel go (issue vs wait) and the instruction mix: two SQ passes
  cd /tmp
  timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/sq_${TAG}_a -o pmc -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 1 --warmup 0 > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_sq_a.log 2>&1
  timeout 600 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQ_INSTS_SMEM SQ_INSTS_LDS --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/sq_${TAG}_b -o pmc -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 1 --warmup 0 > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_sq_b.log 2>&1
  cd $GRAFT_REPO_ROOT
  python tools/pmc_summary.py gpurun_out/${TAG}_sq.json gpurun_out/sq_${TAG}_a gpurun_out/sq_${TAG}_b > gpurun_out/${TAG}_sq_summary.txt 2>&1
  rm -rf gpurun_out/sq_${TAG}_a gpurun_out/sq_${TAG}_b
  head -30 gpurun_out/${TAG}_sq_summary.txt | cut -c1-400
fi
if [ -n "$6" ]; then timeout 600 python tools/ab_probe.py 3000000 libarriba_gpu.so $6 > gpurun_out/${TAG}_ab.json 2> gpurun_out/${TAG}_ab.err; cat gpurun_out/${TAG}_ab.json; fi
