# tools/sq_pass.sh TAG -- what the wavefronts of the kernels do with their cycles: two rocprofv3 passes of SQ counters over one 10^8-fragment sample (no other trace domain beside --kernel-trace)
TAG=$1
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
ARGS="--fragments 100000000 --steps 1 --warmup 0 --no-cpu-baseline --no-deflated-leg --no-stress-leg --no-normal-pairs --keep /tmp/b100"
timeout 900 python bench.py $ARGS > gpurun_out/${TAG}_plain.json 2> gpurun_out/${TAG}_plain.err
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS --kernel-trace --output-format csv -d gpurun_out/sq_${TAG}_a -o pmc -- python bench.py $ARGS > gpurun_out/${TAG}_sq_a.log 2>&1
timeout 900 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d gpurun_out/sq_${TAG}_b -o pmc -- python bench.py $ARGS > gpurun_out/${TAG}_sq_b.log 2>&1
python tools/pmc_summary.py gpurun_out/${TAG}_sq.json gpurun_out/sq_${TAG}_a gpurun_out/sq_${TAG}_b > gpurun_out/${TAG}_sq_summary.txt 2>&1
rm -rf gpurun_out/sq_${TAG}_a gpurun_out/sq_${TAG}_b
tail -5 gpurun_out/${TAG}_sq_summary.txt
