#!/bin/bash
# round 3, fifth GPU session: the output side reworked (candidates picked on the device, pinned staging kept by the session, formatter threads and pileup pages kept), the ingest
# buffers kept between samples, first-pass budget 256; threads of the feed and of the writer; the workflow tests of the GPU tier (they hold the new agpu_select_candidates check)
mkdir -p gpurun_out
T=r03f
export ARRIBA_BENCH_VERBOSE=1 ARRIBA_BENCH_REUSE=1
run() { # name, fragments, directory, steps, extra environment...
  local name=$1 fragments=$2 dir=$3 steps=$4; shift 4
  env "$@" timeout 300 python bench.py --fragments $fragments --steps $steps --warmup 1 --no-cpu-baseline --keep $dir > gpurun_out/${T}_$name.json 2> gpurun_out/${T}_$name.err
  echo "$name exit $?"; grep "step done" gpurun_out/${T}_$name.err | tail -2 | cut -c1-420; grep "^\[writer\] rows\|^\[writer\] thread" gpurun_out/${T}_$name.err | tail -2
  cp $dir/fusions.rank0.tsv /tmp/${T}_$name.tsv
}
timeout 600 python -m pytest tests -x -q -m gpu -k "workflow_from_input_files or workflow_from_the_bam or cpp_workflow or mismapper_stress" > gpurun_out/${T}_pytest.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/${T}_pytest.log
D=/dev/shm/r03f_10m; mkdir -p $D
run 10m 10000000 $D 3 ARRIBA_WRITER_PROFILE=1
env timeout 200 python bench.py --fragments 10000000 --steps 1 --warmup 1 --no-cpu-baseline --python-stages --keep $D > gpurun_out/${T}_10m_python_stages.json 2> gpurun_out/${T}_10m_python_stages.err; grep "step done" gpurun_out/${T}_10m_python_stages.err | tail -1 | cut -c1-420
cmp /tmp/${T}_10m.tsv $D/fusions.rank0.tsv && echo "10 M: fusions.tsv of the C++ workflow == of the Python mirror"
rm -rf $D
D=/dev/shm/r03f_100m; mkdir -p $D
run 100m 100000000 $D 3 ARRIBA_WRITER_PROFILE=1
run 100m_feed64 100000000 $D 1 ARRIBA_FEED_THREADS=64
run 100m_feed128 100000000 $D 1 ARRIBA_FEED_THREADS=128
run 100m_feed16 100000000 $D 1 ARRIBA_FEED_THREADS=16
run 100m_writer32 100000000 $D 1 ARRIBA_WRITER_THREADS=32 ARRIBA_WRITER_PROFILE=1
run 100m_writer64 100000000 $D 1 ARRIBA_WRITER_THREADS=64 ARRIBA_WRITER_PROFILE=1
env timeout 300 python bench.py --fragments 100000000 --steps 1 --warmup 1 --no-cpu-baseline --python-stages --keep $D > gpurun_out/${T}_100m_python_stages.json 2> gpurun_out/${T}_100m_python_stages.err; grep "step done" gpurun_out/${T}_100m_python_stages.err | tail -1 | cut -c1-420
cmp /tmp/${T}_100m.tsv $D/fusions.rank0.tsv && echo "100 M: fusions.tsv of the C++ workflow == of the Python mirror"
rm -rf $D
