#!/bin/bash
# round 3: the writer with the sort's look-ups hoisted and the rows written by all threads (pwrite)
mkdir -p gpurun_out
T=r03z
export ARRIBA_BENCH_VERBOSE=1 ARRIBA_BENCH_REUSE=1
run() { local name=$1 fragments=$2 dir=$3 steps=$4 warmup=$5; shift 5
  env "$@" timeout 300 python bench.py --fragments $fragments --steps $steps --warmup $warmup --no-cpu-baseline --no-normal-pairs --keep $dir > gpurun_out/${T}_$name.json 2> gpurun_out/${T}_$name.err
  echo "$name exit $?"; grep "^\[writer\]" gpurun_out/${T}_$name.err | tail -3 | cut -c1-200; tail -1 gpurun_out/${T}_$name.err | cut -c1-300; }
D=/dev/shm/r03z_10m; mkdir -p $D
run 10m 10000000 $D 3 2 ARRIBA_WRITER_PROFILE=1
rm -rf $D
D=/dev/shm/r03z_100m; mkdir -p $D
run 100m 100000000 $D 2 2 ARRIBA_WRITER_PROFILE=1
rm -rf $D
python3 - <<'PY'
import json
for name in ("10m", "100m"):
    d = json.loads(open("gpurun_out/r03z_%s.json" % name).read().strip().splitlines()[-1])
    print(name, round(d["ms_per_step"]), round(d["value"]), d["seconds_per_step"], d["output_side_seconds"], d.get("self_check", "")[-90:])
PY
