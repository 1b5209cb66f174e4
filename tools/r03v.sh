#!/bin/bash
# round 3: the name order of the fragments told by the windows of the ingest (run_name_order_kernel) instead of a pass behind the last piece
mkdir -p gpurun_out
T=r03v
export ARRIBA_BENCH_VERBOSE=1 ARRIBA_BENCH_REUSE=1
run() { # name, fragments, directory, steps, warmup, extra environment...
  local name=$1 fragments=$2 dir=$3 steps=$4 warmup=$5; shift 5
  env "$@" timeout 400 python bench.py --fragments $fragments --steps $steps --warmup $warmup --no-cpu-baseline --no-normal-pairs --keep $dir > gpurun_out/${T}_$name.json 2> gpurun_out/${T}_$name.err
  echo "$name exit $?"; grep "arriba_workflow_sample" gpurun_out/${T}_$name.err | tail -1 | cut -c1-420; tail -1 gpurun_out/${T}_$name.err | cut -c1-300
}
timeout 600 python -m pytest tests -x -q -m gpu -k "front_of_the_ingest or device_ingest or workflow_from_the_bam or other_kinds or in_parts" > gpurun_out/${T}_pytest.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/${T}_pytest.log | cut -c1-300
D=/dev/shm/r03v_10m; mkdir -p $D
run 10m 10000000 $D 3 2 A=1
rm -rf $D
D=/dev/shm/r03v_100m; mkdir -p $D
run 100m 100000000 $D 3 2 A=1
rm -rf $D
python3 - <<'PY'
import json
for name in ("10m", "100m"):
    try:
        d = json.loads(open("gpurun_out/r03v_%s.json" % name).read().strip().splitlines()[-1])
    except Exception as e:
        print(name, "no line", e); continue
    k = d["kernel_ms"]
    print(name, round(d["ms_per_step"]), round(d["value"]), d["seconds_per_step"], d["read_chimeric_alignments_seconds"], {n: v for n, v in k.items() if any(w in n for w in ("name_order", "fragment_", "group_replay"))}, d.get("self_check", "")[-100:])
PY
