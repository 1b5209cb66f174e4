#!/bin/bash
# round 3: bgzf_crc_kernel in four staged rounds (17 KB of LDS instead of 75)
mkdir -p gpurun_out
T=r03x
export ARRIBA_BENCH_VERBOSE=1 ARRIBA_BENCH_REUSE=1
run() { local name=$1 fragments=$2 dir=$3 steps=$4 warmup=$5; shift 5
  env "$@" timeout 300 python bench.py --fragments $fragments --steps $steps --warmup $warmup --no-cpu-baseline --no-normal-pairs --keep $dir > gpurun_out/${T}_$name.json 2> gpurun_out/${T}_$name.err
  echo "$name exit $?"; tail -1 gpurun_out/${T}_$name.err | cut -c1-300; }
timeout 300 python -m pytest tests -x -q -m gpu -k "crc or front_of_the_ingest or workflow_from_the_bam" > gpurun_out/${T}_pytest.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/${T}_pytest.log | cut -c1-300
D=/dev/shm/r03x_100m; mkdir -p $D
run 100m 100000000 $D 2 2 A=1
rm -rf $D
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r03x_100m.json").read().strip().splitlines()[-1])
k = d["kernel_ms"]
print(round(d["ms_per_step"]), round(d["value"]), d["seconds_per_step"], d["read_chimeric_alignments_seconds"], {n: v for n, v in k.items() if any(w in n for w in ("bgzf", "group_replay"))}, d.get("kernel_launches_per_step"))
PY
