#!/bin/bash
# round 3, fifteenth GPU session: the file read by a thread of its own while the pieces are pushed (four pinned buffers); the vectors of the writer's rows kept from sample to sample
mkdir -p gpurun_out
T=r03q
export ARRIBA_BENCH_VERBOSE=1 ARRIBA_BENCH_REUSE=1
run() { # name, fragments, directory, steps, warmup, extra environment...
  local name=$1 fragments=$2 dir=$3 steps=$4 warmup=$5; shift 5
  env "$@" timeout 400 python bench.py --fragments $fragments --steps $steps --warmup $warmup --no-cpu-baseline --no-normal-pairs --keep $dir > gpurun_out/${T}_$name.json 2> gpurun_out/${T}_$name.err
  echo "$name exit $?"; grep "arriba_workflow_sample" gpurun_out/${T}_$name.err | tail -1 | cut -c1-420; tail -1 gpurun_out/${T}_$name.err | cut -c1-300
}
timeout 600 python -m pytest tests -x -q -m gpu -k "front_of_the_ingest or workflow_from_the_bam or bench_with_one_rank" > gpurun_out/${T}_pytest.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/${T}_pytest.log | cut -c1-400
D=/dev/shm/r03q_10m; mkdir -p $D
run 10m 10000000 $D 3 2 A=1
rm -rf $D
D=/dev/shm/r03q_100m; mkdir -p $D
run 100m 100000000 $D 3 2 A=1
run 100m_writer 100000000 $D 1 1 ARRIBA_WRITER_PROFILE=1
grep -E "^\[rows\]|^\[writer\]|^\[format\]|^\[output\]" gpurun_out/${T}_100m_writer.err | tail -24 > gpurun_out/${T}_writer_laps.txt; cat gpurun_out/${T}_writer_laps.txt
rm -rf $D
python3 - <<'PY'
import json
for name in ("10m", "100m"):
    try:
        d = json.loads(open("gpurun_out/r03q_%s.json" % name).read().strip().splitlines()[-1])
    except Exception as e:
        print(name, "no line", e); continue
    k = d["kernel_ms"]
    print(name, round(d["ms_per_step"]), d["value"], d["seconds_per_step"], d["read_chimeric_alignments_seconds"], d["output_side_seconds"], {n: v for n, v in k.items() if "mismapper" in n}, d.get("self_check", "")[:120])
PY
