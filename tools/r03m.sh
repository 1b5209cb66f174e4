#!/bin/bash
# round 3, eleventh GPU session: the front of the ingest in windows while the pieces arrive (agpu_ingest.hip) -- its tests, the 10 M sample against the reference's golden,
# the 100 M sample both ways (ARRIBA_INGEST_WINDOWS=0: everything behind the last piece, as before); whether pinning the pages of a mapped file scales over threads (feed_probe)
mkdir -p gpurun_out
T=r03m
export ARRIBA_BENCH_VERBOSE=1 ARRIBA_BENCH_REUSE=1
run() { # name, fragments, directory, steps, extra environment...
  local name=$1 fragments=$2 dir=$3 steps=$4; shift 4
  env "$@" timeout 400 python bench.py --fragments $fragments --steps $steps --warmup 1 --no-cpu-baseline --no-normal-pairs --keep $dir > gpurun_out/${T}_$name.json 2> gpurun_out/${T}_$name.err
  echo "$name exit $?"; grep "step done" gpurun_out/${T}_$name.err | tail -2 | cut -c1-420; tail -2 gpurun_out/${T}_$name.err | cut -c1-600
}
timeout 900 python -m pytest tests -x -q -m gpu -k "front_of_the_ingest or device_ingest_builds or at_scale or workflow_from_the_bam or crc" > gpurun_out/${T}_pytest.log 2>&1; echo "pytest exit $?"; tail -15 gpurun_out/${T}_pytest.log | cut -c1-400
D=/dev/shm/r03m_10m; mkdir -p $D
run 10m 10000000 $D 3 A=1
run 10m_at_the_end 10000000 $D 2 ARRIBA_INGEST_WINDOWS=0
rm -rf $D
D=/dev/shm/r03m_100m; mkdir -p $D
run 100m 100000000 $D 3 A=1
run 100m_at_the_end 100000000 $D 2 ARRIBA_INGEST_WINDOWS=0
timeout 200 tools/bin/feed_probe $D/bench.bam 256 16 2>&1 | tail -8 > gpurun_out/${T}_feed_probe.txt; cat gpurun_out/${T}_feed_probe.txt
rm -rf $D
python3 - <<'PY'
import json
for name in ("10m", "10m_at_the_end", "100m", "100m_at_the_end"):
    try:
        d = json.loads(open("gpurun_out/r03m_%s.json" % name).read().strip().splitlines()[-1])
    except Exception as e:
        print(name, "no line", e); continue
    k = d["kernel_ms"]
    print(name, round(d["ms_per_step"]), d["seconds_per_step"], d["read_chimeric_alignments_seconds"], {n: v for n, v in k.items() if any(w in n for w in ("segment", "record_parse", "run_", "group_", "fragment_pack", "name_order"))}, d.get("self_check", "")[:160])
PY
