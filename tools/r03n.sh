#!/bin/bash
# round 3, twelfth GPU session: pieces unwrapped and checked on a stream of their own (the copy of the next piece never waits for a kernel); how the size of the windows of the
# ingest moves the feed (time inside ahost_bam_next / inside agpu_ingest_push*); pinning the pages of a mapped file by several threads in a fresh process (feed_probe d)
mkdir -p gpurun_out
T=r03n
export ARRIBA_BENCH_VERBOSE=1 ARRIBA_BENCH_REUSE=1
run() { # name, fragments, directory, steps, extra environment...
  local name=$1 fragments=$2 dir=$3 steps=$4; shift 4
  env "$@" timeout 400 python bench.py --fragments $fragments --steps $steps --warmup 1 --no-cpu-baseline --no-normal-pairs --keep $dir > gpurun_out/${T}_$name.json 2> gpurun_out/${T}_$name.err
  echo "$name exit $?"; grep "arriba_workflow_sample" gpurun_out/${T}_$name.err | tail -2 | cut -c1-420; tail -1 gpurun_out/${T}_$name.err | cut -c1-300
}
timeout 900 python -m pytest tests -x -q -m gpu -k "front_of_the_ingest or device_ingest_builds or at_scale or workflow_from_the_bam or crc" > gpurun_out/${T}_pytest.log 2>&1; echo "pytest exit $?"; tail -4 gpurun_out/${T}_pytest.log | cut -c1-400
D=/dev/shm/r03n_100m; mkdir -p $D
run 100m 100000000 $D 2 A=1
run 100m_512m 100000000 $D 2 ARRIBA_INGEST_WINDOWS=536870912,1048576
run 100m_1g 100000000 $D 2 ARRIBA_INGEST_WINDOWS=1073741824,1048576
run 100m_at_the_end 100000000 $D 2 ARRIBA_INGEST_WINDOWS=0
( timeout 100 tools/bin/feed_probe $D/bench.bam 256 16 d 8; timeout 100 tools/bin/feed_probe $D/bench.bam 256 16 d 2; timeout 100 tools/bin/feed_probe $D/bench.bam 256 16 d 16; timeout 100 tools/bin/feed_probe $D/bench.bam 64 16 d 16 ) 2>&1 | tail -8 > gpurun_out/${T}_feed_probe.txt; cat gpurun_out/${T}_feed_probe.txt
rm -rf $D
python3 - <<'PY'
import json
for name in ("100m", "100m_512m", "100m_1g", "100m_at_the_end"):
    try:
        d = json.loads(open("gpurun_out/r03n_%s.json" % name).read().strip().splitlines()[-1])
    except Exception as e:
        print(name, "no line", e); continue
    k = d["kernel_ms"]
    print(name, round(d["ms_per_step"]), d["seconds_per_step"], d["read_chimeric_alignments_seconds"], {n: v for n, v in k.items() if any(w in n for w in ("segment", "record_parse", "run_", "group_", "fragment_pack", "name_order", "bgzf"))})
PY
