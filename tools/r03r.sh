#!/bin/bash
# round 3, sixteenth GPU session: filter_mismappers without its first pass (every read to the sweep of the wavefront-per-read kernel), and that kernel with five wavefronts per SIMD
mkdir -p gpurun_out
T=r03r
export ARRIBA_BENCH_VERBOSE=1 ARRIBA_BENCH_REUSE=1
run() { # name, fragments, directory, steps, warmup, extra environment...
  local name=$1 fragments=$2 dir=$3 steps=$4 warmup=$5; shift 5
  env "$@" timeout 400 python bench.py --fragments $fragments --steps $steps --warmup $warmup --no-cpu-baseline --no-normal-pairs --keep $dir > gpurun_out/${T}_$name.json 2> gpurun_out/${T}_$name.err
  echo "$name exit $?"; grep "arriba_workflow_sample" gpurun_out/${T}_$name.err | tail -1 | cut -c1-420; tail -1 gpurun_out/${T}_$name.err | cut -c1-300
}
D=/dev/shm/r03r_10m; mkdir -p $D
run 10m 10000000 $D 2 2 A=1
run 10m_no_first_pass 10000000 $D 2 2 ARRIBA_MISMAPPER_FIRST_PASS=0
rm -rf $D
D=/dev/shm/r03r_100m; mkdir -p $D
run 100m 100000000 $D 2 2 A=1
run 100m_no_first_pass 100000000 $D 2 2 ARRIBA_MISMAPPER_FIRST_PASS=0
run 100m_five_waves 100000000 $D 2 2 ARRIBA_HEAVY_WAVES=5 ARRIBA_HEAVY_WORKGROUPS=5120
run 100m_five_waves_no_first_pass 100000000 $D 2 2 ARRIBA_HEAVY_WAVES=5 ARRIBA_HEAVY_WORKGROUPS=5120 ARRIBA_MISMAPPER_FIRST_PASS=0
rm -rf $D
python3 - <<'PY'
import json
for name in ("10m", "10m_no_first_pass", "100m", "100m_no_first_pass", "100m_five_waves", "100m_five_waves_no_first_pass"):
    try:
        d = json.loads(open("gpurun_out/r03r_%s.json" % name).read().strip().splitlines()[-1])
    except Exception as e:
        print(name, "no line", e); continue
    k = d["kernel_ms"]
    print(name, round(d["ms_per_step"]), round(d["value"]), d["seconds_per_step"], {n: v for n, v in k.items() if "mismapper" in n}, d.get("self_check", "")[-120:])
PY
