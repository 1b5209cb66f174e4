#!/bin/bash
# round 3, eighth GPU session: the whole GPU tier (everything of the round so far), then the 10 M and 100 M lines: replay and pack of the ingest read their slice of the stream from LDS,
# the writer gets its rows in the order of the read lists and piles up packed bases page-wise
mkdir -p gpurun_out
T=r03j
export ARRIBA_BENCH_VERBOSE=1 ARRIBA_BENCH_REUSE=1
run() { # name, fragments, directory, steps, extra environment...
  local name=$1 fragments=$2 dir=$3 steps=$4; shift 4
  env "$@" timeout 300 python bench.py --fragments $fragments --steps $steps --warmup 1 --no-cpu-baseline --keep $dir > gpurun_out/${T}_$name.json 2> gpurun_out/${T}_$name.err
  echo "$name exit $?"; grep "step done" gpurun_out/${T}_$name.err | tail -1 | cut -c1-420; grep "^\[writer\] rows\|^\[writer\] thread\|^\[writer\] fusion" gpurun_out/${T}_$name.err | tail -3
  cp $dir/fusions.rank0.tsv /tmp/${T}_$name.tsv
}
timeout 1200 python -m pytest tests -x -q -m gpu --durations=12 > gpurun_out/${T}_pytest.log 2>&1; echo "pytest exit $?"; tail -18 gpurun_out/${T}_pytest.log | cut -c1-200
D=/dev/shm/r03j_10m; mkdir -p $D
run 10m 10000000 $D 3 ARRIBA_WRITER_PROFILE=2
sha256sum $D/fusions.rank0.tsv $D/bench.bam | cut -c1-80
rm -rf $D
D=/dev/shm/r03j_100m; mkdir -p $D
run 100m 100000000 $D 3 ARRIBA_WRITER_PROFILE=2
sha256sum $D/fusions.rank0.tsv | cut -c1-80
rm -rf $D
python3 - <<'PY'
import json
for name in ("10m", "100m"):
    d = json.loads(open("gpurun_out/r03j_%s.json" % name).read().strip().splitlines()[-1])
    print(name, round(d["ms_per_step"]), d["seconds_per_step"], d["read_chimeric_alignments_seconds"], d["output_side_seconds"])
    print("   ", sorted(d["kernel_ms"].items(), key=lambda kv: -kv[1])[:22])
PY
