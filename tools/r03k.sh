#!/bin/bash
# round 3, ninth GPU session: splice-site bitmap in the walks of the sweep and where the wavefront time of the sweep goes; CRC-32 of the stored blocks (fast version, on by default);
# the 10 M bench sample against the reference's golden; the ingest tests
mkdir -p gpurun_out
T=r03k
export ARRIBA_BENCH_VERBOSE=1 ARRIBA_BENCH_REUSE=1
run() { # name, fragments, directory, steps, extra environment...
  local name=$1 fragments=$2 dir=$3 steps=$4; shift 4
  env "$@" timeout 300 python bench.py --fragments $fragments --steps $steps --warmup 1 --no-cpu-baseline --no-normal-pairs --keep $dir > gpurun_out/${T}_$name.json 2> gpurun_out/${T}_$name.err
  echo "$name exit $?"; grep "mismapper_heavy_kernel\]" gpurun_out/${T}_$name.err | head -2 | cut -c1-500; grep "step done" gpurun_out/${T}_$name.err | tail -1 | cut -c1-420
  cp $dir/fusions.rank0.tsv /tmp/${T}_$name.tsv
}
timeout 900 python -m pytest tests -x -q -m gpu -k "ingest or bench_sample or crc or workflow_from_the_bam or mismapper_stress" > gpurun_out/${T}_pytest.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/${T}_pytest.log
D=/dev/shm/r03k_10m; mkdir -p $D
run 10m 10000000 $D 3 A=1
run 10m_times 10000000 $D 1 ARRIBA_MISMAPPER_TIMES=1
run 10m_no_crc 10000000 $D 1 ARRIBA_VERIFY_CRC=0
rm -rf $D
D=/dev/shm/r03k_100m; mkdir -p $D
run 100m 100000000 $D 3 A=1
run 100m_times 100000000 $D 1 ARRIBA_MISMAPPER_TIMES=1
run 100m_no_crc 100000000 $D 1 ARRIBA_VERIFY_CRC=0
rm -rf $D
python3 - <<'PY'
import json
for name in ("10m", "10m_no_crc", "100m", "100m_no_crc"):
    d = json.loads(open("gpurun_out/r03k_%s.json" % name).read().strip().splitlines()[-1])
    k = d["kernel_ms"]
    print(name, round(d["ms_per_step"]), d["seconds_per_step"], d["read_chimeric_alignments_seconds"], {n: v for n, v in k.items() if any(w in n for w in ("bgzf", "group_replay", "mismapper", "group_names"))}, d.get("self_check", "")[:200])
PY
