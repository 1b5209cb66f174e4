#!/bin/bash
# round 3, seventh GPU session: how the bytes of a file reach HBM fastest (tools/feed_probe.cpp), what the atomics on coverage_t cost the replay of the ingest, CRC-32 of the stored
# blocks on the device (parity + cost), the writer with less work per row, the RCCL compositions on a communicator of one rank, in-vitro summaries and homolog wavefronts as defaults
mkdir -p gpurun_out
T=r03i
export ARRIBA_BENCH_VERBOSE=1 ARRIBA_BENCH_REUSE=1
run() { # name, fragments, directory, steps, extra environment...
  local name=$1 fragments=$2 dir=$3 steps=$4; shift 4
  env "$@" timeout 300 python bench.py --fragments $fragments --steps $steps --warmup 1 --no-cpu-baseline --keep $dir > gpurun_out/${T}_$name.json 2> gpurun_out/${T}_$name.err
  echo "$name exit $?"; grep "step done" gpurun_out/${T}_$name.err | tail -1 | cut -c1-420; grep "^\[writer\] rows\|^\[writer\] thread\|^\[writer\] fusion" gpurun_out/${T}_$name.err | tail -3
  cp $dir/fusions.rank0.tsv /tmp/${T}_$name.tsv
}
timeout 600 python -m pytest tests -x -q -m gpu -k "rccl or homolog or event_level or chain_to or workflow_from_input_files or one_sample" > gpurun_out/${T}_pytest.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/${T}_pytest.log
timeout 120 python tools/r03a_crc.py 2>&1 | grep -v "^WARNING" | tail -3
D=/dev/shm/r03i_10m; mkdir -p $D
run 10m 10000000 $D 3 ARRIBA_WRITER_PROFILE=2
run 10m_crc 10000000 $D 1 ARRIBA_VERIFY_CRC=1
cmp /tmp/${T}_10m.tsv /tmp/${T}_10m_crc.tsv && echo "10 M: identical with the CRC check"
timeout 120 tools/bin/feed_probe $D/bench.bam 256 16 2>&1 | tail -8
timeout 120 tools/bin/feed_probe $D/bench.bam 256 8 2>&1 | head -2
rm -rf $D
D=/dev/shm/r03i_100m; mkdir -p $D
run 100m 100000000 $D 3 ARRIBA_WRITER_PROFILE=2
run 100m_crc 100000000 $D 1 ARRIBA_VERIFY_CRC=1
run 100m_no_coverage 100000000 $D 1 ARRIBA_INGEST_SKIP_COVERAGE=1
rm -rf $D
python3 - <<'PY'
import json
for name in ("10m", "10m_crc", "100m", "100m_crc", "100m_no_coverage"):
    d = json.loads(open("gpurun_out/r03i_%s.json" % name).read().strip().splitlines()[-1])
    k = d["kernel_ms"]
    print(name, round(d["ms_per_step"]), d["seconds_per_step"], {n: v for n, v in k.items() if any(w in n for w in ("bgzf", "group_replay", "in_vitro", "homolog", "gene_read_count"))})
PY
