#!/bin/bash
# round 3, sixth GPU session: thread pools sized by the CPU quota of the container (16 CPUs behind 256 hardware threads), groups of the ingest replayed in stream order,
# fragments per gene counted in LDS; the ingest tests of the GPU tier; the two switches of round 2 (filter_in_vitro with clip summaries, filter_homologs by wavefronts)
mkdir -p gpurun_out
T=r03h
export ARRIBA_BENCH_VERBOSE=1 ARRIBA_BENCH_REUSE=1
run() { # name, fragments, directory, steps, extra environment...
  local name=$1 fragments=$2 dir=$3 steps=$4; shift 4
  env "$@" timeout 300 python bench.py --fragments $fragments --steps $steps --warmup 1 --no-cpu-baseline --keep $dir > gpurun_out/${T}_$name.json 2> gpurun_out/${T}_$name.err
  echo "$name exit $?"; grep "step done" gpurun_out/${T}_$name.err | tail -2 | cut -c1-420; grep "^\[writer\] rows\|^\[writer\] thread" gpurun_out/${T}_$name.err | tail -2
  cp $dir/fusions.rank0.tsv /tmp/${T}_$name.tsv
}
timeout 900 python -m pytest tests -x -q -m gpu -k "ingest or workflow_from_the_bam or one_sample" > gpurun_out/${T}_pytest.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/${T}_pytest.log
D=/dev/shm/r03h_10m; mkdir -p $D
run 10m 10000000 $D 3 ARRIBA_WRITER_PROFILE=1
run 10m_round3_before 10000000 $D 1 ARRIBA_CPU_BUDGET=128
cmp /tmp/${T}_10m.tsv /tmp/${T}_10m_round3_before.tsv && echo "10 M: identical"
ARRIBA_IN_VITRO_SUMMARY=1 timeout 300 python -m pytest tests -x -q -m gpu -k "chain_to or event_level or workflow_from_input_files" > gpurun_out/${T}_pytest_in_vitro_summary.log 2>&1; echo "pytest in_vitro_summary exit $?"; tail -2 gpurun_out/${T}_pytest_in_vitro_summary.log
ARRIBA_HOMOLOG_WAVES=1 timeout 300 python -m pytest tests -x -q -m gpu -k "homolog or workflow_from_input_files" > gpurun_out/${T}_pytest_homolog_waves.log 2>&1; echo "pytest homolog_waves exit $?"; tail -2 gpurun_out/${T}_pytest_homolog_waves.log
run 10m_in_vitro_summary 10000000 $D 1 ARRIBA_IN_VITRO_SUMMARY=1
run 10m_homolog_waves 10000000 $D 1 ARRIBA_HOMOLOG_WAVES=1
cmp /tmp/${T}_10m.tsv /tmp/${T}_10m_in_vitro_summary.tsv && cmp /tmp/${T}_10m.tsv /tmp/${T}_10m_homolog_waves.tsv && echo "10 M: identical with both switches"
rm -rf $D
D=/dev/shm/r03h_100m; mkdir -p $D
run 100m 100000000 $D 3 ARRIBA_WRITER_PROFILE=1
run 100m_switches 100000000 $D 1 ARRIBA_IN_VITRO_SUMMARY=1 ARRIBA_HOMOLOG_WAVES=1
cmp /tmp/${T}_100m.tsv /tmp/${T}_100m_switches.tsv && echo "100 M: identical with both switches"
run 100m_feed8 100000000 $D 1 ARRIBA_FEED_THREADS=8
run 100m_feed12 100000000 $D 1 ARRIBA_FEED_THREADS=12
run 100m_writer8 100000000 $D 1 ARRIBA_WRITER_THREADS=8
rm -rf $D
python3 - <<'PY'
import json
for name in ("10m", "10m_in_vitro_summary", "10m_homolog_waves", "100m", "100m_switches"):
    d = json.loads(open("gpurun_out/r03h_%s.json" % name).read().strip().splitlines()[-1])
    k = d["kernel_ms"]
    print(name, round(d["ms_per_step"]), {n: v for n, v in k.items() if any(w in n for w in ("in_vitro", "clip_summary", "homolog", "gene_read_count", "group_replay", "group_names", "group_head", "fragment_pack", "fragment_layout", "name_order", "group_rank", "first records"))})
PY
