#!/bin/bash
# round 3, last GPU session: CRC-32 of the stored blocks with the payload staged in LDS (traffic of bgzf_crc_kernel: one FETCH_SIZE pass at 10 M), the bench at 10 M and 10^8,
# the whole GPU tier on the round's last state
mkdir -p gpurun_out
T=r03t
R=$GRAFT_REPO_ROOT
export ARRIBA_BENCH_VERBOSE=1 ARRIBA_BENCH_REUSE=1
run() { # name, fragments, directory, steps, warmup, extra environment...
  local name=$1 fragments=$2 dir=$3 steps=$4 warmup=$5; shift 5
  env "$@" timeout 400 python bench.py --fragments $fragments --steps $steps --warmup $warmup --no-cpu-baseline --no-normal-pairs --keep $dir > gpurun_out/${T}_$name.json 2> gpurun_out/${T}_$name.err
  echo "$name exit $?"; grep "arriba_workflow_sample" gpurun_out/${T}_$name.err | tail -1 | cut -c1-420; tail -1 gpurun_out/${T}_$name.err | cut -c1-300
}
D=/dev/shm/r03t_10m; mkdir -p $D
run 10m 10000000 $D 3 2 A=1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${T}_FETCH_SIZE -o pmc -- python $R/bench.py --fragments 10000000 --no-cpu-baseline --no-normal-pairs --steps 1 --warmup 0 --keep $D > $R/gpurun_out/${T}_pmc_FETCH_SIZE.log 2>&1; echo "pmc exit $?"
cd $R
python tools/pmc_summary.py gpurun_out/${T}_pmc_kernels.json gpurun_out/pmc_${T}_FETCH_SIZE > gpurun_out/${T}_pmc_summary.txt 2>&1
rm -rf gpurun_out/pmc_${T}_FETCH_SIZE
grep -E "bgzf_crc_kernel|bgzf_unwrap_kernel|group_replay_kernel" gpurun_out/${T}_pmc_summary.txt | cut -c1-200
rm -rf $D
D=/dev/shm/r03t_100m; mkdir -p $D
run 100m 100000000 $D 3 2 A=1
rm -rf $D
python3 - <<'PY'
import json
for name in ("10m", "100m"):
    try:
        d = json.loads(open("gpurun_out/r03t_%s.json" % name).read().strip().splitlines()[-1])
    except Exception as e:
        print(name, "no line", e); continue
    k = d["kernel_ms"]
    print(name, round(d["ms_per_step"]), round(d["value"]), d["seconds_per_step"], d["read_chimeric_alignments_seconds"], {n: v for n, v in k.items() if any(w in n for w in ("group_replay", "bgzf", "group_names", "record_parse"))}, d.get("self_check", "")[-100:])
PY
timeout 1100 python -m pytest tests -x -q -m gpu > gpurun_out/${T}_pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -4 gpurun_out/${T}_pytest_gpu.log | cut -c1-300
