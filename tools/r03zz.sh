#!/bin/bash
# round 3: the read-name groups of the batch from the QNAME runs the windows have numbered (fragment_layout_kernel without its two record loads per fragment)
mkdir -p gpurun_out
T=r03zz
export ARRIBA_BENCH_VERBOSE=1
timeout 200 python -m pytest tests -x -q -m gpu -k "front_of_the_ingest or device_ingest_builds" > gpurun_out/${T}_pytest.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/${T}_pytest.log | cut -c1-300
timeout 200 python bench.py --fragments 100000000 --steps 2 --warmup 2 --no-cpu-baseline --no-normal-pairs > gpurun_out/${T}_100m.json 2> gpurun_out/${T}_100m.err; echo "100m exit $?"; tail -1 gpurun_out/${T}_100m.err | cut -c1-300
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r03zz_100m.json").read().strip().splitlines()[-1])
k = d["kernel_ms"]
print(round(d["ms_per_step"]), round(d["value"]), d["seconds_per_step"], d["read_chimeric_alignments_seconds"], {n: v for n, v in k.items() if any(w in n for w in ("fragment_", "run_name", "name_order"))})
PY
