bash tools/memo_sweep.sh r06m "ARRIBA_MEMO_SLOTS_LOG2=16 ARRIBA_TASK_CAPACITY_LOG2=15" "ARRIBA_MEMO_SLOTS_LOG2=20 ARRIBA_TASK_CAPACITY_LOG2=17"
for s in "ARRIBA_MEMO_SLOTS_LOG2=16 ARRIBA_TASK_CAPACITY_LOG2=15" "ARRIBA_MEMO_SLOTS_LOG2=20 ARRIBA_TASK_CAPACITY_LOG2=17"; do
  tag=$(echo "$s" | tr ' ' '+')
  env $s timeout 900 python bench.py --stress --fragments 10000000 --steps 2 --warmup 1 --no-cpu-baseline --keep /tmp/s10 > gpurun_out/stress_$tag.json 2> gpurun_out/stress_$tag.err
  python - <<PY
import json
line=[t for t in open("gpurun_out/stress_$tag.json").read().splitlines() if t.startswith("{") and '"metric"' in t]
if line:
    d=json.loads(line[-1]); k=d["kernel_ms_alone"]
    print("stress $tag: ms/step", round(d["ms_per_step"]), "hbm", d.get("hbm_used_GB"), {n:v for n,v in k.items() if "mismapper_heavy" in n})
else:
    print("stress $tag: no line"); print(open("gpurun_out/stress_$tag.err").read()[-600:])
PY
done
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "mismapper or schedule or stress" 2>&1 | tail -3
