# A/B measurements of mismapper_heavy_kernel on one GPU lease: KNOB=VALUE settings, one bench run each on the kept 10^8-fragment sample
mkdir -p gpurun_out/$1
shift
for settings in "$@"; do
  setting=$(echo "$settings" | tr ' ' '+')
  env $settings timeout 900 python bench.py --fragments 100000000 --steps 2 --warmup 1 --no-cpu-baseline --no-deflated-leg --no-stress-leg --no-normal-pairs --keep /tmp/b100 > gpurun_out/ab_$setting.json 2> gpurun_out/ab_$setting.err
  python - <<PY
import json
line=[t for t in open("gpurun_out/ab_$setting.json").read().splitlines() if t.startswith("{") and '"metric"' in t]
if line:
    d=json.loads(line[-1]); k=d["kernel_ms_alone"]
    print("$setting: ms/step", round(d["ms_per_step"]), "heavy", {n:v for n,v in k.items() if "mismapper" in n}, "sum", d["kernel_ms_alone_sum"])
else:
    print("$setting: no line")
PY
done
