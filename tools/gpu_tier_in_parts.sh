# tools/gpu_tier_in_parts.sh PARTS -- the GPU tier as PARTS pytest processes, one after the other, each with its exit code (a process that dies at its end says so here; pytest's summary does not)
PARTS=${1:-4}
cd $GRAFT_REPO_ROOT
python -m pytest tests --collect-only -q -m gpu 2>/dev/null | grep "::" > gpurun_out/gpu_tier_ids.txt
TOTAL=$(wc -l < gpurun_out/gpu_tier_ids.txt)
PER=$(( (TOTAL + PARTS - 1) / PARTS ))
for part in $(seq 1 $PARTS); do
  FIRST=$(( (part - 1) * PER + 1 ))
  sed -n "${FIRST},$(( FIRST + PER - 1 ))p" gpurun_out/gpu_tier_ids.txt > gpurun_out/gpu_tier_part_$part.txt
  timeout 2400 python -m pytest -q -m gpu $(cat gpurun_out/gpu_tier_part_$part.txt | tr '\n' ' ') > gpurun_out/gpu_tier_part_$part.log 2>&1
  echo "part $part ($(head -1 gpurun_out/gpu_tier_part_$part.txt) ...): exit $?; $(grep -E "passed|failed" gpurun_out/gpu_tier_part_$part.log | tail -1); $(grep -E "double free|corruption|Aborted|core" gpurun_out/gpu_tier_part_$part.log | head -2 | tr '\n' ' ')"
done
