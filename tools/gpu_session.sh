#!/bin/bash
# tools/gpu_session.sh TAG LEG [LEG ...] -- one lease of the GPU box (gpurun), every leg under a timeout of its own, everything written under gpurun_out/TAG_*.
# (replaces the per-session scripts tools/r02*.sh / r03*.sh of the earlier rounds.)  A leg is NAME=KIND[,SECONDS]:ARGUMENTS
#   tests   NAME=tests,600:-k "ingest or workflow"      python -m pytest tests -m gpu -x -q ARGUMENTS          -> TAG_NAME.log
#   bench   NAME=bench,300:--fragments 10000000 ...     python bench.py ARGUMENTS                               -> TAG_NAME.json / .err
#   trace   NAME=trace,400:--fragments ... --steps 1    rocprofv3 --kernel-trace --stats -- python bench.py ... -> TAG_NAME_kernel_stats.txt (+ the bench line TAG_NAME.json)
#   pmc     NAME=pmc,600:--fragments ... --steps 1      FETCH_SIZE and WRITE_SIZE in separate passes            -> TAG_NAME_pmc.json, TAG_NAME_pmc_summary.txt
#   sh      NAME=sh,120:any shell command                                                                       -> TAG_NAME.log
# ARRIBA_GIT_HEAD (the commit the device code is at; .git does not travel) goes into the PMC record.  Environment for a leg: put VAR=value in front of the arguments of a sh leg, or export it before calling this script inside the gpurun command.
TAG=$1; shift
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
ROOT=$(pwd)
mkdir -p gpurun_out
export TMPDIR=/tmp
for LEG in "$@"; do
  NAME=${LEG%%=*}; REST=${LEG#*=}; HEAD=${REST%%:*}; ARGS=${REST#*:}
  if [ "$HEAD" = "$REST" ]; then ARGS=""; fi
  KIND=${HEAD%%,*}; LIMIT=${HEAD#*,}; if [ "$LIMIT" = "$HEAD" ]; then LIMIT=600; fi
  OUT=gpurun_out/${TAG}_${NAME}
  STARTED=$(date +%s); RC=0
  case $KIND in
    tests) eval "timeout $LIMIT python -m pytest tests -m gpu -x -q $ARGS" > $OUT.log 2>&1; RC=$?; echo "exit $RC" >> $OUT.log; tail -4 $OUT.log | cut -c1-400 ;;
    bench) eval "timeout $LIMIT python bench.py $ARGS" > $OUT.json 2> $OUT.err; RC=$?; echo "exit $RC" >> $OUT.err; tail -3 $OUT.err | cut -c1-400; tail -1 $OUT.json | cut -c1-600 ;;
    trace) (cd /tmp && eval "timeout $LIMIT rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_${TAG}_${NAME} -o trace -- python $ROOT/bench.py $ARGS" > $ROOT/$OUT.json 2> $ROOT/$OUT.err); echo "exit $?" >> $OUT.err
           find gpurun_out/prof_${TAG}_${NAME} -name '*.db' | head -1 | xargs -I{} python tools/rocprof_summary.py {} "${TAG}: rocprofv3 --kernel-trace --stats -- python bench.py $ARGS" > ${OUT}_kernel_stats.txt 2>&1
           rm -rf gpurun_out/prof_${TAG}_${NAME}; head -12 ${OUT}_kernel_stats.txt | cut -c1-200 ;;
    pmc)   for COUNTER in FETCH_SIZE WRITE_SIZE; do
             (cd /tmp && eval "timeout $LIMIT rocprofv3 --pmc $COUNTER --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_${TAG}_${NAME}_$COUNTER -o pmc -- python $ROOT/bench.py $ARGS" > $ROOT/${OUT}_$COUNTER.json 2> $ROOT/${OUT}_$COUNTER.err)
           done
           python tools/pmc_summary.py ${OUT}_pmc_kernels.json gpurun_out/pmc_${TAG}_${NAME}_FETCH_SIZE gpurun_out/pmc_${TAG}_${NAME}_WRITE_SIZE > ${OUT}_pmc_summary.txt 2>&1
           python - <<PY
import json
kernels = json.load(open("${OUT}_pmc_kernels.json"))
line = json.loads([l for l in open("${OUT}_FETCH_SIZE.json").read().splitlines() if l.startswith("{")][-1])
import sys
sys.path.insert(0, ".")
import bench
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, with --kernel-trace only), bench.py $ARGS; KB summed over the dispatches of a kernel", "fragments": line["config"]["fragments_per_gpu"],
           "git": "${ARRIBA_GIT_HEAD}", "device_code_sha256": bench.device_code_digest(), "kernels": kernels}, open("${OUT}_pmc.json", "w"), indent=1, sort_keys=True)
PY
           rm -rf gpurun_out/pmc_${TAG}_${NAME}_FETCH_SIZE gpurun_out/pmc_${TAG}_${NAME}_WRITE_SIZE; head -30 ${OUT}_pmc_summary.txt | cut -c1-200 ;;
    sh)    eval "timeout $LIMIT $ARGS" > $OUT.log 2>&1; RC=$?; echo "exit $RC" >> $OUT.log; tail -5 $OUT.log | cut -c1-400 ;;
    *)     echo "unknown kind of leg: $KIND" ;;
  esac
  echo "[$TAG $NAME] $KIND took $(( $(date +%s) - STARTED )) s, exit $RC"
  # STOP_ON_FAIL=1: GPU minutes are scarce -- a leg that fails (a hang that ran into its time limit) ends the session instead of letting the legs behind it run into the same wall
  if [ -n "$STOP_ON_FAIL" ] && [ "$RC" != "0" ]; then echo "[$TAG] stopping behind the failed leg $NAME"; break; fi
done
