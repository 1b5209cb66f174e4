#!/bin/bash
# First GPU run of round 3 (written at the end of round 2, when the GPU budget was spent): the measurements DESIGN.md section 8 asks for.
#   /usr/local/graft/bin/gpurun --timeout 2700 -- 'bash tools/r03a.sh'   (about 25 GPU-minutes)
mkdir -p gpurun_out
# 1. the GPU tier of the code as round 2 left it (31 tests; the sort path of agpu_shard_merge and the name-key check run on the GPU for the first time)
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/r03a_pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/r03a_pytest_gpu.log
# 2. 10 M end to end: the output side reworked, the task list of the second mismapper pass (on / off), the writer's own profile
timeout 120 python bench.py --fragments 10000000 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r03a_bench10m.json 2> gpurun_out/r03a_bench10m.err; echo "bench exit $?"; cut -c1-300 gpurun_out/r03a_bench10m.json
ARRIBA_MISMAPPER_WORKLIST=0 timeout 120 python bench.py --fragments 10000000 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r03a_bench10m_recursion.json 2> gpurun_out/r03a_bench10m_recursion.err; echo "bench exit $?"; cut -c1-300 gpurun_out/r03a_bench10m_recursion.json
ARRIBA_WRITER_PROFILE=1 timeout 120 python bench.py --fragments 10000000 --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/r03a_writer_profile.err; grep "writer\]\|step done" gpurun_out/r03a_writer_profile.err | tail -12 | cut -c1-400
# 3. the 100 M sample, one step
timeout 600 python bench.py --fragments 100000000 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r03a_bench100m.json 2> gpurun_out/r03a_bench100m.err; echo "bench exit $?"; cut -c1-300 gpurun_out/r03a_bench100m.json; grep "step done" gpurun_out/r03a_bench100m.err | cut -c1-600
# 4. the same 100 M step with the buffers from the stream-ordered pool (pages kept mapped between the ingest and the stages)
ARRIBA_DEVICE_POOL=1 timeout 600 python bench.py --fragments 100000000 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r03a_bench100m_pool.json 2> gpurun_out/r03a_bench100m_pool.err; echo "bench exit $?"; cut -c1-300 gpurun_out/r03a_bench100m_pool.json; grep "step done" gpurun_out/r03a_bench100m_pool.err | cut -c1-600
# 5. filter_homologs by a wavefront per gene pair: parity first (the tests that hold homologs), then the time at 10 M (homolog_verdict_kernel was 26 ms)
ARRIBA_HOMOLOG_WAVES=1 timeout 300 python -m pytest tests -x -q -m gpu -k "homolog or workflow_from_input_files" > gpurun_out/r03a_pytest_homolog_waves.log 2>&1; echo "pytest exit $?"; tail -2 gpurun_out/r03a_pytest_homolog_waves.log
ARRIBA_HOMOLOG_WAVES=1 timeout 120 python bench.py --fragments 10000000 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r03a_bench10m_homolog_waves.json 2> /dev/null; python3 -c "
import json; d = json.loads(open('gpurun_out/r03a_bench10m_homolog_waves.json').read().strip().splitlines()[-1]); print({k: v for k, v in d['kernel_ms'].items() if 'homolog' in k}, d['ms_per_step'])"
# 6. filter_in_vitro with the clipped ends summarised once per alignment: parity, then the time at 10 M (in_vitro_kernel was 31 ms, 218 GB of traffic)
ARRIBA_IN_VITRO_SUMMARY=1 timeout 300 python -m pytest tests -x -q -m gpu -k "chain_to or event_level or workflow_from_input_files" > gpurun_out/r03a_pytest_in_vitro_summary.log 2>&1; echo "pytest exit $?"; tail -2 gpurun_out/r03a_pytest_in_vitro_summary.log
ARRIBA_IN_VITRO_SUMMARY=1 timeout 120 python bench.py --fragments 10000000 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r03a_bench10m_in_vitro_summary.json 2> /dev/null; python3 -c "
import json; d = json.loads(open('gpurun_out/r03a_bench10m_in_vitro_summary.json').read().strip().splitlines()[-1]); print({k: v for k, v in d['kernel_ms'].items() if 'in_vitro' in k or 'clip_summary' in k}, d['ms_per_step'])"
# 7. CRC-32 of the stored blocks on the device: parity of the check itself, then what it costs at 10 M (bgzf_crc_kernel in kernel_ms)
timeout 120 python tools/r03a_crc.py 2>&1 | grep -v "^WARNING" | tail -2
ARRIBA_VERIFY_CRC=1 timeout 120 python bench.py --fragments 10000000 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r03a_bench10m_crc.json 2> /dev/null; python3 -c "
import json; d = json.loads(open('gpurun_out/r03a_bench10m_crc.json').read().strip().splitlines()[-1]); print({k: v for k, v in d['kernel_ms'].items() if 'bgzf' in k}, d['ms_per_step'])"
# 8. threads that read a piece of the file (32 by default): the feed of the 10 M sample took 0.20 s = 27 GB/s
for threads in 16 64 128; do ARRIBA_FEED_THREADS=$threads timeout 120 python bench.py --fragments 10000000 --steps 2 --warmup 1 --no-cpu-baseline 2> /dev/null | python3 -c "
import json, sys; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('feed threads $threads', d['read_chimeric_alignments_seconds'], d['ms_per_step'])"; done
