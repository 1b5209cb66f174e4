#!/bin/bash
# what the host side of the GPU box gives a process: CPU quota of the container (cgroup), affinity, memory bandwidth of parallel copies
mkdir -p gpurun_out
{
echo "== cgroup"; cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us /sys/fs/cgroup/cpu/cpu.cfs_period_us 2>/dev/null; cat /proc/self/cgroup | head -5
echo "== cpu.stat before"; cat /sys/fs/cgroup/cpu.stat 2>/dev/null | head -8
echo "== affinity"; taskset -p $$ 2>/dev/null; nproc; python3 -c "import os; print(len(os.sched_getaffinity(0)))"
echo "== burn: N threads x 1 s of arithmetic, wall time and CPU time"
python3 - <<'PY'
import os, time, subprocess, sys
code = "import time\nt=time.process_time()\nx=0\nwhile time.process_time()-t<1.0:\n    x+=1\n"
for n in (1, 4, 8, 16, 32, 64, 128):
    t0 = time.time()
    ps = [subprocess.Popen([sys.executable, "-c", code]) for _ in range(n)]
    [p.wait() for p in ps]
    print("threads %3d: wall %.2f s (1.0 = no throttling)" % (n, time.time() - t0))
PY
echo "== cpu.stat after"; cat /sys/fs/cgroup/cpu.stat 2>/dev/null | head -8
echo "== memory"; free -g | head -2; cat /sys/kernel/mm/transparent_hugepage/shmem_enabled 2>/dev/null; df -h /dev/shm | tail -1
} > gpurun_out/r03g_probe.txt 2>&1
cat gpurun_out/r03g_probe.txt
