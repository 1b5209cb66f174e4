cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
(nproc; free -g; df -h /tmp /dev/shm; ulimit -l; cat /proc/meminfo | head -5; rocm-smi --showmeminfo vram | head; ls /opt/rocm/lib | grep -i -E "nvcomp|hipcomp|deflate" ) > gpurun_out/r02a_box.txt 2>&1
python - > gpurun_out/r02a_pcie.txt 2>&1 <<'PY'
import torch, time
x = torch.empty(1<<30, dtype=torch.uint8).pin_memory()
d = torch.empty(1<<30, dtype=torch.uint8, device="cuda")
for _ in range(2): d.copy_(x, non_blocking=True); torch.cuda.synchronize()
t=time.time()
for _ in range(8): d.copy_(x, non_blocking=True)
torch.cuda.synchronize(); dt=time.time()-t
print("H2D pinned GB/s", 8*(1<<30)/dt/1e9)
t=time.time()
for _ in range(8): x.copy_(d, non_blocking=True)
torch.cuda.synchronize(); dt=time.time()-t
print("D2H pinned GB/s", 8*(1<<30)/dt/1e9)
e = torch.empty(1<<30, dtype=torch.uint8, device="cuda")
t=time.time()
for _ in range(20): e.copy_(d)
torch.cuda.synchronize(); dt=time.time()-t
print("D2D copy GB/s (read+write)", 2*20*(1<<30)/dt/1e9)
y = torch.empty(1<<30, dtype=torch.uint8)
t=time.time(); d.copy_(y); torch.cuda.synchronize(); print("H2D pageable GB/s", (1<<30)/(time.time()-t)/1e9)
PY
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r02a_pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/r02a_pytest_gpu.log
timeout 600 python bench.py --fragments 2000000 --steps 3 --warmup 1 > gpurun_out/r02a_bench.json 2> gpurun_out/r02a_bench.err; echo "bench exit $?" >> gpurun_out/r02a_bench.err
tail -3 gpurun_out/r02a_pytest_gpu.log; cat gpurun_out/r02a_box.txt gpurun_out/r02a_pcie.txt; cut -c1-600 gpurun_out/r02a_bench.json; tail -3 gpurun_out/r02a_bench.err
