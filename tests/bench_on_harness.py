"""TEST-ONLY: runs bench.py's main() with the device library replaced by the host stepping harness (tests/emu) and torch.cuda stubbed, so that the control flow
of the bench -- argument handling, the distributed branch (one sample over the ranks, gloo), the JSON line -- is exercised where there is no GPU.  The numbers it
prints mean nothing.  bench.py itself has no such switch: without a GPU it refuses to run."""
import ctypes
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

world = int(os.environ.get("WORLD_SIZE", "1"))
torch.cuda.is_available = lambda: True
torch.cuda.device_count = lambda: max(world, 1)
torch.cuda.set_device = lambda device: None
torch.cuda.synchronize = lambda *args, **kwargs: None
torch.cuda.get_device_properties = lambda device: types.SimpleNamespace(total_memory=288 << 30)
os.environ["ARRIBA_BENCH_BACKEND"] = "gloo"
os.environ["ARRIBA_WORKFLOW_LIBRARY"] = os.path.join(ROOT, "tests", "emu", "libworkflow_on_harness.so")  # arriba_workflow_sample over the harness

from arriba_amd import _capi  # noqa: E402

_bind = _capi.bind_device_api
_harness = ctypes.CDLL(os.path.join(ROOT, "tests", "emu", "libemu.so"))
_capi.bind_device_api = lambda library, prefix: _bind(_harness, "emu_")

import subprocess  # noqa: E402
subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu"), "libworkflow_on_harness.so"], check=True)
import bench  # noqa: E402

if __name__ == "__main__":
    bench.main()
