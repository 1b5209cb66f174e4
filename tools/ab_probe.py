#!/usr/bin/env python3
"""tools/ab_probe.py FRAGMENTS LIB [LIB ...] -- GPU-box probe: per-kernel times of the read-level stages and find_fusions with different builds of the
device library (arriba_amd/lib/libarriba_gpu_<variant>.so from `make variant`), same input.  Prints one JSON object."""
import ctypes
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    fragments = int(sys.argv[1])
    import bench
    from arriba_amd import _capi
    from arriba_amd.pipeline import DevicePipeline
    directory = tempfile.mkdtemp(prefix="ab_")
    session, prefix, _ = bench.generate_and_ingest(fragments, 1000, directory)
    results = {}
    for name in sys.argv[2:]:
        path = os.path.join(ROOT, "arriba_amd", "lib", name)
        if not os.path.exists(path):
            results[name] = "missing"
            continue
        api = _capi.bind_device_api(ctypes.CDLL(path))
        pipeline = DevicePipeline(session, api=api)
        pipeline.run_read_level()
        pipeline.find_fusions()
        pipeline.reset()
        pipeline.set_profiling(True)
        pipeline.run_read_level()
        pipeline.find_fusions()
        kernels = {}
        for kernel, ms, size in pipeline.kernel_profile():
            kernels[kernel] = round(kernels.get(kernel, 0.0) + ms, 3)
        results[name] = {"kernels": {k: v for k, v in sorted(kernels.items(), key=lambda item: -item[1])[:14]}, "stage_ms": {k: round(v["ms"], 3) for k, v in pipeline.timings.items()},
                         "candidates": pipeline.n_candidates}
        pipeline.close()
    print(json.dumps({"fragments": session.fragment_count, "results": results}))


if __name__ == "__main__":
    main()
