"""A/B of the second mismapper pass (workgroups in flight x memo slots) on one 30 M sample: prints the kernel times per setting."""
import json, os, shutil, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import bench
from arriba_amd.pipeline import DevicePipeline, HostSession

fragments = int(sys.argv[1]) if len(sys.argv) > 1 else 30000000
settings = [tuple(int(x) for x in s.split(":")) for s in (sys.argv[2:] or ["1024:21", "4096:21", "8192:20"])]  # workgroups:log2(memo slots)[:steps of the first pass]
directory = tempfile.mkdtemp(prefix="arriba_ab_", dir="/dev/shm")
try:
    prefix, seconds = bench.generate_sample(fragments, 1000, directory)
    print("sample", fragments, "in", round(seconds, 1), "s", flush=True)
    session = HostSession(prefix + ".fa", prefix + ".gtf")
    pipeline = None
    for setting in settings:
        workgroups, slots, first_pass = (list(setting) + [65536])[:3]
        os.environ["ARRIBA_HEAVY_WORKGROUPS"], os.environ["ARRIBA_MEMO_SLOTS_LOG2"], os.environ["ARRIBA_FIRST_PASS_STEPS"] = str(workgroups), str(slots), str(first_pass)
        started = time.perf_counter()
        if pipeline is None:
            pipeline = DevicePipeline(session, device=0, bam=prefix + ".bam", piece_bytes=256 << 20, profiling=True)
        else:
            pipeline.set_profiling(True)
            pipeline.read_chimeric_alignments(prefix + ".bam", piece_bytes=256 << 20)
        pipeline.run_workflow(os.path.join(directory, "f.tsv"), os.path.join(directory, "d.tsv"))
        kernels = pipeline.kernel_profile()
        row = {"workgroups": workgroups, "memo_slots_log2": slots, "first_pass_steps": first_pass, "fusions": sum(1 for _ in open(os.path.join(directory, "f.tsv"))) - 1, "step_s": round(time.perf_counter() - started, 2), "filter_mismappers_ms": round(pipeline.timings["filter_mismappers"]["ms"], 1)}
        for name, ms, size in kernels:
            if name.startswith("mismapper_"):
                row[name] = round(ms, 1)
                row[name + "_reads"] = size // 300
        print(json.dumps(row), flush=True)
finally:
    shutil.rmtree(directory, ignore_errors=True)
