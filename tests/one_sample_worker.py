"""Worker of the multi-process test of arriba_amd/one_sample.py (one rank): its part of the records of the BAM file, the all-gather of the parts, the
whole workflow with filter_mismappers shared out; on rank 0 the comparison with the single-process pipeline over the whole file.
Launched by tests/test_one_sample.py through torch.distributed.run."""
import ctypes
import hashlib
import json
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from arriba_amd import _capi  # noqa: E402
from arriba_amd.one_sample import OneSamplePipeline  # noqa: E402
from arriba_amd.pipeline import DevicePipeline, HostSession  # noqa: E402


def digest_of_batch(pipeline):
    rows, _ = pipeline.batch_rows()
    digest = hashlib.md5()
    for key in sorted(rows):
        if isinstance(rows[key], np.ndarray):
            digest.update(key.encode())
            digest.update(np.ascontiguousarray(rows[key]).tobytes())
    return digest.hexdigest()


def main():
    prefix, backend_api, out_path, bam = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4]
    backend = "gloo"
    device = 0
    if backend_api == "gpu":
        import torch
        device = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
        backend = "nccl" if torch.cuda.device_count() >= int(os.environ.get("WORLD_SIZE", "1")) else "gloo"  # (several ranks on one GPU: the collectives go through the host)
    dist.init_process_group(backend)
    rank, world = dist.get_rank(), dist.get_world_size()
    api = _capi.bind_device_api(ctypes.CDLL(os.path.join(ROOT, "tests", "emu", "libemu.so")), "emu_") if backend_api == "emu" else None
    report = {"rank": rank, "problems": []}
    try:
        session = HostSession(prefix + ".fa", prefix + ".gtf")
        pipeline = OneSamplePipeline(session, bam, api=api, device=device, piece_bytes=4 << 20)
        report.update({"fragments": pipeline.n, "part_bytes": pipeline.exchange["part_bytes"], "records": int(pipeline.ingest_result.records), "mapped_reads": int(pipeline.ingest_result.mapped_reads),
                       "batch": digest_of_batch(pipeline)})
        log = []
        outputs = [out_path + ".fusions.tsv", out_path + ".discarded.tsv"]
        pipeline.run_workflow(outputs[0], outputs[1], log=lambda stage, count: log.append([stage, int(count)]))
        report.update({"log": log, "mismapper_jobs": pipeline.exchange["mismapper_jobs"], "filters": hashlib.md5(pipeline.filters().tobytes()).hexdigest()})
        if rank == 0:
            whole_session = HostSession(prefix + ".fa", prefix + ".gtf")
            whole = DevicePipeline(whole_session, api=api, device=device, bam=bam, piece_bytes=4 << 20)
            expected = {"fragments": whole.n, "records": int(whole.ingest_result.records), "mapped_reads": int(whole.ingest_result.mapped_reads), "batch": digest_of_batch(whole)}
            expected_log = []
            expected_outputs = [out_path + ".expected.fusions.tsv", out_path + ".expected.discarded.tsv"]
            whole.run_workflow(expected_outputs[0], expected_outputs[1], log=lambda stage, count: expected_log.append([stage, int(count)]))
            expected.update({"log": expected_log, "filters": hashlib.md5(whole.filters().tobytes()).hexdigest()})
            for key, value in expected.items():
                if report[key] != value:
                    report["problems"].append([key, report[key], value])
            for mine, theirs in zip(outputs, expected_outputs):
                if open(mine, "rb").read() != open(theirs, "rb").read():
                    report["problems"].append(["output file differs", mine])
            report["fusions"] = sum(1 for _ in open(outputs[0])) - 1
    except Exception as error:  # the test reads the report: say what happened instead of a bare non-zero exit
        report["error"] = str(error)
    json.dump(report, open("%s.rank%d.json" % (out_path, rank), "w"))
    dist.barrier() if "error" not in report else None
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
