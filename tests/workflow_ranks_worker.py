"""TEST-ONLY worker: one sample over the ranks of a job THROUGH THE C++ DRIVER (include/arriba_workflow.h: arriba_workflow_set_communicator) -- started by torch.distributed.run,
one process per rank, gloo.  usage: workflow_ranks_worker.py harness|product fasta gtf out_directory bam [bam ...]
harness: the workflow library built over the host stepping harness (tests/emu); product: libarriba_workflow.so on the GPU (several ranks on one device: the collectives are
torch.distributed's over host memory either way).  Every rank calls arriba_workflow_sample for every file, in a queue (the next file is submitted before the current one is
worked on); rank 0 writes the files, every rank its report.  The test compares the files with those of one rank."""
import ctypes
import json
import os
import sys

import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
mode, fasta, gtf, out = sys.argv[1:5]
bams = sys.argv[5:]
if mode == "harness":
    os.environ["ARRIBA_WORKFLOW_LIBRARY"] = os.path.join(ROOT, "tests", "emu", "libworkflow_on_harness.so")
    from arriba_amd import _capi
    _bind = _capi.bind_device_api
    _harness = ctypes.CDLL(os.path.join(ROOT, "tests", "emu", "libemu.so"))
    _capi.bind_device_api = lambda library, prefix: _bind(_harness, "emu_")
from arriba_amd.pipeline import ArribaError, WorkflowSession  # noqa: E402

os.environ.setdefault("ARRIBA_FEED_PIECE_MB", "1")
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
result = {"rank": rank, "world": world, "samples": []}
try:
    session = WorkflowSession(fasta, gtf)
    if not os.environ.get("WORKFLOW_RANKS_PLAIN"):  # (the files to compare with: one rank, no communicator)
        session.over_ranks()
    session.defer_output(True)   # (ignored over ranks: the writer's thread would issue collectives in an order of its own)
    broken = os.environ.get("WORKFLOW_RANKS_BREAK_RANK") is not None
    if not broken:
        session.submit(bams[0])
    for k, bam in enumerate(bams):
        if k + 1 < len(bams) and not broken:
            session.submit(bams[k + 1])  # this rank's part of the next file is fed beside the stages of this one
        name = os.path.join(out, "sample%d" % k)
        if os.environ.get("WORKFLOW_RANKS_BREAK_RANK") == str(rank):
            bam = bam + ".is_not_there"  # (the test of a failure on one rank)
        try:
            report = session.sample(bam, name + ".tsv", name + ".discarded.tsv")
            result["samples"].append({"report": report, "exchange_parts": session.timing["exchange_parts"], "fragments": dict(report).get("read_chimeric_alignments"),
                                      "shard_fragments": int(session.timing["shard_fragments"]), "exchanged_bytes": int(session.timing["exchanged_bytes"])})
        except ArribaError as error:
            result["samples"].append({"error": str(error)})
    session.close()
except Exception as error:  # noqa: BLE001 (the test reads the report)
    result["error"] = repr(error)
json.dump(result, open(os.path.join(out, "rank%d.json" % rank), "w"))
dist.barrier()
dist.destroy_process_group()
