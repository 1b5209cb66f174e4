"""TEST-ONLY worker: samples in a queue through one resident workflow session (include/arriba_workflow.h: arriba_workflow_submit / _sample / _cancel).
usage: workflow_session_worker.py harness|product fasta gtf bam1 bam2 out_directory
harness: the workflow library built over the host stepping harness (tests/emu), as tests/bench_on_harness.py does; product: libarriba_workflow.so on the GPU.
Writes the files of every sample and checks nothing itself but the order rule; the test compares the files."""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
mode, fasta, gtf, bam1, bam2, out = sys.argv[1:7]
if mode == "harness":
    import subprocess
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu"), "libworkflow_on_harness.so"], check=True)
    os.environ["ARRIBA_WORKFLOW_LIBRARY"] = os.path.join(ROOT, "tests", "emu", "libworkflow_on_harness.so")
    from arriba_amd import _capi
    _bind = _capi.bind_device_api
    _harness = ctypes.CDLL(os.path.join(ROOT, "tests", "emu", "libemu.so"))
    _capi.bind_device_api = lambda library, prefix: _bind(_harness, "emu_")
from arriba_amd import _capi  # noqa: E402
from arriba_amd.pipeline import ArribaError, WorkflowSession  # noqa: E402

os.environ.setdefault("ARRIBA_FEED_PIECE_MB", "1")  # (many pieces through the reader and the pusher of the feed)
result = {}
path = lambda name: os.path.join(out, name)
# one sample at a time
session = WorkflowSession(fasta, gtf)
result["alone1"] = session.sample(bam1, path("alone1.tsv"), path("alone1.discarded.tsv"))
result["alone2"] = session.sample(bam2, path("alone2.tsv"), path("alone2.discarded.tsv"))
session.close()
# in a queue: the next sample is submitted before the current one is worked on
session = WorkflowSession(fasta, gtf)
session.submit(bam1)
session.submit(bam2)
try:
    session.submit(bam1)
    result["third_submit"] = "accepted"
except ArribaError as error:
    result["third_submit"] = str(error)
result["queued1"] = session.sample(bam1, path("queued1.tsv"), path("queued1.discarded.tsv"))
result["feed_overlapped"] = session.timing["feed_total"] > 0
session.submit(bam1)
result["queued2"] = session.sample(bam2, path("queued2.tsv"), path("queued2.discarded.tsv"))
session.submit(bam2)
try:  # bam1 was submitted first
    session.sample(bam2, path("wrong.tsv"))
    result["out_of_order"] = "accepted"
except ArribaError as error:
    result["out_of_order"] = str(error)
result["queued3"] = session.sample(bam1, path("queued3.tsv"), path("queued3.discarded.tsv"))
session.cancel()  # bam2, fed and never asked for
result["queued4"] = session.sample(bam2, path("queued4.tsv"), path("queued4.discarded.tsv"))  # (submits itself)
# the last file of a sample written beside the next sample: complete behind flush()
session.defer_output(True)
session.submit(bam1)
session.submit(bam2)
result["deferred1"] = session.sample(bam1, path("deferred1.tsv"), path("deferred1.discarded.tsv"))
session.submit(bam1)
result["deferred2"] = session.sample(bam2, path("deferred2.tsv"), path("deferred2.discarded.tsv"))
result["deferred3"] = session.sample(bam1, path("deferred3.tsv"), path("deferred3.discarded.tsv"))
session.flush()
session.defer_output(False)
# arriba_workflow_finish_ahead: the feeder of a sample also finishes its ingest, beside the stages of the sample in front (and with the last files deferred: three threads at work)
session.finish_ahead(True)
session.defer_output(True)
session.submit(bam1)
session.submit(bam2)
result["ahead1"] = session.sample(bam1, path("ahead1.tsv"), path("ahead1.discarded.tsv"))
result["ahead_ingest_seconds"] = session.timing["ingest"]
session.submit(bam1)
result["ahead2"] = session.sample(bam2, path("ahead2.tsv"), path("ahead2.discarded.tsv"))
result["ahead3"] = session.sample(bam1, path("ahead3.tsv"), path("ahead3.discarded.tsv"))
session.flush()
session.defer_output(False)
session.submit(bam2)
session.cancel()  # fed AND finished ahead, never asked for
result["ahead4"] = session.sample(bam1, path("ahead4.tsv"), path("ahead4.discarded.tsv"))
session.finish_ahead(False)
result["ahead5"] = session.sample(bam2, path("ahead5.tsv"), path("ahead5.discarded.tsv"))
# the device runs out of memory while two samples are in flight (harness: agpu_debug_fail_allocation_in_finish makes the next agpu_ingest_finish fail as the device library does with
# two lanes): the session throws away what was fed ahead, closes its second lane, runs the sample again alone and submits the other one again
exhaust = _harness.emu_debug_exhaust_memory_in_finish if mode == "harness" else _capi.device_library().agpu_debug_exhaust_memory_in_finish  # (the same hook of the harness and of the device library)
exhaust.argtypes, exhaust.restype = [ctypes.c_int], None
session.submit(bam2)
session.submit(bam1)
exhaust(1)
result["retried1"] = session.sample(bam2, path("retried1.tsv"), path("retried1.discarded.tsv"))
result["retried2"] = session.sample(bam1, path("retried2.tsv"), path("retried2.discarded.tsv"))  # (submitted again by the session behind the retry)
session.submit(bam2)
session.submit(bam1)
exhaust(-1)  # (... and a failure of the retry as well -- a device that is too small for the sample -- is the caller's to hear)
try:
    session.sample(bam2, path("failed.tsv"))
    result["second_failure"] = "accepted"
except ArribaError as error:
    result["second_failure"] = str(error)
exhaust(0)
result["after_failure"] = session.sample(bam1, path("after_failure.tsv"), path("after_failure.discarded.tsv"))
session.submit(bam1)  # left behind: arriba_workflow_close throws it away
session.close()
json.dump(result, open(path("result.json"), "w"))
