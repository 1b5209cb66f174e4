#!/usr/bin/env python3
"""tools/sanitize_round2.py -- what round 2 added, under AddressSanitizer + UBSan (test tooling; run through tools/sanitize_round2.sh): the file side of the
device ingest reading parts of a file (BamFeed::take_part), agpu_shard_export / agpu_shard_merge stepped on the host (with the sort of the merged batch and the
check of the read names), the device ingest and the task list of the mismapper search in the harness, the output writer's pileups on reads with insertions,
deletions and non-template bases -- each compared with what it must equal."""
import ctypes, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import datasets, parity
import test_host_and_device_logic as T
from arriba_amd import _capi
from arriba_amd.pipeline import DevicePipeline, HostSession
api = _capi.bind_device_api(ctypes.CDLL(os.environ["ARRIBA_EMU_LIBRARY"]), "emu_")
tmp = tempfile.mkdtemp(prefix="asan2_")
for name, extra in (("toy3k", []), ("scrambled3k", [])):
    os.makedirs(os.path.join(tmp, name)); prefix = datasets.generate(datasets.DATASETS[name], os.path.join(tmp, name))
    session = HostSession(prefix + ".fa", prefix + ".gtf")
    expected = T._device_batch_columns(session, DevicePipeline(session, api=api, bam=prefix + ".bam"))
    payload = T._bam_payload(prefix + ".bam")
    small = prefix + ".small.bam"; T._write_bgzf(small, payload, 1, block=997)
    raw = prefix + ".raw.bam"; open(raw, "wb").write(payload)
    for path in (prefix + ".bam", small, raw):
        for parts in (2, 5, 33):
            s2, merged, _ = T._ingest_in_parts(prefix, path, parts, api)
            assert T._device_batch_columns(s2, merged) == expected
    print(name, "in parts: equal to the whole file")
os.environ["EMU_MISMAPPER_BUDGET"] = "64"
for kind in ("indels_and_non_template_bases", "soft_clips_and_n_bases", "single_end"):
    os.makedirs(os.path.join(tmp, kind))
    stages = T.check_library_against_the_live_reference(kind, os.path.join(tmp, kind), api)
    print(kind, "equal to the live reference", stages[-1])
