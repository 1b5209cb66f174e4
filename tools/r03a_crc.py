"""bgzf_crc_kernel on the GPU (ARRIBA_VERIFY_CRC=1): the intact golden file is accepted, whole and in parts, a flipped payload byte is found; then the time at 10 M"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["ARRIBA_VERIFY_CRC"] = "1"
import datasets
import test_host_and_device_logic as T
from arriba_amd import _capi
from arriba_amd.pipeline import ArribaError, DevicePipeline, HostSession
api = _capi.bind_device_api(_capi.device_library(), "agpu_")
directory = tempfile.mkdtemp(prefix="r03a_crc_")
prefix = datasets.generate(datasets.DATASETS["toy3k"], directory)
session = HostSession(prefix + ".fa", prefix + ".gtf")
expected = T._device_batch_columns(session, DevicePipeline(session, bam=prefix + ".bam"))
merged_session, merged, _ = T._ingest_in_parts(prefix, prefix + ".bam", 5, api)
assert T._device_batch_columns(merged_session, merged) == expected
raw = bytearray(open(prefix + ".bam", "rb").read())
at, blocks = 0, []
while at + 18 <= len(raw):
    size = int.from_bytes(raw[at + 16:at + 18], "little") + 1
    blocks.append((at, size)); at += size
start, size = blocks[len(blocks) // 2]
raw[start + 18 + 5 + 2000] ^= 0x11
open(prefix + ".damaged.bam", "wb").write(bytes(raw))
try:
    DevicePipeline(HostSession(prefix + ".fa", prefix + ".gtf"), bam=prefix + ".damaged.bam")
    print("CRC: the damaged block was NOT found")
except ArribaError as error:
    print("CRC: intact file accepted (whole, 5 parts), damaged block found:", error)
