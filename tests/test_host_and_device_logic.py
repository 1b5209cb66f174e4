"""CPU tier: host stages against the golden dumps of the reference, the C ABI surface, and the device code's
per-fragment logic single-stepped on the host (tests/emu) against the same dumps."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import conftest
import datasets
import parity

ROOT = conftest.ROOT


def declared_symbols(header, prefix):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(%s\w+)\s*\(" % prefix, text)))


def test_device_library_exports_every_declared_symbol(built):
    import ctypes
    lib = ctypes.CDLL(os.path.join(ROOT, "arriba_amd", "lib", "libarriba_gpu.so"))
    symbols = declared_symbols("arriba_gpu.h", "agpu_")
    assert len(symbols) >= 20
    missing = [s for s in symbols if not hasattr(lib, s)]
    assert not missing, missing
    lib.agpu_api_version.restype = ctypes.c_int
    assert lib.agpu_api_version() == 1


@pytest.mark.parametrize("header", ["arriba_gpu.h", "arriba_host.h", "arriba_workflow.h"])
def test_public_headers_are_plain_c(header, tmp_path):
    """The drop-in boundary is a C ABI: the headers must compile as C99 on their own (no C++ types, no torch types in the signatures)."""
    import subprocess
    source = tmp_path / "include_check.c"
    source.write_text('#include "%s"\nint main(void) { return 0; }\n' % header)
    result = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-I" + os.path.join(conftest.ROOT, "include"), str(source)],
                            stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True)
    assert result.returncode == 0, result.stdout


def test_host_library_exports_every_declared_symbol(built):
    import ctypes
    lib = ctypes.CDLL(os.path.join(ROOT, "arriba_amd", "lib", "libarriba_host.so"))
    symbols = declared_symbols("arriba_host.h", "ahost_")
    missing = [s for s in symbols if not hasattr(lib, s)]
    assert len(symbols) >= 15 and not missing, missing


def test_product_fails_loudly_without_gpu(built):
    """No CPU fallback: creating a device context without a GPU must raise, not silently compute elsewhere."""
    from arriba_amd import _capi
    api = _capi.bind_device_api(_capi.device_library())
    if api.device_count() > 0:
        pytest.skip("a GPU is visible")
    assert not api.create(0, None)
    assert b"no HIP device" in api.last_error() or b"fallback" in api.last_error()


@pytest.mark.parametrize("name", ["toy3k", "shuffled2k"])
def test_ingest_matches_reference_read_table(name, dataset_files):
    session = parity.open_session(dataset_files(name))
    assert parity.check_ingest(session, conftest.golden_dir(name)) > 1000


@pytest.mark.parametrize("name", ["toy3k", "shuffled2k", "mid30k", "stacked4k"])
def test_device_logic_on_host_matches_reference(name, dataset_files, emu_api):
    golden = conftest.golden_dir(name)
    session, pipeline = parity.run_read_level(parity.open_session, dataset_files(name), api=emu_api)
    parity.check_gene_table(pipeline, golden)
    parity.check_read_filters(session, pipeline, golden)
    parity.check_scalars(pipeline, golden)
    if name != "mid30k":
        parity.check_annotation(session, pipeline, golden)
    if name == "stacked4k":
        counts = np.concatenate([pipeline.gene_sets(slot)[0] for slot in range(3)])
        assert (counts > 4).sum() > 500 and counts.max() >= 9  # the memory-resident tail of the gene sets is exercised
        pipeline.find_fusions()
        assert parity.check_candidates(session, pipeline, golden) > 5000
    if name == "toy3k":
        pipeline.find_fusions()
        assert parity.check_candidates(session, pipeline, golden) > 1000
        assert parity.check_evalues(session, pipeline, golden) > 1000
        assert parity.check_mismappers(session, pipeline, golden) == 2
    if name == "mid30k":
        assert pipeline.scalars["estimated"] and pipeline.scalars["mate_gap_samples"] >= 10000


def _batch_columns(session):
    """every column of the packed batch as bytes (the pointers of the C view copied out), plus the scalars of the ingest"""
    import ctypes
    view = session.batch_view.contents
    n = int(view.n)
    columns = {"n": n, "mapped_reads": session.mapped_reads, "names": session.fragment_names()}
    for name, width in (("n_aln", 1), ("fbits", 1), ("group", 4)):
        columns[name] = ctypes.string_at(getattr(view, name), n * width)
    for slot in range(3):
        for name, width in (("contig", 2), ("start", 4), ("end", 4), ("abits", 1), ("cigar_offset", 4), ("cigar_count", 2)):
            columns["%s%d" % (name, slot)] = ctypes.string_at(getattr(view, name)[slot], n * width)
    for slot in range(2):
        for name in ("seq_offset", "seq_length"):
            columns["%s%d" % (name, slot)] = ctypes.string_at(getattr(view, name)[slot], n * 4)
    columns["cigar_pool"] = ctypes.string_at(view.cigar_pool, int(view.cigar_pool_size) * 4)
    columns["seq_pool"] = ctypes.string_at(view.seq_pool, int(view.seq_pool_size))
    return columns


def test_ingest_with_several_threads_equals_single_threaded(built, tmp_path, monkeypatch):
    """The reader deals the records to the workers by the hash of the read name (arriba_amd/csrc/host/ingest.cpp); batch, counters and the
    coverage-dependent verdicts must not depend on the number of workers.  Shuffled names and separated mates: the first mate waits parked."""
    spec = {"args": ["--seed", "17", "--fragments", "40000", "--normal-mult", "0.5", "--contigs", "5", "--contig-len", "400000", "--junctions", "400", "--dup", "0.2", "--shuffle", "--separate-mates"]}
    prefix = datasets.generate(spec, str(tmp_path))
    results = []
    for threads in ("1", "2", "5"):
        monkeypatch.setenv("ARRIBA_INGEST_THREADS", threads)
        session = parity.open_session(prefix)
        columns = _batch_columns(session)
        pairs = np.array([[c, 0] for c in range(len(session.contig_names()))], dtype=np.uint32).reshape(-1)
        top, low = session.viral_verdicts(pairs, np.zeros(4, dtype=np.uint8))  # reads the coverage windows and the per-contig viral read counts
        columns["verdicts"] = top.tobytes() + low.tobytes()
        columns["strandedness"] = session.detect_strandedness()
        columns["coverage"] = int(session._lib.ahost_coverage_checksum(session._session))
        results.append(columns)
    assert results[0]["n"] > 30000
    for other in results[1:]:
        different = [key for key in results[0] if results[0][key] != other[key]]
        assert not different, different


def _bam_payload(path):
    """the uncompressed BAM stream of a BGZF file"""
    import gzip
    return gzip.open(path, "rb").read()


def _write_bgzf(path, payload, level, block=0xff00):
    import struct
    import zlib
    with open(path, "wb") as out:
        for at in list(range(0, len(payload), block)) + [len(payload)]:
            chunk = payload[at:at + block] if at < len(payload) else b""  # the empty block at the end: the BGZF end-of-file marker
            compressor = zlib.compressobj(level, zlib.DEFLATED, -15)
            data = compressor.compress(chunk) + compressor.flush()
            out.write(struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, ord("B"), ord("C"), 2, len(data) + 25))
            out.write(data + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))


def test_ingest_reads_every_container_of_a_bam_stream(built, dataset_files, tmp_path):
    """The block-parallel BGZF reader (stored blocks as the generator and STAR --outBAMcompression 0 write them, deflated blocks, small blocks that
    straddle the batches), the zlib fallback for a plain gzip stream, and the errors: a truncated block, a corrupted checksum"""
    import gzip
    from arriba_amd.pipeline import ArribaError, HostSession
    prefix = dataset_files("toy3k")
    payload = _bam_payload(prefix + ".bam")
    variants = {"deflated": str(tmp_path / "deflated.bam"), "small_blocks": str(tmp_path / "small.bam"), "plain_gzip": str(tmp_path / "plain.bam")}
    _write_bgzf(variants["deflated"], payload, 6)
    _write_bgzf(variants["small_blocks"], payload, 1, block=997)
    with gzip.open(variants["plain_gzip"], "wb") as out:
        out.write(payload)
    def ingest(path):
        session = HostSession(prefix + ".fa", prefix + ".gtf")
        session.read_chimeric_alignments(path)
        columns = _batch_columns(session)
        columns["coverage"] = int(session._lib.ahost_coverage_checksum(session._session))
        return columns
    expected = ingest(prefix + ".bam")
    assert expected["n"] > 2000
    for name, path in variants.items():
        assert ingest(path) == expected, name
    raw = open(variants["deflated"], "rb").read()
    open(str(tmp_path / "truncated.bam"), "wb").write(raw[:len(raw) // 2])
    damaged = bytearray(raw)
    damaged[len(raw) // 3] ^= 0x5A
    open(str(tmp_path / "damaged.bam"), "wb").write(bytes(damaged))
    for name in ("truncated.bam", "damaged.bam"):
        with pytest.raises(ArribaError):
            ingest(str(tmp_path / name))


def test_ingest_reads_pipes_and_standard_input(built, dataset_files, tmp_path):
    """The reference's standard invocation is a pipe: `STAR ... | arriba -x /dev/stdin` (run_arriba.sh:42; sam_open reads any path once,
    source/read_chimeric_alignments.cpp:563).  The container must be recognised from the one open stream: a named pipe fed by the generator
    (raw BAM, what bench.py does), and /dev/stdin / "-" of a child process with raw, BGZF and plain-gzip streams."""
    import gzip
    import subprocess
    import sys
    prefix = dataset_files("toy3k")
    payload = _bam_payload(prefix + ".bam")
    session = parity.open_session(prefix)
    expected = _batch_columns(session)
    expected["coverage"] = int(session._lib.ahost_coverage_checksum(session._session))
    from arriba_amd.pipeline import HostSession
    # (a) a FIFO fed by gen_synth --raw-bam-to, exactly as bench.py streams its workload
    fifo = str(tmp_path / "records.fifo")
    os.mkfifo(fifo)
    producer = subprocess.Popen([datasets.GEN_SYNTH, "--out", str(tmp_path / "unused"), "--raw-bam-to", fifo] + datasets.DATASETS["toy3k"]["args"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    piped = HostSession(prefix + ".fa", prefix + ".gtf")
    piped.read_chimeric_alignments(fifo)
    assert producer.wait() == 0
    columns = _batch_columns(piped)
    columns["coverage"] = int(piped._lib.ahost_coverage_checksum(piped._session))
    assert columns == expected
    # (b) standard input of a child process: raw, BGZF (stored blocks), plain gzip; as /dev/stdin and as "-"
    streams = {"raw": payload, "bgzf": open(prefix + ".bam", "rb").read(), "gzip": gzip.compress(payload, 1)}
    child = ("import sys; sys.path[:0] = %r; import parity; from arriba_amd.pipeline import HostSession; "
             "s = HostSession(%r, %r); s.read_chimeric_alignments(sys.argv[1]); "
             "print(s.fragment_count, s.mapped_reads, s._lib.ahost_coverage_checksum(s._session))") % ([conftest.ROOT, os.path.join(conftest.ROOT, "tests")], prefix + ".fa", prefix + ".gtf")
    for name, data in streams.items():
        for path in ("/dev/stdin", "-"):
            result = subprocess.run([sys.executable, "-c", child, path], input=data, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
            assert result.returncode == 0, (name, path, result.stderr.decode()[-500:])
            assert result.stdout.decode().split() == [str(expected["n"]), str(expected["mapped_reads"]), str(expected["coverage"])], (name, path)
    # a stream that is cut in the middle of a record is an error, not a short batch
    result = subprocess.run([sys.executable, "-c", child, "/dev/stdin"], input=payload[:len(payload) // 2 + 7], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert result.returncode != 0 and b"failed to load alignments" in result.stderr


def _device_batch_columns(session, pipeline):
    """the batch a device ingest built (all rows fetched through agpu_gather_rows_*), keyed like _batch_columns"""
    arrays, _ = pipeline.batch_rows()
    names = arrays["names"].tobytes().decode()
    offsets = arrays["name_offset"]
    columns = {"n": arrays["n"], "mapped_reads": session.mapped_reads, "names": [names[offsets[i]:offsets[i + 1]] for i in range(arrays["n"])]}
    for key in ("n_aln", "fbits", "group", "cigar_pool", "seq_pool"):
        columns[key] = arrays[key].tobytes()
    for slot in range(3):
        for key in ("contig", "start", "end", "abits", "cigar_offset", "cigar_count"):
            columns["%s%d" % (key, slot)] = arrays["%s%d" % (key, slot)].tobytes()
    for slot in range(2):
        for key in ("seq_offset", "seq_length"):
            columns["%s%d" % (key, slot)] = arrays["%s%d" % (key, slot)].tobytes()
    columns["coverage"] = int(session._lib.ahost_coverage_checksum(session._session))
    return columns


DEVICE_INGEST_DATASETS = {
    "toy3k": datasets.DATASETS["toy3k"]["args"], "shuffled2k": datasets.DATASETS["shuffled2k"]["args"], "stacked4k": datasets.DATASETS["stacked4k"]["args"], "itd6k": datasets.DATASETS["itd6k"]["args"],
    "mid30k": datasets.DATASETS["mid30k"]["args"],
    "shuffled_dups_40k": ["--seed", "17", "--fragments", "40000", "--normal-mult", "0.5", "--contigs", "5", "--contig-len", "400000", "--junctions", "400", "--dup", "0.2", "--shuffle", "--separate-mates"],
    "stranded_multimappers_20k": ["--seed", "23", "--fragments", "20000", "--normal-mult", "1.5", "--contigs", "4", "--contig-len", "300000", "--junctions", "200", "--multimap", "0.2", "--stranded", "--itd-hotspots", "2"],
}


@pytest.mark.parametrize("name", sorted(DEVICE_INGEST_DATASETS))
def test_device_ingest_builds_the_batch_of_the_host_ingest(name, built, emu_api, tmp_path):
    """read_chimeric_alignments on the device (ingest_core.hpp stepped on the host): records cut from the stream, collated by name key, the reference's loop
    body replayed per name, sanity check, name order, pack -- every column, pool and name of the batch, coverage_t, the counters, the strandedness vote and
    the viral verdicts equal what the host ingest (itself byte-identical to the reference's read table) produces"""
    from arriba_amd.pipeline import DevicePipeline, HostSession
    prefix = datasets.generate({"args": DEVICE_INGEST_DATASETS[name]}, str(tmp_path))
    host = parity.open_session(prefix)
    expected = _batch_columns(host)
    expected["coverage"] = int(host._lib.ahost_coverage_checksum(host._session))
    session = HostSession(prefix + ".fa", prefix + ".gtf")
    pipeline = DevicePipeline(session, api=emu_api, bam=prefix + ".bam", piece_bytes=1 << 20)
    columns = _device_batch_columns(session, pipeline)
    different = [key for key in expected if expected[key] != columns[key]]
    assert not different, different
    assert pipeline.ingest_result.names_were_sorted == (0 if "--shuffle" in DEVICE_INGEST_DATASETS[name] else 1)
    assert pipeline.detect_strandedness() == host.detect_strandedness()
    pairs = np.array([[c, 0] for c in range(len(session.contig_names()))], dtype=np.uint32).reshape(-1)
    for mine, theirs in zip(session.viral_verdicts(pairs, np.zeros(4, dtype=np.uint8)), host.viral_verdicts(pairs, np.zeros(4, dtype=np.uint8))):
        assert mine.tobytes() == theirs.tobytes()
    assert expected["n"] > 1500


def test_device_ingest_reads_every_container_and_pipes(built, dataset_files, emu_api, tmp_path, monkeypatch):
    """the file side of the device ingest (BamFeed): stored BGZF handed on raw with its block table, deflated BGZF inflated by all threads, a switch from stored
    to deflated blocks in the middle of a file, plain gzip, raw BAM from a named pipe; truncated and damaged files are errors"""
    import gzip
    import subprocess
    from arriba_amd.pipeline import ArribaError, DevicePipeline, HostSession
    prefix = dataset_files("toy3k")
    payload = _bam_payload(prefix + ".bam")
    def ingest(path, piece_bytes=1 << 20):
        session = HostSession(prefix + ".fa", prefix + ".gtf")
        return _device_batch_columns(session, DevicePipeline(session, api=emu_api, bam=path, piece_bytes=piece_bytes))
    expected = ingest(prefix + ".bam")
    assert expected["n"] > 2000
    variants = {"deflated": str(tmp_path / "deflated.bam"), "small_blocks": str(tmp_path / "small.bam"), "plain_gzip": str(tmp_path / "plain.bam"), "raw": str(tmp_path / "raw.bam"), "mixed": str(tmp_path / "mixed.bam")}
    _write_bgzf(variants["deflated"], payload, 6)
    _write_bgzf(variants["small_blocks"], payload, 1, block=997)
    with gzip.open(variants["plain_gzip"], "wb") as out:
        out.write(payload)
    open(variants["raw"], "wb").write(payload)
    stored = open(prefix + ".bam", "rb").read()
    # mixed: the first stored blocks of the generator's file, then the rest of the stream deflated
    cut_blocks, at, taken = 5, 0, 0
    for _ in range(cut_blocks):
        size = int.from_bytes(stored[at + 16:at + 18], "little") + 1
        taken += int.from_bytes(stored[at + size - 4:at + size], "little")
        at += size
    _write_bgzf(str(tmp_path / "tail.bam"), payload[taken:], 6)
    open(variants["mixed"], "wb").write(stored[:at] + open(str(tmp_path / "tail.bam"), "rb").read())
    for name, path in variants.items():
        assert ingest(path) == expected, name
    monkeypatch.setenv("ARRIBA_INGEST_THREADS", "3")
    assert ingest(variants["deflated"], piece_bytes=3 << 20) == expected
    fifo = str(tmp_path / "records.fifo")
    os.mkfifo(fifo)
    producer = subprocess.Popen([datasets.GEN_SYNTH, "--out", str(tmp_path / "unused"), "--raw-bam-to", fifo] + datasets.DATASETS["toy3k"]["args"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    assert ingest(fifo) == expected
    assert producer.wait() == 0
    raw = open(variants["deflated"], "rb").read()
    open(str(tmp_path / "truncated.bam"), "wb").write(raw[:len(raw) // 2])
    damaged = bytearray(raw)
    damaged[len(raw) // 3] ^= 0x5A
    open(str(tmp_path / "damaged.bam"), "wb").write(bytes(damaged))
    open(str(tmp_path / "cut_record.bam"), "wb").write(payload[:len(payload) // 2 + 7])
    open(str(tmp_path / "cut_stored.bam"), "wb").write(stored[:len(stored) // 2])
    for name in ("truncated.bam", "damaged.bam", "cut_record.bam", "cut_stored.bam"):
        with pytest.raises(ArribaError):
            ingest(str(tmp_path / name))


def _ingest_in_parts(prefix, path, parts, api, piece_bytes=1 << 20):
    """what arriba_amd/one_sample.py does, without the processes: every part of the records of `path` ingested by a context of its own, the exported blocks
    laid out as an all-gather leaves them, merged by the context of part 0; returns (session, pipeline of the merged batch, fragments of the parts)"""
    from ctypes import byref, c_uint64
    from arriba_amd import _capi
    from arriba_amd.pipeline import DevicePipeline, HostSession

    class Part(DevicePipeline):
        def __init__(self, session, part, **kw):
            self.part = part
            super().__init__(session, **kw)

        def read_chimeric_alignments(self, bam, external_duplicate_marking=False, max_itd_length=100, piece_bytes=1 << 20):
            self.config, result, _ = self._ingest_records(bam, external_duplicate_marking, max_itd_length, piece_bytes, part=self.part, parts=parts)
            self.n = int(result.fragments)
            return self.n
    pipelines = [Part(HostSession(prefix + ".fa", prefix + ".gtf"), part, api=api, bam=path, piece_bytes=piece_bytes) for part in range(parts)]
    sizes = []
    for pipeline in pipelines:
        size = c_uint64()
        pipeline._check(api.shard_export_size(pipeline.ctx, byref(size)))
        sizes.append(size.value)
    stride = (max(sizes) + 15) & ~15
    blocks = np.zeros(stride * parts, dtype=np.uint8)
    for pipeline in pipelines:
        pipeline._check(api.shard_export(pipeline.ctx, blocks.ctypes.data + pipeline.part * stride, stride))
    merged, result = pipelines[0], _capi.IngestResult()
    merged._check(api.shard_merge(merged.ctx, blocks.ctypes.data, stride, parts, byref(result)))
    merged._adopt_ingest(merged.config, result)
    return merged.session, merged, [pipeline.n for pipeline in pipelines[1:]] + [None]


def test_device_ingest_in_parts_gives_the_batch_of_the_whole_file(built, dataset_files, emu_api, tmp_path):
    """One sample over several GPUs starts with every rank reading its part of the records (BamFeed::take_part cuts the file between read names at places found
    from the bytes alone) and one merge of the parts (agpu_shard_export / agpu_shard_merge): whatever the container and the number of parts -- more parts than
    the file has blocks included -- the merged batch, the counters and coverage_t are those of a single ingest of the whole file."""
    import gzip
    from arriba_amd.pipeline import ArribaError, DevicePipeline, HostSession
    for name in ("toy3k", "itd6k"):
        prefix = dataset_files(name)
        session = HostSession(prefix + ".fa", prefix + ".gtf")
        expected = _device_batch_columns(session, DevicePipeline(session, api=emu_api, bam=prefix + ".bam"))
        payload = _bam_payload(prefix + ".bam")
        variants = {"stored": prefix + ".bam", "raw": str(tmp_path / (name + ".raw.bam")), "deflated": str(tmp_path / (name + ".deflated.bam")), "small_blocks": str(tmp_path / (name + ".small.bam"))}
        open(variants["raw"], "wb").write(payload)
        _write_bgzf(variants["deflated"], payload, 6)
        _write_bgzf(variants["small_blocks"], payload, 1, block=997)  # (most records lie across block boundaries)
        for container, path in variants.items():
            for parts in ((1, 2, 3, 7, 64) if name == "toy3k" else (5,)):
                merged_session, merged, fragments = _ingest_in_parts(prefix, path, parts, emu_api)
                assert _device_batch_columns(merged_session, merged) == expected, (name, container, parts)
                if 1 < parts < 10:
                    assert all(n is None or n > 0 for n in fragments), (container, parts, fragments)  # every part got records
    prefix = dataset_files("toy3k")
    plain = str(tmp_path / "plain.bam")
    with gzip.open(plain, "wb") as out:
        out.write(_bam_payload(prefix + ".bam"))
    with pytest.raises(ArribaError, match="part of a sample"):  # not seekable by blocks: every rank would have to inflate the whole file
        _ingest_in_parts(prefix, plain, 2, emu_api)
    # read names in the order of a FASTQ file, not of std::string (what STAR writes): the parts interleave in name order, the merged batch is sorted
    scrambled = dataset_files("scrambled3k")
    session = HostSession(scrambled + ".fa", scrambled + ".gtf")
    whole = DevicePipeline(session, api=emu_api, bam=scrambled + ".bam")
    assert whole.ingest_result.names_were_sorted == 0
    expected = _device_batch_columns(session, whole)
    for parts in (2, 5):
        merged_session, merged, fragments = _ingest_in_parts(scrambled, scrambled + ".bam", parts, emu_api)
        assert merged.ingest_result.names_were_sorted == 0 and all(n is None or n > 0 for n in fragments)
        assert _device_batch_columns(merged_session, merged) == expected, parts
    # a file whose alignments are not grouped by read name (mates apart): a name ends up in two parts -- refused, not merged wrongly
    apart = dataset_files("shuffled2k")
    with pytest.raises(ArribaError, match="more than one place"):
        _ingest_in_parts(apart, apart + ".bam", 3, emu_api)
    # a header that declares the file sorted by coordinate: said at once, before any part is read (one GPU reads such a file as ever: it collates by name)
    import struct
    payload = _bam_payload(apart + ".bam")
    text_length = struct.unpack_from("<I", payload, 4)[0]
    line = b"@HD\tVN:1.6\tSO:coordinate\n"
    declared = payload[:4] + struct.pack("<I", text_length + len(line)) + line + payload[8:]
    by_coordinate = str(tmp_path / "by_coordinate.bam")
    _write_bgzf(by_coordinate, declared, 0)
    with pytest.raises(ArribaError, match="sorted by coordinate"):
        _ingest_in_parts(apart, by_coordinate, 2, emu_api)
    session = HostSession(apart + ".fa", apart + ".gtf")
    assert _device_batch_columns(session, DevicePipeline(session, api=emu_api, bam=by_coordinate)) == _device_batch_columns(session, DevicePipeline(session, api=emu_api, bam=apart + ".bam"))


def test_stored_blocks_are_checked_against_their_crc(built, dataset_files, emu_api, tmp_path, monkeypatch):
    """Stored BGZF blocks go to the device as they are; their payload is checked against the CRC-32 of the trailer (ARRIBA_VERIFY_CRC=0 switches the check off) (crc32_core.hpp, stepped
    here as the kernel does it: 256-byte chunks joined in a tree; the core is checked against zlib on random blocks).  One flipped base of one read -- a record that
    still parses -- is found; the intact file, whole and in parts (whose first and last blocks are delivered in part and carry no CRC), is accepted."""
    import zlib
    from arriba_amd.pipeline import ArribaError, DevicePipeline, HostSession
    monkeypatch.setenv("ARRIBA_VERIFY_CRC", "1")
    prefix = dataset_files("toy3k")
    session = HostSession(prefix + ".fa", prefix + ".gtf")
    expected = _device_batch_columns(session, DevicePipeline(session, api=emu_api, bam=prefix + ".bam"))
    merged_session, merged, _ = _ingest_in_parts(prefix, prefix + ".bam", 5, emu_api)
    assert _device_batch_columns(merged_session, merged) == expected
    raw = bytearray(open(prefix + ".bam", "rb").read())
    at, blocks = 0, []
    while at + 18 <= len(raw):
        size = int.from_bytes(raw[at + 16:at + 18], "little") + 1
        blocks.append((at, size))
        at += size
    start, size = blocks[len(blocks) // 2]
    payload_at = start + 18 + 5 + 2000  # somewhere inside the sequence / quality bytes of a record in the middle of the file
    assert zlib.crc32(bytes(raw[start + 23:start + size - 8])) == int.from_bytes(raw[start + size - 8:start + size - 4], "little")
    raw[payload_at] ^= 0x11
    damaged = str(tmp_path / "damaged.bam")
    open(damaged, "wb").write(bytes(raw))
    with pytest.raises(ArribaError, match="failed to load alignments"):
        DevicePipeline(HostSession(prefix + ".fa", prefix + ".gtf"), api=emu_api, bam=damaged)
    monkeypatch.setenv("ARRIBA_VERIFY_CRC", "0")
    DevicePipeline(HostSession(prefix + ".fa", prefix + ".gtf"), api=emu_api, bam=damaged)  # (without the check -- it is on by default, as in htslib -- the damage goes unnoticed: a quality value or a base differs)


def test_device_ingest_survives_a_false_record_start(built, dataset_files, emu_api, tmp_path):
    """The record chain is cut by segments that GUESS their first record.  A record whose aux array holds two well-formed record headers exactly at the start of
    an 8 KB segment makes that guess wrong (and flags the segment behind it, whose own guess is right): one repair pass must settle the chain -- not one pass per
    segment until the end of the stream -- and the batch must be the host ingest's."""
    import ctypes
    import struct
    from arriba_amd.pipeline import DevicePipeline, HostSession
    prefix = dataset_files("toy3k")
    payload = _bam_payload(prefix + ".bam")
    l_text = struct.unpack_from("<I", payload, 4)[0]
    at = 8 + l_text
    n_ref = struct.unpack_from("<I", payload, at)[0]
    at += 4
    for _ in range(n_ref):
        at += 4 + struct.unpack_from("<I", payload, at)[0] + 4
    base = at
    cut = base  # a record boundary ~100 kB into the stream
    while cut < base + 100000:
        cut += 4 + struct.unpack_from("<I", payload, cut)[0]
    fake = struct.pack("<IiiBBHHHiiii", 40, 0, 5, 2, 0, 0, 0, 0, 0, -1, -1, 0) + b"y\0" + b"\0" * 6   # block_size 40: a complete, plausible record of 44 bytes
    assert len(fake) == 44
    carrier_head = 4 + 32 + 2 + 8  # block_size, fixed fields, name "x\0", aux tag + 'B' + 'C' + count
    pad = (base - cut - carrier_head) % 8192
    body = b"\xAA" * pad + fake + fake + b"\xFF" * 64
    carrier = struct.pack("<iiBBHHHiiii", -1, -1, 2, 0, 0, 0, 4, 0, -1, -1, 0) + b"x\0" + b"ZZBC" + struct.pack("<I", len(body)) + body  # an unmapped record: no stage looks at it
    stream = payload[:cut] + struct.pack("<I", len(carrier)) + carrier + payload[cut:]
    assert (cut + carrier_head + pad - base) % 8192 == 0
    path = str(tmp_path / "false_start.bam")
    open(path, "wb").write(stream)
    host = HostSession(prefix + ".fa", prefix + ".gtf")
    host.read_chimeric_alignments(path)
    expected = _batch_columns(host)
    expected["coverage"] = int(host._lib.ahost_coverage_checksum(host._session))
    session = HostSession(prefix + ".fa", prefix + ".gtf")
    pipeline = DevicePipeline(session, api=emu_api, bam=path, piece_bytes=1 << 20)
    assert _device_batch_columns(session, pipeline) == expected and expected["n"] > 2000
    harness = ctypes.CDLL(os.path.join(conftest.ROOT, "tests", "emu", "libemu.so"))
    harness.emu_ingest_repair_passes.restype = ctypes.c_uint64
    assert harness.emu_ingest_repair_passes() == 1


def test_ingest_result_survives_save_and_load(built, dataset_files, tmp_path):
    from arriba_amd.pipeline import ArribaError, HostSession
    prefix = dataset_files("shuffled2k")
    session = parity.open_session(prefix)
    path = str(tmp_path / "ingest.bin")
    session.save_ingest(path)
    restored = HostSession(prefix + ".fa", prefix + ".gtf")
    assert restored.load_ingest(path) == session.fragment_count
    assert _batch_columns(session) == _batch_columns(restored)
    assert session._lib.ahost_coverage_checksum(session._session) == restored._lib.ahost_coverage_checksum(restored._session)
    assert session.contig_names() == restored.contig_names() and session.detect_strandedness() == restored.detect_strandedness()
    with open(path, "r+b") as handle:
        handle.write(b"XXXX")
    with pytest.raises(ArribaError, match="not an ingest file"):
        HostSession(prefix + ".fa", prefix + ".gtf").load_ingest(path)


def test_empty_like_inputs_are_rejected_like_the_reference(built, tmp_path):
    """The reference exits with 'no normal reads found' on a BAM without mapped reads (source/read_chimeric_alignments.cpp:759)."""
    import struct
    from arriba_amd.pipeline import ArribaError, HostSession
    prefix = datasets.generate({"args": ["--seed", "2", "--fragments", "10", "--contigs", "2", "--contig-len", "150000", "--junctions", "5", "--reference-only"]}, str(tmp_path))
    session = HostSession(prefix + ".fa", prefix + ".gtf")
    names = [b"1", b"2"]
    header = b"BAM\x01" + struct.pack("<i", 0) + struct.pack("<i", len(names))
    for name in names:
        header += struct.pack("<i", len(name) + 1) + name + b"\x00" + struct.pack("<i", 150000)
    with pytest.raises(ArribaError, match="no normal reads found"):
        session.read_chimeric_alignments(header)


def test_bench_cpu_baseline_leg_reports_the_contract_fields(built, tmp_path):
    """bench.py's cpu_baseline object: the reference binary on a bounded sample, 1 core, its loading phase told apart from its work on the sample"""
    import sys
    sys.path.insert(0, conftest.ROOT)
    import bench
    if not datasets.reference_available():
        pytest.skip("oracle/_ref/arriba_ref is not built")
    baseline = bench.cpu_baseline(1000, str(tmp_path), sample_fragments=50000)
    assert baseline["kind"] == "reference" and baseline["cores"] == 1 and baseline["unit"] == "chimeric reads/s"
    assert baseline["value"] > 1000 and "chimeric fragments of the same synthetic workload" in baseline["sample"]
    assert baseline["value_without_loading"] > baseline["value"] and 0 < baseline["loading_seconds"] < baseline["seconds"]
    same_reference, other_reads = bench.workload_args(1000, 7, 0), bench.workload_args(1000, 7, 9)
    assert same_reference[:2] == other_reads[:2] and same_reference[2:4] != other_reads[2:4]  # shards of one sample: one genome seed, different read seeds


def test_bench_host_only_leg_runs_without_a_gpu(built):
    """bench.py --host-only: the file side of the device ingest and the host ingest from a file and from a named pipe (the leg that would have caught round 1's
    open-read-close of a pipe before the driver did)"""
    import json
    import subprocess
    import sys
    result = subprocess.run([sys.executable, os.path.join(conftest.ROOT, "bench.py"), "--host-only", "--fragments", "20000"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=600)
    assert result.returncode == 0, result.stderr[-1000:]
    line = json.loads(result.stdout.strip().splitlines()[-1])
    assert line["host_ingest_from_fifo"]["fragments"] == line["host_ingest_from_file"]["fragments"] > 15000
    assert line["bam_feed"]["stream_bytes"] > 0.99 * line["bam_bytes"] - 100000 and line["bam_feed"]["n_targets"] == 27


def test_gene_set_capacity_is_reported_not_truncated(built, emu_api, tmp_path):
    """An alignment annotated with more genes than a device gene set holds (16) must end the stage with an error, never with a truncated set."""
    from arriba_amd.pipeline import ArribaError
    prefix = datasets.generate({"args": ["--seed", "13", "--fragments", "3000", "--contigs", "3", "--contig-len", "300000", "--junctions", "80", "--genes-per-mb", "40", "--gene-stack", "24"]}, str(tmp_path))
    with pytest.raises(ArribaError, match="gene set exceeded the device capacity"):
        parity.run_read_level(parity.open_session, prefix, api=emu_api)


def test_event_level_predicates_match_reference(dataset_files, emu_api):
    """filter_both_intronic, filter_short_anchor, filter_end_to_end_fusions, filter_no_coverage (event_core.hpp) against the reference's dumps"""
    session, pipeline = parity.run_read_level(parity.open_session, dataset_files("toy3k"), api=emu_api)
    discarded = parity.check_event_predicates(session, pipeline, conftest.golden_dir("toy3k"))
    assert discarded["both_intronic"] > 20 and discarded["filter_in_vitro"] > 20 and discarded["select_most_supported_breakpoints"] > 100
    session, pipeline = parity.run_read_level(parity.open_session, dataset_files("toy3k"), api=emu_api)
    counts = parity.check_event_chain(session, pipeline, conftest.golden_dir("toy3k"))  # the nine stages in one go, state injected only in front
    assert counts[0] > counts[-1] > 0


def test_internal_tandem_duplications(dataset_files, emu_api):
    """A dataset with recurrent internal tandem duplications: the read lists merge_adjacent_fusions appends to the absorbing candidates, and
    recover_internal_tandem_duplication (candidates recovered, reads un-filtered, first candidate in iteration order counts a shared read)"""
    golden = conftest.golden_dir("itd6k")
    session, pipeline = parity.run_read_level(parity.open_session, dataset_files("itd6k"), api=emu_api)
    pipeline.find_fusions()
    assert parity.check_candidates(session, pipeline, golden) > 2000
    pipeline.merge_adjacent_fusions()
    assert parity.check_read_lists(session, pipeline, golden, "merge_adjacent_fusions") > 2000  # list contents incl. the appended entries
    recovered, cleared = parity.check_recover_itd(session, pipeline, golden)
    assert recovered >= 3 and cleared > 200


@pytest.mark.parametrize("name", ["toy3k", "itd6k"])
def test_chain_to_no_coverage_without_injected_state(name, dataset_files, emu_api):
    """find_fusions ... filter_no_coverage (the reference's stages 18-35, default filters): nothing taken from the reference"""
    session, pipeline = parity.run_read_level(parity.open_session, dataset_files(name), api=emu_api)
    counts = parity.check_chain_to_no_coverage(session, pipeline, conftest.golden_dir(name))
    assert counts[0] > counts[-1] > 0


def test_homologs_and_chain_to_the_last_filter(dataset_files, emu_api):
    """filter_homologs on a sample with families of homologous genes: thousands of candidates through the elimination (the reference run with the
    filters in front switched off, state injected behind them), then the reference's stages 18-40 (find_fusions ... filter_homologs ->
    filter_mismappers -> select_most_supported_breakpoints -> recover_isoforms, default filters: every filter of the
    candidate-level workflow) as one chain with nothing taken from the reference"""
    session, pipeline = parity.run_read_level(parity.open_session, dataset_files("homologs8k_open"), api=emu_api)
    entering, discarded = parity.check_homologs(session, pipeline, conftest.golden_dir("homologs8k_open"), state_from="recover_many_spliced")
    assert entering > 3000 and discarded > 400
    for name in ("homologs8k", "toy3k"):
        session, pipeline = parity.run_read_level(parity.open_session, dataset_files(name), api=emu_api)
        counts, reads_discarded, confidence_levels = parity.check_chain_to_isoforms(session, pipeline, conftest.golden_dir(name))
        assert counts[0] > counts[-1] > 0
    assert counts[-5] == counts[-4] == 46 and counts[-1] == 48  # toy3k: no homologs, two isoforms recovered
    session, pipeline = parity.run_read_level(parity.open_session, dataset_files("toy3k"), api=emu_api)
    assert parity.check_isoforms(session, pipeline, conftest.golden_dir("toy3k")) == (46, 2)
    assert confidence_levels[1] > 0 and confidence_levels[2] > 0 and sum(confidence_levels) == pipeline.n_candidates  # assign_confidence at the end of the chain
    session, pipeline = parity.run_read_level(parity.open_session, dataset_files("toy3k"), api=emu_api)
    assert parity.check_confidence(session, pipeline, conftest.golden_dir("toy3k")) == confidence_levels
    session, pipeline = parity.run_read_level(parity.open_session, dataset_files("homologs8k"), api=emu_api)
    pipeline.find_fusions()
    from arriba_amd.pipeline import ArribaError
    with pytest.raises(ArribaError):  # needs the k-mer index
        pipeline.filter_homologs()


def test_blacklist_and_known_fusions(dataset_files, emu_api):
    """filter_blacklisted_ranges and recover_known_fusions: the parser (every kind of item, keywords, contig prefixes, malformed lines) and the
    per-candidate match through the genome bins, from injected state and inside the chain find_fusions ... assign_confidence"""
    prefix = dataset_files("rules8k")
    golden = conftest.golden_dir("rules8k")
    session, pipeline = parity.run_read_level(parity.open_session, prefix, api=emu_api)
    rules, count = pipeline.load_range_rules(prefix + ".blacklist.tsv", True)
    lines = [line for line in open(prefix + ".blacklist.tsv").read().split("\n") if line and not line.startswith("#")]
    assert count == len(lines) - 10  # the ten malformed lines the generator appends are skipped (the reference warns about exactly those)
    kinds = {(rules[k].first.type, rules[k].second.type) for k in range(count)}
    assert {pair[0] for pair in kinds} == {0, 1, 2} and len({pair[1] for pair in kinds}) >= 9  # ranges, positions, genes; keywords in the second column
    assert any(rules[k].first.strand_defined for k in range(count))
    rules, count = pipeline.load_range_rules(prefix + ".known_fusions.tsv", False)
    assert count == len([line for line in open(prefix + ".known_fusions.tsv") if not line.startswith("#")]) - 2  # "any" is not a keyword there
    recovered, blacklisted = parity.check_range_rules(session, pipeline, golden, prefix)
    assert recovered > 50 and blacklisted > 10
    session, pipeline = parity.run_read_level(parity.open_session, prefix, api=emu_api)
    counts, reads_discarded, confidence_levels = parity.check_chain_to_isoforms(session, pipeline, golden, rules_prefix=prefix)
    assert len(counts) == 19 and counts[-1] > 0
    from arriba_amd.pipeline import ArribaError
    with pytest.raises(ArribaError):
        pipeline.load_range_rules(prefix + ".no_such_file.tsv", True)


@pytest.mark.parametrize("name", ["toy3k", "rules8k"])
def test_output_files_equal_the_reference(name, dataset_files, emu_api, tmp_path):
    """After the whole chain both output files equal the reference's byte for byte: discarded.tsv (thousands of lines in the iteration order of
    fusions_t) and fusions.tsv (the reference's sort order; fusion transcript from the read pileups, best transcripts, peptide, reading frame)"""
    prefix = dataset_files(name)
    session, pipeline = parity.run_read_level(parity.open_session, prefix, api=emu_api)
    parity.check_chain_to_isoforms(session, pipeline, conftest.golden_dir(name), rules_prefix=prefix if name == "rules8k" else None)
    fusions, discarded = parity.check_output_files(session, pipeline, conftest.golden_dir(name), str(tmp_path), rules_prefix=prefix if name == "rules8k" else None)
    assert fusions > 40 and discarded > 1500


@pytest.mark.parametrize("name", ["toy3k", "rules8k", "toy3k_fill", "wgs8k"])
def test_workflow_from_input_files_to_output_files(name, dataset_files, emu_api, tmp_path):
    """FASTA + GTF + BAM (+ blacklist and known fusions) -> fusions.tsv + discarded.tsv through DevicePipeline.run_workflow with the reference's default
    parameters and nothing taken from the reference: both files byte-identical, every "(remaining=N)" of the reference's log reproduced"""
    stages = parity.check_workflow(dataset_files(name), conftest.golden_dir(name), str(tmp_path), api=emu_api, rules=name in ("rules8k", "wgs8k"), fill_sequence_gaps=name == "toy3k_fill",
                                   structural_variants=name == "wgs8k")  # toy3k_fill: -I; wgs8k: -d (structural variants from WGS: marked, filtered, recovered, confidence, output columns)
    assert len(stages) >= 18 and stages[-1][0] == "recover_isoforms" and stages[-1][1] > 40


@pytest.mark.parametrize("name", ["toy3k", "rules8k", "toy3k_fill", "wgs8k"])
def test_workflow_from_the_bam_file_through_the_device_ingest(name, dataset_files, emu_api, tmp_path):
    """the same, with read_chimeric_alignments on the device (the host feeds bytes; strandedness, the float sum of the read lengths and the rows of the
    supporting reads the writer prints come back from the device): both output files byte-identical to the reference's"""
    stages = parity.check_workflow(dataset_files(name), conftest.golden_dir(name), str(tmp_path), api=emu_api, rules=name in ("rules8k", "wgs8k"), fill_sequence_gaps=name == "toy3k_fill",
                                   structural_variants=name == "wgs8k", device_ingest=True)
    assert len(stages) >= 18 and stages[-1][0] == "recover_isoforms" and stages[-1][1] > 40


@pytest.mark.skipif(not datasets.reference_available(), reason="needs the oracle build of the reference (oracle/_ref)")
@pytest.mark.parametrize("device_ingest", [False, True])
@pytest.mark.parametrize("fragments", [1, 5, 40, 200])
def test_workflow_on_tiny_inputs_against_the_live_reference(fragments, device_ingest, emu_api, tmp_path):
    """one to a few hundred chimeric fragments: no candidate survives, or none exists at all; the output files still equal the reference's"""
    spec = {"args": ["--seed", "71", "--fragments", str(fragments), "--contigs", "3", "--contig-len", "200000", "--junctions", "10", "--normal-mult", "1.0"]}
    prefix = datasets.generate(spec, str(tmp_path))
    dump = str(tmp_path / "dump")
    os.makedirs(dump)
    with open(os.path.join(dump, "reference.log"), "w") as out:
        out.write(datasets.run_reference(prefix, dump, spec))
    os.makedirs(str(tmp_path / "mine"))
    stages = parity.check_workflow(prefix, dump, str(tmp_path / "mine"), api=emu_api, reference_prefix=prefix, device_ingest=device_ingest)
    assert stages[0][1] >= 1


@pytest.mark.skipif(not datasets.reference_available(), reason="needs the oracle build of the reference (oracle/_ref)")
def test_workflow_with_non_default_options_against_the_live_reference(emu_api, tmp_path):
    """the options of the reference reach the stages: 19 of them away from their defaults, two filters off, every optional input file given"""
    stages = parity.check_workflow_with_non_default_options(20000, str(tmp_path), api=emu_api)
    assert dict(stages)["mark_genomic_support"] > 100 and stages[-1][1] > 50


@pytest.mark.skipif(not datasets.reference_available(), reason="needs the oracle build of the reference (oracle/_ref)")
@pytest.mark.parametrize("label,options,params,workflow_options,ingest", [
    ("duplicates_marked_externally_reverse_stranded_viral", ["-u", "-s", "reverse", "-T", "2", "-C", "0.2", "-F", "150", "-X"], {"fragment_length": 150, "external_duplicate_marking": 1},
     {"strandedness": 2, "top_viral_contigs": 2, "viral_contig_min_covered_fraction": 0.2, "print_extra_info_for_discarded_fusions": True}, {"external_duplicate_marking": True}),
    ("stranded", ["-s", "yes"], {}, {"strandedness": 1}, {}),
    ("unstranded", ["-s", "no"], {}, {"strandedness": 0}, {}),
    ("device_ingest_duplicates_marked_externally_short_itd", ["-u", "-l", "40", "-X"], {"external_duplicate_marking": 1, "max_itd_length": 40}, {"print_extra_info_for_discarded_fusions": True, "max_itd_length": 40},
     {"external_duplicate_marking": True, "max_itd_length": 40, "device_ingest": True}),
    ("device_ingest_auto_strandedness", [], {}, {}, {"device_ingest": True})])
def test_workflow_with_library_options_against_the_live_reference(label, options, params, workflow_options, ingest, emu_api, tmp_path):
    """-u, -s, -T, -C, -F and -X (fusion transcripts and read identifiers for the discarded candidates, too) on a stranded library"""
    spec = {"args": ["--seed", "73", "--fragments", "12000", "--normal-mult", "0.4", "--contigs", "5", "--contig-len", "400000", "--junctions", "200", "--dup", "0.2", "--stranded"]}
    prefix = datasets.generate(spec, str(tmp_path))
    dump = str(tmp_path / "dump")
    os.makedirs(dump)
    with open(os.path.join(dump, "reference.log"), "w") as out:
        out.write(datasets.run_reference(prefix, dump, spec, extra_args=options))
    os.makedirs(str(tmp_path / "mine"))
    stages = parity.check_workflow(prefix, dump, str(tmp_path / "mine"), api=emu_api, reference_prefix=prefix, params=params, workflow_options=workflow_options, **ingest)
    assert stages[-1][1] > 100


# kinds of libraries and reads the golden datasets do not hold, compared with the reference run live (both tiers)
LIBRARY_SPECS = {
    # every mate differs from the assembly by a short insertion or deletion (CIGAR operations I and D), half of the junctions have bases between the genes that belong to neither
    "indels_and_non_template_bases": ["--seed", "303", "--fragments", "30000", "--normal-mult", "0.4", "--contigs", "5", "--contig-len", "400000", "--junctions", "200", "--dup", "0.1", "--indels", "1.0", "--non-template", "0.5"],
    # a single-end library: split read + supplementary alignment, no mates, no pairing flags
    "single_end": ["--seed", "505", "--fragments", "30000", "--normal-mult", "0.4", "--contigs", "5", "--contig-len", "400000", "--junctions", "200", "--dup", "0.1", "--single-end"],
    # reads of 150 nt, many multi-mapping reads
    "long_reads_multimappers": ["--seed", "606", "--fragments", "20000", "--normal-mult", "0.4", "--contigs", "5", "--contig-len", "400000", "--junctions", "200", "--dup", "0.1", "--read-len", "150", "--indels", "0.3",
                                "--non-template", "0.3", "--multimap", "0.2"],
    # supplementary alignments with soft clips and the whole read (STAR --chimOutType WithinBAM SoftClip), 5 % of the bases N, indels, non-template bases
    "soft_clips_and_n_bases": ["--seed", "911", "--fragments", "20000", "--normal-mult", "0.4", "--contigs", "5", "--contig-len", "400000", "--junctions", "200", "--dup", "0.1", "--n-bases", "0.05",
                               "--soft-clip-supplementary", "--non-template", "0.4", "--indels", "0.3"],
    # a fifth of the fragments multi-mapping, half of those with secondary alignments that lack the HI tag (ignored with a warning), contigs outside the interesting set
    "missing_hi_tags": ["--seed", "1616", "--fragments", "20000", "--normal-mult", "0.4", "--contigs", "26", "--contig-len", "150000", "--junctions", "200", "--dup", "0.1", "--multimap", "0.2", "--missing-hi", "0.5"],
    # short single-end reads of a stranded library
    "short_stranded_single_end": ["--seed", "707", "--fragments", "20000", "--normal-mult", "0.4", "--contigs", "5", "--contig-len", "400000", "--junctions", "200", "--dup", "0.1", "--read-len", "60", "--clip-min", "12",
                                  "--clip-max", "30", "--single-end", "--stranded", "--multimap", "0.1"],
}
INDEL_SPEC = {"args": LIBRARY_SPECS["indels_and_non_template_bases"]}


def check_library_against_the_live_reference(kind, directory, api=None):
    spec = {"args": LIBRARY_SPECS[kind]}
    prefix = datasets.generate(spec, directory)
    dump = os.path.join(directory, "dump")
    os.makedirs(dump)
    with open(os.path.join(dump, "reference.log"), "w") as out:
        out.write(datasets.run_reference(prefix, dump, spec))
    os.makedirs(os.path.join(directory, "mine"))
    stages = parity.check_workflow(prefix, dump, os.path.join(directory, "mine"), api=api, reference_prefix=prefix, device_ingest=True)
    transcripts = [line.split("\t")[27] for line in open(prefix + ".fusions.tsv") if not line.startswith("#")]
    assert stages[-1][1] > 100
    if kind == "indels_and_non_template_bases":  # inserted bases in brackets, deleted ones as dashes, uncertain ones as question marks, non-template bases between pipes
        assert sum("[" in t for t in transcripts) > 10 and sum("-" in t for t in transcripts) > 10 and sum("?" in t for t in transcripts) > 0 and sum(t.count("|") > 1 for t in transcripts) > 30
    return stages


@pytest.mark.skipif(not datasets.reference_available(), reason="needs the oracle build of the reference (oracle/_ref)")
@pytest.mark.parametrize("kind", sorted(LIBRARY_SPECS))
def test_workflow_on_other_kinds_of_libraries_against_the_live_reference(kind, emu_api, tmp_path):
    """Insertions and deletions in the reads, non-template bases at the junctions, single-end libraries, other read lengths: the device ingest, the mismatch
    filters, find_fusions on fragments of two alignments, the pileups of the fusion transcripts -- every count and both files equal the reference's"""
    check_library_against_the_live_reference(kind, str(tmp_path), emu_api)


@pytest.mark.skipif(not datasets.reference_available(), reason="needs the oracle build of the reference (oracle/_ref)")
@pytest.mark.parametrize("disabled", [["mismappers"], ["homologs", "mismappers"], []])
def test_workflow_with_mismappers_switched_off_against_the_live_reference(disabled, emu_api, tmp_path):
    """-f mismappers: the reference skips the stage (source/arriba.cpp:562), reads and candidates keep their state; clipped segments copied from the
    partner gene make the stage fire when it is on (the same dataset with the default filters discards reads as mis-mappers)"""
    spec = {"args": ["--seed", "79", "--fragments", "15000", "--normal-mult", "0.4", "--contigs", "5", "--contig-len", "400000", "--junctions", "200", "--partner-clip", "0.5", "--clip-min", "40", "--clip-max", "70",
                     "--homolog-families", "4"]}
    prefix = datasets.generate(spec, str(tmp_path))
    dump = str(tmp_path / "dump")
    os.makedirs(dump)
    with open(os.path.join(dump, "reference.log"), "w") as out:
        out.write(datasets.run_reference(prefix, dump, spec, disable_filters=disabled))
    os.makedirs(str(tmp_path / "mine"))
    stages = parity.check_workflow(prefix, dump, str(tmp_path / "mine"), api=emu_api, reference_prefix=prefix, params={"disable_filters": disabled})
    counts = dict(stages)
    if disabled:
        assert counts["filter_mismappers"] == counts["filter_homologs"] and stages[-1][1] > 20
    else:  # the stage on: hundreds of re-alignments decide, candidates go (the budgeted two-pass schedule of the device is stepped with EMU_MISMAPPER_BUDGET, too)
        assert counts["filter_mismappers"] <= counts["filter_homologs"]  # (both files and every count equal the reference's: check_workflow)


@pytest.mark.skipif(not datasets.reference_available(), reason="needs the oracle build of the reference (oracle/_ref)")
@pytest.mark.parametrize("schedule", [{"EMU_MISMAPPER_BUDGET": "64"}, {"EMU_MISMAPPER_BUDGET": "64", "EMU_MISMAPPER_ROUNDS": "64"}, {"EMU_MISMAPPER_BUDGET": "64", "EMU_MISMAPPER_WORKLIST": "0"},
                                      {"EMU_MISMAPPER_BUDGET": "64", "EMU_MISMAPPER_WORKLIST_CAPACITY": "6"},
                                      {"EMU_MISMAPPER_BUDGET": "64", "EMU_MISMAPPER_FRONT_SLOTS": "2"}, {"EMU_MISMAPPER_BUDGET": "64", "EMU_MISMAPPER_FRONT_SLOTS": "0"},
                                      {"EMU_MISMAPPER_BUDGET": "64", "EMU_MISMAPPER_STRANDS_TOGETHER": "0"}, {"EMU_MISMAPPER_BUDGET": "64", "EMU_MISMAPPER_ROUNDS": "64", "EMU_MISMAPPER_STRANDS_TOGETHER": "0"},
                                      {"EMU_MISMAPPER_BUDGET": "64", "EMU_MISMAPPER_CHARS": "0"}])
def test_every_schedule_of_the_mismapper_search_gives_the_reference(schedule, emu_api, tmp_path, monkeypatch):
    """The verdict of a read is a pure function of the read, whoever computes it in whatever order: with a step budget of 64 nearly every read goes to the second
    pass, which is stepped as the list of tasks the device uses (nested calls listed and deduplicated by the memo; taken one by one, and in rounds of 64 as a wavefront
    takes them -- the harness fails if a listed task was never run), as the recursion with the memo of failed calls, and with a list of 6 tasks that overflows so that
    the recursion takes over; with a front of the memo (the table in LDS of round 6) of 16 slots, of two -- nearly every key spills to the table behind it -- and
    without one; the sweep over both strands of a segment at once (the default since round 6) and strand by strand -- reads discarded, candidates and both files equal the reference's."""
    for key, value in schedule.items():
        monkeypatch.setenv(key, value)
    spec = {"args": ["--seed", "79", "--fragments", "15000", "--normal-mult", "0.4", "--contigs", "5", "--contig-len", "400000", "--junctions", "200", "--partner-clip", "0.5", "--clip-min", "40", "--clip-max", "70",
                     "--homolog-families", "4"]}
    prefix = datasets.generate(spec, str(tmp_path))
    dump = str(tmp_path / "dump")
    os.makedirs(dump)
    with open(os.path.join(dump, "reference.log"), "w") as out:
        out.write(datasets.run_reference(prefix, dump, spec, extra_args=["-U", "32767"]))
    os.makedirs(str(tmp_path / "mine"))
    stages = parity.check_workflow(prefix, dump, str(tmp_path / "mine"), api=emu_api, reference_prefix=prefix, params={"subsampling_threshold": 32767})
    assert dict(stages)["filter_mismappers"] <= dict(stages)["filter_homologs"]  # (every count, the read filters of the written candidates and both files equal the reference's: check_workflow)


@pytest.mark.parametrize("name", ["toy3k", "wgs8k", "mid30k in pieces of 1 MB"])
def test_cpp_workflow_driver_over_the_c_abis(name, dataset_files, emu_api, tmp_path, monkeypatch):
    """arriba_workflow_run (arriba_amd/csrc/workflow: the reference's main() behind its option parser, C++ over the two C ABIs, no Python in the loop),
    linked against the stepping harness in place of the device library: both output files equal the reference's byte for byte"""
    import gzip
    import subprocess
    directory = os.path.join(conftest.ROOT, "tests", "emu")
    subprocess.run(["make", "-s", "-C", directory, "workflow_on_harness"], check=True)
    if " in pieces" in name:  # the file goes through the reader thread and the pusher of the driver in ~20 pieces (four pinned buffers in turn) instead of one
        name = name.split()[0]
        monkeypatch.setenv("ARRIBA_FEED_PIECE_MB", "1")
    prefix = dataset_files(name)
    outputs = [str(tmp_path / "fusions.tsv"), str(tmp_path / "discarded.tsv")]
    optional = [prefix + suffix for suffix in (".blacklist.tsv", ".known_fusions.tsv", ".tags.tsv", ".protein_domains.gff3", ".sv.tsv")] if name == "wgs8k" else []
    result = subprocess.run([os.path.join(directory, "workflow_on_harness"), prefix + ".fa", prefix + ".gtf", prefix + ".bam"] + outputs + optional, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, universal_newlines=True)
    assert result.returncode == 0, result.stdout
    stages = dict(line.split("\t") for line in result.stdout.strip().split("\n"))
    assert int(stages["find_fusions"]) > 1000 and int(stages["recover_isoforms"]) > 40
    for mine, reference in zip(outputs, ("fusions.tsv.gz", "discarded.tsv.gz")):
        assert open(mine).read() == gzip.open(os.path.join(conftest.golden_dir(name), reference), "rt").read(), reference


def test_command_line_of_the_reference(dataset_files, emu_api, tmp_path):
    """arriba_gpu_workflow with the reference's flags (source/options.cpp:282: -x -g -a -o -O -b -k -t -p -d -f ...), the BAM file on standard input as in run_arriba.sh:42:
    the progress lines equal the reference's log line by line (text and counts, time stamps aside), both output files equal its files; the reference's checks of the
    arguments end with "ERROR: ..." and exit code 1"""
    import gzip
    import re
    import subprocess
    directory = os.path.join(conftest.ROOT, "tests", "emu")
    subprocess.run(["make", "-s", "-C", directory, "workflow_on_harness"], check=True)
    driver = os.path.join(directory, "workflow_on_harness")
    prefix = dataset_files("wgs8k")
    golden = conftest.golden_dir("wgs8k")
    outputs = [str(tmp_path / "fusions.tsv"), str(tmp_path / "discarded.tsv")]
    command = [driver, "-x", "/dev/stdin", "-g", prefix + ".gtf", "-a", prefix + ".fa", "-o", outputs[0], "-O", outputs[1], "-b", prefix + ".blacklist.tsv", "-k", prefix + ".known_fusions.tsv",
               "-t", prefix + ".tags.tsv", "-p", prefix + ".protein_domains.gff3", "-d", prefix + ".sv.tsv"]
    result = subprocess.run(command, stdin=open(prefix + ".bam", "rb"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True)
    assert result.returncode == 0, result.stderr[-2000:]
    for mine, reference in zip(outputs, ("fusions.tsv.gz", "discarded.tsv.gz")):
        assert open(mine).read() == gzip.open(os.path.join(golden, reference), "rt").read(), reference
    # (the golden log merges stdout and stderr: a warning lands inside the line that was being written; quoted paths differ between the runs)
    def progress_lines(text):
        text = re.sub(r"=(?:WARNING:[^\n]*\n)+", "=", text)
        text = re.sub(r" WARNING:[^\n]*", "", text)
        lines = [re.sub(r"'[^']*'", "''", re.sub(r"^\[[^\]]*\] ", "", line)).rstrip() for line in text.splitlines()]
        return [line for line in lines if "(remaining=" in line or "(total=" in line or "(marked=" in line or line.startswith("Detecting strandedness") or line.startswith("Estimating fragment length")]
    mine, theirs = progress_lines(result.stdout), progress_lines(open(os.path.join(golden, "reference.log")).read())
    assert mine == theirs and len(mine) > 40
    assert result.stdout.rstrip().splitlines()[-1].split("] ", 1)[1].startswith("Done (elapsed time=00:00:")
    # the output file may be a pipe (the writer puts the rows of a file that can seek to their places from all threads; a pipe gets them one after the other)
    fifo = str(tmp_path / "fusions.fifo")
    os.mkfifo(fifo)
    reader = subprocess.Popen(["cat", fifo], stdout=subprocess.PIPE)
    piped = subprocess.run(command[:7] + ["-o", fifo] + command[9:], stdin=open(prefix + ".bam", "rb"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True)
    text = reader.communicate(timeout=60)[0].decode()
    assert piped.returncode == 0 and text == gzip.open(os.path.join(golden, "fusions.tsv.gz"), "rt").read()
    # -f: a filter that is off prints no line; unknown names are errors
    quiet = subprocess.run(command[:11] + ["-f", "blacklist,mismappers,homologs", "-X", "-I", "-u", "-U", "100"], stdin=open(prefix + ".bam", "rb"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True)
    assert quiet.returncode == 0 and "Re-aligning chimeric reads" not in quiet.stdout and "Filtering genes with" not in quiet.stdout and "Filtering duplicates" in quiet.stdout
    for arguments, message in ((["-g", prefix + ".gtf", "-a", prefix + ".fa", "-o", outputs[0], "-f", "blacklist"], "missing mandatory option -x"),
                               (command[:9], "filter 'blacklist' enabled, but missing option -b"),
                               (command[:9] + ["-f", "no_such_filter"], "invalid argument to option -f: no_such_filter"),
                               (command[:9] + ["-f", "blacklist", "-E", "-1"], "argument to -E must be greater than 0"),
                               (command[:9] + ["-f", "blacklist", "-f", "duplicates"], "option -f specified too often"),
                               (command[:9] + ["-f", "blacklist", "-U", "40000"], "argument to -U must be an integer between 1 and 32767"),
                               (command[:9] + ["-f", "blacklist", "-q"], "unknown option: -q"),
                               (command[:9] + ["-f", "blacklist", "-b"], "option -b requires an argument"),
                               (command[:3] + ["-g", str(tmp_path / "none.gtf")], "file not found/readable"),
                               (command[:9] + ["-f", "blacklist", "-c", prefix + ".bam"], "option -c")):
        failed = subprocess.run([driver] + arguments[1:] if arguments[0] == driver else [driver] + arguments, stdin=subprocess.DEVNULL, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True)
        assert failed.returncode == 1 and failed.stderr.startswith("ERROR: ") and message in failed.stderr, (arguments, failed.stderr)


def test_cpp_workflow_library_fails_loudly_without_a_gpu(built, dataset_files, tmp_path):
    """libarriba_workflow.so exports what include/arriba_workflow.h declares; without a GPU arriba_gpu_workflow stops with the device library's error, no CPU fallback"""
    import ctypes
    import subprocess
    import torch
    lib = ctypes.CDLL(os.path.join(conftest.ROOT, "arriba_amd", "lib", "libarriba_workflow.so"))
    for symbol in ("arriba_workflow_default_options", "arriba_workflow_run", "arriba_workflow_last_error"):
        assert hasattr(lib, symbol), symbol
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    prefix = dataset_files("toy3k")
    result = subprocess.run([os.path.join(conftest.ROOT, "arriba_amd", "lib", "arriba_gpu_workflow"), prefix + ".fa", prefix + ".gtf", prefix + ".bam", str(tmp_path / "fusions.tsv")],
                            stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True)
    assert result.returncode == 1 and "ERROR" in result.stderr and not os.path.exists(str(tmp_path / "fusions.tsv")), (result.returncode, result.stderr[-500:])


@pytest.mark.skipif(not datasets.reference_available(), reason="needs the oracle build of the reference (oracle/_ref)")
@pytest.mark.parametrize("seed", [1, 2])
def test_fuzzed_rule_files_against_the_live_reference(seed, emu_api, tmp_path):
    """blacklist, known-fusions and tags files made of random items (contigs with strand, chr prefix, asterisk; positions and ranges that are empty,
    negative, reversed, hexadecimal, padded with blanks; gene names; keywords in the wrong place; missing and extra columns; DOS line ends): whatever the
    reference's parser makes of a line, the output files must not differ"""
    import gzip
    import random
    rng = random.Random(seed)
    spec = dict(datasets.DATASETS["rules8k"])
    prefix = datasets.generate(spec, str(tmp_path))
    genes = [line.split("\t")[0] for line in gzip.open(os.path.join(conftest.golden_dir("rules8k"), "fusions.tsv.gz"), "rt").read().split("\n")[1:40] if line]
    contigs = ["1", "2", "3", "4", "chr1", "chr2", "1*", "NC_*", "AC_*", "GL*", "X", "chrX", "nowhere", "+1", "-2", "+chr3", "-4*"]
    def position():
        return rng.choice(["%d" % rng.randint(1, 300000), "%d-%d" % (rng.randint(1, 150000), rng.randint(150000, 300000)), "0", "-5", "5-", "-", "abc", "1e3", "0x10", "+5", " 5", "5 ", "200000-1000", "", "1-2-3"])
    def item():
        r = rng.random()
        if r < 0.3:
            return rng.choice(genes)
        if r < 0.9:
            return rng.choice(contigs) + ":" + position()
        return rng.choice(["any", "split_read_donor", "read_through", "low_support", "filter_spliced", "not_both_spliced", "discordant_mates", "split_read_any", "split_read_acceptor", "ANY", "", ":5", "1:", "::", "1:2:3"])
    with open(prefix + ".blacklist.tsv", "w") as out:
        out.write("\n".join("\t".join(item() for _ in range(rng.choice([2, 2, 2, 2, 1, 3]))) + rng.choice(["", "", "\r", "\t"]) for _ in range(400)) + "\n")
    with open(prefix + ".known_fusions.tsv", "w") as out:
        out.write("\n".join("\t".join(item() for _ in range(rng.choice([2, 2, 3]))) for _ in range(300)) + "\n")
    with open(prefix + ".tags.tsv", "w") as out:
        out.write("\n".join("\t".join([item(), item(), rng.choice(["tag A", "x,y", "", "ok", "t\x01b"])][:rng.choice([3, 3, 2])]) for _ in range(300)) + "\n")
    dump = str(tmp_path / "dump")
    os.makedirs(dump)
    with open(os.path.join(dump, "reference.log"), "w") as out:
        out.write(datasets.run_reference(prefix, dump, spec))
    os.makedirs(str(tmp_path / "mine"))
    stages = dict(parity.check_workflow(prefix, dump, str(tmp_path / "mine"), api=emu_api, rules=True, reference_prefix=prefix))
    assert stages["filter_blacklisted_ranges"] < stages["recover_many_spliced"]  # some of the random lines hit


@pytest.mark.skipif(not datasets.reference_available(), reason="needs the oracle build of the reference (oracle/_ref)")
@pytest.mark.parametrize("seed", [1, 2])
def test_fuzzed_variant_and_domain_files_against_the_live_reference(seed, emu_api, tmp_path):
    """structural variants (four-column and VCF lines with fields swapped for nonsense, truncated, extended, DOS line ends) and protein domains (GFF3 lines
    with broken attributes, percent escapes, bad coordinates): whatever the reference's parsers make of a line, the output files must not differ"""
    import random
    rng = random.Random(seed)
    spec = dict(datasets.DATASETS["wgs8k"])
    prefix = datasets.generate(spec, str(tmp_path))
    variants = [line for line in open(prefix + ".sv.tsv").read().split("\n") if line and not line.startswith("#")]
    def mutated_variant(line):
        fields, r = line.split("\t"), rng.random()
        if r < 0.15 and len(fields) > 1:
            fields[rng.randrange(len(fields))] = rng.choice(["", ".", "PASS", "upstream", "-", "+", "x:y", "1:abc", "SVTYPE=DEL", "SVTYPE=DEL;END=", "SVTYPE=INV;END=500", "END=9;SVTYPE=DUP", "N[1:5[", "]2:77]N", "[3:9[N", "N]4:1]", "N.", ".N", "<DEL>"])
        elif r < 0.25:
            fields = fields[:rng.randrange(1, len(fields) + 1)]
        elif r < 0.3:
            fields = fields + ["extra"]
        elif r < 0.35:
            return "\t".join(fields).replace(":", " :", 1)
        return "\t".join(fields) + rng.choice(["", "", "\r"])
    with open(prefix + ".sv.tsv", "w") as out:
        out.write("#comment\n" + "\n".join(mutated_variant(rng.choice(variants)) for _ in range(600)) + "\n")
    domains = [line for line in open(prefix + ".protein_domains.gff3").read().split("\n") if line and not line.startswith("#")]
    def mutated_domain(line):
        fields, r = line.split("\t"), rng.random()
        if r < 0.2 and len(fields) > 8:
            fields[8] = rng.choice([fields[8].replace("Name=", "name="), fields[8] + ";Name=Second%2", "Name=%ZZbad%4;gene_name=SYN1;gene_id=E", "gene_id=X;gene_name=SYN2;Name=a b|c,d",
                                    "Name=Q%41%2c;gene_name=SYN3;gene_id=ENSG00000000003.12", fields[8].replace("gene_id=ENSG", "gene_id=XNSG")])
        elif r < 0.3:
            fields[rng.randrange(len(fields))] = rng.choice(["", "x", "-1", "0", "+", "?"])
        elif r < 0.35:
            fields = fields[:rng.randrange(1, len(fields))]
        return "\t".join(fields) + rng.choice(["", "\r"])
    with open(prefix + ".protein_domains.gff3", "w") as out:
        out.write("##gff-version 3\n" + "\n".join(mutated_domain(rng.choice(domains)) for _ in range(400)) + "\n")
    dump = str(tmp_path / "dump")
    os.makedirs(dump)
    with open(os.path.join(dump, "reference.log"), "w") as out:
        out.write(datasets.run_reference(prefix, dump, spec))
    os.makedirs(str(tmp_path / "mine"))
    stages = dict(parity.check_workflow(prefix, dump, str(tmp_path / "mine"), api=emu_api, rules=True, reference_prefix=prefix, structural_variants=True))
    assert stages["mark_genomic_support"] > 500


def test_chain_to_relative_support_without_injected_state(dataset_files, emu_api):
    """find_fusions -> merge_adjacent_fusions -> e-value -> candidate predicates -> filter_relative_support, nothing taken from the reference in between"""
    golden = conftest.golden_dir("toy3k_chain")
    session, pipeline = parity.run_read_level(parity.open_session, dataset_files("toy3k_chain"), api=emu_api)
    pipeline.find_fusions()
    assert parity.check_merge_adjacent(session, pipeline, golden) >= 0
    session, pipeline = parity.run_read_level(parity.open_session, dataset_files("toy3k_chain"), api=emu_api)
    assert parity.check_chain_to_relative_support(session, pipeline, golden) > 1000
    # with the default filters: filter_multimappers in the chain
    golden = conftest.golden_dir("toy3k")
    session, pipeline = parity.run_read_level(parity.open_session, dataset_files("toy3k"), api=emu_api)
    assert parity.check_multimappers(session, pipeline, golden) > 50
    session, pipeline = parity.run_read_level(parity.open_session, dataset_files("toy3k"), api=emu_api)
    assert parity.check_chain_to_relative_support(session, pipeline, golden, multimappers=True) > 1000


def check_samples_in_a_queue(mode, tmp_path):
    """Two different samples of one genome through a resident session, one at a time and in a queue (arriba_workflow_submit), with the last files deferred and with the ingest finished
    ahead by the feeder (arriba_workflow_finish_ahead): the same files; the order rule and arriba_workflow_cancel behave as include/arriba_workflow.h says."""
    import json
    arguments = ["--seed", "11", "--fragments", "3000", "--contigs", "4", "--contig-len", "300000", "--junctions", "60"]
    for k, read_seed in ((1, "0"), (2, "5")):
        subprocess.run([datasets.GEN_SYNTH, "--out", str(tmp_path / ("s%d" % k)), "--read-seed", read_seed] + arguments, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    out = tmp_path / "out"
    out.mkdir()
    result = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "workflow_session_worker.py"), mode, str(tmp_path / "s1.fa"), str(tmp_path / "s1.gtf"), str(tmp_path / "s1.bam"), str(tmp_path / "s2.bam"), str(out)],
                            stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, timeout=1200)
    assert result.returncode == 0, result.stdout[-3000:]
    report = json.load(open(str(out / "result.json")))
    assert "two samples are submitted already" in report["third_submit"] and "order they were submitted" in report["out_of_order"], report
    assert report["feed_overlapped"]
    assert report["alone1"] != report["alone2"]  # (two different samples)
    read = lambda name: open(str(out / name), "rb").read()
    assert report["ahead_ingest_seconds"] == 0  # (the sample was finished by its feeder: nothing of the ingest was left for the call that asked for it)
    for queued, alone in (("queued1", "alone1"), ("queued2", "alone2"), ("queued3", "alone1"), ("queued4", "alone2"), ("deferred1", "alone1"), ("deferred2", "alone2"), ("deferred3", "alone1"),
                          ("ahead1", "alone1"), ("ahead2", "alone2"), ("ahead3", "alone1"), ("ahead4", "alone1"), ("ahead5", "alone2")):
        assert report[queued] == report[alone], queued
        assert read(queued + ".tsv") == read(alone + ".tsv") and read(queued + ".discarded.tsv") == read(alone + ".discarded.tsv"), queued
    assert len(read("alone1.tsv").splitlines()) > 5
    if True:  # the device out of memory with two lanes: run again alone, once (workflow.cpp: arriba_workflow_sample; keyed on AGPU_ERR_NO_MEMORY) -- on the harness and, since round 6, on the GPU
        assert report["retried1"] == report["alone2"] and report["retried2"] == report["alone1"] and report["after_failure"] == report["alone1"]
        assert "hipMalloc failed" in report["second_failure"], report["second_failure"]
        for retried, alone in (("retried1", "alone2"), ("retried2", "alone1"), ("after_failure", "alone1")):
            assert read(retried + ".tsv") == read(alone + ".tsv") and read(retried + ".discarded.tsv") == read(alone + ".discarded.tsv"), retried


def test_samples_in_a_queue_through_one_session(built, emu_api, tmp_path):
    check_samples_in_a_queue("harness", tmp_path)


def test_crc32_by_chunks_against_zlib(built):
    """the CRC-32 of a BGZF block as bgzf_crc_kernel computes it (crc32_core.hpp: the payload as the end of a virtual 64 KB block, raw CRCs of 64 chunks, two table-driven
    operators) against zlib's crc32: every length up to 300, lengths around the chunk and word boundaries up to 64 KB, every alignment, random / zero / 0xFF bytes"""
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu"), "crc_check"], check=True)
    result = subprocess.run([os.path.join(ROOT, "tests", "emu", "crc_check")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, timeout=600)
    assert result.returncode == 0 and " 0 failures" in result.stdout, result.stdout[-2000:]


def test_inflate_core_against_zlib(built):
    """the DEFLATE decoder of bgzf_inflate_kernel (inflate_core.hpp), stepped with one lane: 1120 blocks of seven kinds of data x sizes x levels x strategies equal zlib's bytes"""
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu"), "inflate_check"], check=True)
    result = subprocess.run([os.path.join(ROOT, "tests", "emu", "inflate_check")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, timeout=600)
    assert result.returncode == 0 and "0 failures" in result.stdout, result.stdout[-2000:]


def test_word_wise_name_helpers_against_their_definitions(built):
    """find_byte, first_difference, qname_length, same_name, compare_names and the three-word load_record of ingest_core.hpp (eight bytes of the stream per load instead of one)
    give what the byte-by-byte definitions give, on every length and alignment; compare_names against std::string::compare of "QNAME,HI[ITD]"; is_tandem_duplication with its
    short cut over the first sixteen bases against the loop as written (60 000 clipped reads, half of them hits); add_fragment_to_coverage as ranges in the difference array +
    prefix sums against one increment per window (120 000 fragments: pairs, deletions, introns, both ends of a contig, contigs without windows); the hit index kept beside a
    record against the walk over its aux fields (every integer type of HI, values beyond 32 bits, no HI)"""
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu"), "words_check"], check=True)
    result = subprocess.run([os.path.join(ROOT, "tests", "emu", "words_check")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, timeout=600)
    assert result.returncode == 0 and "words_check: ok" in result.stdout, result.stdout[-2000:]


@pytest.mark.parametrize("device_ingest", [False, True])
def test_last_file_written_from_a_detached_sample(device_ingest, dataset_files, emu_api, tmp_path):
    """ahost_detach_sample / ahost_write_fusions_of / ahost_release_sample (the deferred writer of a session with two lanes): the file written from the detached sample is the file
    ahost_write_fusions writes; the session holds no sample afterwards, ingests the next one as if nothing had happened and writes the same file again"""
    from arriba_amd.pipeline import ArribaError, DevicePipeline, HostSession
    prefix = dataset_files("toy3k")
    files = {"fa": prefix + ".fa", "gtf": prefix + ".gtf", "bam": prefix + ".bam"}
    def run(directory, detached):
        os.makedirs(directory)
        session = HostSession(files["fa"], files["gtf"])
        if device_ingest:
            pipeline = DevicePipeline(session, api=emu_api, bam=files["bam"], piece_bytes=1 << 20)
        else:
            session.read_chimeric_alignments(files["bam"])
            pipeline = DevicePipeline(session, api=emu_api)
        pipeline.run_workflow(os.path.join(directory, "fusions.tsv"), None)
        pipeline.write_fusions(os.path.join(directory, "discarded.tsv"), discarded=True, print_extra_info=True, detached=detached)
        return session, pipeline
    run(str(tmp_path / "attached"), False)
    session, pipeline = run(str(tmp_path / "detached"), True)
    for name in ("fusions.tsv", "discarded.tsv"):
        assert open(str(tmp_path / "attached" / name), "rb").read() == open(str(tmp_path / "detached" / name), "rb").read(), name
    assert len(open(str(tmp_path / "detached" / "discarded.tsv")).readlines()) > 50
    with pytest.raises(ArribaError):  # nothing left to write from
        pipeline.write_fusions(str(tmp_path / "again.tsv"), discarded=True, print_extra_info=True)
    pipeline.close()
    if device_ingest:  # the session goes on with the next sample
        again = DevicePipeline(session, api=emu_api, bam=files["bam"], piece_bytes=1 << 20)
        again.run_workflow(str(tmp_path / "again_fusions.tsv"), str(tmp_path / "again_discarded.tsv"), print_extra_info_for_discarded_fusions=True)
        assert open(str(tmp_path / "again_discarded.tsv"), "rb").read() == open(str(tmp_path / "attached" / "discarded.tsv"), "rb").read()


def test_long_read_names_through_the_whole_workflow(built, emu_api, tmp_path):
    """read names of 45 characters (an Illumina run's; the golden datasets have 11): 64-bit name offsets in the batch -- the device ingest (stepped) builds the batch of the host
    ingest, and the whole workflow gives the reference's files"""
    arguments = ["--seed", "11", "--fragments", "3000", "--contigs", "4", "--contig-len", "300000", "--junctions", "60", "--name-length", "45"]
    prefix = str(tmp_path / "long")
    subprocess.run([datasets.GEN_SYNTH, "--out", prefix] + arguments, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    from arriba_amd.pipeline import DevicePipeline, HostSession
    host = HostSession(prefix + ".fa", prefix + ".gtf")
    host.read_chimeric_alignments(prefix + ".bam")
    expected = _batch_columns(host)
    session = HostSession(prefix + ".fa", prefix + ".gtf")
    columns = _device_batch_columns(session, DevicePipeline(session, api=emu_api, bam=prefix + ".bam", piece_bytes=1 << 20))
    assert [key for key in expected if expected[key] != columns[key]] == []
    assert all(len(name) >= 45 for name in columns["names"][:10])
    if os.path.exists(datasets.ARRIBA_REF):
        reference = str(tmp_path / "reference"); os.makedirs(reference)
        result = subprocess.run([datasets.ARRIBA_REF, "-x", prefix + ".bam", "-g", prefix + ".gtf", "-a", prefix + ".fa", "-o", os.path.join(reference, "fusions.tsv"), "-O", os.path.join(reference, "discarded.tsv"), "-f", "blacklist"],
                                stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True)
        assert result.returncode == 0, result.stdout[-2000:]
        mine = str(tmp_path / "mine"); os.makedirs(mine)
        other = HostSession(prefix + ".fa", prefix + ".gtf")
        pipeline = DevicePipeline(other, api=emu_api, bam=prefix + ".bam", piece_bytes=1 << 20)
        pipeline.run_workflow(os.path.join(mine, "fusions.tsv"), os.path.join(mine, "discarded.tsv"))
        for name in ("fusions.tsv", "discarded.tsv"):
            assert open(os.path.join(mine, name), "rb").read() == open(os.path.join(reference, name), "rb").read(), name


def test_no_launch_of_a_wavefront_per_item_outside_a_chunk():
    """A static audit of the device sources (review of round 5, item 8): a launch takes fewer than 2^32 work-items, so a grid of one wavefront (or one workgroup of 64 lanes) per
    data-dependent item must go through for_each_wave_chunk (device_utils.hpp), which launches < 2^26 items at a time -- or loop over its items inside a bounded grid (std::min).
    Every `<<<` whose grid multiplies an item count by 64, or takes an item count as its number of 64-lane workgroups, is looked at."""
    import glob
    import re
    offenders = []
    for path in sorted(glob.glob(os.path.join(ROOT, "arriba_amd", "csrc", "device", "*.hip"))):
        for number, line in enumerate(open(path), 1):
            for launch in re.finditer(r"<<<(.*?)>>>", line):
                arguments = launch.group(1)
                grid = arguments.split(",")[0]
                per_item_wave = re.search(r"\*\s*64", grid) is not None
                block_of_64 = re.match(r"\s*[^,]+,\s*64\s*,", arguments) is not None and not re.match(r"\s*(1|workgroups|groups)\s*$", grid)
                if (per_item_wave or block_of_64) and "for_each_wave_chunk" not in line and "std::min" not in grid:
                    offenders.append("%s:%d: %s" % (os.path.basename(path), number, line[max(0, launch.start() - 40):launch.end()].strip()[:160]))
    # the kernels of the container take the blocks of ONE pushed piece (at most 256 MB / 4 KB of them)
    offenders = [entry for entry in offenders if "bgzf_inflate" not in entry]
    assert offenders == [], offenders


def test_exonic_lengths_with_more_exons_than_a_contig_id_holds(built, tmp_path):
    """The bug the hg38-size GPU test of round 6 found: the exon index has one slot per FEATURE (the reference sizes it so, source/annotation.t.hpp:26), compute_exonic_length
    (source/arriba.cpp:166-184) walks all slots, and the accessor of a slot took a 16-bit contig id -- with more than 65 535 exons the walk wrapped and every real contig was
    counted once per 65 536 slots (nine times for GENCODE's 550 k exons): every exonic length, hence every e-value, was wrong, and no sample on the 5.8 k-gene genome of the other
    tests could show it.  Here: 2 x 40 Mb with 400 genes per Mb, > 65 536 exons, against the gene table the live reference dumps."""
    import golden_io
    from arriba_amd.pipeline import HostSession
    if not os.path.exists(datasets.ARRIBA_REF_DUMP):
        pytest.skip("oracle/_ref/arriba_ref_dump is not built")
    prefix = str(tmp_path / "many")
    subprocess.run([datasets.GEN_SYNTH, "--out", prefix, "--seed", "3", "--fragments", "2000", "--contigs", "2", "--contig-len", "40000000", "--genes-per-mb", "400", "--junctions", "50"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    dump = str(tmp_path / "dump")
    os.makedirs(dump)
    switches = {"ARRIBA_ORACLE_DUMP_LISTS": "0", "ARRIBA_ORACLE_DUMP_READS": "0", "ARRIBA_ORACLE_DUMP_STAGES": "key"}
    os.environ.update(switches)
    try:
        datasets.run_reference(prefix, dump)
    finally:
        for key in switches:
            del os.environ[key]
    genes = golden_io.read_genes(os.path.join(dump, "genes.tsv"))
    session = HostSession(prefix + ".fa", prefix + ".gtf")
    view = session._lib.ahost_annotation_view(session._session).contents
    assert view.n_exons > 65536 and view.n_genes > 5000, (view.n_exons, view.n_genes)
    different = [(g["id"], view.gene_exonic_length[g["id"]], g["exonic_length"]) for g in genes[:view.n_genes] if view.gene_exonic_length[g["id"]] != g["exonic_length"]]
    assert not different, (len(different), different[:5])
