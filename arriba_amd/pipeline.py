"""Host-side mirror of the reference's stage order for the hot path (reference: source/arriba.cpp:119-413).

``HostSession`` reads FASTA/GTF/BAM through libarriba_host.so and owns the structure-of-arrays views.
``DevicePipeline`` pushes them through the C ABI of libarriba_gpu.so (include/arriba_gpu.h) and calls the
device stages where the reference calls the corresponding stage function.  Method names follow the reference.
"""
import ctypes
from ctypes import POINTER, byref, c_float, c_int32, c_uint32, c_uint64

import numpy as np

from . import _capi


class ArribaError(RuntimeError):
    pass


class HostSession(object):
    """load_assembly + read_annotation_gtf + read_chimeric_alignments (reference: source/arriba.cpp:91-130)."""

    def __init__(self, fasta, gtf, interesting_contigs=None, viral_contigs=None, gtf_features=None):
        self._lib = _capi.host_library()
        enc = lambda s: None if s is None else s.encode()
        self._session = self._lib.ahost_open(enc(fasta), enc(gtf), enc(interesting_contigs), enc(viral_contigs), enc(gtf_features))
        if not self._session:
            raise ArribaError("ERROR: " + self._lib.ahost_last_error().decode())
        self._keepalive = None

    def close(self):
        if self._session:
            self._lib.ahost_close(self._session)
            self._session = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def read_chimeric_alignments(self, bam, external_duplicate_marking=False, max_itd_length=100):
        """bam: path to a BAM file, or a bytes-like object / numpy uint8 array holding the raw (inflated) BAM stream."""
        if isinstance(bam, str):
            status = self._lib.ahost_ingest_bam_file(self._session, bam.encode(), int(external_duplicate_marking), max_itd_length)
        else:
            array = np.frombuffer(bam, dtype=np.uint8) if not isinstance(bam, np.ndarray) else bam
            self._keepalive = array
            status = self._lib.ahost_ingest_bam_memory(self._session, array.ctypes.data, array.size, int(external_duplicate_marking), max_itd_length)
        if status != 0:
            raise ArribaError("ERROR: " + self._lib.ahost_last_error().decode())
        return self.fragment_count

    def load_tags(self, path):
        """-t: tags for the column `tags` of the output files (reference: load_tags, source/annotate_tags.cpp:11-44)"""
        if self._lib.ahost_load_tags(self._session, path.encode()) != 0:
            raise ArribaError("ERROR: " + self._lib.ahost_last_error().decode())

    def load_protein_domains(self, path):
        """-p: protein domains (GFF3) for the column `retained_protein_domains` (reference: load_protein_domains, source/annotate_protein_domains.cpp:33-121)"""
        if self._lib.ahost_load_protein_domains(self._session, path.encode()) != 0:
            raise ArribaError("ERROR: " + self._lib.ahost_last_error().decode())

    def save_ingest(self, path):
        """the ingest result (batch, counters, coverage) as a file; load_ingest() of a session on the same FASTA/GTF restores it without parsing"""
        if self._lib.ahost_save_ingest(self._session, path.encode()) != 0:
            raise ArribaError("ERROR: " + self._lib.ahost_last_error().decode())

    def load_ingest(self, path):
        if self._lib.ahost_load_ingest(self._session, path.encode()) != 0:
            raise ArribaError("ERROR: " + self._lib.ahost_last_error().decode())
        return self.fragment_count

    @property
    def fragment_count(self):
        return int(self._lib.ahost_fragment_count(self._session))

    @property
    def mapped_reads(self):
        return int(self._lib.ahost_mapped_reads(self._session))

    @property
    def annotation_view(self):
        return self._lib.ahost_annotation_view(self._session)

    @property
    def genome_view(self):
        return self._lib.ahost_genome_view(self._session)

    @property
    def batch_view(self):
        view = self._lib.ahost_batch_view(self._session)
        if not view:
            raise ArribaError("no BAM ingested yet")
        return view

    def contig_names(self):
        return [self._lib.ahost_contig_name(self._session, c).decode() for c in range(self._lib.ahost_contig_count(self._session))]

    def fragment_names(self):
        names = []
        length = c_uint32()
        for i in range(self.fragment_count):
            pointer = self._lib.ahost_fragment_name(self._session, i, byref(length))
            names.append(ctypes.string_at(pointer, length.value).decode())
        return names

    def detect_strandedness(self):
        return int(self._lib.ahost_detect_strandedness(self._session))

    def viral_verdicts(self, pairs, gene_bits, top_count=5, min_covered_fraction=0.05):
        n_contigs = self._lib.ahost_contig_count(self._session)
        top = np.zeros(n_contigs, dtype=np.uint8)
        low = np.zeros(n_contigs, dtype=np.uint8)
        pairs = np.ascontiguousarray(pairs, dtype=np.uint32)
        gene_bits = np.ascontiguousarray(gene_bits, dtype=np.uint8)
        self._lib.ahost_viral_verdicts(self._session, pairs.ctypes.data, pairs.size // 2, gene_bits.ctypes.data, gene_bits.size, top_count, min_covered_fraction, top.ctypes.data, low.ctypes.data)
        return top, low

    def estimate_fragment_length(self, mate_gaps, fragments_visited, default_fragment_length=200):
        mate_gaps = np.ascontiguousarray(mate_gaps, dtype=np.int32)
        mean, stddev, read_length = c_float(), c_float(), c_float()
        max_mate_gap = c_int32()
        estimated = self._lib.ahost_estimate_fragment_length(self._session, mate_gaps.ctypes.data, mate_gaps.size, fragments_visited, default_fragment_length,
                                                             byref(mean), byref(stddev), byref(read_length), byref(max_mate_gap))
        return {"estimated": bool(estimated), "mate_gap_mean": mean.value, "mate_gap_stddev": stddev.value, "read_length_mean": read_length.value, "max_mate_gap": max_mate_gap.value}


class WorkflowSession(object):
    """The product path as a resident service: arriba_workflow_open once (assembly, annotation, device context), arriba_workflow_sample per BAM file -- the reference's
    main() in C++ over the two C ABIs (arriba_amd/csrc/workflow/workflow.cpp; reference: source/arriba.cpp:84-615), no Python between the stages.  bench.py times
    `sample`; `device` is a DevicePipeline view of the session's device context for the kernel profile and the post-conditions."""

    def __init__(self, fasta, gtf, params=None, device=0, api=None, **options):
        self._lib = _capi.workflow_library()
        self.options = _capi.WorkflowOptions()
        self._lib.arriba_workflow_default_options(byref(self.options))
        self.options.assembly_file, self.options.gene_annotation_file = fasta.encode(), gtf.encode()
        self.options.device_index = device
        for key, value in (params or {}).items():
            if key == "disable_filters":
                for name in value:
                    self.options.device.filter_enabled[_capi.FILTER_NAMES.index(name)] = 0
            else:
                setattr(self.options.device, key, value)
        for key, value in options.items():
            setattr(self.options, key, value.encode() if isinstance(value, str) else value)
        self._session = self._lib.arriba_workflow_open(byref(self.options))
        if not self._session:
            raise ArribaError(self._lib.arriba_workflow_last_error().decode())
        self.api = api if api is not None else _capi.bind_device_api(_capi.device_library(), "agpu_")
        self.ctx = self._lib.arriba_workflow_device(self._session)
        self.timing, self.report, self.n, self.n_candidates, self.records = {}, [], 0, 0, -1
        self.timings, self._profiling_on, self.ingest_result = {}, False, None  # (what bench.py reads from a DevicePipeline)
        self._profiled = set()

    def submit(self, bam):
        """the sample that comes after the one `sample` is called for next: its file is fed (PCIe, the front of read_chimeric_alignments) while the stages of that one run"""
        if self._lib.arriba_workflow_submit(self._session, bam.encode()) != 0:
            raise ArribaError(self._lib.arriba_workflow_last_error().decode())
        if self._profiling_on:
            self.set_profiling(True, only_new_lanes=True)

    def finish_ahead(self, on=True):
        """the ingest of a sample that was submitted ahead is finished by its feeder thread, beside the stages of the sample in front (the lanes keep their batch buffers: ~25 GB more at 10^8 fragments)"""
        if self._lib.arriba_workflow_finish_ahead(self._session, int(on)) != 0:
            raise ArribaError(self._lib.arriba_workflow_last_error().decode())

    def over_ranks(self, group=None):
        """One sample over the ranks of `group` (arriba_workflow_set_communicator, include/arriba_workflow.h): `sample` becomes a collective call -- every rank feeds its part of
        the file, one all-gather of the parts, the stages on every rank, filter_mismappers and the rows of the output files shared out, rank 0 writes.  The three collectives
        over host memory the C++ driver asks for are torch.distributed's (gloo in the CPU tests; with the nccl backend the host bytes go through tensors on this rank's GPU)."""
        import numpy as np
        import torch
        import torch.distributed as dist
        rank, size = dist.get_rank(group), dist.get_world_size(group)
        device = torch.device("cuda", self.options.device_index) if dist.get_backend(group) == "nccl" else torch.device("cpu")

        def host_array(pointer, count, dtype):
            return np.ctypeslib.as_array(ctypes.cast(pointer, ctypes.POINTER(np.ctypeslib.as_ctypes_type(dtype))), shape=(count,))

        def guarded(work):  # an exception must not cross the C frames above a callback: the driver is told with a status and ends the call on every rank
            def call(*arguments):
                try:
                    work(*arguments)
                    return 0
                except Exception as problem:
                    self.exchange_error = problem
                    return 1
            return call

        def all_gather(state, mine, everybody, n_bytes):
            if n_bytes == 0:
                return
            sent = torch.from_numpy(host_array(mine, n_bytes, np.uint8)).to(device)
            parts = [torch.empty(n_bytes, dtype=torch.uint8, device=device) for _ in range(size)]
            dist.all_gather(parts, sent, group=group)
            received = host_array(everybody, size * n_bytes, np.uint8)
            for r in range(size):
                received[r * n_bytes:(r + 1) * n_bytes] = parts[r].cpu().numpy()

        def all_reduce_int64(state, values, count, operation):
            array = host_array(values, count, np.int64)
            tensor = torch.from_numpy(array.copy()).to(device)
            dist.all_reduce(tensor, op={_capi.WORKFLOW_MAX: dist.ReduceOp.MAX, _capi.WORKFLOW_MIN: dist.ReduceOp.MIN, _capi.WORKFLOW_SUM: dist.ReduceOp.SUM}[operation], group=group)
            array[:] = tensor.cpu().numpy()

        def all_reduce_max_bytes(state, values, count):
            array = host_array(values, count, np.uint8)
            tensor = torch.from_numpy(array.copy()).to(device)
            dist.all_reduce(tensor, op=dist.ReduceOp.MAX, group=group)
            array[:] = tensor.cpu().numpy()

        self.exchange_error = None
        self._communicator = _capi.WorkflowCommunicator(rank, size, None, _capi.ALL_GATHER(guarded(all_gather)), _capi.ALL_REDUCE_INT64(guarded(all_reduce_int64)),
                                                        _capi.ALL_REDUCE_MAX_BYTES(guarded(all_reduce_max_bytes)), None)  # (kept: the session calls into it)
        if self._lib.arriba_workflow_set_communicator(self._session, byref(self._communicator)) != 0:
            raise ArribaError(self._lib.arriba_workflow_last_error().decode())
        self.rank, self.world = rank, size

    def join_rccl(self, unique_id, rank, size):
        """One sample over `size` GPUs with RCCL alone (arriba_workflow_join_rccl): `unique_id` = rccl_unique_id() of rank 0, brought to every rank by whatever started them; the
        parts of the batch and the verdicts of filter_mismappers travel in device memory, sizes, status words and row texts are bounced through the device."""
        buffer = (ctypes.c_ubyte * _capi.RCCL_ID_BYTES).from_buffer_copy(bytes(unique_id))
        if self._lib.arriba_workflow_join_rccl(self._session, buffer, rank, size) != 0:
            raise ArribaError(self._lib.arriba_workflow_last_error().decode())
        self.rank, self.world = rank, size

    def rccl_unique_id(self):
        buffer = (ctypes.c_ubyte * _capi.RCCL_ID_BYTES)()
        if self._lib.arriba_workflow_rccl_unique_id(buffer) != 0:
            raise ArribaError(self._lib.arriba_workflow_last_error().decode())
        return bytes(buffer)

    def defer_output(self, on=True):
        """the last output file of a sample is written by a thread of the session while the next sample is worked on (complete behind flush())"""
        self._lib.arriba_workflow_defer_output(self._session, int(on))

    def flush(self):
        """waits for the deferred writers; returns the seconds the last one took"""
        seconds = ctypes.c_double()
        if self._lib.arriba_workflow_flush(self._session, byref(seconds)) != 0:
            raise ArribaError(self._lib.arriba_workflow_last_error().decode())
        return seconds.value

    def cancel(self):
        """throws away what was submitted and not worked on"""
        self._lib.arriba_workflow_cancel(self._session)

    def _lane_contexts(self):
        return [ctx for ctx in (self._lib.arriba_workflow_lane_device(self._session, lane) for lane in (0, 1)) if ctx]

    def sample(self, bam, output_file, discarded_output_file=None):
        """one sample, BAM file -> fusions.tsv (and discarded.tsv); returns the stages with their "(remaining=N)" counts"""
        report, timing = _capi.WorkflowReport(), _capi.WorkflowTiming()
        if self._lib.arriba_workflow_sample(self._session, bam.encode(), output_file.encode(), discarded_output_file.encode() if discarded_output_file else None, byref(report), byref(timing)) != 0:
            raise ArribaError(self._lib.arriba_workflow_last_error().decode())
        self.ctx = self._lib.arriba_workflow_device(self._session)  # (the lane that worked on this sample)
        self.timing = {name: getattr(timing, name) for name, _ in _capi.WorkflowTiming._fields_}
        self.report = [(report.stages[k].stage.decode(), int(report.stages[k].count)) for k in range(report.n_stages)]
        return self.report

    def set_profiling(self, enabled, only_new_lanes=False):
        for ctx in self._lane_contexts():
            if only_new_lanes and ctx in self._profiled:
                continue
            self._check(self.api.set_profiling(ctx, int(enabled)))
            self._profiled.add(ctx)
        self._profiling_on = bool(enabled)

    def kernel_profile(self):
        """the launches of both lanes (a sample submitted ahead runs in the other lane)"""
        mine, launches = self.ctx, []
        try:
            for ctx in self._lane_contexts():
                self.ctx = ctx
                launches += DevicePipeline.kernel_profile(self)
        finally:
            self.ctx = mine
        return launches

    def gene_sets(self, slot):
        """of the fragments the device context of the last sample holds: all of them, or -- one sample over several ranks, the reads sharded -- this rank's share"""
        held = int(self.timing["shard_fragments"]) if getattr(self, "timing", None) and self.timing.get("shard_fragments", 0) > 0 else self.n
        everything, self.n = self.n, held
        try:
            return DevicePipeline.gene_sets(self, slot)
        finally:
            self.n = everything

    def _check(self, status):
        if status != 0:
            raise ArribaError("ERROR: " + self.api.last_error().decode() + " (status %d)" % status)

    def close(self):
        if self._session:
            self._lib.arriba_workflow_close(self._session)
            self._session = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DevicePipeline(object):
    """The device stages of the hot path in the reference's order.

    `api` defaults to the product library (libarriba_gpu.so, prefix agpu_); the CPU-only tests pass the host
    stepping harness of tests/emu instead.
    """

    def __init__(self, session, params=None, api=None, device=0, batch_view=None, bam=None, external_duplicate_marking=False, max_itd_length=100, piece_bytes=64 << 20, profiling=False):
        """bam: path of the BAM file -> read_chimeric_alignments runs on the device (agpu_ingest_*): the host session only opens the file, parses the
        header and feeds the bytes; the batch never exists on the host.  Without it the batch of the session's host ingest is uploaded."""
        self.session = session
        self.api = api if api is not None else _capi.bind_device_api(_capi.device_library(), "agpu_")
        self.params = _capi.Params()
        self.api.default_params(byref(self.params))
        if params:
            for key, value in params.items():
                if key == "disable_filters":
                    for name in value:
                        self.params.filter_enabled[_capi.FILTER_NAMES.index(name)] = 0
                else:
                    setattr(self.params, key, value)
        self.ctx = self.api.create(device, byref(self.params))
        if not self.ctx:
            raise ArribaError("ERROR: " + self.api.last_error().decode())
        self.timings = {}
        self._profiling_on = False
        self.wall_ms = {}
        import time
        self._last_record = time.perf_counter()
        self._check(self.api.upload_annotation(self.ctx, session.annotation_view))
        self.device_ingest = bam is not None
        self.ingest_result = None
        if profiling:
            self.set_profiling(True)  # before the ingest: its kernels are part of the profile
        if self.device_ingest:
            self.read_chimeric_alignments(bam, external_duplicate_marking, max_itd_length, piece_bytes)
        else:
            self._check(self.api.upload_genome(self.ctx, session.genome_view))
            self._check(self.api.upload_batch(self.ctx, batch_view if batch_view is not None else session.batch_view))
            self.n = int(batch_view.contents.n) if batch_view is not None else session.fragment_count
        self.n_real_genes = session.annotation_view.contents.n_genes
        self.n_dummy_genes = 0
        self.scalars = {}

    def _free_pieces(self):
        if getattr(self, "_pieces", None) is not None:
            for pointer in self._pieces[1]:
                self.api.host_free(pointer)
            self._pieces = None

    def close(self):
        if self.ctx:
            self._free_pieces()
            self.api.destroy(self.ctx)
            self.ctx = None

    def read_chimeric_alignments(self, bam, external_duplicate_marking=False, max_itd_length=100, piece_bytes=64 << 20):
        """reference: read_chimeric_alignments, source/read_chimeric_alignments.cpp:560-773, on the device: the host opens the file, parses the BAM header
        and feeds the bytes in pieces (two pinned buffers in turn); records are cut, collated by name, classified, sanity-checked, sorted and packed in HBM"""
        import time
        started = time.perf_counter()
        config, result, fed = self._ingest_records(bam, external_duplicate_marking, max_itd_length, piece_bytes)
        finished = time.perf_counter()
        self._record("read_chimeric_alignments")
        self._adopt_ingest(config, result)
        self.ingest_seconds = {"feed": fed - started, "device": finished - fed, "adopt": time.perf_counter() - finished}
        return self.n

    def _ingest_records(self, bam, external_duplicate_marking, max_itd_length, piece_bytes, part=None, parts=None):
        """opens the file (or part `part` of `parts` of its records), feeds the pieces, runs agpu_ingest_finish; returns (config, result, time the last piece was fed)"""
        import time
        lib, handle = self.session._lib, self.session._session
        host_error = lambda: ArribaError("ERROR: " + lib.ahost_last_error().decode())
        config = _capi.IngestConfig()
        if part is None:
            status = lib.ahost_bam_open(handle, bam.encode(), int(external_duplicate_marking), max_itd_length, byref(config))
        else:
            status = lib.ahost_bam_open_part(handle, bam.encode(), int(external_duplicate_marking), max_itd_length, part, parts, byref(config))
        if status != 0:
            raise host_error()
        try:
            if getattr(self, "_genome_contigs", None) != config.n_contigs:
                self._check(self.api.upload_genome(self.ctx, self.session.genome_view))  # the contigs of the BAM header are part of the run now
                self._genome_contigs = config.n_contigs
            self._check(self.api.ingest_begin(self.ctx, byref(config)))
            block_capacity = piece_bytes // 4096 + 16
            # the two pinned buffers of the pieces stay with the pipeline (pinning 2 x 256 MB costs as much as feeding a gigabyte): a resident service reads sample after sample
            if getattr(self, "_pieces", None) is None or self._pieces[0] != piece_bytes:
                self._free_pieces()
                buffers = []
                for _ in range(2):
                    pointer = self.api.host_alloc(piece_bytes)
                    if not pointer:
                        for allocated in buffers:
                            self.api.host_free(allocated)
                        raise ArribaError("ERROR: " + self.api.last_error().decode())
                    buffers.append(pointer)
                self._pieces = (piece_bytes, buffers, [(_capi.BgzfBlock * block_capacity)(), (_capi.BgzfBlock * block_capacity)()])
            _, buffers, tables = self._pieces
            piece = _capi.BamPiece()
            pushes = 0
            while True:
                status = lib.ahost_bam_next(handle, buffers[pushes & 1], piece_bytes, tables[pushes & 1], block_capacity, byref(piece))
                if status < 0:
                    raise host_error()
                if status == 0:
                    break
                if piece.stored_bgzf:
                    self._check(self.api.ingest_push_bgzf(self.ctx, buffers[pushes & 1], piece.bytes, tables[pushes & 1], piece.n_blocks, piece.stream_bytes))
                else:
                    self._check(self.api.ingest_push(self.ctx, buffers[pushes & 1], piece.bytes))
                pushes += 1
            fed = time.perf_counter()
            result = _capi.IngestResult()
            self._check(self.api.ingest_finish(self.ctx, byref(result)))
        finally:
            lib.ahost_bam_close(handle)
        # (config.coverage_window_offset points into the session: read it before anything else touches the session)
        self._coverage_windows = int(config.coverage_window_offset[config.n_contigs]) if config.n_contigs else 0
        return config, result, fed

    def _adopt_ingest(self, config, result):
        """hands what the host's sequential stages and its writer need from the ingest on the device to the host session (reference: the tail of
        read_chimeric_alignments, source/read_chimeric_alignments.cpp:759-771, and coverage_t)"""
        lib, handle = self.session._lib, self.session._session
        n_contigs = config.n_contigs
        viral = np.zeros(max(n_contigs, 1), dtype=np.uint64)
        self._check(self.api.get_viral_read_counts(self.ctx, viral.ctypes.data))
        windows = self._coverage_windows
        coverage, starts, ends = np.zeros(max(windows, 1), dtype=np.uint16), np.zeros(max(windows, 1), dtype=np.uint8), np.zeros(max(windows, 1), dtype=np.uint8)
        self._check(self.api.get_coverage(self.ctx, coverage.ctypes.data, starts.ctypes.data, ends.ctypes.data))
        if lib.ahost_adopt_device_ingest(handle, byref(result), viral.ctypes.data, coverage.ctypes.data, starts.ctypes.data, ends.ctypes.data) != 0:
            raise ArribaError("ERROR: " + lib.ahost_last_error().decode())
        self.ingest_result = result
        self.n = int(result.fragments)
        self.device_ingest = True
        self.n_dummy_genes = 0
        self.scalars = {}
        return self.n

    def batch_rows(self, fragments=None):
        """rows of the batch on the device as numpy arrays (fragments=None: all) -- what agpu_gather_rows_* hands to the host's writer; also the test's view of a batch built on the device"""
        if fragments is not None:
            fragments = np.ascontiguousarray(fragments, dtype=np.uint32)
        sizes = [c_uint64(), c_uint64(), c_uint64()]
        n = self.n if fragments is None else fragments.size
        self._check(self.api.gather_rows_begin(self.ctx, None if fragments is None else fragments.ctypes.data, n, byref(sizes[0]), byref(sizes[1]), byref(sizes[2])))
        rows = _capi.BatchRows()
        arrays = {"n_aln": np.zeros(max(n, 1), np.uint8), "fbits": np.zeros(max(n, 1), np.uint8), "group": np.zeros(max(n, 1), np.uint32),
                  "cigar_pool": np.zeros(max(sizes[0].value, 1), np.uint32), "seq_pool": np.zeros(max(sizes[1].value, 4), np.uint8), "name_offset": np.zeros(n + 1, np.uint32), "names": np.zeros(max(sizes[2].value, 1), np.uint8)}
        for slot in range(3):
            for name, dtype in (("contig", np.uint16), ("start", np.int32), ("end", np.int32), ("abits", np.uint8), ("cigar_offset", np.uint32), ("cigar_count", np.uint16)):
                arrays["%s%d" % (name, slot)] = np.zeros(max(n, 1), dtype)
                getattr(rows, name)[slot] = arrays["%s%d" % (name, slot)].ctypes.data
        for slot in range(2):
            for name in ("seq_offset", "seq_length"):
                arrays["%s%d" % (name, slot)] = np.zeros(max(n, 1), np.uint32)
                getattr(rows, name)[slot] = arrays["%s%d" % (name, slot)].ctypes.data
        for name in ("n_aln", "fbits", "group", "cigar_pool", "seq_pool", "name_offset", "names"):
            setattr(rows, name, arrays[name].ctypes.data)
        self._check(self.api.gather_rows_copy(self.ctx, byref(rows)))
        arrays["n"] = n
        arrays["cigar_pool"] = arrays["cigar_pool"][:rows.cigar_pool_size]; arrays["seq_pool"] = arrays["seq_pool"][:rows.seq_pool_size]; arrays["names"] = arrays["names"][:rows.names_size]
        for key in list(arrays):
            if key not in ("n", "cigar_pool", "seq_pool", "names", "name_offset"):
                arrays[key] = arrays[key][:n]
        return arrays, rows

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, status):
        if status != 0:
            raise ArribaError("ERROR: " + self.api.last_error().decode() + " (status %d)" % status)

    def _record(self, stage):
        import time
        now = time.perf_counter()
        self.wall_ms[stage] = self.wall_ms.get(stage, 0.0) + (now - self._last_record) * 1e3  # host wall time since the previous stage ended
        self._last_record = now
        ms, size = c_float(), c_uint64()
        self.api.last_kernel_ms(self.ctx, byref(ms))
        self.api.last_kernel_bytes(self.ctx, byref(size))
        self.timings[stage] = {"ms": ms.value, "bytes": size.value}

    def reset(self):
        """back to the state right after the upload (for repeated timed passes over the same resident batch)"""
        self._check(self.api.reset(self.ctx))
        self._record("reset")
        self.n_dummy_genes = 0

    # ---- stages, named after the reference functions they replace ---------------------------------

    def mark_multimappers(self):
        marked = c_uint64()
        self._check(self.api.mark_multimappers(self.ctx, byref(marked)))
        self._record("mark_multimappers")
        return marked.value

    def detect_strandedness(self):
        """reference: detect_strandedness, source/read_stats.cpp:94-143 -- on the host's batch, or on the device when the batch was built there"""
        if not self.device_ingest:
            return self.session.detect_strandedness()
        verdict = ctypes.c_int()
        self._check(self.api.detect_strandedness(self.ctx, byref(verdict)))
        return verdict.value

    def annotate_alignments(self, strandedness=None):
        if strandedness is None:
            strandedness = self.detect_strandedness()
        self.scalars["strandedness"] = strandedness
        self.params.strandedness = strandedness
        self._check(self.api.set_params(self.ctx, byref(self.params)))
        n_dummy = c_uint32()
        self._check(self.api.annotate(self.ctx, byref(n_dummy)))
        self._record("annotate")
        self.n_dummy_genes = n_dummy.value
        return self.n_dummy_genes

    def filter_duplicates_and_contigs(self, top_viral_contigs=5, viral_contig_min_covered_fraction=0.05):
        count = c_uint64()
        self._check(self.api.get_viral_integration_sites(self.ctx, None, 0, byref(count)))
        pairs = np.zeros(2 * max(count.value, 1), dtype=np.uint32)
        self._check(self.api.get_viral_integration_sites(self.ctx, pairs.ctypes.data, count.value, byref(count)))
        gene_bits = self.gene_table()["bits"]
        top, low = self.session.viral_verdicts(pairs[:2 * count.value], gene_bits, top_viral_contigs, viral_contig_min_covered_fraction)
        self._check(self.api.read_filters_stage1(self.ctx, top.ctypes.data, low.ctypes.data))
        self._record("read_filters_stage1")

    def estimate_fragment_length(self):
        gaps = np.zeros(100001, dtype=np.int32)
        n_samples, visited = c_uint32(), c_uint64()
        self._check(self.api.fragment_length_samples(self.ctx, gaps.ctypes.data, byref(n_samples), byref(visited)))
        self._record("fragment_length_samples")
        if self.device_ingest:
            # the reference's sequential float sum of the read lengths (hazard H4) runs on the host over the lengths of the fragments its loop visited
            count = min(visited.value, self.n)
            lengths = [np.zeros(max(count, 1), dtype=np.uint32), np.zeros(max(count, 1), dtype=np.uint32)]
            self._check(self.api.get_read_lengths(self.ctx, 0, count, lengths[0].ctypes.data, lengths[1].ctypes.data))
            lib = self.session._lib
            total = lib.ahost_read_length_sum_of(c_float(0), lengths[0].ctypes.data, lengths[1].ctypes.data, count)
            samples = np.ascontiguousarray(gaps[:n_samples.value], dtype=np.int32)
            mean, stddev, read_length, max_mate_gap = c_float(), c_float(), c_float(), c_int32()
            estimated = lib.ahost_estimate_fragment_length_from_sums(samples.ctypes.data, samples.size, c_float(total), count, self.params.fragment_length, byref(mean), byref(stddev), byref(read_length), byref(max_mate_gap))
            estimate = {"estimated": bool(estimated), "mate_gap_mean": mean.value, "mate_gap_stddev": stddev.value, "read_length_mean": read_length.value, "max_mate_gap": max_mate_gap.value}
        else:
            estimate = self.session.estimate_fragment_length(gaps[:n_samples.value], visited.value, self.params.fragment_length)
        self.scalars.update(estimate)
        self.scalars["mate_gap_samples"] = n_samples.value
        return estimate

    def filter_reads(self):
        remaining = np.zeros(_capi.FILTER_COUNT, dtype=np.uint64)
        self._check(self.api.read_filters_stage2(self.ctx, remaining.ctypes.data))
        self._record("read_filters_stage2")
        self.remaining = {_capi.FILTER_NAMES[f]: int(remaining[f]) for f in (1, 30, 31, 32, 33, 4, 2, 3, 6, 7, 5, 8, 10, 36)}
        return self.remaining

    def run_read_level(self, strandedness=None, top_viral_contigs=5, viral_contig_min_covered_fraction=0.05):
        """mark_multimappers ... filter_low_entropy (reference: source/arriba.cpp:141-409)."""
        self.scalars["marked_multimappers"] = self.mark_multimappers()
        self.annotate_alignments(strandedness)
        self.filter_duplicates_and_contigs(top_viral_contigs, viral_contig_min_covered_fraction)
        self.estimate_fragment_length()
        return self.filter_reads()

    def run_workflow(self, output_file, discarded_output_file=None, blacklist_file=None, known_fusions_file=None, tags_file=None, protein_domains_file=None, genomic_breakpoints_file=None,
                     max_genomic_breakpoint_distance=100000, strandedness=None, evalue_cutoff=0.3,
                     min_itd_support=10, min_itd_allele_fraction=0.07, high_expression_quantile=0.998, min_spliced_events=4, min_anchor_length=23,
                     max_homolog_identity=0.3, max_itd_length=100, fill_sequence_gaps=False, top_viral_contigs=5, viral_contig_min_covered_fraction=0.05,
                     print_extra_info_for_discarded_fusions=False, log=None):
        """The reference's main() behind read_chimeric_alignments (source/arriba.cpp:119-610) with its default parameters: the read-level cascade, find_fusions,
        every candidate-level filter in the reference's order, assign_confidence, and the two output files.  `log` receives (stage, remaining) pairs --
        the numbers the reference prints as "(remaining=N)".  Filters switched off with -f are skipped by the stages themselves (agpu_params.filter_enabled)."""
        note = log if log is not None else (lambda stage, remaining: None)
        self.run_read_level(strandedness, top_viral_contigs, viral_contig_min_covered_fraction)
        note("find_fusions", self.find_fusions())
        self.upload_coverage()
        if genomic_breakpoints_file:
            note("mark_genomic_support", self.mark_genomic_support(genomic_breakpoints_file, max_genomic_breakpoint_distance))
        note("merge_adjacent_fusions", self.merge_adjacent_fusions())
        note("filter_multimappers", self.filter_multimappers()[0])
        self.estimate_expected_fusions()
        self.filter_candidate_predicates()
        note("filter_relative_support", self.filter_relative_support())
        note("recover_internal_tandem_duplication", self.recover_internal_tandem_duplication(min_itd_support, min_itd_allele_fraction))
        note("filter_both_intronic", self.filter_both_intronic())
        if known_fusions_file:
            note("recover_known_fusions", self.recover_known_fusions(known_fusions_file))
        note("filter_in_vitro", self.filter_in_vitro(high_expression_quantile))
        note("recover_both_spliced", self.recover_both_spliced())
        note("select_most_supported_breakpoints", self.select_most_supported_breakpoints())
        note("filter_marginal_read_through", self.filter_marginal_read_through())
        note("recover_many_spliced", self.recover_many_spliced(min_spliced_events))
        if genomic_breakpoints_file:
            self.assign_confidence()  # filter_no_genomic_support looks at the confidence (source/arriba.cpp:516-523)
            note("filter_no_genomic_support", self.filter_no_genomic_support())
        if blacklist_file:
            note("filter_blacklisted_ranges", self.filter_blacklisted_ranges(blacklist_file, evalue_cutoff))
        note("filter_short_anchor", self.filter_short_anchor(min_anchor_length))
        note("filter_end_to_end", self.filter_end_to_end())
        note("filter_no_coverage", self.filter_no_coverage())
        self.make_kmer_index()
        note("filter_homologs", self.filter_homologs(max_homolog_identity))
        note("filter_mismappers", self.filter_mismappers()[0])
        if genomic_breakpoints_file:
            note("recover_genomic_support", self.recover_genomic_support())
        note("select_most_supported_breakpoints", self.select_most_supported_breakpoints())
        note("recover_isoforms", self.recover_isoforms())
        if tags_file:
            self.session.load_tags(tags_file)
        if protein_domains_file:
            self.session.load_protein_domains(protein_domains_file)
        self.write_fusions(output_file, discarded=False, max_itd_length=max_itd_length, fill_sequence_gaps=fill_sequence_gaps)
        if discarded_output_file:
            self.write_fusions(discarded_output_file, discarded=True, print_extra_info=print_extra_info_for_discarded_fusions, max_itd_length=max_itd_length, fill_sequence_gaps=fill_sequence_gaps)

    def find_fusions(self, max_mate_gap=None):
        """reference: find_fusions, source/fusions.cpp:203-473; returns the number of candidates"""
        if max_mate_gap is None:
            max_mate_gap = self.scalars["max_mate_gap"]
        count = c_uint64()
        self._check(self.api.find_fusions(self.ctx, max_mate_gap, byref(count)))
        self._record("find_fusions")
        self.n_candidates = count.value
        return self.n_candidates

    def candidates(self, lists=True):
        """Candidate table in the reference's insertion order; lists as (offset[3n+1], reads).  lists=False: the columns only."""
        n = self.n_candidates
        u32 = lambda: np.zeros(max(n, 1), dtype=np.uint32)
        i32 = lambda: np.zeros(max(n, 1), dtype=np.int32)
        table = {"gene1": u32(), "gene2": u32(), "contigs": u32(), "breakpoint1": i32(), "breakpoint2": i32(), "flags": u32(), "filter": np.zeros(max(n, 1), dtype=np.uint8),
                 "split_reads1": u32(), "split_reads2": u32(), "discordant_mates": u32(), "anchor_start1": i32(), "anchor_start2": i32(), "list_offset": np.zeros(3 * n + 1, dtype=np.uint64)}
        order = ["gene1", "gene2", "contigs", "breakpoint1", "breakpoint2", "flags", "filter", "split_reads1", "split_reads2", "discordant_mates", "anchor_start1", "anchor_start2", "list_offset"]
        self._check(self.api.get_candidates(self.ctx, *[table[k].ctypes.data if (lists or k != "list_offset") else None for k in order]))
        if not lists:
            for key in order[:-1]:
                table[key] = table[key][:n]
            del table["list_offset"]
            return table
        total = c_uint64()
        self._check(self.api.get_candidate_read_lists(self.ctx, None, 0, byref(total)))
        reads = np.zeros(max(total.value, 1), dtype=np.uint32)
        self._check(self.api.get_candidate_read_lists(self.ctx, reads.ctypes.data, total.value, byref(total)))
        for key in order[:-1]:
            table[key] = table[key][:n]
        table["read_lists"] = reads[:total.value]
        return table

    def selected_candidates(self, discarded=False):
        """the candidates an output file holds (-o: filter == none, -O: the others), picked on the device with the columns the writer prints (agpu_select_candidates)"""
        count = c_uint64()
        self._check(self.api.select_candidates(self.ctx, int(discarded), byref(count)))
        n = count.value
        dtypes = {"candidate": np.uint32, "gene1": np.uint32, "gene2": np.uint32, "contigs": np.uint32, "breakpoint1": np.int32, "breakpoint2": np.int32, "flags": np.uint32, "filter": np.uint8, "split_reads1": np.uint32,
                  "split_reads2": np.uint32, "discordant_mates": np.uint32, "evalue": np.float32, "confidence": np.uint8, "iteration_rank": np.uint32, "closest_genomic_breakpoint1": np.int32, "closest_genomic_breakpoint2": np.int32}
        columns = {key: np.zeros(max(n, 1), dtype=dtype) for key, dtype in dtypes.items()}
        view = _capi.SelectedCandidates()
        for key, column in columns.items():
            setattr(view, key, column.ctypes.data)
        self._check(self.api.get_selected_candidates(self.ctx, byref(view)))
        return {key: column[:n] for key, column in columns.items()}

    def filters_of(self, fragments):
        """the filters of the given fragments (agpu_get_filters_of)"""
        fragments = np.ascontiguousarray(fragments, dtype=np.uint32)
        out = np.zeros(max(fragments.size, 1), dtype=np.uint8)
        self._check(self.api.get_filters_of(self.ctx, fragments.ctypes.data if fragments.size else None, fragments.size, out.ctypes.data))
        return out[:fragments.size]

    def candidate_read_lists_of(self, candidates):
        """(list_offset[3n+1] starting at 0, reads) of the given candidates only"""
        candidates = np.ascontiguousarray(candidates, dtype=np.uint32)
        n = candidates.size
        offsets = np.zeros(3 * n + 1, dtype=np.uint64)
        total = c_uint64()
        self._check(self.api.get_candidate_read_lists_of(self.ctx, candidates.ctypes.data if n else None, n, offsets.ctypes.data, None, 0, byref(total)))
        reads = np.zeros(max(total.value, 1), dtype=np.uint32)
        if total.value:
            self._check(self.api.get_candidate_read_lists_of(self.ctx, candidates.ctypes.data, n, offsets.ctypes.data, reads.ctypes.data, total.value, byref(total)))
        return offsets, reads[:total.value]

    def merge_adjacent_fusions(self, max_distance=5):
        """reference: merge_adjacent_fusions, source/merge_adjacent_fusions.cpp:19-108; returns the number of unfiltered candidates"""
        remaining = c_uint64()
        self._check(self.api.merge_adjacent_fusions(self.ctx, max_distance, byref(remaining)))
        self._record("merge_adjacent_fusions")
        return remaining.value

    def filter_multimappers(self):
        """reference: filter_multimappers, source/filter_multimappers.cpp:109-221; returns (remaining candidates, fragments discarded)"""
        remaining, discarded = c_uint64(), c_uint64()
        self._check(self.api.filter_multimappers(self.ctx, byref(remaining), byref(discarded)))
        self._record("filter_multimappers")
        return remaining.value, discarded.value

    # event-level predicates behind filter_relative_support (each returns the reference's "(remaining=N)")
    def upload_coverage(self):
        if self.device_ingest:
            return  # coverage_t was built on the device
        view = self.session._lib.ahost_coverage_view(self.session._session)
        if not view:
            raise ArribaError("ERROR: " + self.session._lib.ahost_last_error().decode())
        self._check(self.api.upload_coverage(self.ctx, view))

    def _event_stage(self, name, *arguments):
        remaining = c_uint64()
        self._check(getattr(self.api, name)(self.ctx, *arguments, byref(remaining)))
        self._record(name)
        return remaining.value

    def recover_internal_tandem_duplication(self, min_supporting_reads=10, min_fraction_of_coverage=0.07):
        """reference: recover_internal_tandem_duplication, source/recover_internal_tandem_duplication.cpp:11-85 (-Z, -z)"""
        return self._event_stage("recover_internal_tandem_duplication", min_supporting_reads, c_float(min_fraction_of_coverage))

    def filter_both_intronic(self):
        """reference: filter_both_intronic, source/filter_both_intronic.cpp:18-36"""
        return self._event_stage("filter_both_intronic")

    def filter_short_anchor(self, min_length=23):
        """reference: filter_short_anchor, source/filter_short_anchor.cpp:7-24 (-A, default 23)"""
        return self._event_stage("filter_short_anchor", min_length)

    def filter_end_to_end(self):
        """reference: filter_end_to_end_fusions, source/filter_end_to_end.cpp:28-78"""
        return self._event_stage("filter_end_to_end")

    def filter_no_coverage(self):
        """reference: filter_no_coverage, source/filter_no_coverage.cpp:9-103"""
        return self._event_stage("filter_no_coverage")

    def select_most_supported_breakpoints(self):
        """reference: select_most_supported_breakpoints, source/select_best.cpp:21-80"""
        return self._event_stage("select_most_supported_breakpoints")

    def filter_in_vitro(self, high_expression_quantile=0.998):
        """reference: filter_in_vitro, source/filter_in_vitro.cpp:82-228 (-Q, default 0.998)"""
        return self._event_stage("filter_in_vitro", c_float(high_expression_quantile))

    def recover_both_spliced(self, max_fusions_to_recover=200, high_expression_quantile=0.998, max_exon_size=1000, max_coverage=1000):
        """reference: recover_both_spliced, source/recover_both_spliced.cpp:72-182 (the arguments of the call at source/arriba.cpp:491)"""
        return self._event_stage("recover_both_spliced", max_fusions_to_recover, c_float(high_expression_quantile), max_exon_size, max_coverage)

    def load_range_rules(self, path, allow_keywords):
        """a blacklist (allow_keywords=True) or known-fusions file -> (pointer to rules, count); reference: parse_blacklist_item, source/filter_blacklisted_ranges.cpp:83-118"""
        rules, count = POINTER(_capi.RangeRule)(), c_uint32()
        if self.session._lib.ahost_load_range_rules(self.session._session, path.encode(), int(allow_keywords), byref(rules), byref(count)) != 0:
            raise ArribaError("ERROR: " + self.session._lib.ahost_last_error().decode())
        return rules, count.value

    def filter_blacklisted_ranges(self, path, evalue_cutoff=0.3, max_mate_gap=None):
        """reference: filter_blacklisted_ranges, source/filter_blacklisted_ranges.cpp:227-301 (-b)"""
        rules, count = self.load_range_rules(path, True)
        gap = int(self.scalars["max_mate_gap"] if max_mate_gap is None else max_mate_gap)
        return self._event_stage("filter_blacklisted_ranges", rules, count, c_float(evalue_cutoff), gap)

    def recover_known_fusions(self, path, max_mate_gap=None):
        """reference: recover_known_fusions, source/recover_known_fusions.cpp:14-100 (-k)"""
        rules, count = self.load_range_rules(path, False)
        gap = int(self.scalars["max_mate_gap"] if max_mate_gap is None else max_mate_gap)
        return self._event_stage("recover_known_fusions", rules, count, gap)

    def write_fusions(self, path, discarded=False, print_extra_info=None, max_itd_length=100, fill_sequence_gaps=False, detached=False):
        """reference: write_fusions_to_file, source/output_fusions.cpp:1043-1261 (-o / -O); the device's results are fetched and formatted by the host library.
        detached: the way the C++ session writes the last file of a sample beside the next sample -- what the writer reads of the sample leaves the host session first
        (ahost_detach_sample), which then holds no sample until its next ingest"""
        if print_extra_info is None:
            print_extra_info = not discarded
        import time
        marks = [("start", time.perf_counter())]
        mark = lambda name: marks.append((name, time.perf_counter()))
        # only the candidates the file will hold travel to the host with their read lists (fusions.tsv: the few thousand that passed every filter, of millions);
        # discarded.tsv counts the discarded reads of every discarded candidate by filter, so it takes the lists of all of them
        table = self.candidates(lists=False)
        mark("candidate columns")
        written = np.flatnonzero((table["filter"] != 0) if discarded else (table["filter"] == 0)).astype(np.uint32)
        list_offset, read_lists = self.candidate_read_lists_of(written)
        mark("read lists")
        n = written.size
        columns = {key: np.ascontiguousarray(table[key][written]) for key in ("gene1", "gene2", "contigs", "breakpoint1", "breakpoint2", "flags", "filter", "split_reads1", "split_reads2", "discordant_mates")}
        columns["list_offset"], columns["read_lists"] = list_offset, np.ascontiguousarray(read_lists)
        columns["evalue"] = np.ascontiguousarray(self.evalues()[written], dtype=np.float32)
        columns["confidence"] = np.ascontiguousarray(self.assign_confidence()[written])
        columns["iteration_rank"] = np.ascontiguousarray(self.candidate_iteration_order()[written], dtype=np.uint32)
        mark("evalue, confidence, order")
        columns["read_filter"] = np.ascontiguousarray(self.filters(), dtype=np.uint8)
        mark("read filters")
        columns["closest_genomic_breakpoint1"], columns["closest_genomic_breakpoint2"] = (np.ascontiguousarray(column[written]) for column in self.genomic_support())
        genes = self.gene_table()
        columns["gene_contig"], columns["gene_start"], columns["gene_end"] = (np.ascontiguousarray(genes[key]) for key in ("contig", "start", "end"))
        view = _capi.FusionTable()
        view.n_candidates = n
        view.n_genes = len(columns["gene_contig"])
        for key, column in columns.items():
            setattr(view, key, column.ctypes.data if column.size else None)
        if self.device_ingest and print_extra_info:
            # the host's writer works on the rows of the supporting reads of the candidates it writes: fetched from the device now
            lib = self.session._lib
            count = c_uint64()
            if lib.ahost_fusion_table_reads(byref(view), int(discarded), None, 0, byref(count)) != 0:
                raise ArribaError("ERROR: " + lib.ahost_last_error().decode())
            fragments = np.zeros(max(count.value, 1), dtype=np.uint32)
            if lib.ahost_fusion_table_reads(byref(view), int(discarded), fragments.ctypes.data, count.value, byref(count)) != 0:
                raise ArribaError("ERROR: " + lib.ahost_last_error().decode())
            fragments = fragments[:count.value]
            mark("reads of the table")
            arrays, rows = self.batch_rows(fragments)
            mark("rows from the device")
            if lib.ahost_set_batch_rows(self.session._session, byref(rows), fragments.ctypes.data if fragments.size else None) != 0:
                raise ArribaError("ERROR: " + lib.ahost_last_error().decode())
            mark("rows into the session")
        self._emit_fusions(view, path, discarded, print_extra_info, max_itd_length, fill_sequence_gaps, detached)
        mark("ahost_write_fusions")
        self.writer_seconds = {name: round(at - marks[k][1], 4) for k, (name, at) in enumerate(marks[1:])}  # where the time of the output side went (bench.py reports it)

    def _emit_fusions(self, view, path, discarded, print_extra_info, max_itd_length, fill_sequence_gaps, detached=False):
        """the rows of the file from the table of the candidates it holds (the host library's writer)"""
        if detached:
            lib = self.session._lib
            sample = lib.ahost_detach_sample(self.session._session)
            if not sample:
                raise ArribaError("ERROR: " + lib.ahost_last_error().decode())
            status = lib.ahost_write_fusions_of(sample, byref(view), path.encode(), int(discarded), int(print_extra_info), max_itd_length, int(self.scalars["max_mate_gap"]), int(fill_sequence_gaps))
            message = lib.ahost_last_error().decode() if status != 0 else ""
            lib.ahost_release_sample(sample)
            if status != 0:
                raise ArribaError("ERROR: " + message)
            return
        if self.session._lib.ahost_write_fusions(self.session._session, byref(view), path.encode(), int(discarded), int(print_extra_info), max_itd_length, int(self.scalars["max_mate_gap"]), int(fill_sequence_gaps)) != 0:
            raise ArribaError("ERROR: " + self.session._lib.ahost_last_error().decode())

    def mark_genomic_support(self, path, max_distance=100000):
        """reference: mark_genomic_support, source/filter_genomic_support.cpp:81-219 (-d, -D); returns the number of candidates with a supporting structural variant"""
        variants, count, marked = POINTER(_capi.GenomicBreakpoint)(), c_uint32(), c_uint64()
        if self.session._lib.ahost_load_genomic_breakpoints(self.session._session, path.encode(), byref(variants), byref(count)) != 0:
            raise ArribaError("ERROR: " + self.session._lib.ahost_last_error().decode())
        self._check(self.api.mark_genomic_support(self.ctx, variants, count.value, max_distance, byref(marked)))
        self._record("mark_genomic_support")
        return marked.value

    def genomic_support(self):
        """closest genomic breakpoints of every candidate (-1 = none)"""
        closest1, closest2 = np.zeros(max(self.n_candidates, 1), dtype=np.int32), np.zeros(max(self.n_candidates, 1), dtype=np.int32)
        self._check(self.api.get_genomic_support(self.ctx, closest1.ctypes.data, closest2.ctypes.data))
        return closest1[:self.n_candidates], closest2[:self.n_candidates]

    def filter_no_genomic_support(self):
        """reference: filter_no_genomic_support, source/filter_genomic_support.cpp:401-417 (behind assign_confidence)"""
        return self._event_stage("filter_no_genomic_support")

    def recover_genomic_support(self):
        """reference: recover_genomic_support, source/filter_genomic_support.cpp:419-444"""
        return self._event_stage("recover_genomic_support")

    def assign_confidence(self):
        """reference: assign_confidence, source/filter_genomic_support.cpp:222-399; returns the confidence (0 low, 1 medium, 2 high) of every candidate"""
        confidence = np.zeros(max(self.n_candidates, 1), dtype=np.uint8)
        self._check(self.api.assign_confidence(self.ctx, confidence.ctypes.data))
        self._record("assign_confidence")
        return confidence[:self.n_candidates]

    def recover_isoforms(self):
        """reference: recover_isoforms, source/recover_isoforms.cpp:10-47"""
        return self._event_stage("recover_isoforms")

    def filter_homologs(self, max_identity_fraction=0.3):
        """reference: filter_homologs, source/filter_homologs.cpp:68-141 (-L, default 0.3); after make_kmer_index"""
        return self._event_stage("filter_homologs", c_float(max_identity_fraction))

    def recover_many_spliced(self, min_spliced_events=4):
        """reference: recover_many_spliced, source/recover_many_spliced.cpp:8-51 (-M, default 4)"""
        return self._event_stage("recover_many_spliced", min_spliced_events)

    def filter_marginal_read_through(self):
        """reference: filter_marginal_read_through, source/filter_marginal_read_through.cpp:8-46"""
        return self._event_stage("filter_marginal_read_through")

    def candidate_iteration_order(self):
        """rank of every candidate in the iteration order of the reference's fusions_t (hazard H2), computed on the device"""
        rank = np.zeros(max(self.n_candidates, 1), dtype=np.uint32)
        self._check(self.api.candidate_iteration_order(self.ctx, rank.ctypes.data))
        self._record("candidate_iteration_order")
        return rank[:self.n_candidates]

    def candidate_iteration_order_on_host(self, table=None):
        """the same order from the literal std::unordered_map of the host library (the check of the device computation)"""
        table = table if table is not None else self.candidates()
        rank = np.zeros(max(self.n_candidates, 1), dtype=np.uint32)
        columns = [np.ascontiguousarray(table[k]) for k in ("gene1", "gene2", "contigs", "breakpoint1", "breakpoint2", "flags")]
        if _capi.host_library().ahost_candidate_iteration_order(self.n_candidates, *[c.ctypes.data for c in columns], rank.ctypes.data) != 0:
            raise ArribaError("ERROR: " + _capi.host_library().ahost_last_error().decode())
        return rank[:self.n_candidates]

    def set_candidate_state(self, filter=None, split_reads1=None, split_reads2=None, discordant_mates=None):
        """push the columns an event-level host stage changed (reference: merge_adjacent_fusions, filter_multimappers)"""
        arrays = [None if a is None else np.ascontiguousarray(a, dtype=t) for a, t in ((filter, np.uint8), (split_reads1, np.uint32), (split_reads2, np.uint32), (discordant_mates, np.uint32))]
        self._check(self.api.set_candidate_state(self.ctx, *[None if a is None else a.ctypes.data for a in arrays]))

    def estimate_expected_fusions(self, mapped_reads=None, iteration_rank=None):
        """reference: estimate_expected_fusions, source/filter_relative_support.cpp:17-207; returns the e-values (float32)"""
        if mapped_reads is None:
            mapped_reads = self.session.mapped_reads
        if iteration_rank is None:
            self._check(self.api.candidate_iteration_order(self.ctx, None))
            self._record("candidate_iteration_order")
            pointer = None
        else:
            iteration_rank = np.ascontiguousarray(iteration_rank, dtype=np.uint32)
            pointer = iteration_rank.ctypes.data
        self._check(self.api.estimate_expected_fusions(self.ctx, mapped_reads, pointer))
        self._record("estimate_expected_fusions")
        evalue = np.zeros(max(self.n_candidates, 1), dtype=np.float32)
        self._check(self.api.get_evalues(self.ctx, evalue.ctypes.data))
        return evalue[:self.n_candidates]

    def evalues(self):
        """the e-value of every candidate (after estimate_expected_fusions)"""
        evalue = np.zeros(max(self.n_candidates, 1), dtype=np.float32)
        self._check(self.api.get_evalues(self.ctx, evalue.ctypes.data))
        return evalue[:self.n_candidates]

    def filter_candidate_predicates(self):
        """reference: filter_non_coding_neighbors, filter_intragenic_both_exonic, filter_min_support (source/arriba.cpp:437-455);
        returns the number of candidates each of the three discarded"""
        discarded = np.zeros(3, dtype=np.uint64)
        self._check(self.api.filter_candidate_predicates(self.ctx, discarded.ctypes.data))
        self._record("filter_candidate_predicates")
        return {"non_coding_neighbors": int(discarded[0]), "intragenic_exonic": int(discarded[1]), "min_support": int(discarded[2])}

    def filter_relative_support(self):
        remaining = c_uint64()
        self._check(self.api.filter_relative_support(self.ctx, byref(remaining)))
        self._record("filter_relative_support")
        return remaining.value

    def set_read_filters(self, filters):
        """push the read-level filter ids a host stage changed (reference: filter_multimappers)"""
        filters = np.ascontiguousarray(filters, dtype=np.uint8)
        assert filters.size == self.n
        self._check(self.api.set_read_filters(self.ctx, filters.ctypes.data))

    def make_kmer_index(self, padding=None):
        """reference: make_kmer_index, source/filter_mismappers.cpp:47-84; padding as in source/arriba.cpp:552"""
        if padding is None:
            padding = int(np.float32(self.scalars["max_mate_gap"]) + np.float32(2) * np.float32(self.scalars["read_length_mean"]))
        positions = c_uint64()
        self._check(self.api.make_kmer_index(self.ctx, padding, byref(positions)))
        self._record("make_kmer_index")
        return positions.value

    def filter_mismappers(self, max_mate_gap=None):
        """reference: filter_mismappers, source/filter_mismappers.cpp:272-359; returns (remaining candidates, reads discarded)"""
        if max_mate_gap is None:
            max_mate_gap = self.scalars["max_mate_gap"]
        remaining, discarded = c_uint64(), c_uint64()
        self._check(self.api.filter_mismappers(self.ctx, max_mate_gap, byref(remaining), byref(discarded)))
        self._record("filter_mismappers")
        return remaining.value, discarded.value

    def fusion_stats(self):
        stats = np.zeros(5, dtype=np.uint64)
        self._check(self.api.get_fusion_stats(self.ctx, stats.ctypes.data))
        return dict(zip(("emissions", "candidates", "list_entries", "discordant_emissions", "wave_buckets"), (int(v) for v in stats)))

    def set_profiling(self, enabled):
        """per-kernel HIP-event timing on the launch stream (reset on every call)"""
        self._check(self.api.set_profiling(self.ctx, int(enabled)))
        self._profiling_on = bool(enabled)

    def kernel_profile(self):
        """[(kernel name, ms, algorithmic bytes)] for every launch since set_profiling(True)"""
        count = c_uint32()
        self._check(self.api.get_kernel_profile(self.ctx, None, None, None, 0, byref(count)))
        n = count.value
        names = ctypes.create_string_buffer(max(n, 1) * _capi.KERNEL_NAME_LENGTH)
        ms = np.zeros(max(n, 1), dtype=np.float32)
        size = np.zeros(max(n, 1), dtype=np.uint64)
        self._check(self.api.get_kernel_profile(self.ctx, names, ms.ctypes.data, size.ctypes.data, n, byref(count)))
        raw = names.raw
        return [(raw[k * _capi.KERNEL_NAME_LENGTH:(k + 1) * _capi.KERNEL_NAME_LENGTH].split(b"\0")[0].decode(), float(ms[k]), int(size[k])) for k in range(n)]

    def discordant_swapped(self):
        out = np.zeros(self.n, dtype=np.uint8)
        self._check(self.api.get_discordant_swapped(self.ctx, out.ctypes.data))
        return out

    # ---- result access -------------------------------------------------------------------------------

    def filters(self):
        out = np.zeros(self.n, dtype=np.uint8)
        self._check(self.api.get_filters(self.ctx, out.ctypes.data))
        return out

    def alignment_bits(self, slot):
        out = np.zeros(self.n, dtype=np.uint8)
        self._check(self.api.get_alignment_bits(self.ctx, slot, out.ctypes.data))
        return out

    def fragment_bits(self):
        out = np.zeros(self.n, dtype=np.uint8)
        self._check(self.api.get_fragment_bits(self.ctx, out.ctypes.data))
        return out

    def gene_sets(self, slot):
        """Returns (count[n], genes[total]) -- CSR of the gene ids annotated to alignment `slot` of every fragment."""
        total = c_uint64()
        count = np.zeros(self.n, dtype=np.uint8)
        self._check(self.api.get_gene_sets(self.ctx, slot, count.ctypes.data, None, 0, byref(total)))
        genes = np.zeros(max(total.value, 1), dtype=np.uint32)
        self._check(self.api.get_gene_sets(self.ctx, slot, count.ctypes.data, genes.ctypes.data, total.value, byref(total)))
        return count, genes[:total.value]

    def gene_table(self):
        total = self.n_real_genes + self.n_dummy_genes
        table = {"contig": np.zeros(total, dtype=np.uint16), "start": np.zeros(total, dtype=np.int32), "end": np.zeros(total, dtype=np.int32),
                 "bits": np.zeros(total, dtype=np.uint8), "exonic_length": np.zeros(total, dtype=np.int32)}
        self._check(self.api.get_gene_table(self.ctx, 0, total, table["contig"].ctypes.data, table["start"].ctypes.data, table["end"].ctypes.data,
                                            table["bits"].ctypes.data, table["exonic_length"].ctypes.data))
        return table
