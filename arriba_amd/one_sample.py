"""One sample over the GPUs of a node (SURVEY.md section 8 row e; BASELINE.json config 4: "100 M chimeric-read BAM sharded 8 x MI355X").

Where the time of a large sample goes on one GPU (profiles/r02g_bench100m.json: 90.7 s for 100 M fragments) decides what is shared out:

  read_chimeric_alignments   11.5 s   every rank reads its part of the records of the file and builds its part of the batch
                                      (ahost_bam_open_part: the file is cut between read names at places every rank finds by itself)
  -- ONE all-gather of the parts (agpu_shard_export -> all_gather -> agpu_shard_merge): every rank then holds the batch of the whole sample,
     bit for bit the batch a single ingest of all records builds (fragments concatenate in name order; counters and coverage add up)
  all stages up to filter_homologs   1.3 s   run on every rank over the whole batch: identical results without a single exchange
  filter_mismappers          72 s     the re-alignments are independent per read: rank r takes the jobs r, r + N, r + 2N, ...
  -- ONE all-reduce (max) of the verdict bytes
  output files               5.8 s    every rank formats the rows r, r + N, ... (the fusion transcripts from the read pileups), rank 0 gathers the texts and writes

So the two exchanges are large and few, as point-to-point xGMI links like them, and everything the reference computes in an order-dependent way
(source/fusions.cpp, source/filter_duplicates.cpp, the sequential float sums of source/read_stats.cpp) runs unsharded and stays bit-identical by
construction.  (arriba_amd/sharded.py keeps the other design -- the cheap stages sharded as well, four exchanges -- for samples that do not fit one GPU.)

`torch.distributed` carries the collectives (backend "nccl" == RCCL on the GPUs, "gloo" in the CPU tests); buffers handed to them are torch tensors on the
device of the backend, the C ABI writes into / reads from them directly.
"""
from ctypes import byref, c_uint64

import torch
import torch.distributed as dist

from . import _capi
from .pipeline import ArribaError, DevicePipeline


class OneSamplePipeline(DevicePipeline):
    """DevicePipeline whose read_chimeric_alignments and filter_mismappers are shared out over the ranks of `group`; after the constructor every rank holds
    the whole batch.  The rows of the output files are formatted by all ranks and written by rank 0."""

    def __init__(self, session, bam, params=None, api=None, device=0, group=None, external_duplicate_marking=False, max_itd_length=100, piece_bytes=64 << 20, profiling=False):
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.collective_device = torch.device("cuda", device) if dist.get_backend(group) == "nccl" else torch.device("cpu")
        self.exchange = {}
        super().__init__(session, params=params, api=api, device=device, bam=bam, external_duplicate_marking=external_duplicate_marking, max_itd_length=max_itd_length,
                         piece_bytes=piece_bytes, profiling=profiling)

    def _sync(self):
        """The C ABI works on its own HIP stream: torch's work on the buffers (fills, the collective) must be done before it touches them."""
        if self.collective_device.type == "cuda":
            torch.cuda.synchronize(self.collective_device)

    def _all_gather_int(self, value):
        mine = torch.tensor([int(value)], dtype=torch.int64, device=self.collective_device)
        out = [torch.zeros_like(mine) for _ in range(self.world)]
        dist.all_gather(out, mine, group=self.group)
        return [int(t.item()) for t in out]

    def _together(self, work):
        """runs work() and tells the other ranks how it went before anybody enters the next collective: an error on one rank (a damaged block in its part of the file,
        no memory) ends the run on all of them with a message instead of leaving the others waiting in an all-gather"""
        error, result = None, None
        try:
            result = work()
        except Exception as problem:  # (also what torch or numpy raise inside work(): the other ranks must not be left waiting)
            error = problem
        ok = torch.tensor([0 if error else 1], dtype=torch.int64, device=self.collective_device)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.group)
        if error:
            raise error
        if int(ok.item()) == 0:
            raise ArribaError("ERROR: another rank of the sample failed (its own message says why)")
        return result

    def read_chimeric_alignments(self, bam, external_duplicate_marking=False, max_itd_length=100, piece_bytes=64 << 20):
        """reference: read_chimeric_alignments (source/read_chimeric_alignments.cpp:560-773): this rank's part of the records, then the parts of all ranks"""
        import time
        started = time.perf_counter()
        config, _, fed = self._together(lambda: self._ingest_records(bam, external_duplicate_marking, max_itd_length, piece_bytes, part=self.rank, parts=self.world))
        ingested = time.perf_counter()
        self._record("read_chimeric_alignments")
        size = c_uint64()
        self._together(lambda: self._check(self.api.shard_export_size(self.ctx, byref(size))))
        sizes = self._all_gather_int(size.value)
        stride = (max(sizes) + 15) & ~15
        mine = torch.empty(stride, dtype=torch.uint8, device=self.collective_device)
        blocks = torch.empty(self.world * stride, dtype=torch.uint8, device=self.collective_device)
        self._sync()
        self._together(lambda: self._check(self.api.shard_export(self.ctx, mine.data_ptr(), stride)))
        exported = time.perf_counter()
        if self.collective_device.type == "cuda":
            dist.all_gather_into_tensor(blocks, mine, group=self.group)
        else:
            dist.all_gather(list(blocks.view(self.world, stride).unbind(0)), mine, group=self.group)
        self._sync()
        gathered = time.perf_counter()
        result = _capi.IngestResult()
        self._together(lambda: self._check(self.api.shard_merge(self.ctx, blocks.data_ptr(), stride, self.world, byref(result))))
        del blocks, mine
        if self.collective_device.type == "cuda":
            torch.cuda.empty_cache()  # the buffers of the exchange (the size of the whole batch) go back to the device: the stages allocate through the C ABI, not through torch
        merged = time.perf_counter()
        self._record("shard_merge")
        self._adopt_ingest(config, result)
        self.exchange["part_bytes"] = sizes
        self.ingest_seconds = {"feed": fed - started, "device": ingested - fed, "export": exported - ingested, "all_gather": gathered - exported, "merge": merged - gathered,
                               "adopt": time.perf_counter() - merged}
        return self.n

    def filter_mismappers(self, max_mate_gap=None):
        """reference: filter_mismappers (source/filter_mismappers.cpp:272-359); the re-alignments shared out, one all-reduce of the verdicts"""
        if max_mate_gap is None:
            max_mate_gap = self.scalars["max_mate_gap"]
        n_jobs = c_uint64()
        self._together(lambda: self._check(self.api.mismapper_jobs(self.ctx, byref(n_jobs))))
        self._record("filter_mismappers")
        spent = dict(self.timings["filter_mismappers"])
        verdicts = torch.zeros(max(n_jobs.value, 1), dtype=torch.uint8, device=self.collective_device)
        self._sync()
        self._together(lambda: self._check(self.api.mismapper_verdicts(self.ctx, max_mate_gap, self.rank, self.world, verdicts.data_ptr())))
        self._record("filter_mismappers")
        spent["ms"] += self.timings["filter_mismappers"]["ms"]
        self._sync()
        dist.all_reduce(verdicts, op=dist.ReduceOp.MAX, group=self.group)
        self._sync()
        remaining, discarded = c_uint64(), c_uint64()
        self._together(lambda: self._check(self.api.filter_mismappers_apply(self.ctx, verdicts.data_ptr(), byref(remaining), byref(discarded))))
        del verdicts
        self._record("filter_mismappers")
        spent["ms"] += self.timings["filter_mismappers"]["ms"]
        spent["bytes"] = self.timings["filter_mismappers"]["bytes"]
        self.timings["filter_mismappers"] = spent
        self.exchange["mismapper_jobs"] = n_jobs.value
        return remaining.value, discarded.value

    def _emit_fusions(self, view, path, discarded, print_extra_info, max_itd_length, fill_sequence_gaps, detached=False):
        """Every rank holds the same candidates and the rows of their supporting reads: rank r formats the rows r, r + N, r + 2N, ... of the file (the fusion transcripts
        from the pileups of the supporting reads are the expensive part), the texts are gathered on rank 0, which interleaves them and writes the file."""
        import ctypes
        import numpy as np
        lib = self.session._lib
        text, size = ctypes.c_void_p(), c_uint64()
        def format_rows():
            if lib.ahost_format_fusions(self.session._session, byref(view), int(discarded), int(print_extra_info), max_itd_length, int(self.scalars["max_mate_gap"]), int(fill_sequence_gaps),
                                        self.rank, self.world, byref(text), byref(size)) != 0:
                raise ArribaError("ERROR: " + lib.ahost_last_error().decode())
        self._together(format_rows)
        sizes = self._all_gather_int(size.value)
        width = max(max(sizes), 1)
        mine = torch.zeros(width, dtype=torch.uint8)
        if size.value:
            mine[:size.value] = torch.from_numpy(np.ctypeslib.as_array(ctypes.cast(text, ctypes.POINTER(ctypes.c_uint8)), shape=(size.value,)))
        mine = mine.to(self.collective_device)
        parts = [torch.zeros(width, dtype=torch.uint8, device=self.collective_device) for _ in range(self.world)] if self.rank == 0 else None
        dist.gather(mine, parts, dst=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
        if self.rank != 0:
            return
        texts = [parts[r][:sizes[r]].cpu().numpy().tobytes() for r in range(self.world)]
        header, _, texts[0] = texts[0].partition(b"\n")
        rows = [text.split(b"\n")[:-1] if text else [] for text in texts]  # (every row ends with a newline)
        with open(path, "wb") as out:
            out.write(header + b"\n")
            for k in range(sum(len(r) for r in rows)):
                out.write(rows[k % self.world][k // self.world] + b"\n")
