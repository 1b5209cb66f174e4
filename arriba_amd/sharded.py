"""One process per GPU, each holding a contiguous range of the chimeric fragments in name order.

The per-fragment stages run on the shard through the same C ABI as the single-GPU path.  The reference's results depend on
fragments of other shards in exactly four places, and these are the exchanges made here with ``torch.distributed``
(backend ``nccl`` == RCCL over xGMI on the GPUs, ``gloo`` in the CPU tests):

  unmapped positions   all-gather   -> the dummy genes are cut from the sorted positions of the whole sample (source/arriba.cpp:207-260)
  duplicate winners    all-gather   -> the first fragment of a key in the name order of the whole sample survives (source/filter_duplicates.cpp)
  mate-gap samples     all-gather   -> the first 100001 samples in name order + the sequential float read-length sum (source/read_stats.cpp:11-92)
  gene-pair emissions  all-to-all   -> the owner of a gene pair builds all its candidates (source/fusions.cpp:203-473)

followed by one all-gather of the candidate tables; sorting them by first occurrence gives the reference's insertion order.
Buffers handed to the collectives are torch tensors on the device of the backend; the C ABI copies into / out of them directly.
"""
import ctypes
from ctypes import byref, c_float, c_int32, c_uint32, c_uint64

import numpy as np
import torch
import torch.distributed as dist

from . import _capi
from .pipeline import ArribaError, DevicePipeline

EMISSION_BYTES = 36
DUPLICATE_ENTRY_BYTES = 16
MAX_SAMPLES = 100001


def shard_ranges(session, world):
    """Cuts [0, n) into `world` contiguous ranges that keep the fragments of one read name together."""
    n = session.fragment_count
    cuts = [0]
    for r in range(1, world):
        cuts.append(max(cuts[-1], int(session._lib.ahost_shard_boundary(session._session, (n * r) // world))))
    cuts.append(n)
    return [(cuts[r], cuts[r + 1] - cuts[r]) for r in range(world)]


class ShardedPipeline(DevicePipeline):
    """DevicePipeline over the fragments [first, first + count) of `session`; the sample is the concatenation of the shards of all ranks."""

    def __init__(self, session, first, count, params=None, api=None, device=0, group=None, independent_sessions=False):
        """independent_sessions: every rank ingested only its own shard (its host session knows nothing about the other shards), as opposed to
        every rank holding the host session of the whole sample and driving a slice of it"""
        self.group = group
        self.independent_sessions = independent_sessions
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.first_local, self.count_local = first, count
        view = session._lib.ahost_batch_slice_view(session._session, first, count)
        if not view:
            raise ArribaError("ERROR: " + session._lib.ahost_last_error().decode())
        super().__init__(session, params=params, api=api, device=device, batch_view=view)
        self.collective_device = torch.device("cuda", device) if dist.get_backend(group) == "nccl" else torch.device("cpu")
        counts = self._all_gather_int(count)
        self.shard_counts = counts
        self.first_rank = int(sum(counts[:self.rank]))
        self.global_n = int(sum(counts))
        self._check(self.api.set_shard(self.ctx, self.first_rank, self.global_n))

    # ---- collectives over raw bytes --------------------------------------------------------------------

    def _sync(self):
        """The C ABI works on its own HIP stream: make the collective's result (and torch's fills before it) visible to it."""
        if self.collective_device.type == "cuda":
            torch.cuda.synchronize(self.collective_device)

    def _all_gather_int(self, value):
        tensor = torch.tensor([int(value)], dtype=torch.int64, device=self.collective_device)
        out = [torch.zeros_like(tensor) for _ in range(self.world)]
        dist.all_gather(out, tensor, group=self.group)
        return [int(t.item()) for t in out]

    def _all_gather_tensor(self, mine):
        """all-gather of a variable number of bytes per rank (uint8 tensor on the collective device) -> (all bytes in rank order, sizes)"""
        sizes = self._all_gather_int(mine.numel())
        width = max(max(sizes), 1)
        padded = torch.zeros(width, dtype=torch.uint8, device=self.collective_device)
        padded[:mine.numel()] = mine
        parts = [torch.zeros(width, dtype=torch.uint8, device=self.collective_device) for _ in range(self.world)]
        dist.all_gather(parts, padded, group=self.group)
        self._sync()
        if sum(sizes) == 0:
            return torch.zeros(0, dtype=torch.uint8, device=self.collective_device), sizes
        return torch.cat([parts[r][:sizes[r]] for r in range(self.world)]), sizes

    def _all_gather_bytes(self, n_bytes, fill):
        """`fill(pointer)` writes this rank's n_bytes straight into the collective's buffer (host or device memory)"""
        mine = torch.zeros(max(n_bytes, 1), dtype=torch.uint8, device=self.collective_device)
        self._sync()
        if n_bytes:
            fill(mine.data_ptr())
        return self._all_gather_tensor(mine[:n_bytes])

    def _all_gather_array(self, array):
        array = np.ascontiguousarray(array)
        raw = torch.from_numpy(array.view(np.uint8).reshape(-1).copy()).to(self.collective_device)
        gathered, sizes = self._all_gather_tensor(raw)
        return np.frombuffer(gathered.cpu().numpy().tobytes(), dtype=array.dtype), [size // array.dtype.itemsize for size in sizes]

    def _broadcast_float_chain(self, update):
        """running = update(rank, running) evaluated for rank 0, 1, ... in order (a sequential float accumulation over the shards)"""
        running = torch.zeros(1, dtype=torch.float32, device=self.collective_device)
        for r in range(self.world):
            if r == self.rank:
                running[0] = update(float(running.item()))
            dist.broadcast(running, src=dist.get_global_rank(self.group, r) if self.group is not None else r, group=self.group)
        return float(running.item())

    # ---- stages with an exchange -------------------------------------------------------------------------

    def annotate_alignments(self, strandedness=None):
        if strandedness is None:  # the vote looks at the first fragments of the sample in name order: rank 0 decides
            if getattr(self, "independent_sessions", False):
                # every rank ingested its own shard: rank 0's session sees only the first shard, which may hold fewer than the 100 informative split reads the
                # reference's vote wants while the whole sample has them (source/read_stats.cpp:94-143).  The vote needs the sample: ask the caller.
                raise ArribaError("ERROR: strandedness must be given (yes/no/reverse) when every rank ingests its own shard: the automatic vote looks at the first fragments of the whole sample")
            value = torch.tensor([self.session.detect_strandedness() if self.rank == 0 else 0], dtype=torch.int64, device=self.collective_device)
            dist.broadcast(value, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
            strandedness = int(value.item())
        self.scalars["strandedness"] = strandedness
        self.params.strandedness = strandedness
        self._check(self.api.set_params(self.ctx, byref(self.params)))
        unmapped = c_uint64()
        self._check(self.api.annotate_begin(self.ctx, byref(unmapped)))
        positions, _ = self._all_gather_bytes(unmapped.value * 8, lambda pointer: self._check(self.api.copy_unmapped_positions(self.ctx, pointer)))
        n_dummy = c_uint32()
        self._check(self.api.annotate_finish(self.ctx, positions.data_ptr() if positions.numel() else None, positions.numel() // 8, byref(n_dummy)))
        self._record("annotate")
        self.n_dummy_genes = n_dummy.value
        return self.n_dummy_genes

    def filter_duplicates_and_contigs(self, top_viral_contigs=5, viral_contig_min_covered_fraction=0.05):
        count = c_uint64()
        self._check(self.api.get_viral_integration_sites(self.ctx, None, 0, byref(count)))
        pairs = np.zeros(2 * max(count.value, 1), dtype=np.uint32)
        self._check(self.api.get_viral_integration_sites(self.ctx, pairs.ctypes.data, count.value, byref(count)))
        all_pairs, _ = self._all_gather_array(pairs[:2 * count.value])
        top, low = self.session.viral_verdicts(all_pairs, self.gene_table()["bits"], top_viral_contigs, viral_contig_min_covered_fraction)
        entries = c_uint64()
        self._check(self.api.duplicates_begin(self.ctx, byref(entries)))
        gathered, _ = self._all_gather_bytes(entries.value * DUPLICATE_ENTRY_BYTES, lambda pointer: self._check(self.api.copy_duplicate_entries(self.ctx, pointer)))
        self._check(self.api.read_filters_stage1_global(self.ctx, gathered.data_ptr() if gathered.numel() else None, gathered.numel() // DUPLICATE_ENTRY_BYTES, top.ctypes.data, low.ctypes.data))
        self._record("read_filters_stage1")

    def estimate_fragment_length(self):
        """The reference collects the mate gaps of the first 100001 qualifying fragments in name order and sums the read lengths of every
        fragment it visits on the way, sequentially in float (source/read_stats.cpp:11-44): shard by shard here."""
        gaps = np.zeros(MAX_SAMPLES, dtype=np.int32)
        n_samples, visited = c_uint32(), c_uint64()
        self._check(self.api.fragment_length_samples_limited(self.ctx, MAX_SAMPLES, gaps.ctypes.data, byref(n_samples), byref(visited)))
        counts = self._all_gather_int(n_samples.value)
        before = sum(counts[:self.rank])
        if before >= MAX_SAMPLES:            # the loop stopped in an earlier shard
            wanted, visited_local = 0, 0
        elif n_samples.value >= MAX_SAMPLES - before:  # the loop stops in this shard, right behind the fragment that delivers sample number 100001
            wanted = MAX_SAMPLES - before
            if wanted < MAX_SAMPLES:
                self._check(self.api.fragment_length_samples_limited(self.ctx, wanted, gaps.ctypes.data, byref(n_samples), byref(visited)))
            visited_local = visited.value
        else:                                # the whole shard is visited
            wanted, visited_local = n_samples.value, self.count_local
        self._record("fragment_length_samples")
        all_gaps, _ = self._all_gather_array(gaps[:wanted])
        visited_by_shard = self._all_gather_int(visited_local)
        read_length_sum = self._broadcast_float_chain(lambda running: float(self.session._lib.ahost_read_length_sum(self.session._session, running, self.first_local, visited_by_shard[self.rank])))
        mean, stddev, read_length = c_float(), c_float(), c_float()
        max_mate_gap = c_int32()
        all_gaps = np.ascontiguousarray(all_gaps, dtype=np.int32)
        estimated = self.session._lib.ahost_estimate_fragment_length_from_sums(all_gaps.ctypes.data, all_gaps.size, read_length_sum, sum(visited_by_shard), self.params.fragment_length,
                                                                               byref(mean), byref(stddev), byref(read_length), byref(max_mate_gap))
        estimate = {"estimated": bool(estimated), "mate_gap_mean": mean.value, "mate_gap_stddev": stddev.value, "read_length_mean": read_length.value, "max_mate_gap": max_mate_gap.value}
        self.scalars.update(estimate)
        self.scalars["mate_gap_samples"] = int(all_gaps.size)
        return estimate

    def filter_reads(self):
        remaining = super().filter_reads()  # counts of this shard; the sample's "(remaining=N)" is their sum
        self.remaining_local = dict(remaining)
        names = list(remaining)
        totals = torch.tensor([remaining[k] for k in names], dtype=torch.int64, device=self.collective_device)
        dist.all_reduce(totals, group=self.group)
        self.remaining = {k: int(v) for k, v in zip(names, totals.tolist())}
        return self.remaining

    def mark_multimappers(self):
        marked = super().mark_multimappers()  # shards are cut between read names, so no pair of neighbours is separated
        total = torch.tensor([marked], dtype=torch.int64, device=self.collective_device)
        dist.all_reduce(total, group=self.group)
        return int(total.item())

    def find_fusions(self, max_mate_gap=None):
        """emissions -> all-to-all by gene-pair owner -> candidates of the gene pairs this rank owns"""
        if max_mate_gap is None:
            max_mate_gap = self.scalars["max_mate_gap"]
        counts = np.zeros(self.world, dtype=np.uint64)
        self._check(self.api.build_emissions(self.ctx, self.world, counts.ctypes.data))
        send_counts = [int(c) for c in counts]
        send = torch.zeros(max(sum(send_counts) * EMISSION_BYTES, 1), dtype=torch.uint8, device=self.collective_device)
        self._sync()
        if sum(send_counts):
            self._check(self.api.copy_emissions(self.ctx, send.data_ptr()))
        # how much every rank sends to every rank
        matrix = torch.tensor(send_counts, dtype=torch.int64, device=self.collective_device)
        rows = [torch.zeros_like(matrix) for _ in range(self.world)]
        dist.all_gather(rows, matrix, group=self.group)
        receive_counts = [int(rows[source][self.rank].item()) for source in range(self.world)]
        received = torch.zeros(max(sum(receive_counts) * EMISSION_BYTES, 1), dtype=torch.uint8, device=self.collective_device)
        dist.all_to_all_single(received[:sum(receive_counts) * EMISSION_BYTES], send[:sum(send_counts) * EMISSION_BYTES],
                               [c * EMISSION_BYTES for c in receive_counts], [c * EMISSION_BYTES for c in send_counts], group=self.group)
        self._sync()
        count = c_uint64()
        self._check(self.api.find_fusions_from_emissions(self.ctx, received.data_ptr() if sum(receive_counts) else None, sum(receive_counts), max_mate_gap, byref(count)))
        self._record("find_fusions")
        self.n_candidates = count.value
        self.exchange = {"emissions_sent": sum(send_counts), "emissions_received": sum(receive_counts)}
        return self.n_candidates

    def replicate_candidates(self):
        """all-gather of the owners' candidate columns and import in the reference's insertion order (ascending first occurrence), so that the
        candidate-level stages (iteration order, e-value, relative support) run replicated on every rank.  Everything stays on the collective
        device: the C ABI copies the columns straight into / out of the collective's buffers.  The read lists stay with the owners."""
        n = self.n_candidates
        device = self.collective_device
        names = ("gene1", "gene2", "contigs", "breakpoint1", "breakpoint2", "flags", "filter", "split_reads1", "split_reads2", "discordant_mates", "anchor_start1", "anchor_start2")
        dtypes = {name: (torch.uint8 if name == "filter" else torch.int32) for name in names}  # 32-bit columns travel as int32 bit patterns
        local = {name: torch.zeros(max(n, 1), dtype=dtypes[name], device=device) for name in names}
        first = torch.zeros(max(n, 1), dtype=torch.int64, device=device)
        self._sync()
        if n:
            self._check(self.api.get_candidates(self.ctx, *[local[name].data_ptr() for name in names], None))
            self._check(self.api.get_candidate_first_occurrence(self.ctx, first.data_ptr()))
        counts = self._all_gather_int(n)
        width = max(max(counts), 1)

        def gather(tensor):
            padded = torch.zeros(width, dtype=tensor.dtype, device=device)
            padded[:n] = tensor[:n]
            parts = [torch.zeros(width, dtype=tensor.dtype, device=device) for _ in range(self.world)]
            dist.all_gather(parts, padded, group=self.group)
            return torch.cat([parts[r][:counts[r]] for r in range(self.world)])

        all_first = gather(first)
        order = torch.argsort(all_first, stable=True)  # (read rank << 8 | ordinal) is non-negative as int64
        columns = {name: gather(local[name])[order].contiguous() for name in names}
        total = int(all_first.numel())
        self._sync()
        self._check(self.api.import_candidates(self.ctx, total, *[columns[name].data_ptr() if total else None for name in names]))
        # where the candidates this rank built ended up: their read lists stay here (filter_multimappers needs them)
        inverse = torch.empty_like(order)
        inverse[order] = torch.arange(total, dtype=order.dtype, device=device)
        base = int(sum(counts[:self.rank]))
        owned = inverse[base:base + n].to(torch.int32).contiguous()
        self._sync()
        self._check(self.api.set_owned_candidates(self.ctx, owned.data_ptr() if n else None, n))
        self.n_owned_candidates = n
        self.n_candidates = total
        return total

    def filter_multimappers(self):
        """reference: filter_multimappers (source/filter_multimappers.cpp:109-221) over the shards; needs replicate_candidates() first.
        Exchanges: multi-mapper flags (all-gather), best candidate rank per multi-mapping read (all-reduce MIN), reads discarded
        (all-gather), candidate counters (all-reduce MIN).  Returns (remaining candidates, fragments discarded in the whole sample)."""
        device = self.collective_device
        flags, _ = self._all_gather_bytes(self.count_local, lambda pointer: self._check(self.api.copy_multimapper_flags(self.ctx, pointer)))
        n_multimappers = c_uint64()
        self._check(self.api.multimappers_begin(self.ctx, flags.data_ptr() if flags.numel() else None, byref(n_multimappers)))
        best = torch.full((max(n_multimappers.value, 1),), 0x7FFFFFFF, dtype=torch.int32, device=device)
        self._sync()
        self._check(self.api.multimappers_partial_best(self.ctx, best.data_ptr()))
        dist.all_reduce(best, op=dist.ReduceOp.MIN, group=self.group)
        self._sync()
        discarded_local = torch.zeros(max(self.count_local, 1), dtype=torch.uint8, device=device)
        discarded = c_uint64()
        self._sync()
        self._check(self.api.multimappers_resolve(self.ctx, best.data_ptr(), discarded_local.data_ptr(), byref(discarded)))
        discarded_global, _ = self._all_gather_tensor(discarded_local[:self.count_local])
        counters = torch.zeros(max(3 * self.n_candidates, 1), dtype=torch.int32, device=device)
        self._sync()
        self._check(self.api.multimappers_recount(self.ctx, discarded_global.data_ptr() if discarded_global.numel() else None, counters.data_ptr()))
        dist.all_reduce(counters, op=dist.ReduceOp.MIN, group=self.group)
        self._sync()
        remaining = c_uint64()
        self._check(self.api.multimappers_finish(self.ctx, counters.data_ptr(), byref(remaining)))
        self._record("filter_multimappers")
        total = torch.tensor([discarded.value], dtype=torch.int64, device=device)
        dist.all_reduce(total, group=self.group)
        return remaining.value, int(total.item())

    def estimate_expected_fusions(self, mapped_reads=None, iteration_rank=None):
        if mapped_reads is None and self.independent_sessions:  # mapped reads of the whole sample
            total = torch.tensor([self.session.mapped_reads], dtype=torch.int64, device=self.collective_device)
            dist.all_reduce(total, group=self.group)
            mapped_reads = int(total.item())
        return super().estimate_expected_fusions(mapped_reads, iteration_rank)

    def first_occurrence(self):
        out = np.zeros(max(self.n_candidates, 1), dtype=np.uint64)
        self._check(self.api.get_candidate_first_occurrence(self.ctx, out.ctypes.data))
        return out[:self.n_candidates]

    def gather_candidates(self):
        """The candidate table of the whole sample in the reference's insertion order (on every rank): all-gather of the owners' tables,
        sorted by first occurrence.  Read lists hold global name ranks."""
        table = self.candidates()
        first = self.first_occurrence()
        merged = {}
        all_first, sizes = self._all_gather_array(first)
        order = np.argsort(all_first, kind="stable")
        for key in ("gene1", "gene2", "contigs", "breakpoint1", "breakpoint2", "flags", "filter", "split_reads1", "split_reads2", "discordant_mates", "anchor_start1", "anchor_start2"):
            column, _ = self._all_gather_array(table[key])
            merged[key] = column[order]
        # read lists: per candidate three lists; gather the sizes and the concatenated entries
        offsets = table["list_offset"].astype(np.int64)
        list_sizes = (offsets[1:] - offsets[:-1]).astype(np.uint32) if self.n_candidates else np.zeros(0, dtype=np.uint32)
        all_sizes, _ = self._all_gather_array(list_sizes)
        all_lists, _ = self._all_gather_array(table["read_lists"].astype(np.uint32))
        all_sizes = all_sizes.reshape(-1, 3).astype(np.int64)
        starts = np.concatenate([[0], np.cumsum(all_sizes.sum(axis=1))])[:-1]
        chunks, new_sizes = [], []
        for c in order:
            chunks.append(all_lists[starts[c]:starts[c] + all_sizes[c].sum()])
            new_sizes.extend(all_sizes[c])
        merged["read_lists"] = np.concatenate(chunks) if chunks else np.zeros(0, dtype=np.uint32)
        merged["list_offset"] = np.concatenate([[0], np.cumsum(np.array(new_sizes, dtype=np.int64))]).astype(np.uint64)
        merged["first_occurrence"] = all_first[order]
        return merged
