"""Worker of the multi-process test (one rank): runs the sharded pipeline over its range of the fragments and, on rank 0, compares the
merged result with the single-process pipeline over the whole sample.  Launched by tests/test_sharded.py through torch.distributed.run."""
import ctypes
import json
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from arriba_amd import _capi  # noqa: E402
from arriba_amd.pipeline import DevicePipeline  # noqa: E402
from arriba_amd.sharded import ShardedPipeline, shard_ranges  # noqa: E402
import parity  # noqa: E402


def main():
    prefix, backend_api, out_path = sys.argv[1], sys.argv[2], sys.argv[3]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    api = _capi.bind_device_api(ctypes.CDLL(os.path.join(ROOT, "tests", "emu", "libemu.so")), "emu_") if backend_api == "emu" else None
    session = parity.open_session(prefix)
    first, count = shard_ranges(session, world)[rank]
    sharded = ShardedPipeline(session, first, count, api=api)
    remaining = sharded.run_read_level()
    sharded.find_fusions()
    merged = sharded.gather_candidates()
    local_filters = sharded.filters()
    merged_remaining = sharded.merge_adjacent_fusions()  # on the owners: a cluster of adjacent breakpoints lives inside one gene pair
    sharded.replicate_candidates()
    multimappers = sharded.filter_multimappers()
    filters_after_multimappers = sharded.filters()
    evalue = sharded.estimate_expected_fusions()
    predicates = sharded.filter_candidate_predicates()
    relative_support_remaining = sharded.filter_relative_support()
    candidate_filters = sharded.candidates()["filter"]
    report = {"rank": rank, "first": first, "count": count, "exchange": sharded.exchange, "multimappers": multimappers, "owned_candidates": sharded.n_owned_candidates}
    if rank == 0:
        whole = DevicePipeline(session, api=api)
        expected_remaining = whole.run_read_level()
        whole.find_fusions()
        table = whole.candidates()
        problems = []
        if remaining != expected_remaining:
            problems.append(("remaining", remaining, expected_remaining))
        for key in ("marked_multimappers", "strandedness", "max_mate_gap", "estimated", "mate_gap_mean", "mate_gap_stddev", "read_length_mean"):
            if sharded.scalars[key] != whole.scalars[key]:
                problems.append((key, sharded.scalars[key], whole.scalars[key]))
        if sharded.n_dummy_genes != whole.n_dummy_genes:
            problems.append(("dummy genes", sharded.n_dummy_genes, whole.n_dummy_genes))
        if not np.array_equal(local_filters, whole.filters()[first:first + count]):
            problems.append(("read filters of shard 0",))
        for key in ("gene1", "gene2", "contigs", "breakpoint1", "breakpoint2", "flags", "filter", "split_reads1", "split_reads2", "discordant_mates", "anchor_start1", "anchor_start2", "list_offset", "read_lists"):
            if not np.array_equal(np.asarray(merged[key], dtype=np.int64), np.asarray(table[key], dtype=np.int64)):
                problems.append(("candidates." + key, len(merged[key]), len(table[key])))
        whole.merge_adjacent_fusions()
        expected_multimappers = whole.filter_multimappers()
        if tuple(multimappers) != tuple(expected_multimappers):
            problems.append(("filter_multimappers", multimappers, expected_multimappers))
        if not np.array_equal(filters_after_multimappers, whole.filters()[first:first + count]):
            problems.append(("read filters of shard 0 after filter_multimappers",))
        expected_evalue = whole.estimate_expected_fusions()
        if not np.array_equal(evalue.view(np.uint32), expected_evalue.view(np.uint32)):
            problems.append(("e-values", int((evalue.view(np.uint32) != expected_evalue.view(np.uint32)).sum())))
        if predicates != whole.filter_candidate_predicates():
            problems.append(("candidate predicates", predicates))
        if relative_support_remaining != whole.filter_relative_support() or not np.array_equal(candidate_filters, whole.candidates()["filter"]):
            problems.append(("filter_relative_support",))
        report.update({"problems": problems, "candidates": int(whole.n_candidates), "fragments": int(whole.n)})
    with open("%s.rank%d.json" % (out_path, rank), "w") as out:
        json.dump(report, out, default=str)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
