#!/usr/bin/env python3
"""tools/stage_probe.py -- GPU-box probe: per-kernel times of the read-level cascade with individual filters switched off (-f), to see
which predicate the time of stage2_kernel goes to.  Prints one JSON object."""
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    fragments = int(sys.argv[1]) if len(sys.argv) > 1 else 3000000
    import bench
    from arriba_amd.pipeline import DevicePipeline
    directory = tempfile.mkdtemp(prefix="probe_")
    session, prefix, _ = bench.generate_and_ingest(fragments, 1000, directory)
    results = {}
    configurations = [("all", []), ("no_mismatches", ["mismatches"]), ("no_homopolymer", ["homopolymer"]), ("no_hairpin_long_gap", ["hairpin", "long_gap"]),
                      ("no_read_through_same_gene", ["read_through", "same_gene"]), ("only_mismatches", ["read_through", "inconsistently_clipped", "homopolymer", "small_insert_size", "long_gap", "same_gene", "hairpin"]),
                      ("none", ["read_through", "inconsistently_clipped", "homopolymer", "small_insert_size", "long_gap", "same_gene", "hairpin", "mismatches", "low_entropy"])]
    for name, disabled in configurations:
        pipeline = DevicePipeline(session, params={"disable_filters": disabled})
        pipeline.run_read_level()
        pipeline.reset()
        pipeline.set_profiling(True)
        pipeline.run_read_level()
        kernels = {}
        for kernel, ms, size in pipeline.kernel_profile():
            kernels[kernel] = round(kernels.get(kernel, 0.0) + ms, 3)
        results[name] = {"kernels": kernels, "remaining": pipeline.remaining}
        pipeline.close()
    # the candidate-level stages once, with per-kernel times
    pipeline = DevicePipeline(session)
    pipeline.run_read_level()
    pipeline.set_profiling(True)
    pipeline.find_fusions()
    remaining_after_merge = pipeline.merge_adjacent_fusions()
    remaining_after_multimappers, multimapper_reads = pipeline.filter_multimappers()
    pipeline.estimate_expected_fusions()
    remaining_after_evalue = pipeline.filter_relative_support()
    positions = pipeline.make_kmer_index()
    remaining, discarded = pipeline.filter_mismappers()
    kernels = {}
    for kernel, ms, size in pipeline.kernel_profile():
        kernels[kernel] = round(kernels.get(kernel, 0.0) + ms, 3)
    results["candidate_stages"] = {"kernels": kernels, "candidates": pipeline.n_candidates, "remaining_after_merge_adjacent": remaining_after_merge, "remaining_after_multimappers": remaining_after_multimappers, "reads_discarded_as_multimappers": multimapper_reads, "remaining_after_relative_support": remaining_after_evalue, "kmer_positions": positions,
                                   "remaining_after_mismappers": remaining, "reads_discarded_as_mismappers": discarded, "stage_ms": {k: round(v["ms"], 3) for k, v in pipeline.timings.items()}}
    print(json.dumps({"fragments": session.fragment_count, "results": results}))


if __name__ == "__main__":
    main()
