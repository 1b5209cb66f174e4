#!/usr/bin/env python3
"""What the read-sharded split of one sample over N GPUs can be expected to take, from the MEASURED phases of a one-GPU line of bench.py:
    python tools/project_scaling.py profiles/r06b_bench100m.json [N ...]

No multi-GPU box was available to any round: this is a projection, and it says which of its terms are measurements (every kernel's own time in the sample that ran alone:
`kernel_ms_alone`; the host phases of that sample: `one_sample_alone`) and which are assumptions (the rates of the exchanges over xGMI, how the host's file reads scale).
A kernel is priced by where its work lives in the split:
  reads      it runs over the fragments (or records, or re-alignment jobs) a rank holds: time / N
  candidates it runs over the candidate table, the emissions of all ranks or the read lists, which every rank holds whole: time x 1
The line of the driver's 8-GPU box, when one runs `bench.py --gpus 8`, is the measurement this stands in for."""
import json
import sys

# kernels whose work is the reads of a rank (prefixes); everything else counts as replicated
BY_READS = ("bgzf_", "segment_", "record_parse", "group_", "run_", "name_", "fragment_", "qname_", "window_", "coverage_", "mark_multimappers", "annotate_stage", "dummy_", "stage1_kernel", "stage2_kernel",
            "duplicate_", "low_entropy", "sample_", "select_fragments", "emission_", "rocprim::exclusive_scan(emission", "select_multimappers", "multimapper_score", "multimapper_group", "mismapper_heavy", "mismapper_verdict",
            "clip_summary", "in_vitro_kernel", "in_vitro_wave", "in_vitro_partial", "gene_read_count", "gather_", "read_state_export", "strandedness_", "packed_name", "iota", "shard_")
XGMI_ALL_GATHER_GB_PER_S = 300.0   # assumption: algorithm bandwidth of a ring all-gather over 7 x 153 GB/s links (MI355X_MICROARCH.md); the exchanges are 1-5 GB
HOST_READ_GB_PER_S = 180.0         # assumption: what N readers get from the page cache of one host together (~180 GB/s at N = 8)


def main():
    line = json.loads([text for text in open(sys.argv[1]).read().splitlines() if text.startswith("{") and '"metric"' in text][-1])
    ranks = [int(value) for value in sys.argv[2:]] or [2, 4, 8]
    kernels = line["kernel_ms_alone"]
    by_reads = sum(ms for name, ms in kernels.items() if name.startswith(BY_READS))
    replicated = sum(ms for name, ms in kernels.items() if not name.startswith(BY_READS))
    other = line["kernel_ms_alone_sum"] - by_reads - replicated  # (the kernels behind the 48 the line lists)
    alone = line["one_sample_alone"]["parts"]
    fragments = line["config"]["fragments_per_gpu"]
    bam_gb = line["bam_GB_per_s_end_to_end"] * line["seconds_per_step"]["total"]
    emissions_gb = 36e-9 * 1.3 * fragments            # 36 bytes per read x gene pair, ~1.3 emissions per fragment of this workload (agpu_get_fusion_stats at 10^8: 1.3e8)
    states_gb = 3 * 1e-9 * fragments                   # one byte per fragment, three times per sample
    duplicates_gb = 16e-9 * 0.7 * fragments            # the winners of the duplicate keys: 16 bytes, ~0.7 distinct keys per fragment (30 % exact duplicates)
    report = {"source": sys.argv[1], "fragments": fragments, "measured": {"kernel_ms_alone_sum": line["kernel_ms_alone_sum"], "kernels_over_the_reads_of_a_rank_ms": round(by_reads, 1),
              "kernels_over_candidates_and_lists_ms": round(replicated, 1), "kernels_not_listed_ms": round(other, 1), "one_sample_alone": alone, "step_of_a_queue_s": line["ms_per_step"] / 1e3},
              "assumed": {"xgmi_all_gather_GB_per_s": XGMI_ALL_GATHER_GB_PER_S, "host_read_GB_per_s": HOST_READ_GB_PER_S, "emissions_GB": round(emissions_gb, 2), "read_states_GB": round(states_gb, 2), "duplicate_winners_GB": round(duplicates_gb, 2)},
              "projected": {}}
    # host phases of the sample that ran alone: feed (file -> HBM), what is left of the ingest, stages + filter_mismappers (device work, priced by the kernels below), output
    device_alone = (alone["stages"] + alone["filter_mismappers"] + alone["ingest"]) * 1e3
    host_in_stages = max(0.0, device_alone - (line["kernel_ms_alone_sum"] - sum(ms for name, ms in kernels.items() if name.startswith(("bgzf_", "segment_", "record_parse")))))  # launches, read-backs, the sequential host scalars
    for n in ranks:
        feed = max(alone["feed"] / n, bam_gb / HOST_READ_GB_PER_S)
        exchanges = (emissions_gb + states_gb + duplicates_gb) * (n - 1) / n / XGMI_ALL_GATHER_GB_PER_S
        device = (by_reads / n + replicated + other) / 1e3 + host_in_stages / 1e3
        output = alone["output_results"] + alone["output_rows"] / n + alone["output_format"] / n + 0.02  # (rows gathered from their ranks, every rank formats every n-th row; + the gather of the texts)
        latency = feed + device + exchanges + output
        # samples in a queue: the feed of the next sample runs beside the device work of the current one
        step = max(feed, device + exchanges + alone["output_results"] + alone["output_rows"] / n)
        report["projected"][str(n)] = {"latency_s": round(latency, 3), "speedup_of_one_sample": round(alone["total"] / latency, 2), "step_of_a_queue_s": round(step, 3), "speedup_of_the_queue": round(line["ms_per_step"] / 1e3 / step, 2),
                                       "parts_s": {"feed": round(feed, 3), "device_and_launches": round(device, 3), "exchanges": round(exchanges, 3), "output": round(output, 3)},
                                       "hbm_per_rank_GB": round(29 + 5 + 4.7 + 12 + (54 + 35 + 25) / n, 0)}
    report["reading"] = ("the kernels over candidates and read lists (%.0f ms: find_fusions from the emissions of all ranks, merge_adjacent_fusions, the list walks, select_best, both_spliced, homologs, ...) do not shrink with N: "
                         "they bound the speed-up at %.1fx however many GPUs; sharding THEM is a partition of the candidates by gene pair (arriba_amd/sharded.py has its exchange), not of the reads"
                         % (replicated, line["kernel_ms_alone_sum"] / (replicated + other)))
    report["hbm_per_rank_GB_is"] = "29 GB of working arrays of find_fusions over the emissions of ALL ranks (220 B per emission; the 2.7 GB of memo tables of filter_mismappers live in the same buffer later) + 5 GB of task lists (per rank, whatever N; 62 GB of tables and lists until round 6) + 4.7 GB of emissions + ~12 GB of candidates and read lists + (stream 54 + tables of the ingest 35 + batch and gene sets 25) / N"
    print(json.dumps(report, indent=1))


if __name__ == "__main__":
    main()
