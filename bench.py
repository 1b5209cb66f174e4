#!/usr/bin/env python3
"""bench.py -- throughput of the MI355X hot path on BASELINE.json's synthetic 10 M chimeric-read configuration.

A "step" is one pass of the hot path (mark_multimappers -> annotate -> read-level filter cascade -> find_fusions) over one
batch of synthetic chimeric fragments that is already resident in HBM; the batch comes from the deterministic generator
(tools/gen_synth.cpp, BAM records streamed through the host ingest).  With --gpus N every rank processes its own shard
(weak scaling: reads shard naturally by read).  Rank 0 prints ONE JSON line.

The CPU baseline is the oracle build of the UNMODIFIED reference (oracle/_ref/arriba_ref, 1 core, it is single-threaded by
design) timed on a bounded sample of the same workload; if that binary did not travel, the field says so.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def workload_args(fragments, seed, read_seed=0):
    # 2x100 bp, 55 % split-read triplets / 35 % discordant pairs / 10 % read-through, Zipf junction
    # support, 30 % PCR duplicates, synthetic 24-contig genome with GENCODE-like annotation (no hg38 offline)
    return ["--seed", str(seed), "--read-seed", str(read_seed), "--fragments", str(fragments), "--normal-mult", "0", "--contigs", "24", "--contig-len", "12000000", "--genes-per-mb", "20",
            "--junctions", str(max(1000, fragments // 50))]


def generate_and_ingest(fragments, seed, directory, read_seed=0):
    """Writes the reference (FASTA/GTF) to `directory`, streams the BAM records through a pipe into the host ingest."""
    from arriba_amd.pipeline import HostSession
    import datasets
    prefix = os.path.join(directory, "bench")
    subprocess.run([datasets.GEN_SYNTH, "--out", prefix, "--reference-only"] + workload_args(fragments, seed, read_seed), check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    session = HostSession(prefix + ".fa", prefix + ".gtf")
    # ARRIBA_BENCH_CACHE=<directory>: the ingested batch is kept as a file, so that the profiling passes of one GPU session (each a new process
    # of this script) do not generate and parse the same 10 M fragments again (~1 minute each)
    cache = os.environ.get("ARRIBA_BENCH_CACHE")
    cache_file = os.path.join(cache, "ingest_%d_%d_%d.bin" % (fragments, seed, read_seed)) if cache else None
    if cache_file and os.path.exists(cache_file):
        session.load_ingest(cache_file)
        return session, prefix, None
    fifo = prefix + ".bam.fifo"
    os.mkfifo(fifo)
    producer = subprocess.Popen([datasets.GEN_SYNTH, "--out", prefix, "--raw-bam-to", fifo] + workload_args(fragments, seed, read_seed), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    started = time.time()
    session.read_chimeric_alignments(fifo)
    producer.wait()
    elapsed = time.time() - started
    if cache_file:
        os.makedirs(cache, exist_ok=True)
        session.save_ingest(cache_file)
    return session, prefix, elapsed


def cpu_baseline(seed, directory):
    """The unmodified reference (oracle/_ref/arriba_ref) on a bounded sample of the same workload, 1 core."""
    import datasets
    if not os.path.exists(datasets.ARRIBA_REF):
        return {"value": None, "unit": "chimeric reads/s", "cores": 1, "kind": "reference", "sample": "oracle/_ref/arriba_ref not present on this box"}
    sample_fragments = 200000
    prefix = os.path.join(directory, "cpu")
    subprocess.run([datasets.GEN_SYNTH, "--out", prefix] + workload_args(sample_fragments, seed), check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    started = time.time()
    result = subprocess.run([datasets.ARRIBA_REF, "-x", prefix + ".bam", "-g", prefix + ".gtf", "-a", prefix + ".fa", "-o", prefix + ".fusions.tsv", "-f", "blacklist"],
                            stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True)
    elapsed = time.time() - started
    if result.returncode != 0:
        return {"value": None, "unit": "chimeric reads/s", "cores": 1, "kind": "reference", "sample": "reference failed: " + result.stdout[-200:]}
    import re
    total = re.search(r"Reading chimeric alignments from .*\(total=(\d+)\)", result.stdout.replace("\n", " "))
    chimeric = int(total.group(1)) if total else sample_fragments
    # the host ingest of this repository on the same file (BAM records -> SoA batch; reader + worker threads, arriba_amd/csrc/host/ingest.cpp)
    from arriba_amd.pipeline import HostSession
    session = HostSession(prefix + ".fa", prefix + ".gtf")
    ingest_started = time.time()
    session.read_chimeric_alignments(prefix + ".bam")
    ingest_elapsed = time.time() - ingest_started
    return {"value": chimeric / elapsed, "unit": "chimeric reads/s", "cores": 1, "kind": "reference",
            "sample": "%d chimeric fragments of the same synthetic workload, whole reference binary BAM->fusions.tsv, %.1f s wall" % (chimeric, elapsed),
            "host_ingest_same_sample": {"value": session.fragment_count / ingest_elapsed, "unit": "chimeric reads/s", "threads": "1 reader + min(16, cores - 1) workers", "seconds": round(ingest_elapsed, 2)}}


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--gpus", type=int, default=1)
    parser.add_argument("--steps", type=int, default=3)
    parser.add_argument("--warmup", type=int, default=1)
    parser.add_argument("--fragments", type=int, default=10000000, help="chimeric fragments per GPU (BASELINE.json config 2: 10 M)")
    parser.add_argument("--no-cpu-baseline", action="store_true")
    parser.add_argument("--workflow", action="store_true", help="after the timed steps, also run the whole workflow once (every candidate-level filter, assign_confidence, both output files) and report its wall time; 1 GPU only")
    args = parser.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    distributed = world > 1
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    local_rank %= torch.cuda.device_count()  # (several ranks share a device only in the single-GPU dry run of the N > 1 path)
    torch.cuda.set_device(local_rank)
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # nccl == RCCL over xGMI; ARRIBA_BENCH_BACKEND=gloo is the dry run of the N > 1 path on a box with fewer GPUs than ranks
        dist.init_process_group(backend=os.environ.get("ARRIBA_BENCH_BACKEND", "nccl"))

    import __graft_entry__
    if rank == 0:
        __graft_entry__.build()
    if distributed:
        dist.barrier()
    from arriba_amd.pipeline import DevicePipeline

    directory = tempfile.mkdtemp(prefix="bench_r%d_" % rank)
    # every rank draws its own reads from the same genome and annotation: the shards of one sample
    session, prefix, ingest_seconds = generate_and_ingest(args.fragments, 1000, directory, read_seed=0 if world == 1 else 7000 + rank)
    if distributed:
        # one shard per rank; the sample is the concatenation of the shards in rank order (arriba_amd/sharded.py: four exchanges over RCCL)
        from arriba_amd.sharded import ShardedPipeline
        pipeline = ShardedPipeline(session, 0, session.fragment_count, device=local_rank, independent_sessions=True)
    else:
        pipeline = DevicePipeline(session, device=local_rank)
    n = pipeline.n

    step_stats = {}

    def step():
        pipeline.reset()
        pipeline.run_read_level()
        pipeline.find_fusions()
        pipeline.merge_adjacent_fusions()      # clusters live inside one gene pair: with shards, every owner merges its own candidates
        if distributed:
            step_stats.update(pipeline.fusion_stats())
            pipeline.replicate_candidates()    # all-gather of the owners' candidate columns: the candidate-level stages run replicated
        pipeline.filter_multimappers()         # best alignment of every multi-mapping read; with shards: two all-gathers + two all-reduce MIN
        pipeline.estimate_expected_fusions()   # includes the device computation of the reference container's iteration order (hazard H2)
        pipeline.filter_candidate_predicates() # non_coding_neighbors, intragenic_exonic, min_support (source/arriba.cpp:437-455)
        pipeline.filter_relative_support()

    for _ in range(args.warmup):
        step()
    stage_ms = {}
    pipeline.set_profiling(True)  # HIP events around every kernel launch, recorded on the launch stream
    pipeline.wall_ms.clear()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    started = time.time()
    for _ in range(args.steps):
        step()
        for stage, timing in pipeline.timings.items():
            stage_ms.setdefault(stage, []).append(timing)
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    elapsed = time.time() - started
    if distributed:
        collective_device = "cuda" if dist.get_backend() == "nccl" else "cpu"
        tensor = torch.tensor([elapsed], device=collective_device, dtype=torch.float64)
        dist.all_reduce(tensor, op=dist.ReduceOp.MAX)
        elapsed = float(tensor.item())
        counts = torch.tensor([n], device=collective_device, dtype=torch.int64)
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
        total_fragments = int(counts.item())
    else:
        total_fragments = n

    # post-conditions of the last step on the full-size batch (untimed; no oracle involved): a kernel that silently skipped a part of the
    # batch would leave alignments without a gene or a read filter count that does not add up
    import numpy as np
    self_check = []
    for slot in (0, 1):
        without_gene = int((pipeline.gene_sets(slot)[0] == 0).sum())
        if without_gene:
            self_check.append("%d alignments in slot %d have no gene after annotate" % (without_gene, slot))
    unfiltered = int((pipeline.filters() == 0).sum())
    remaining_local = pipeline.remaining_local["low_entropy"] if hasattr(pipeline, "remaining_local") else pipeline.remaining["low_entropy"]
    multimapper_discards = int((pipeline.filters() == 9).sum())
    if unfiltered + multimapper_discards != remaining_local:
        self_check.append("fragments without a filter (%d) + discarded as multi-mappers (%d) != remaining after the read filters (%d)" % (unfiltered, multimapper_discards, remaining_local))
    if self_check:
        raise SystemExit("bench self-check failed: " + "; ".join(self_check))

    if rank == 0:
        per_stage = {stage: {"ms": sum(t["ms"] for t in ts) / len(ts), "bytes": ts[-1]["bytes"]} for stage, ts in stage_ms.items()}
        # per-kernel launch durations of the timed steps (HIP events on the launch stream); the dominant kernel is the one with the largest total
        kernels = {}
        for name, ms, size in pipeline.kernel_profile():
            entry = kernels.setdefault(name, {"launches": 0, "ms": 0.0, "bytes": 0})
            entry["launches"] += 1
            entry["ms"] += ms
            entry["bytes"] += size
        dominant = max(kernels, key=lambda name: kernels[name]["ms"])
        launches = kernels[dominant]["launches"]
        dominant_ms = kernels[dominant]["ms"] / launches
        dominant_bytes = kernels[dominant]["bytes"] / launches
        achieved = dominant_bytes / (dominant_ms * 1e-3) / 1e9 if dominant_ms > 0 else 0.0
        # HBM traffic of the dominant kernel from the committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE are collected in separate
        # runs, tools/gpu_round.sh; KB per kernel over all dispatches).  Calibration in the same passes on the runtime's copy kernel with a
        # known byte count: FETCH_SIZE reports half of the bytes read (as MI355X_MICROARCH.md states for wide loads), WRITE_SIZE all of them.
        traffic = None
        pmc_path = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc_path):
            pmc = json.load(open(pmc_path))
            entry = pmc.get("kernels", {}).get(dominant.split("(")[0])
            if entry and entry.get("dispatches") and pmc.get("fragments") == n:
                traffic = (2.0 * entry.get("FETCH_SIZE", 0.0) + entry.get("WRITE_SIZE", 0.0)) * 1024.0 / entry["dispatches"]
        cascade_bytes = sum(values["bytes"] for values in per_stage.values())
        cascade_ms = sum(values["ms"] for values in per_stage.values())
        stats = step_stats if distributed else pipeline.fusion_stats()
        line = {
            "metric": "chimeric reads/s end-to-end (BAM->fusions.tsv), synthetic, device hot path with inputs resident in HBM",
            "value": total_fragments * args.steps / elapsed,
            "unit": "chimeric reads/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": "synthetic %d chimeric fragments per GPU (2x100 bp, 24-contig synthetic genome, GENCODE-like GTF), default filters" % args.fragments,
                       "fragments_per_gpu": n, "candidates": pipeline.n_candidates, "parallelism": ("%d shards by read: all-gather of unmapped positions, duplicate winners and mate-gap samples, all-to-all of gene-pair emissions, all-gather of candidate columns (RCCL)" % world) if distributed else "1 GPU", "gene_pair_emissions": stats["emissions"], "read_list_entries": stats["list_entries"],
                       "stages_timed": "mark_multimappers, annotate, read filters (14), fragment-length samples, find_fusions, merge_adjacent_fusions, filter_multimappers, fusions_t iteration order, estimate_expected_fusions, filter_non_coding_neighbors, filter_intragenic_both_exonic, filter_min_support, filter_relative_support",
                       "generate_and_ingest_reads_per_s": (n / ingest_seconds) if ingest_seconds else "batch loaded from ARRIBA_BENCH_CACHE"},
            "stage_ms": {stage: round(values["ms"], 3) for stage, values in per_stage.items()},
            "stage_wall_ms": {stage: round(value / args.steps, 3) for stage, value in pipeline.wall_ms.items()},
            "kernel_ms": {name: round(values["ms"] / values["launches"], 3) for name, values in sorted(kernels.items(), key=lambda item: -item[1]["ms"])[:12]},
            "roofline": {"bound": "hbm", "kernel": dominant, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "launch_ms": dominant_ms, "algorithmic_bytes_per_launch": dominant_bytes,
                         "cascade": {"algorithmic_bytes": cascade_bytes, "kernel_ms": cascade_ms, "achieved": cascade_bytes / (cascade_ms * 1e-3) / 1e9, "frac": cascade_bytes / (cascade_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}},
        }
        line["self_check"] = "every alignment has a gene; unfiltered + multi-mapper discards == remaining after the read filters"
        line["cpu_baseline"] = {"value": None, "unit": "chimeric reads/s", "cores": 1, "kind": "reference", "sample": "skipped"} if args.no_cpu_baseline else cpu_baseline(1000, directory)
        if args.workflow and not distributed:
            # untimed extra (not part of `value`): the reference's main() behind the ingest, to the two output files, once over the same batch
            try:
                remaining = []
                pipeline.reset()
                torch.cuda.synchronize()
                workflow_started = time.time()
                pipeline.run_workflow(os.path.join(directory, "workflow.fusions.tsv"), os.path.join(directory, "workflow.discarded.tsv"), log=lambda stage, count: remaining.append((stage, count, round(time.time() - workflow_started, 3))))
                torch.cuda.synchronize()
                line["workflow"] = {"seconds": time.time() - workflow_started, "stages": remaining, "fusions": remaining[-1][1] if remaining else None}
            except Exception as error:  # the candidate-level stages must not cost the bench line
                line["workflow"] = {"error": str(error)}
        print(json.dumps(line))
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
