#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on one MI355X: chimeric reads/s END TO END, BAM file -> fusions.tsv.

A "step" is one whole run of the workflow over one sample -- at N = 1 one call of arriba_workflow_sample of libarriba_workflow.so, the C++ driver over the two C ABIs with a resident
session (--python-stages: the ctypes mirror of the stage order instead) -- (BASELINE.json config 2/3: synthetic chimeric-read BAM, 2x100 bp, uncompressed BGZF as STAR
--outBAMcompression 0 writes it, run_arriba.sh:34): the BAM file (resident in the page cache / tmpfs, as the reference would read it) is opened, its
header parsed, its bytes fed to the GPU, read_chimeric_alignments runs in HBM (agpu_ingest.hip), then every stage of the reference's main() in its order --
the read-level cascade, find_fusions, merge_adjacent, filter_multimappers, e-value, every candidate-level filter, make_kmer_index + filter_homologs +
filter_mismappers, recover_isoforms, assign_confidence -- and the output file fusions.tsv is written (-O discarded.tsv with --discarded).  `value` =
chimeric fragments of the sample (the reference's "(total=N)") / seconds per step.  Loading the assembly and the annotation (ahost_open) and creating the
device context happen once before the timed region, as in a resident service; the reference binary timed beside it pays for them inside its own time and
the line says how much that is.  The time from the resident batch to the end of filter_relative_support -- what round 1 reported -- is the secondary
field `device_resident_step`.

With --gpus N the N ranks work on ONE sample of the same size (BASELINE.json config 4; arriba_amd/one_sample.py): every rank ingests its part of the records of
the file, one all-gather puts the batch together on every GPU, the stages up to filter_homologs run on every rank, the re-alignments of filter_mismappers are shared out
with one all-reduce of the verdicts, rank 0 writes the files (with the one-GPU step of round 3 the replicated stages cap this at ~1.7x for N = 8: DESIGN.md section 6); value = fragments of the sample / the slowest rank's
time ("scaling": "strong").  --per-rank-samples gives every rank a sample of its own instead (no collective on the data path; "weak").  Rank 0 prints ONE JSON line.

The CPU baseline is the oracle build of the UNMODIFIED reference (oracle/_ref/arriba_ref, single-threaded by design) on a bounded sample of the same workload.
--host-only: no GPU needed -- generates a small sample, runs the file side of the device ingest (BamFeed) and the classic host ingest from a named pipe and
from a file, prints their rates; run it in the CPU container before committing a change under arriba_amd/csrc/host/.
"""
import argparse
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


BENCH_STARTED = time.time()


def progress(text):
    """where the time goes, on stderr (stdout carries the one JSON line): a run that is cut off by a time limit still says how far it got"""
    sys.stderr.write("[bench %7.1f s] %s\n" % (time.time() - BENCH_STARTED, text))
    sys.stderr.flush()


NAME_LENGTH = [0]  # --name-length: read names padded to the length of an Illumina run's (45 characters: 4.7 GB of names at 10^8 fragments)


def workload_args(fragments, seed, read_seed=0, stress=False):
    # 2x100 bp, 55 % split-read triplets / 35 % discordant pairs / 10 % read-through, Zipf junction support, 30 % PCR
    # duplicates, synthetic 24-contig genome with GENCODE-like annotation (no hg38 offline).  stress = config 3: long clips copied from the partner gene
    args = ["--seed", str(seed), "--read-seed", str(read_seed), "--fragments", str(fragments), "--normal-mult", "0", "--contigs", "24", "--contig-len", "12000000", "--genes-per-mb", "20",
            "--junctions", str(max(1000, fragments // 50))]
    if stress:
        args += ["--clip-min", "40", "--clip-max", "70", "--partner-clip", "0.5"]
    if NAME_LENGTH[0]:
        args += ["--name-length", str(NAME_LENGTH[0])]
    return args


def device_code_digest():
    """SHA-256 over the device sources (arriba_amd/csrc/device/*.hip, *.hpp): PMC passes are taken in runs of their own and committed (profiles/pmc_latest.json); the line
    quotes their traffic only if they were taken at this very device code (.git does not travel to the GPU box, the sources do)"""
    import glob
    import hashlib
    digest = hashlib.sha256()
    for path in sorted(glob.glob(os.path.join(ROOT, "arriba_amd", "csrc", "device", "*.h*"))):
        digest.update(os.path.basename(path).encode() + b"\0" + open(path, "rb").read())
    return digest.hexdigest()


def cpu_budget():
    """the processors this process may use at once: os.cpu_count() less what affinity and the CPU quota of the container (cgroup) allow -- the GPU box shows 256 hardware threads
    behind a quota of 16 CPUs, and 64 busy threads there run for a sixth of the time and stall for the rest (profiles/r03g_probe.txt)"""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            cores = min(cores, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            quota, period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()), int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0 and period > 0:
                cores = min(cores, max(1, quota // period))
        except (OSError, ValueError):
            pass
    return max(1, cores)


def scratch_directory(need_bytes):
    """a directory for the sample: tmpfs if it has the room (the BAM is read from memory, as from the page cache), else the default temporary directory"""
    for base in ("/dev/shm", tempfile.gettempdir()):
        try:
            if shutil.disk_usage(base).free > need_bytes * 1.3 + (4 << 30):
                return tempfile.mkdtemp(prefix="arriba_bench_", dir=base)
        except OSError:
            pass
    return tempfile.mkdtemp(prefix="arriba_bench_")


def generate_sample(fragments, seed, directory, read_seed=0, stress=False, threads=None):
    import datasets
    prefix = os.path.join(directory, "bench")
    threads = threads or min(64, cpu_budget())
    started = time.time()
    if os.environ.get("ARRIBA_BENCH_REUSE") and all(os.path.exists(prefix + suffix) for suffix in (".bam", ".fa", ".gtf")):
        return prefix, 0.0  # (A/B measurements on one GPU lease: the sample of an earlier run with --keep)
    subprocess.run([datasets.GEN_SYNTH, "--out", prefix, "--threads", str(threads)] + workload_args(fragments, seed, read_seed, stress), check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return prefix, time.time() - started


def run_reference(command, cwd=None):
    """runs the reference binary; returns (return code, its stdout, wall seconds, seconds until it starts on the BAM file, the N of its "(total=N)").  stdout is read as it comes:
    the reference flushes "Reading chimeric alignments from '...' " before it opens the file (source/arriba.cpp:124-129) and ends that line only when the file is read, so a
    line-wise reader would count the reading as loading; stderr goes to a file of its own (its warnings land in the middle of that very line otherwise)"""
    started = time.time()
    with tempfile.TemporaryFile() as errors:
        process = subprocess.Popen(command, cwd=cwd, stdout=subprocess.PIPE, stderr=errors)
        loaded_at, chunks = None, []
        while True:
            chunk = os.read(process.stdout.fileno(), 65536)
            if not chunk:
                break
            chunks.append(chunk)
            if loaded_at is None and b"Reading chimeric alignments" in b"".join(chunks[-2:]):
                loaded_at = time.time()
        process.wait()
        elapsed = time.time() - started
        errors.seek(0)
        output = b"".join(chunks).decode(errors="replace")
        total = re.search(r"Reading chimeric alignments from [^\n]*\(total=(\d+)\)", output)
        return process.returncode, output + errors.read().decode(errors="replace"), elapsed, (loaded_at - started) if loaded_at else 0.0, int(total.group(1)) if total else None


def cpu_baseline_fit():
    """the reference's time at 1 / 5 / 10 / 20 M fragments of this workload, run once where the repository was built (tools/make_bench_golden.py), and the fit SURVEY.md 8(d) asks for"""
    try:
        record = json.load(open(os.path.join(ROOT, "tests", "golden", "cpu_baseline_fit.json")))
    except (OSError, ValueError):
        return None
    return {"where": record.get("where"), "points": [{key: p.get(key) for key in ("workload", "chimeric_fragments", "seconds", "loading_seconds", "peak_memory_gb")} for p in record.get("points", [])],
            "fit": {key: value for key, value in record.items() if key.startswith("fit_")}}


def cpu_baseline(seed, directory, stress=False, sample_fragments=800000, subsampling=None, timed_fragments=None):
    """The unmodified reference (oracle/_ref/arriba_ref) on a bounded sample of the same workload, 1 core.  `value` is its rate AT THE SIZE OF THE TIMED SAMPLE (review of round 4,
    item 7b: the reference slows down with the sample, and the rate of the small live sample flatters it): the fit a*N + b*N*log2(N) through its runs at 1-20 M fragments
    (tests/golden/cpu_baseline_fit.json, another CPU), scaled by what this box's core does on the live sample against the fit at that size; the live point itself is `live`."""
    import datasets
    unit = "chimeric reads/s"
    if not os.path.exists(datasets.ARRIBA_REF):
        return {"value": None, "unit": unit, "cores": 1, "kind": "reference", "sample": "oracle/_ref/arriba_ref not present on this box"}
    prefix = os.path.join(directory, "cpu")
    subprocess.run([datasets.GEN_SYNTH, "--out", prefix] + workload_args(sample_fragments, seed, stress=stress), check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    command = [datasets.ARRIBA_REF, "-x", prefix + ".bam", "-g", prefix + ".gtf", "-a", prefix + ".fa", "-o", prefix + ".fusions.tsv", "-f", "blacklist"] + (["-U", str(subsampling)] if subsampling not in (None, 300) else [])
    returncode, output, elapsed, loading, total = run_reference(command)  # (the moment the reference starts on the BAM file separates its loading phase from its per-sample work)
    if returncode != 0:
        return {"value": None, "unit": unit, "cores": 1, "kind": "reference", "sample": "reference failed: " + output[-200:]}
    chimeric = total if total else sample_fragments
    live = {"value": chimeric / elapsed, "value_without_loading": chimeric / max(elapsed - loading, 1e-9), "chimeric_fragments": chimeric, "seconds": round(elapsed, 2), "loading_seconds": round(loading, 2)}
    sample = "%d chimeric fragments of the same synthetic workload, whole reference binary BAM->fusions.tsv%s, %.1f s wall of which %.1f s load the assembly and the annotation" % (chimeric, (" with -U %d" % subsampling) if subsampling not in (None, 300) else "", elapsed, loading)
    fit = cpu_baseline_fit()
    value, at_size = live["value"], None
    model = (fit or {}).get("fit", {}).get("fit_stress" if stress else "fit_config2")
    if model and timed_fragments and timed_fragments > chimeric:
        import math
        seconds_of = lambda count: model["a"] * count + model["b"] * count * math.log2(count)
        if seconds_of(chimeric) > 0 and seconds_of(timed_fragments) > 0:
            speed_of_this_core = seconds_of(chimeric) / max(elapsed - loading, 1e-9)  # (> 1: this box's core is faster than the one the fit was measured on)
            seconds_at_size = seconds_of(timed_fragments) / speed_of_this_core + loading
            value = timed_fragments / seconds_at_size
            at_size = {"chimeric_fragments": timed_fragments, "seconds": round(seconds_at_size, 1), "speed_of_this_core_against_the_fit": round(speed_of_this_core, 3),
                       "what": "extrapolated: the reference would need ~1 GB per million fragments (documentation/10-Current-limitations.md:14) and does not fit this box at that size"}
            sample = "extrapolated to the %d chimeric fragments of the timed sample with the fit of tests/golden/cpu_baseline_fit.json (%s), scaled by this box's core on the live sample (x%.2f); live: %s" % (timed_fragments, model["model"], speed_of_this_core, sample)
    return {"value": value, "unit": unit, "cores": 1, "kind": "reference, extrapolated" if at_size else "reference", "measured_on_this_box": live["value"], "sample": sample, "live": live, "at_the_timed_size": at_size,
            "value_without_loading": live["value_without_loading"], "seconds": live["seconds"], "loading_seconds": live["loading_seconds"],
            # the reference slows down with the sample (its containers are trees and hash maps of pointers): the points measured once in the build container and their fit
            "at_larger_samples": fit}


def normal_pairs_leg(pipeline, directory, fragments=10000000, steps=2):
    """SURVEY.md section 8(d)-2 specifies the sample of config 2 with 4 N ordinary proper pairs beside the N chimeric fragments (coverage and mapped_reads are then non-trivial and
    the ingest skips four records in five): the 10 M sample with them -- 107 M records, 21.6 GB -- through the same resident session, reported beside the headline number (a smaller main sample: a leg of its size)."""
    prefix = os.path.join(directory, "normal")
    started = time.time()
    arguments = workload_args(fragments, 1000)
    arguments[arguments.index("--normal-mult") + 1] = "4"
    subprocess.run([__import__("datasets").GEN_SYNTH, "--out", prefix, "--threads", str(min(64, cpu_budget()))] + arguments, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    generated = time.time() - started
    bam_bytes = os.path.getsize(prefix + ".bam")
    seconds = []
    for _ in range(steps + 1):  # (the first one is warm-up)
        started = time.perf_counter()
        pipeline.sample(prefix + ".bam", prefix + ".fusions.tsv")
        seconds.append(time.perf_counter() - started)
    counts = dict(pipeline.report)
    timing = pipeline.timing
    os.remove(prefix + ".bam")
    return {"what": "config 2 with 4 N ordinary proper pairs (SURVEY.md 8d-2): %d chimeric fragments among %d BAM records, %.1f GB" % (counts.get("read_chimeric_alignments", 0), counts.get("bam_records", 0), bam_bytes / 1e9),
            "chimeric_reads_per_s": counts.get("read_chimeric_alignments", 0) / (sum(seconds[1:]) / steps), "bam_records_per_s": counts.get("bam_records", 0) / (sum(seconds[1:]) / steps), "bam_GB_per_s": bam_bytes / 1e9 / (sum(seconds[1:]) / steps),
            "seconds_per_step": round(sum(seconds[1:]) / steps, 4), "steps": steps, "last_step": {key: round(value, 4) for key, value in timing.items()}, "generate_seconds": round(generated, 1)}


def stress_leg(fragments, steps=2, warmup=1):
    """BASELINE.json's config 3 at its stated size in the default run (review of round 5): the mismapper stress -- clipped segments of 40-70 nt copied from the partner gene,
    `-U 32767`, so that filter_mismappers sees every read and the candidates list 92.7 G supporting reads at 10^8 fragments (discordant lists implicit) -- as a run of this script
    of its own (a session of its own: -U is a parameter of the session, and the sample wants the device to itself), behind the session of the headline sample.  Its line, in short."""
    command = [sys.executable, os.path.abspath(__file__), "--stress", "--fragments", str(fragments), "--steps", str(steps), "--warmup", str(warmup), "--no-cpu-baseline", "--no-deflated-leg"]
    started = time.time()
    result = subprocess.run(command, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, env=dict(os.environ, ARRIBA_BENCH_NO_STRESS_LEG="1"))
    lines = [line for line in result.stdout.splitlines() if line.startswith("{") and '"metric"' in line]
    if result.returncode != 0 or not lines:
        return {"error": "the run of config 3 failed: " + result.stderr[-500:], "seconds": round(time.time() - started, 1)}
    line = json.loads(lines[-1])
    return {"what": line["config"]["workload"], "chimeric_reads_per_s": line["value"], "seconds_per_step": round(line["ms_per_step"] / 1e3, 3), "steps": line["steps"], "warmup": line["warmup"],
            "fragments": line["config"]["fragments_per_gpu"], "candidates": line["config"]["candidates"], "fusions": line["config"]["fusions"], "hbm_used_GB": line.get("hbm_used_GB"),
            "seconds_per_step_by_part": line.get("seconds_per_step"), "roofline": {key: line["roofline"].get(key) for key in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "launch_ms", "algorithmic_bytes_per_launch", "launches_per_step")},
            "kernel_ms": dict(list((line.get("kernel_ms") or {}).items())[:12]), "self_check": line.get("self_check"), "seconds_of_the_leg": round(time.time() - started, 1)}


def deflated_leg(pipeline, directory, fragments, stress, stored_output, steps=2, finish_ahead=False):
    """the same sample as a BAM file with deflated BGZF blocks (zlib level 1: what STAR writes by default and what samtools writes -- the reference reads any BAM htslib opens,
    source/read_chimeric_alignments.cpp:563-566): a quarter of the bytes cross PCIe, bgzf_inflate_kernel makes the record stream in HBM.  The same records: the same fusions.tsv."""
    import hashlib
    prefix = os.path.join(directory, "deflated")
    started = time.time()
    subprocess.run([__import__("datasets").GEN_SYNTH, "--out", prefix, "--threads", str(min(64, cpu_budget())), "--bam-only", "--bgzf-level", "1"] + workload_args(fragments, 1000, stress=stress), check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    generated = time.time() - started
    bam_bytes = os.path.getsize(prefix + ".bam")
    output = prefix + ".fusions.tsv"
    seconds = []
    pipeline.finish_ahead(finish_ahead)  # (as in the timed steps)
    pipeline.submit(prefix + ".bam")
    for k in range(steps + 1):  # (the first one is warm-up; the samples in a queue, as in the timed steps)
        if k == 1:
            pipeline.set_profiling(True)  # (a new epoch: the launches of the timed steps of this leg)
        pipeline.submit(prefix + ".bam")
        started = time.perf_counter()
        pipeline.sample(prefix + ".bam", output)
        seconds.append(time.perf_counter() - started)
    pipeline.cancel()
    pipeline.flush()
    pipeline.finish_ahead(False)
    container_kernels = {}
    for name, ms, size in pipeline.kernel_profile():  # (the launches of the steps, and of the sample that was fed ahead and thrown away: the sums are per fed sample)
        if name.startswith("bgzf_"):
            entry = container_kernels.setdefault(name, {"launches": 0, "ms": 0.0, "bytes": 0})
            entry["launches"] += 1; entry["ms"] += ms; entry["bytes"] += size
    timing = pipeline.timing
    counts = dict(pipeline.report)
    digest = lambda path: hashlib.sha256(open(path, "rb").read()).hexdigest()
    same = digest(output) == digest(stored_output)
    os.remove(prefix + ".bam")
    per_step = sum(seconds[1:]) / steps
    return {"what": "the same %d fragments as a BAM file with deflated BGZF blocks (zlib level 1, %.1f GB: %.2f x smaller), inflated on the device (bgzf_inflate_tokens_kernel + bgzf_inflate_resolve_kernel)" % (counts.get("read_chimeric_alignments", 0), bam_bytes / 1e9, counts.get("bam_stream_bytes", 0) / max(bam_bytes, 1)),
            "chimeric_reads_per_s": counts.get("read_chimeric_alignments", 0) / per_step, "seconds_per_step": round(per_step, 4), "steps": steps, "bam_GB": round(bam_bytes / 1e9, 2),
            "last_step": {key: round(value, 4) for key, value in timing.items()}, "fusions_tsv_equals_the_stored_sample's": same, "generate_seconds": round(generated, 1),
            # the kernels of the container (event times: they run beside the stages of the sample in front): ms per launch, GB/s of their algorithmic bytes (compressed in + stream out for
            # pass 1, the stream for pass 2 and the CRC)
            "container_kernels": {name: {"launches": values["launches"], "ms_per_launch": round(values["ms"] / values["launches"], 3), "ms_per_fed_sample": round(values["ms"] / (steps + 1), 1),
                                         "GB_per_s": round(values["bytes"] / max(values["ms"], 1e-9) / 1e6, 1)} for name, values in container_kernels.items()}}


def host_only(args):
    """the host side alone (no GPU): rates of the file side of the device ingest and of the classic host ingest; a smoke test for changes under csrc/host/"""
    import ctypes
    import datasets
    import __graft_entry__
    from arriba_amd import _capi
    from arriba_amd.pipeline import HostSession
    directory = scratch_directory(args.fragments * 600)
    try:
        prefix, generate_seconds = generate_sample(args.fragments, 1000, directory)
        lib = _capi.host_library()
        line = {"mode": "host-only", "fragments": args.fragments, "bam_bytes": os.path.getsize(prefix + ".bam"), "generate_seconds": round(generate_seconds, 2)}
        session = HostSession(prefix + ".fa", prefix + ".gtf")
        # (a) BamFeed: header + all pieces, stored BGZF handed on raw
        config = _capi.IngestConfig()
        started = time.time()
        if lib.ahost_bam_open(session._session, (prefix + ".bam").encode(), 0, 100, ctypes.byref(config)) != 0:
            raise SystemExit("ahost_bam_open: " + lib.ahost_last_error().decode())
        piece_bytes = 64 << 20
        buffer = ctypes.create_string_buffer(piece_bytes)
        blocks = (_capi.BgzfBlock * (piece_bytes // 4096 + 16))()
        piece = _capi.BamPiece()
        fed = stream = pieces = 0
        while True:
            status = lib.ahost_bam_next(session._session, buffer, piece_bytes, blocks, len(blocks), ctypes.byref(piece))
            if status < 0:
                raise SystemExit("ahost_bam_next: " + lib.ahost_last_error().decode())
            if status == 0:
                break
            fed += piece.bytes; stream += piece.stream_bytes; pieces += 1
        lib.ahost_bam_close(session._session)
        seconds = time.time() - started
        line["bam_feed"] = {"seconds": round(seconds, 3), "file_GB_per_s": fed / seconds / 1e9, "pieces": pieces, "stream_bytes": stream, "first_record_offset": int(config.first_record_offset), "n_targets": int(config.n_targets)}
        # (b) the classic host ingest from a file and from a named pipe (what round 1's bench streamed through)
        for source in ("file", "fifo"):
            host = HostSession(prefix + ".fa", prefix + ".gtf")
            path = prefix + ".bam"
            producer = None
            if source == "fifo":
                path = prefix + ".fifo"
                os.mkfifo(path)
                producer = subprocess.Popen(["sh", "-c", 'cat "$0" > "$1"', prefix + ".bam", path])  # (the shell opens the pipe: opening it here would block until the reader is there)
            started = time.time()
            host.read_chimeric_alignments(path)
            seconds = time.time() - started
            if producer:
                producer.wait()
            line["host_ingest_from_" + source] = {"seconds": round(seconds, 3), "chimeric_reads_per_s": host.fragment_count / seconds, "fragments": host.fragment_count}
        print(json.dumps(line))
    finally:
        shutil.rmtree(directory, ignore_errors=True)


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--gpus", type=int, default=1)
    parser.add_argument("--steps", type=int, default=None, help="timed steps (default 4; 2 for --stress at 50 M fragments and more, where a sample takes ~40 s)")
    parser.add_argument("--warmup", type=int, default=None, help="default 2 (1 for --stress at 50 M fragments and more); untimed steps in front of the timed ones; the session has two lanes that take the samples in turn, and a lane allocates its buffers with its first sample: with fewer than two that first sample is a timed step (2.99 s instead of 1.92 at 10^8 fragments, profiles/r04f_driver.err)")
    parser.add_argument("--fragments", type=int, default=None, help="chimeric fragments per GPU (default: 100 M, BASELINE.json's 100 M-read synthetic, if the box has the memory for the 54 GB file; 20000 with --host-only)")
    parser.add_argument("--stress", action="store_true", help="BASELINE.json config 3: clipped segments of 40-70 nt copied from the partner gene, -U 32767 (filter_mismappers sees every read)")
    parser.add_argument("--subsampling-threshold", type=int, default=None, help="-U of the reference (source/options.cpp:422-423); default 300, with --stress 32767 as SURVEY.md 8(d) config 3 says")
    parser.add_argument("--name-length", type=int, default=0, help="read names of this many characters (the generator's default: 11); 45 = an Illumina run's")
    parser.add_argument("--discarded", action="store_true", help="also write discarded.tsv (-O) inside the step")
    parser.add_argument("--host-ingest", action="store_true", help="read_chimeric_alignments by the multi-threaded host ingest instead of on the device (round 1's path)")
    parser.add_argument("--python-stages", action="store_true", help="time the ctypes mirror of the stage order (arriba_amd/pipeline.py) instead of arriba_workflow_sample of the product library")
    parser.add_argument("--no-pipeline", action="store_true", help="one sample at a time: without it the file of the next sample is fed (arriba_workflow_submit) while the stages of the current one run")
    parser.add_argument("--no-deferred-output", action="store_true", help="with the samples in a queue: fusions.tsv of a sample is written before arriba_workflow_sample returns (without it: by a thread of the session, beside the next sample)")
    parser.add_argument("--no-cpu-baseline", action="store_true")
    parser.add_argument("--no-deflated-leg", action="store_true", help="skip the secondary measurement on the same sample with deflated BGZF blocks (value_deflated: inflated on the device)")
    parser.add_argument("--no-stress-leg", action="store_true", help="skip the secondary measurement of BASELINE.json's config 3 (value_stress: the mismapper stress with -U 32767 at the size of the headline sample, a run of its own behind it)")
    parser.add_argument("--no-normal-pairs", action="store_true", help="skip the secondary measurement on the 10 M sample with 4 N ordinary proper pairs (value_with_normal_pairs)")
    parser.add_argument("--host-only", action="store_true")
    parser.add_argument("--keep", help="keep the sample and the output files in this directory")
    parser.add_argument("--per-rank-samples", action="store_true", help="with --gpus N: every rank works on a sample of its own (weak scaling, no collective) instead of all ranks on one sample")
    args = parser.parse_args()
    # config 3 at its stated size (-U 32767: implicit discordant lists, ~40 s per sample): fewer steps and no deflated leg unless asked for, so that the plain command finishes in minutes
    large_stress = args.stress and args.fragments is not None and args.fragments >= 50000000
    if args.steps is None:
        args.steps = 2 if large_stress else 4
    if args.warmup is None:
        args.warmup = 1 if large_stress else 2
    if large_stress:
        args.no_deflated_leg = True
    NAME_LENGTH[0] = args.name_length
    if args.host_only:
        args.fragments = args.fragments or 20000
        return host_only(args)

    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # called the way the driver calls it at N = 1 (`python bench.py --gpus N ...`): the N ranks are launched from here, one per GPU over RCCL, exactly as the driver's
        # own command would (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...`)
        import socket
        with socket.socket() as probe:
            probe.bind(("127.0.0.1", 0))
            port = probe.getsockname()[1]
        command = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(sys.argv[0])] + sys.argv[1:]  # (sys.argv[0]: bench.py, or the harness the CPU tier runs it on)
        progress("launching %d ranks: %s" % (args.gpus, " ".join(command)))
        raise SystemExit(subprocess.run(command, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))).returncode)
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d was started with WORLD_SIZE=%d: the line would report a number of GPUs that did not take part" % (args.gpus, world))
    distributed = world > 1
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=os.environ.get("ARRIBA_BENCH_BACKEND", "nccl"))

    import __graft_entry__
    if rank == 0:
        __graft_entry__.build()
    if distributed:
        dist.barrier()
    import numpy as np
    from arriba_amd.pipeline import DevicePipeline, HostSession, WorkflowSession
    # --gpus N: BASELINE.json config 4 -- ONE sample over the N GPUs (arriba_amd/one_sample.py: every rank ingests its part of the file, one all-gather, the
    # re-alignments of filter_mismappers shared out, rank 0 writes the files); the same total work for every N = strong scaling
    one_sample = distributed and not args.per_rank_samples and not args.host_ingest

    fallback_reason = os.environ.get("ARRIBA_BENCH_FALLBACK_REASON")
    large_sample_record = None
    if args.fragments is None:
        # BASELINE.json quotes the metric on the 100 M-read synthetic: that is the default where the box can hold it (54 GB of BAM in memory per GPU, ~140 GB of HBM
        # at the peak); otherwise config 2 (10 M).
        memory = os.sysconf("SC_PAGE_SIZE") * os.sysconf("SC_PHYS_PAGES")
        shm_free = shutil.disk_usage("/dev/shm").free if os.path.isdir("/dev/shm") else 0
        device_memory = torch.cuda.get_device_properties(local_rank).total_memory
        samples = 1 if one_sample else world
        fits = memory > samples * (160 << 30) and max(shm_free, shutil.disk_usage(tempfile.gettempdir()).free) > samples * (80 << 30) and device_memory > (200 << 30)
        if distributed:  # (every rank must come to the same conclusion)
            verdict = torch.tensor([int(fits)], device="cuda" if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(verdict, op=dist.ReduceOp.MIN)
            fits = bool(verdict.item())
        args.fragments = 100000000 if fits else 10000000
        if not fits:
            fallback_reason = "10 M fragments (BASELINE.json config 2) instead of the 100 M sample: host memory %.0f GB, tmpfs %.0f GB free, HBM %.0f GB" % (memory / 2**30, shm_free / 2**30, device_memory / 2**30)
        elif not distributed and not os.environ.get("ARRIBA_BENCH_CHILD"):
            # the large sample runs in a child with a time limit: if it does not come back with a line (a time-out, an error), the line of config 2 is printed instead,
            # with the reason -- a bench without a line is worth nothing
            command = [sys.executable, os.path.abspath(__file__), "--fragments", str(args.fragments), "--steps", str(args.steps), "--warmup", str(args.warmup)]
            command += (["--subsampling-threshold", str(args.subsampling_threshold)] if args.subsampling_threshold is not None else []) + (["--name-length", str(args.name_length)] if args.name_length else [])
            command += [flag for flag, on in (("--stress", args.stress), ("--discarded", args.discarded), ("--host-ingest", args.host_ingest), ("--python-stages", args.python_stages), ("--no-cpu-baseline", args.no_cpu_baseline), ("--no-normal-pairs", args.no_normal_pairs), ("--no-deflated-leg", args.no_deflated_leg), ("--no-pipeline", args.no_pipeline), ("--no-deferred-output", args.no_deferred_output)) if on]
            # the driver gives a bench run 1800 s; the large sample gets what is left of ~1500 s after a reserve for the line of config 2 (generation, 25 steps of ~1 s, the
            # reference on its bounded sample: ~150 s), and its child decides after every step whether the steps asked for still fit (ARRIBA_BENCH_DEADLINE)
            total_limit = float(os.environ.get("ARRIBA_BENCH_TOTAL_LIMIT", "1500"))
            limit = float(os.environ.get("ARRIBA_BENCH_LARGE_LIMIT", str(max(60.0, total_limit - 150.0 - (time.time() - BENCH_STARTED)))))
            child_scratch = scratch_directory(args.fragments * 600)  # the child's sample lives here; removed below whatever happens to the child
            try:
                child = subprocess.run(command, stdout=subprocess.PIPE, env=dict(os.environ, ARRIBA_BENCH_CHILD="1", ARRIBA_BENCH_SCRATCH=child_scratch, ARRIBA_BENCH_DEADLINE=str(limit - 40.0)), timeout=limit, universal_newlines=True)
                lines = [line for line in child.stdout.splitlines() if line.startswith("{")]
                if child.returncode == 0 and lines and '"metric"' in lines[-1]:
                    print(lines[-1])
                    return
                # the steps of the large sample that did run are not thrown away: they go into the line of config 2 as `large_sample`
                for line in lines:
                    try:
                        large_sample_record = json.loads(line).get("large_sample", large_sample_record)
                    except ValueError:
                        pass
                fallback_reason = ("the %d steps + %d warm-up steps asked for do not fit the time limit of the run at the measured step time of the 100 M sample (see large_sample)" % (args.steps, args.warmup)) if child.returncode == 3 else "the 100 M sample ended with exit code %d and no line" % child.returncode
            except subprocess.TimeoutExpired:
                fallback_reason = "the 100 M sample did not finish within %.0f s" % limit
            finally:
                shutil.rmtree(child_scratch, ignore_errors=True)
            progress("falling back to 10 M fragments: " + fallback_reason)
            args.fragments = 10000000
    if one_sample:  # rank 0 makes the sample where every rank of the node finds it
        shared = [args.keep or os.environ.get("ARRIBA_BENCH_SCRATCH") or scratch_directory(args.fragments * 600)] if rank == 0 else [None]
        dist.broadcast_object_list(shared, src=0)
        directory = shared[0]
    else:
        directory = args.keep or os.environ.get("ARRIBA_BENCH_SCRATCH") or scratch_directory(args.fragments * 600)
    os.makedirs(directory, exist_ok=True)
    try:
        if one_sample:
            timing = [None, None]
            if rank == 0:
                timing = list(generate_sample(args.fragments, 1000, directory, stress=args.stress))
            dist.broadcast_object_list(timing, src=0)  # (also the barrier behind which the files exist)
            prefix, generate_seconds = timing
        else:
            prefix, generate_seconds = generate_sample(args.fragments, 1000, directory, read_seed=0 if world == 1 else 7000 + rank, stress=args.stress, threads=max(1, min(64, cpu_budget() // world)))
        bam_bytes = os.path.getsize(prefix + ".bam")
        progress("sample generated: %d fragments, %.1f GB BAM in %.1f s (%s)" % (args.fragments, bam_bytes / 1e9, generate_seconds, directory))
        subsampling = args.subsampling_threshold if args.subsampling_threshold is not None else (32767 if args.stress else 300)
        params = {"subsampling_threshold": subsampling} if subsampling != 300 else None
        pipeline = None
        outputs = [os.path.join(directory, "fusions.rank%d.tsv" % rank), os.path.join(directory, "discarded.rank%d.tsv" % rank) if args.discarded else None]
        stage_log, step_seconds, ingest_parts, steps_done, all_steps = [], [], [], [0], []
        # One GPU per sample: the step is arriba_workflow_sample of the product library (libarriba_workflow.so: the reference's main() in C++ over the two C ABIs, resident session);
        # --python-stages times the ctypes mirror of the same stage order instead (arriba_amd/pipeline.py, what rounds 1-2 timed), as do --host-ingest and one sample over N GPUs
        # (round 5: also with --gpus N -- the same call of the same library is timed at N = 1 and N = 8, arriba_workflow_sample as a collective call over a communicator:
        # RCCL alone on the GPUs, arriba_workflow_join_rccl; torch.distributed's gloo through callbacks where the harness stands in for the device, tests/bench_on_harness.py)
        through_workflow_library = not args.host_ingest and not args.python_stages
        pipelined, primed, finish_ahead, collectives = through_workflow_library and not args.no_pipeline, [False], [False], [None]
        if through_workflow_library:
            pipeline = WorkflowSession(prefix + ".fa", prefix + ".gtf", params=params, device=local_rank)
            session = None
            if one_sample:
                rccl_alone = dist.get_backend() == "nccl" and os.environ.get("ARRIBA_BENCH_COLLECTIVES") != "torch"
                if rccl_alone:  # the communicator of RCCL alone; if a rank cannot have it (librccl not found), all of them take torch.distributed's collectives through the callbacks instead
                    joined = 1
                    try:
                        unique = [pipeline.rccl_unique_id() if rank == 0 else None]
                    except Exception as problem:
                        unique, joined = [None], 0
                        progress("arriba_workflow_rccl_unique_id: %s" % problem)
                    dist.broadcast_object_list(unique, src=0)
                    if unique[0] is None:
                        joined = 0
                    else:
                        try:
                            pipeline.join_rccl(unique[0], rank, world)
                        except Exception as problem:
                            joined = 0
                            progress("arriba_workflow_join_rccl: %s" % problem)
                    verdict = torch.tensor([joined], device="cuda")
                    dist.all_reduce(verdict, op=dist.ReduceOp.MIN)
                    rccl_alone = bool(verdict.item())
                if not rccl_alone:
                    pipeline.over_ranks()
                collectives[0] = "RCCL: arriba_workflow_join_rccl" if rccl_alone else "torch.distributed (%s) through arriba_workflow_set_communicator" % dist.get_backend()
        else:
            session = HostSession(prefix + ".fa", prefix + ".gtf")  # assembly + annotation, once (resident)
        progress("assembly and annotation loaded")

        def step():
            nonlocal pipeline
            started = time.perf_counter()
            del stage_log[:]
            if through_workflow_library:
                if pipelined:  # a resident service with a queue of samples: the next one is submitted before this one is worked on
                    if not primed[0]:
                        pipeline.defer_output(not args.no_deferred_output and not one_sample)
                        # the ingest of the next sample finished by its feeder, beside the stages of this one (arriba_workflow_finish_ahead: the lanes keep their batch buffers).
                        # On by itself up to 5e7 fragments on a 288 GB device: 10 M fragments 0.43 -> 0.32 s per step (profiles/r05k_*); at 10^8 it fits (289 of 295 GB in use,
                        # profiles/r05l_ahead100m.json) but gains nothing -- there the device is busy either way, 2.01 s against 1.99-2.03 s -- and leaves no room for the deflated leg:
                        # off unless ARRIBA_FINISH_AHEAD=1 asks for it
                        knob = os.environ.get("ARRIBA_FINISH_AHEAD")
                        finish_ahead[0] = not one_sample and ((knob == "1") if knob in ("0", "1") else (torch.cuda.get_device_properties(local_rank).total_memory > (250 << 30) and not args.stress and args.fragments <= 50_000_000))
                        pipeline.finish_ahead(finish_ahead[0])
                        pipeline.submit(prefix + ".bam")
                        primed[0] = True
                    pipeline.submit(prefix + ".bam")
                report = pipeline.sample(prefix + ".bam", outputs[0], outputs[1])
                finished = time.perf_counter()
                timing = pipeline.timing
                counts = dict(report)
                pipeline.n, pipeline.n_candidates, pipeline.records = counts.get("read_chimeric_alignments", 0), counts.get("find_fusions", 0), counts.get("bam_records", -1)
                pipeline.writer_seconds = {key: round(timing[key], 4) for key in ("output_results", "output_rows", "output_format")}
                stage_log.extend((stage, count, None) for stage, count in report)
                ingest_parts.append({"feed": timing["feed"], "device": timing["ingest"], "adopt": timing["adopt"], "feed_read": timing.get("feed_read", 0.0), "feed_push": timing.get("feed_push", 0.0), "feed_total": timing.get("feed_total", 0.0)})
                if one_sample:
                    ingest_parts[-1].update({"exchange_parts": timing["exchange_parts"], "exchange_verdicts": timing["exchange_verdicts"], "exchange_rows": timing["exchange_rows"]})
                ingested = started + timing["feed"] + timing["ingest"] + timing["adopt"]
                step_seconds.append({"ingest": ingested - started, "workflow": finished - ingested, "total": finished - started, "stages": timing["stages"], "filter_mismappers": timing["filter_mismappers"], "output": timing["output"]})
                if verbose:
                    progress("arriba_workflow_sample: %s" % {key: round(value, 3) for key, value in timing.items()})
                return after_step(started, finished, ingested)
            if args.host_ingest:
                session.read_chimeric_alignments(prefix + ".bam")
                if pipeline is not None:
                    pipeline.close()
                pipeline = DevicePipeline(session, params=params, device=local_rank)
                pipeline.set_profiling(profiling[0])
                ingest_parts.append({"host_ingest": time.perf_counter() - started})
            elif pipeline is None and one_sample:
                from arriba_amd.one_sample import OneSamplePipeline
                pipeline = OneSamplePipeline(session, prefix + ".bam", params=params, device=local_rank, piece_bytes=256 << 20, profiling=profiling[0] or args.warmup == 0)
                ingest_parts.append(dict(pipeline.ingest_seconds))
            elif pipeline is None:
                pipeline = DevicePipeline(session, params=params, device=local_rank, bam=prefix + ".bam", piece_bytes=256 << 20, profiling=profiling[0] or args.warmup == 0)
                ingest_parts.append(dict(pipeline.ingest_seconds))
            else:
                pipeline.read_chimeric_alignments(prefix + ".bam", piece_bytes=256 << 20)
                ingest_parts.append(dict(pipeline.ingest_seconds))
            ingested = time.perf_counter()
            if verbose:
                progress("read_chimeric_alignments done: %d fragments in %.2f s %s" % (pipeline.n, ingested - started, ingest_parts[-1]))

            def note(stage, count):
                stage_log.append((stage, count, round(time.perf_counter() - started, 4)))
                if verbose:
                    progress("%s: %d (%.2f s into the step)" % (stage, count, time.perf_counter() - started))
            pipeline.run_workflow(outputs[0], outputs[1], log=note)
            finished = time.perf_counter()
            step_seconds.append({"ingest": ingested - started, "workflow": finished - ingested, "total": finished - started})
            return after_step(started, finished, ingested)

        def after_step(started, finished, ingested):
            steps_done[0] += 1
            remaining = args.warmup + args.steps - steps_done[0]
            deadline = os.environ.get("ARRIBA_BENCH_DEADLINE")
            if os.environ.get("ARRIBA_BENCH_CHILD") and deadline and remaining > 0:
                # the large sample: do the steps still to come fit?  Priced with the fastest step so far (the first one is cold: allocations, first touches of the
                # pinned buffers), plus the reference on its bounded sample behind the steps
                all_steps.append(finished - started)
                per_step = min(all_steps)
                if (time.time() - BENCH_STARTED) + remaining * per_step + (0 if args.no_cpu_baseline else 45) > float(deadline):
                    progress("a step of the %d-fragment sample takes %.1f s: %d more steps do not fit the time limit" % (args.fragments, per_step, remaining))
                    print(json.dumps({"large_sample": {"fragments": pipeline.n, "seconds_per_step": [round(x, 3) for x in all_steps], "steps_run": len(all_steps), "chimeric_reads_per_s": pipeline.n / per_step,
                                                       "read_chimeric_alignments_seconds": ingest_parts[-1], "stage_kernel_ms": {stage: round(values["ms"], 1) for stage, values in pipeline.timings.items()},
                                                       "output_side_seconds": getattr(pipeline, "writer_seconds", None)}}))
                    sys.stdout.flush()
                    raise SystemExit(3)  # (through the `finally` below: the 54 GB sample must not stay behind)
            progress("step done: %.2f s; read_chimeric_alignments %.2f s %s, workflow %.2f s; %s; output side: %s" % (finished - started, ingested - started, {k: round(v, 3) for k, v in ingest_parts[-1].items()}, finished - ingested,
                     ("stages %.2f s, filter_mismappers %.2f s" % (step_seconds[-1]["stages"], step_seconds[-1]["filter_mismappers"])) if through_workflow_library else
                     "slowest stages: %s" % sorted(((round(v["ms"]), k) for k, v in pipeline.timings.items()), reverse=True)[:4], getattr(pipeline, "writer_seconds", None)))

        profiling = [False]
        verbose = args.fragments >= 30000000 or bool(os.environ.get("ARRIBA_BENCH_VERBOSE"))  # large samples: every stage reports on stderr, so that a run cut off by a time limit says where it was
        if verbose:
            import faulthandler
            faulthandler.dump_traceback_later(240, repeat=True, file=sys.stderr)
        for _ in range(args.warmup):
            step()
        profiling[0] = True
        if pipeline is not None:
            pipeline.set_profiling(True)  # HIP events around every kernel launch, recorded on the launch stream
        del step_seconds[:], ingest_parts[:]
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()
        started = time.time()
        for _ in range(args.steps):
            step()
            if pipeline is not None and not pipeline._profiling_on:
                pipeline.set_profiling(True)
        deferred_writer_seconds = pipeline.flush() if pipelined else None  # (inside the timed region: the file of the last step is complete before the clock stops)
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        elapsed = time.time() - started
        n = pipeline.n
        sample_alone, timed_profile, alone_profile = None, None, None
        if pipelined:  # the sample submitted behind the last timed step is thrown away; then one sample alone, for the time from its file to its fusions.tsv when nothing overlaps it
            pipeline.cancel()
            pipeline.defer_output(False)
            pipeline.finish_ahead(False)
            timed_profile = pipeline.kernel_profile()  # (the launches of the timed steps; the sample alone is profiled on its own)
            pipeline.set_profiling(True)
            primed[0], pipelined = False, False
            alone_started = time.perf_counter()
            pipeline.sample(prefix + ".bam", outputs[0], outputs[1])
            sample_alone = {"seconds": round(time.perf_counter() - alone_started, 4), "parts": {key: round(value, 4) for key, value in pipeline.timing.items()}}
            alone_profile = pipeline.kernel_profile()
            pipelined = True
        if distributed:
            device = "cuda" if dist.get_backend() == "nccl" else "cpu"
            tensor = torch.tensor([elapsed], device=device, dtype=torch.float64)
            dist.all_reduce(tensor, op=dist.ReduceOp.MAX)
            elapsed = float(tensor.item())
            counts = torch.tensor([n], device=device, dtype=torch.int64)
            dist.all_reduce(counts, op=dist.ReduceOp.SUM)
            total_fragments = n if one_sample else int(counts.item())  # one sample: every rank holds (and counted) all of its fragments
        else:
            total_fragments = n

        progress("timed steps done: %.2f s per step" % (elapsed / args.steps))
        try:
            free_bytes, total_bytes = torch.cuda.mem_get_info(local_rank)
            hbm_used_gb = round((total_bytes - free_bytes) / 1e9, 1)
        except Exception:
            hbm_used_gb = None
        # post-conditions of the last step on the full-size batch (untimed; no oracle involved): a kernel that silently skipped a part of the batch would
        # leave alignments without a gene, or read filter counts that do not add up; the output file must exist and hold the fusions the log counted
        self_check = []
        for slot in (0, 1):
            without_gene = int((pipeline.gene_sets(slot)[0] == 0).sum())
            if without_gene:
                self_check.append("%d alignments in slot %d have no gene after annotate" % (without_gene, slot))
        writes_files = rank == 0 or not one_sample
        fusion_lines = sum(1 for line in open(outputs[0]) if not line.startswith("#")) if writes_files else stage_log[-1][1]
        if not stage_log or stage_log[-1][0] != "recover_isoforms" or fusion_lines != stage_log[-1][1]:
            self_check.append("fusions.tsv holds %d fusions, the last stage counted %s" % (fusion_lines, stage_log[-1:] or None))
        # the sample of config 2 (10 M fragments) is the one the unmodified reference was run on once (tests/golden/bench10m): the file written by the last timed step must be its file
        reference_check = None
        golden = os.path.join(ROOT, "tests", "golden", ("stress%dm" if args.stress else "bench%dm") % (args.fragments // 1000000), "meta.json")
        if writes_files and args.fragments % 1000000 == 0 and subsampling == (32767 if args.stress else 300) and not args.discarded and not args.name_length and os.path.exists(golden):
            import hashlib
            meta = json.load(open(golden))
            if hashlib.sha256(open(outputs[0], "rb").read()).hexdigest() != meta["fusions_tsv_sha256"]:
                self_check.append("fusions.tsv differs from the file the unmodified reference writes for this sample (%s)" % os.path.relpath(os.path.dirname(golden), ROOT))
            reference_check = "fusions.tsv of the last timed step is byte-identical (SHA-256) to the file the unmodified reference wrote for this very sample (%s: %d fusions, reference run time %.0f s where the repository was built)" % (os.path.relpath(os.path.dirname(golden), ROOT), meta["fusions"], meta["reference_seconds_in_the_build_container"])
        if self_check:
            raise SystemExit("bench self-check failed: " + "; ".join(self_check))

        if rank == 0:
            kernels = {}
            for name, ms, size in (timed_profile if timed_profile is not None else pipeline.kernel_profile()):
                entry = kernels.setdefault(name, {"launches": 0, "ms": 0.0, "bytes": 0})
                entry["launches"] += 1
                entry["ms"] += ms
                entry["bytes"] += size
            modelled = {name: values for name, values in kernels.items() if values["bytes"] > 0}
            # Which kernel the line prices (review of round 4, item 7a): the one with the most time OF ITS OWN.  With the samples in a queue the kernels of the next sample's ingest run
            # at the lowest priority beside the stages of the sample in front, and their event times are mostly waiting (group_replay_kernel: 849 ms of events per step for 239 ms
            # of work) -- so the choice is made in the sample that ran alone behind the timed steps, where nothing overlaps; `achieved` is still what the HIP events of the timed
            # steps say about that kernel, `alone` what it does by itself.
            alone_kernels = {}
            for name, ms, size in (alone_profile or []):
                entry = alone_kernels.setdefault(name, {"launches": 0, "ms": 0.0, "bytes": 0})
                entry["launches"] += 1
                entry["ms"] += ms
                entry["bytes"] += size
            alone_modelled = {name: values for name, values in alone_kernels.items() if values["bytes"] > 0 and name in modelled}
            dominant = max(alone_modelled, key=lambda name: alone_modelled[name]["ms"]) if alone_modelled else max(modelled, key=lambda name: modelled[name]["ms"])
            pmc_path = os.path.join(ROOT, "profiles", "pmc_latest.json")
            pmc = json.load(open(pmc_path)) if os.path.exists(pmc_path) else None

            def roofline_of(kernel):
                """algorithmic bytes per launch / mean launch time from the HIP events of the timed steps; HBM traffic per launch from the committed PMC passes (FETCH_SIZE and WRITE_SIZE in
                separate runs, tools/gpu_session.sh) when they were taken at this device code and sample size; the same kernel in the sample that ran alone"""
                launches = kernels[kernel]["launches"]
                launch_ms = kernels[kernel]["ms"] / launches
                launch_bytes = kernels[kernel]["bytes"] / launches
                achieved = launch_bytes / (launch_ms * 1e-3) / 1e9 if launch_ms > 0 else 0.0
                traffic, traffic_note = None, "no PMC passes committed (profiles/pmc_latest.json)"
                if pmc is not None:
                    entry = pmc.get("kernels", {}).get(kernel.split("(")[0])
                    if pmc.get("device_code_sha256") != device_code_digest():
                        traffic_note = "profiles/pmc_latest.json was taken at other device code (its device_code_sha256 is not that of arriba_amd/csrc/device here): not quoted"
                    elif pmc.get("fragments") != n:
                        traffic_note = "profiles/pmc_latest.json was taken on a sample of %s fragments, this one has %d: not quoted" % (pmc.get("fragments"), n)
                    elif entry and entry.get("dispatches"):
                        traffic = (2.0 * entry.get("FETCH_SIZE", 0.0) + entry.get("WRITE_SIZE", 0.0)) * 1024.0 / entry["dispatches"]
                        traffic_note = "2 x FETCH_SIZE + WRITE_SIZE per launch (rocprofv3 --pmc, separate passes; %s)" % pmc.get("source", "profiles/pmc_latest.json")
                alone = None
                if kernel in alone_kernels and alone_kernels[kernel]["ms"] > 0:
                    ms_alone = alone_kernels[kernel]["ms"] / alone_kernels[kernel]["launches"]
                    bytes_alone = alone_kernels[kernel]["bytes"] / alone_kernels[kernel]["launches"]
                    alone = {"achieved": bytes_alone / (ms_alone * 1e-3) / 1e9, "frac": bytes_alone / (ms_alone * 1e-3) / 1e9 / HBM_PEAK_GBS, "launch_ms": ms_alone, "launches": alone_kernels[kernel]["launches"],
                             "ms_per_sample": alone_kernels[kernel]["ms"], "what": "the same kernel in one sample that ran alone behind the timed steps (nothing of another sample beside it)"}
                return {"bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_note": traffic_note, "alone": alone,
                        "launch_ms": launch_ms, "algorithmic_bytes_per_launch": launch_bytes, "launches_per_step": launches / args.steps}
            roofline = roofline_of(dominant)
            roofline["chosen_by"] = "most kernel time of its own in the sample that ran alone" if alone_modelled else "most summed event time in the timed steps (no sample ran alone)"
            # beside it: the search kernel of filter_mismappers (dependent look-ups: priced with the bytes of its reads and hits it is nowhere near the roofline, and its real traffic is
            # what `traffic` says) and the streaming kernel with the most time of its own
            search_kernels = [name for name in modelled if name.startswith("mismapper_heavy_kernel")]
            streaming = [name for name in (alone_modelled or modelled) if not name.startswith("mismapper_") and not name.startswith("group_replay") and not name.startswith("bgzf_inflate") and name != dominant]
            also = []
            if search_kernels and search_kernels[0] != dominant:
                also.append(roofline_of(max(search_kernels, key=lambda name: modelled[name]["ms"])))
            if streaming:
                also.append(roofline_of(max(streaming, key=lambda name: (alone_modelled or modelled)[name]["ms"])))
            roofline["also"] = also
            # the cascade as a whole (review of round 5, hygiene): the algorithmic bytes of a sample -- SURVEY.md 8(d): 433 B per fragment through the read-level cascade and the key
            # emission, plus the BAM stream the ingest reads once -- over the device work of a sample (the sum of its kernels when nothing runs beside them), against the HBM peak
            if alone_kernels:
                cascade_bytes = 433.0 * n + bam_bytes
                cascade_ms = sum(values["ms"] for values in alone_kernels.values())
                roofline["cascade"] = {"algorithmic_bytes_per_sample": cascade_bytes, "kernel_ms_alone_sum": round(cascade_ms, 1), "achieved": cascade_bytes / (cascade_ms * 1e-3) / 1e9, "unit": "GB/s",
                                       "frac": cascade_bytes / (cascade_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "what": "433 B x fragments (SURVEY.md 8d) + the bytes of the BAM stream, over the sum of the kernel times of one sample that ran alone"}
            kernel_ms_per_step = sum(values["ms"] for values in kernels.values()) / args.steps
            mean = lambda key: sum(s[key] for s in step_seconds) / len(step_seconds)
            resident_stages = ("mark_multimappers", "annotate", "read_filters_stage1", "fragment_length_samples", "read_filters_stage2", "find_fusions", "merge_adjacent_fusions", "filter_multimappers",
                               "candidate_iteration_order", "estimate_expected_fusions", "filter_candidate_predicates", "filter_relative_support")
            resident_ms = sum(pipeline.timings[stage]["ms"] for stage in resident_stages if stage in pipeline.timings)
            line = {
                "metric": "chimeric reads/s end-to-end (BAM->fusions.tsv), synthetic",
                "value": total_fragments * args.steps / elapsed,
                "unit": "chimeric reads/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": elapsed / args.steps * 1e3,
                "higher_is_better": True, "scaling": "strong" if one_sample else "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
                "config": {"workload": "synthetic %d chimeric fragments " % args.fragments + ("in one sample" if one_sample else "per GPU") + " (%d BAM records, %.1f GB uncompressed BGZF; 2x100 bp, 24-contig synthetic genome, GENCODE-like GTF)%s, default filters, BAM file in memory -> fusions.tsv%s"
                                       % (pipeline.records if through_workflow_library else pipeline.ingest_result.records if pipeline.ingest_result else -1, bam_bytes / 1e9, (", mismapper stress (clips of 40-70 nt copied from the partner gene, -U %d)" % subsampling) if args.stress else ((", -U %d" % subsampling) if subsampling != 300 else "") + ((", read names of %d characters" % args.name_length) if args.name_length else ""),
                                          " + discarded.tsv" if args.discarded else ""),
                           "fragments_per_gpu": n, "candidates": pipeline.n_candidates, "fusions": fusion_lines,
                           "read_chimeric_alignments": "host ingest (multi-threaded) + upload" if args.host_ingest else "on the device (agpu_ingest.hip), the host feeds the bytes of the file",
                           "timed_call": ("arriba_workflow_sample of libarriba_workflow.so (C++ over the two C ABIs; resident session)" + ("; samples in a queue: arriba_workflow_submit(next) before arriba_workflow_sample(current), so the file of the next sample is fed under the stages of the current one -- every timed step is one whole sample, BAM file -> fusions.tsv, and carries the feed of its successor; fusions.tsv of a sample is formatted and written by a thread of the session beside the next sample (arriba_workflow_defer_output), the last one complete before the clock stops (arriba_workflow_flush)" + ("; the ingest of a sample is finished by the thread that feeds it, beside the stages of the sample in front (arriba_workflow_finish_ahead)" if finish_ahead[0] else "") if pipelined else "; one sample at a time")) if through_workflow_library else "the ctypes mirror of the stage order (arriba_amd/pipeline.py)",
                           "parallelism": ((("one sample over %d GPUs in the C++ driver (arriba_workflow_sample as a collective call; %s), THE READS SHARDED: every rank ingests its part of the file and keeps its fragments (this rank: %d of %d) through the read-level cascade, ONE all-gather of the emissions of find_fusions, candidates and read lists built on every rank, one byte of state per fragment replicated when filters change, "
                                             "multi-mapper scores / re-alignments of filter_mismappers / clipped mates of filter_in_vitro where the reads are, rows of the written candidates gathered from their ranks, rows formatted by all ranks, rank 0 writes; exchanges %.3f s, %.2f GB received by this rank per sample"
                                             % (world, collectives[0], int(pipeline.timing.get("shard_fragments", 0)), n, ingest_parts[-1].get("exchange_parts", 0.0), pipeline.timing.get("exchanged_bytes", 0) / 1e9)) if pipeline.timing.get("shard_fragments", 0) > 0 else
                                            ("one sample over %d GPUs in the C++ driver (arriba_workflow_sample as a collective call; %s): every rank ingests its part of the file, one all-gather of the parts (%.3f s with export and merge), the stages on every rank, filter_mismappers shared out (one all-reduce of the verdict bytes), the rows of the files formatted by all ranks, rank 0 writes (the names of the file are not in order, or ARRIBA_RANKS_SPLIT=replicated)"
                                             % (world, collectives[0], ingest_parts[-1].get("exchange_parts", 0.0)))) if through_workflow_library else
                                           ("one sample over %d GPUs: every rank ingests its part of the file, one all-gather of the parts (%s), filter_mismappers shared out (one all-reduce of %d verdict bytes), rank 0 writes"
                                            % (world, "%.2f GB per rank" % (max(pipeline.exchange["part_bytes"]) / 1e9), pipeline.exchange.get("mismapper_jobs", 0)))) if one_sample
                                          else ("%d samples, one per GPU, no collective on the data path" % world) if distributed else "1 GPU",
                           "outside_the_step": "loading assembly + annotation (ahost_open), device context; generating the sample took %.1f s" % generate_seconds,
                           "names_were_sorted": bool(pipeline.ingest_result.names_were_sorted) if pipeline.ingest_result else None,
                           "why_not_the_100M_sample": fallback_reason, "large_sample": large_sample_record},
                "seconds_per_step": dict({"read_chimeric_alignments": round(mean("ingest"), 4), "workflow_to_output_files": round(mean("workflow"), 4), "total": round(mean("total"), 4)},
                                         **({key: round(mean(key), 4) for key in ("stages", "filter_mismappers", "output")} if through_workflow_library else {})),
                "read_chimeric_alignments_seconds": {key: round(sum(p.get(key, 0.0) for p in ingest_parts) / len(ingest_parts), 4) for key in ingest_parts[-1]},
                "output_side_seconds": getattr(pipeline, "writer_seconds", None),
                "latency_s": sample_alone["seconds"] if sample_alone else None,  # one sample alone, BAM file -> fusions.tsv, nothing of another sample beside it (`value` is the throughput of samples in a queue)
                "hbm_used_GB": hbm_used_gb,  # (of the device, behind the timed steps: every buffer of the session is grow-only and stays)
                "samples_pipelined": bool(pipelined), "ingest_finished_ahead": bool(finish_ahead[0]), "output_deferred": bool(pipelined and not args.no_deferred_output), "deferred_writer_seconds": deferred_writer_seconds, "one_sample_alone": sample_alone,
                "bam_GB_per_s_end_to_end": bam_bytes / mean("total") / 1e9,
                "stages": stage_log,
                # per step: the sum over the launches of one step (the front of the ingest runs window by window: ~200 launches of its kernels in a step of 10^8 fragments)
                "kernel_ms": {name: round(values["ms"] / args.steps, 3) for name, values in sorted(kernels.items(), key=lambda item: -item[1]["ms"])[:48]},
                # ... and of the sample that ran alone behind the timed steps: what every kernel takes when nothing of another sample runs beside it (their sum is the device work of a sample)
                "kernel_ms_alone": ({name: round(values["ms"], 3) for name, values in sorted(alone_kernels.items(), key=lambda item: -item[1]["ms"])[:48]} if alone_kernels else None),
                "kernel_ms_alone_sum": round(sum(values["ms"] for values in alone_kernels.values()), 1) if alone_kernels else None,
                "kernel_launches_per_step": {name: round(values["launches"] / args.steps, 1) for name, values in sorted(kernels.items(), key=lambda item: -item[1]["ms"])[:12]},
                "kernel_ms_per_step": round(kernel_ms_per_step, 2),
                "roofline": roofline,
            }
            # the search kernel that dominates is bound by the latency of dependent look-ups, not by bandwidth; beside it the best streaming kernel of the step, priced the same way
            # (round 2 printed the fastest one here, a 0.1 ms kernel whose input was still in the caches; now the one the step spends most time in)
            streaming = {name: values["bytes"] / values["ms"] / 1e6 for name, values in modelled.items() if not name.startswith("mismapper_") and name != dominant}
            if streaming:
                best = max(streaming, key=lambda name: modelled[name]["ms"])
                line["roofline_streaming"] = {"bound": "hbm", "kernel": best, "achieved": streaming[best], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": streaming[best] / HBM_PEAK_GBS,
                                              "launch_ms": kernels[best]["ms"] / kernels[best]["launches"], "algorithmic_bytes_per_launch": kernels[best]["bytes"] / kernels[best]["launches"]}
            if not through_workflow_library:  # (arriba_workflow_sample does not time the stages one by one: its own laps are in seconds_per_step)
                line["stage_kernel_ms"] = {stage: round(values["ms"], 3) for stage, values in pipeline.timings.items()}
                line["device_resident_step"] = {"what": "round 1's figure: resident batch -> filter_relative_support (kernel time of the stages between the ingest and the candidate-level filters)", "ms": round(resident_ms, 3),
                                                "chimeric_reads_per_s": n / (resident_ms * 1e-3) if resident_ms > 0 else None}
            line["self_check"] = "every alignment has a gene; fusions.tsv holds the fusions the last stage counted" + ("; " + reference_check if reference_check else
                "; NO file of the reference exists for a sample of this size (it needs ~1 GB per million fragments): parity at this size rests on the same code having produced the reference's fusions.tsv and discarded.tsv "
                "at 10 M and 20 M fragments of config 2 and at 3 M of config 3 (SHA-256, GPU tier: tests/golden/bench10m, bench20m, stress3m) and on the legs of this line giving one another's file")
            progress("self-check done, kernel profile read")
            if through_workflow_library and not distributed and not args.no_deflated_leg:
                progress("the same sample with deflated BGZF blocks")
                line["value_deflated"] = deflated_leg(pipeline, directory, args.fragments, args.stress, outputs[0], finish_ahead=finish_ahead[0])
                if not line["value_deflated"]["fusions_tsv_equals_the_stored_sample's"]:
                    raise SystemExit("bench self-check failed: the deflated sample gives another fusions.tsv than the stored one")
            if through_workflow_library and not distributed and not args.stress and not args.no_normal_pairs:
                progress("the sample with 4 N ordinary proper pairs")
                line["value_with_normal_pairs"] = normal_pairs_leg(pipeline, directory, fragments=min(10000000, args.fragments))
            if args.no_cpu_baseline or distributed:  # (timed at N = 1 only)
                line["cpu_baseline"] = {"value": None, "unit": "chimeric reads/s", "cores": 1, "kind": "reference", "sample": "skipped: the reference is timed by the run with 1 GPU" if distributed else "skipped"}
            else:
                line["cpu_baseline"] = cpu_baseline(1000, directory, stress=args.stress, subsampling=subsampling, timed_fragments=total_fragments, sample_fragments=150000 if args.stress else 800000)  # (config 3: the reference needs 5 1/2 minutes for 1 M fragments)
            if through_workflow_library and not distributed and not args.stress and not args.no_stress_leg and not os.environ.get("ARRIBA_BENCH_NO_STRESS_LEG") and args.fragments >= 10000000:
                progress("BASELINE.json config 3 at the size of the headline sample: the mismapper stress with -U 32767, in a session of its own")
                pipeline.close()  # (the stress sample wants the device to itself: 92.7 G list entries at 10^8 fragments)
                line["value_stress"] = stress_leg(args.fragments)
            print(json.dumps(line))
    finally:
        if one_sample:
            try:
                dist.barrier()  # nobody removes the sample while another rank still reads it
            except Exception:
                pass
        if not args.keep and (rank == 0 or not one_sample):
            shutil.rmtree(directory, ignore_errors=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
