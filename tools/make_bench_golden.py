#!/usr/bin/env python3
"""tools/make_bench_golden.py -- run the UNMODIFIED reference (oracle/_ref/arriba_ref, oracle/Makefile) once on samples of bench.py's
workload (workload_args(N, 1000): BASELINE.json config 2, or config 3 with --stress) where the repository is built, and keep what pins
parity and the CPU baseline at that size: SHA-256 of the BAM file it read and of the fusions.tsv and discarded.tsv it wrote, its log in full, wall seconds
with and without its loading phase, peak memory (the reference's own last line, source/arriba.cpp:616-628).

    python tools/make_bench_golden.py --fragments 20000000 --golden bench20m      # tests/golden/bench20m/{meta.json,reference.log}
    python tools/make_bench_golden.py --fragments 1000000 --fit                   # one point of tests/golden/cpu_baseline_fit.json
    python tools/make_bench_golden.py --fragments 5000000 --normal-mult 4 --golden normal5m   # with 4 N ordinary pairs beside the N chimeric fragments

The points of the fit (a*N + b*N*log2 N over the per-sample seconds, loading excluded) are what bench.py quotes beside the live
800 k-fragment baseline: SURVEY.md section 8(d), "run 1 M / 5 M / 10 M / 20 M, fit, report the extrapolation as such".
Test tooling: nothing of the product imports this."""
import argparse
import hashlib
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
FIT_PATH = os.path.join(ROOT, "tests", "golden", "cpu_baseline_fit.json")


def sha256(path):
    digest = hashlib.sha256()
    with open(path, "rb") as handle:
        for block in iter(lambda: handle.read(1 << 24), b""):
            digest.update(block)
    return digest.hexdigest()


def fit_points(points):
    """least squares of seconds = a*N + b*N*log2(N) through the points [(N, seconds)]"""
    import numpy as np
    n = np.array([p[0] for p in points], dtype=np.float64)
    t = np.array([p[1] for p in points], dtype=np.float64)
    design = np.stack([n, n * np.log2(n)], axis=1)
    (a, b), *_ = np.linalg.lstsq(design, t, rcond=None)
    return float(a), float(b)


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--fragments", type=int, required=True)
    parser.add_argument("--stress", action="store_true")
    parser.add_argument("--golden", help="name of the directory under tests/golden/ that gets meta.json + reference.log")
    parser.add_argument("--fit", action="store_true", help="add the point to tests/golden/cpu_baseline_fit.json and redo the fit")
    parser.add_argument("--scratch", default=None)
    parser.add_argument("--threads", type=int, default=4)
    parser.add_argument("--normal-mult", type=int, default=None, help="ordinary proper pairs per chimeric fragment beside them (SURVEY.md section 8d-2 asks for 4; bench.py's main sample has none)")
    args = parser.parse_args()
    import bench
    import datasets
    directory = args.scratch or tempfile.mkdtemp(prefix="bench_golden_", dir="/tmp")
    os.makedirs(directory, exist_ok=True)
    prefix = os.path.join(directory, "s")
    generator = bench.workload_args(args.fragments, 1000, stress=args.stress)
    if args.normal_mult is not None:
        generator[generator.index("--normal-mult") + 1] = str(args.normal_mult)
    subprocess.run([datasets.GEN_SYNTH, "--out", prefix, "--threads", str(args.threads)] + generator, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    bam_sha = sha256(prefix + ".bam")
    # (-O: the discarded candidates as well -- review of round 4, item 7c: the SHA pins of the large samples did not pin discarded.tsv)
    command = [datasets.ARRIBA_REF, "-x", "s.bam", "-g", "s.gtf", "-a", "s.fa", "-o", "ref_fusions.tsv", "-O", "ref_discarded.tsv", "-f", "blacklist"] + (["-U", "32767"] if args.stress else [])
    returncode, log, elapsed, loading, total = bench.run_reference(command, cwd=directory)
    if returncode != 0:
        raise SystemExit("the reference failed:\n" + log[-2000:])
    peak = re.search(r"peak memory=([0-9.]+)gb", log)
    fusions = sum(1 for line in open(os.path.join(directory, "ref_fusions.tsv")) if not line.startswith("#"))
    meta = {
        "what": "BASELINE.json config %s exactly as bench.py generates it (workload_args(%d, 1000%s)): the UNMODIFIED reference (oracle/_ref/arriba_ref, built by oracle/Makefile) run once on this sample in the build container; its fusions.tsv is kept by its SHA-256, its log in full"
                % ("3" if args.stress else "2", args.fragments, ", stress=True" if args.stress else ""),
        "generator": "arriba_amd/lib/gen_synth " + " ".join(generator) + " (any --threads: the sample does not depend on them)",
        "command": " ".join(["oracle/_ref/arriba_ref"] + command[1:]),
        "bam_sha256": bam_sha,
        "fusions_tsv_sha256": sha256(os.path.join(directory, "ref_fusions.tsv")),
        "fusions": fusions,
        "discarded_tsv_sha256": sha256(os.path.join(directory, "ref_discarded.tsv")),
        "discarded": sum(1 for line in open(os.path.join(directory, "ref_discarded.tsv")) if not line.startswith("#")),
        "chimeric_fragments": total,
        "reference_seconds_in_the_build_container": round(elapsed, 1),
        "reference_loading_seconds": round(loading, 1),
        "reference_peak_memory_gb": float(peak.group(1)) if peak else None,
    }
    print(json.dumps(meta, indent=1))
    if args.golden:
        target = os.path.join(ROOT, "tests", "golden", args.golden)
        os.makedirs(target, exist_ok=True)
        json.dump(meta, open(os.path.join(target, "meta.json"), "w"), indent=1)
        lines = log.splitlines()
        kept = [line for line in lines if not line.startswith("WARNING: encountered early stop codon")]  # (the synthetic GTF's coding sequences are random: hundreds of these)
        if len(kept) < len(lines):
            kept.append("(%d lines 'WARNING: encountered early stop codon in transcript ...' of the reference's stderr left out)" % (len(lines) - len(kept)))
        open(os.path.join(target, "reference.log"), "w").write("\n".join(kept) + "\n")
    if args.fit:
        record = json.load(open(FIT_PATH)) if os.path.exists(FIT_PATH) else {"points": []}
        key = "stress" if args.stress else "config2"
        record["points"] = [p for p in record["points"] if not (p["fragments"] == args.fragments and p["workload"] == key)]
        record["points"].append({"workload": key, "fragments": args.fragments, "chimeric_fragments": meta["chimeric_fragments"], "seconds": meta["reference_seconds_in_the_build_container"],
                                 "loading_seconds": meta["reference_loading_seconds"], "peak_memory_gb": meta["reference_peak_memory_gb"], "fusions": fusions})
        record["points"].sort(key=lambda p: (p["workload"], p["fragments"]))
        for workload in sorted(set(p["workload"] for p in record["points"])):
            points = [(p["chimeric_fragments"], p["seconds"] - p["loading_seconds"]) for p in record["points"] if p["workload"] == workload]
            if len(points) >= 2:
                a, b = fit_points(points)
                record["fit_" + workload] = {"model": "seconds_without_loading = a*N + b*N*log2(N), N = chimeric fragments (total=N of the reference's log)", "a": a, "b": b,
                                             "extrapolated_seconds_at_1e8": a * 1e8 + b * 1e8 * 26.575424759098897, "points_used": len(points)}
        record["where"] = "the build container (8 host cores visible, the reference uses 1); source/arriba.cpp:616-628 prints the elapsed time"
        json.dump(record, open(FIT_PATH, "w"), indent=1)
    if not args.scratch:
        shutil.rmtree(directory, ignore_errors=True)


if __name__ == "__main__":
    main()
