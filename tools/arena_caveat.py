#!/usr/bin/env python3
"""tools/arena_caveat.py -- what the parity claim leaves out (hazard H1; review of round 3, "what's weak" #1).

The reference keeps the genes / exons / transcripts of an alignment in vectors sorted BY POINTER VALUE of std::list nodes
(source/common.hpp:128-146,156-160,180), so wherever it takes "the first gene" of such a set, or loops over gene1 x gene2, heap addresses decide.  The oracle
build serves those nodes from a bump arena (oracle/arena.cpp): pointer order == creation order == GTF order, dummy genes last -- a defined order, the one this
repository reproduces bit for bit.  A stock build of the reference (glibc malloc) has SOME order that depends on the allocator's state; ARRIBA_ORACLE_ARENA=0
makes oracle/_ref/arriba_ref behave like that stock build.  This tool runs the same binary both ways on generated samples and reports which lines of the log
and which rows of fusions.tsv differ, so that the claim can be stated with its bound:

    identical to the reference with allocation-ordered annotation nodes; against a stock build, k of n rows differ on <sample>

Sites of the reference where the order of a pointer-sorted set decides a result (each checked by reading the source):
    source/read_stats.cpp:32          estimate_fragment_length: spliced distance along *genes.begin() of the forward mate -> mate gap mean / stddev -> max_mate_gap
    source/read_stats.cpp:118-122     detect_strandedness: splice site and strand of *genes.begin()
    source/arriba.cpp:290-306         the encompassing dummy gene of an intergenic pair: genes[0]
    source/fusions.cpp:330-331        find_fusions: candidates are inserted gene1 x gene2 in set order -> insertion order of fusions_t -> its iteration order (hazard H2), on which
    source/filter_relative_support.cpp:20-41   the first candidate of an (gene, breakpoint1, breakpoint2) overlap group registers the fusion partner -> partner counts -> e-value,
    source/select_best.cpp, source/recover_isoforms.cpp:10-47, source/filter_homologs.cpp   "first / last in iteration order wins" folds, and the order of equal rows in the output file depend
    source/annotation.cpp:379-429     is_breakpoint_spliced / get_spliced_distance: the first exon of a pointer-sorted exon set that matches

Usage: python tools/arena_caveat.py [--out profiles/r04_arena_caveat.json] [--fragments 400000 ...]   (test tooling; needs oracle/_ref)"""
import argparse
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

SITES = ["source/read_stats.cpp:32", "source/read_stats.cpp:118-122", "source/arriba.cpp:290-306", "source/fusions.cpp:330-331", "source/filter_relative_support.cpp:20-41",
         "source/recover_isoforms.cpp:10-47", "source/annotation.cpp:379-429"]


def stage_lines(log):
    """the lines of the reference's log without time stamps, file names and the resource line"""
    lines = []
    for line in log.splitlines():
        line = re.sub(r"^\[[^\]]*\] ", "", line)
        if line.startswith(("Writing ", "Done ", "Launching ", "Loading ", "Freeing ", "Reading ")) and "total=" not in line:
            continue
        lines.append(line)
    return lines


def compare(prefix, directory, extra=()):
    """runs oracle/_ref/arriba_ref on prefix.{bam,gtf,fa} with the arena (the defined order) and without (stock glibc order); returns the report of the differences"""
    import datasets
    outputs = {}
    for mode in ("arena", "stock"):
        out = os.path.join(directory, mode)
        command = [datasets.ARRIBA_REF, "-x", prefix + ".bam", "-g", prefix + ".gtf", "-a", prefix + ".fa", "-o", out + ".tsv", "-O", out + ".discarded.tsv", "-f", "blacklist"] + list(extra)
        result = subprocess.run(command, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, env=dict(os.environ, ARRIBA_ORACLE_ARENA="1" if mode == "arena" else "0"))
        if result.returncode != 0:
            raise SystemExit("the reference failed (%s):\n%s" % (mode, result.stdout[-2000:]))
        rows = [line.rstrip("\n") for line in open(out + ".tsv") if not line.startswith("#")]
        outputs[mode] = {"log": stage_lines(result.stdout), "rows": rows}
    arena, stock = outputs["arena"], outputs["stock"]
    differing_stages = [{"arena": a, "stock": s} for a, s in zip(arena["log"], stock["log"]) if a != s]
    first = next((k for k, (a, s) in enumerate(zip(arena["log"], stock["log"])) if a != s), None)
    # rows as the reference identifies a fusion: genes, breakpoints, supporting reads, confidence (columns 1-15); the order of the rows is compared separately
    key = lambda row: "\t".join(row.split("\t")[:15])
    arena_rows, stock_rows = set(map(key, arena["rows"])), set(map(key, stock["rows"]))
    return {"rows_arena": len(arena["rows"]), "rows_stock": len(stock["rows"]), "rows_only_with_arena": len(arena_rows - stock_rows), "rows_only_in_stock_build": len(stock_rows - arena_rows),
            "rows_in_both": len(arena_rows & stock_rows), "same_rows_in_another_order": arena_rows == stock_rows and [key(r) for r in arena["rows"]] != [key(r) for r in stock["rows"]],
            "log_lines": len(arena["log"]), "log_lines_that_differ": differing_stages, "first_line_that_differs": arena["log"][first] if first is not None else None,
            "lines_in_front_of_it_identical": first if first is not None else len(arena["log"])}


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--out", default=None)
    parser.add_argument("--fragments", type=int, nargs="*", default=[400000, 2000000])
    args = parser.parse_args()
    import bench
    import datasets
    report = {"what": __doc__.split("\n\n")[1].replace("\n", " "), "sites": SITES, "samples": []}
    directory = tempfile.mkdtemp(prefix="arena_caveat_", dir="/tmp")
    try:
        for fragments in args.fragments:
            prefix = os.path.join(directory, "s%d" % fragments)
            subprocess.run([datasets.GEN_SYNTH, "--out", prefix, "--threads", "4"] + bench.workload_args(fragments, 1000), check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            entry = compare(prefix, directory)
            entry["sample"] = "bench.py's workload (config 2), %d fragments, seed 1000" % fragments
            report["samples"].append(entry)
            print(json.dumps({k: v for k, v in entry.items() if k != "log_lines_that_differ"}))
            os.remove(prefix + ".bam")
        name = "mid30k"
        prefix = datasets.generate(datasets.DATASETS[name], directory)
        entry = compare(prefix, directory)
        entry["sample"] = "tests/datasets.py: " + name
        report["samples"].append(entry)
        print(json.dumps({k: v for k, v in entry.items() if k != "log_lines_that_differ"}))
    finally:
        shutil.rmtree(directory, ignore_errors=True)
    if args.out:
        json.dump(report, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
