"""Parsers for the TSV dumps written by the oracle build of the reference (oracle/ref_hooks.cpp)."""
import glob
import gzip
import os


def _open(path):
    if os.path.exists(path):
        return open(path, "rt")
    return gzip.open(path + ".gz", "rt")


def find_dump(directory, kind, stage, which=0):
    """Path (without .gz) of <kind>.<nn>_<stage>.tsv in a dump directory; a stage that runs twice: which=0 the first, which=-1 the last dump."""
    import re
    pattern = re.compile(r"^%s\.\d+_%s\.tsv(\.gz)?$" % (re.escape(kind), re.escape(stage)))  # the stage name must match completely
    matches = sorted(path for path in glob.glob(os.path.join(directory, "%s.*_%s.tsv*" % (kind, stage))) if pattern.match(os.path.basename(path)))
    if not matches:
        raise FileNotFoundError("%s.*_%s.tsv in %s" % (kind, stage, directory))
    path = matches[which]
    return path[:-3] if path.endswith(".gz") else path


def read_reads(path):
    """-> list of dicts: name, filter, single_end, multimapper, duplicate, alignments[ {supplementary, first_in_pair, exonic, strand,
    predicted_strand, ambiguous, contig, start, end, cigar, sequence, genes[]} ]"""
    reads = []
    with _open(path) as handle:
        for line in handle:
            if line.startswith("#"):
                continue
            f = line.rstrip("\n").split("\t")
            read = {"name": f[0], "filter": int(f[1]), "single_end": int(f[2]), "multimapper": int(f[3]), "duplicate": int(f[4]), "alignments": []}
            for a in range(int(f[5])):
                g = f[6 + 12 * a: 18 + 12 * a]
                read["alignments"].append({
                    "supplementary": int(g[0]), "first_in_pair": int(g[1]), "exonic": int(g[2]), "strand": int(g[3]), "predicted_strand": int(g[4]),
                    "ambiguous": int(g[5]), "contig": int(g[6]), "start": int(g[7]), "end": int(g[8]), "cigar": g[9], "sequence": g[10],
                    "genes": [] if g[11] == "." else [int(x) for x in g[11].split(",")]})
            reads.append(read)
    return reads


def read_filters(path):
    names, filters = [], []
    with _open(path) as handle:
        for line in handle:
            name, value = line.rstrip("\n").split("\t")
            names.append(name)
            filters.append(int(value))
    return names, filters


def read_scalars(path):
    scalars = {}
    with _open(path) as handle:
        for line in handle:
            key, value = line.rstrip("\n").split("\t")
            scalars[key] = value
    return scalars


def read_genes(path):
    genes = []
    with _open(path) as handle:
        for line in handle:
            if line.startswith("#"):
                continue
            f = line.rstrip("\n").split("\t")
            genes.append({"id": int(f[0]), "contig": int(f[1]), "start": int(f[2]), "end": int(f[3]), "strand": int(f[4]), "is_dummy": int(f[5]),
                          "is_protein_coding": int(f[6]), "exonic_length": int(f[7]), "name": f[8], "gene_id": f[9]})
    return genes


FUSION_COLUMNS = ["gene1", "gene2", "contig1", "contig2", "breakpoint1", "breakpoint2", "direction1", "direction2", "filter", "split_reads1", "split_reads2",
                  "discordant_mates", "spliced1", "spliced2", "exonic1", "exonic2", "predicted_strand1", "predicted_strand2", "predicted_strands_ambiguous",
                  "transcript_start", "transcript_start_ambiguous", "confidence", "evalue_bits", "evalue", "anchor_start1", "anchor_start2",
                  "closest_genomic_breakpoint1", "closest_genomic_breakpoint2", "split_read1_list", "split_read2_list", "discordant_mate_list"]


def read_fusions(path):
    fusions = []
    with _open(path) as handle:
        for line in handle:
            if line.startswith("#"):
                continue
            f = line.rstrip("\n").split("\t")
            fusion = {}
            for column, value in zip(FUSION_COLUMNS, f):
                if column.endswith("_list"):
                    fusion[column] = [] if value == "." else value.split(";")
                elif column == "evalue_bits":
                    fusion[column] = int(value, 16)
                elif column == "evalue":
                    fusion[column] = float(value)
                else:
                    fusion[column] = int(value)
            fusions.append(fusion)
    return fusions
