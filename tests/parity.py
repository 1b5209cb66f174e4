"""Comparison helpers shared by the CPU-tier (host stepping harness) and GPU-tier (HIP kernels) parity tests."""
import os

import numpy as np

import datasets
import golden_io


def run_read_level(session_factory, prefix, api=None):
    from arriba_amd.pipeline import DevicePipeline
    session = session_factory(prefix)
    pipeline = DevicePipeline(session, api=api)
    pipeline.run_read_level()
    return session, pipeline


def open_session(prefix):
    from arriba_amd.pipeline import HostSession
    session = HostSession(prefix + ".fa", prefix + ".gtf")
    session.read_chimeric_alignments(prefix + ".bam")
    return session


def check_ingest(session, golden):
    """packed batch vs the reference's read table after annotation (ingest-time fields only)"""
    reads = golden_io.read_reads(golden_io.find_dump(golden, "reads", "annotated"))
    names = session.fragment_names()
    assert names == [r["name"] for r in reads]
    view = session.batch_view.contents
    ops = "MIDNSHP=XB"
    codes = "=ACMGRSVTWYHKDBN"
    pool = np.ctypeslib.as_array(view.seq_pool, shape=(max(int(view.seq_pool_size), 1),))
    cigar_pool = np.ctypeslib.as_array(view.cigar_pool, shape=(max(int(view.cigar_pool_size), 1),))
    for i, read in enumerate(reads):
        assert view.n_aln[i] == len(read["alignments"]), read["name"]
        assert (view.fbits[i] & 1) == read["single_end"] and ((view.fbits[i] >> 2) & 1) == read["duplicate"]
        for slot, alignment in enumerate(read["alignments"]):
            bits = view.abits[slot][i]
            assert (bits & 1, (bits >> 1) & 1, (bits >> 2) & 1) == (alignment["strand"], alignment["first_in_pair"], alignment["supplementary"]), read["name"]
            assert (view.contig[slot][i], view.start[slot][i], view.end[slot][i]) == (alignment["contig"], alignment["start"], alignment["end"]), read["name"]
            offset, count = view.cigar_offset[slot][i], view.cigar_count[slot][i]
            cigar = "".join("%d%s" % (c >> 4, ops[c & 15]) for c in cigar_pool[offset:offset + count])
            assert cigar == alignment["cigar"], read["name"]
            if slot < 2:
                length = view.seq_length[slot][i]
                packed = pool[view.seq_offset[slot][i] * 4: view.seq_offset[slot][i] * 4 + (length + 1) // 2]
                sequence = "".join(codes[(packed[b >> 1] >> ((~b & 1) << 2)) & 15] for b in range(length))
                assert sequence == alignment["sequence"], read["name"]
    return len(reads)


def check_annotation(session, pipeline, golden):
    reads = golden_io.read_reads(golden_io.find_dump(golden, "reads", "annotated"))
    names = session.fragment_names()
    assert names == [r["name"] for r in reads]
    mismatches = []
    for slot in range(3):
        count, genes = pipeline.gene_sets(slot)
        bits = pipeline.alignment_bits(slot)
        offsets = np.concatenate([[0], np.cumsum(count.astype(np.int64))]).astype(np.int64)
        for i, read in enumerate(reads):
            if slot >= len(read["alignments"]):
                continue
            alignment = read["alignments"][slot]
            mine = [int(g) for g in genes[offsets[i]:offsets[i + 1]]]
            exonic, ambiguous = (bits[i] >> 3) & 1, (bits[i] >> 5) & 1
            predicted = 0 if ambiguous else (bits[i] >> 4) & 1
            if mine != alignment["genes"] or (exonic, ambiguous, predicted) != (alignment["exonic"], alignment["ambiguous"], alignment["predicted_strand"]):
                mismatches.append((read["name"], slot, mine, alignment["genes"]))
    assert not mismatches, mismatches[:10]
    fragment_bits = pipeline.fragment_bits()
    assert [int((b >> 1) & 1) for b in fragment_bits] == [r["multimapper"] for r in reads]


def check_gene_table(pipeline, golden):
    table = pipeline.gene_table()
    genes = golden_io.read_genes(os.path.join(golden, "genes.tsv"))
    assert len(genes) == len(table["start"])
    for g in genes:
        i = g["id"]
        assert (table["contig"][i], table["start"][i], table["end"][i], table["bits"][i] & 1, (table["bits"][i] >> 1) & 1, (table["bits"][i] >> 2) & 1, table["exonic_length"][i]) == \
               (g["contig"], g["start"], g["end"], g["strand"], g["is_dummy"], g["is_protein_coding"], g["exonic_length"]), g


def check_read_filters(session, pipeline, golden):
    names, filters = golden_io.read_filters(golden_io.find_dump(golden, "filters", "read_filters_final"))
    assert session.fragment_names() == names
    mine = pipeline.filters()
    different = [(names[i], int(mine[i]), filters[i]) for i in range(len(names)) if mine[i] != filters[i]]
    assert not different, different[:10]


def check_scalars(pipeline, golden):
    scalars = golden_io.read_scalars(os.path.join(golden, "scalars.tsv"))
    assert pipeline.scalars["marked_multimappers"] == int(scalars["marked_multimappers"])
    assert pipeline.scalars["strandedness"] == int(scalars["strandedness"])
    assert pipeline.scalars["max_mate_gap"] == int(scalars["max_mate_gap"])
    assert int(pipeline.scalars["estimated"]) == int(scalars["fragment_length_estimated"])
    if pipeline.scalars["estimated"]:
        bits = lambda value: int(np.float32(value).view(np.uint32))
        assert bits(pipeline.scalars["mate_gap_mean"]) == int(scalars["mate_gap_mean_bits"], 16)
        assert bits(pipeline.scalars["mate_gap_stddev"]) == int(scalars["mate_gap_stddev_bits"], 16)
        assert bits(pipeline.scalars["read_length_mean"]) == int(scalars["read_length_mean_bits"], 16)
    log = open(os.path.join(golden, "reference.log")).read()
    assert pipeline.remaining == datasets.parse_remaining(log)


def check_candidates(session, pipeline, golden):
    """candidate table after find_fusions vs the reference's fusions_t dump (every field and the three read lists)"""
    fusions = golden_io.read_fusions(golden_io.find_dump(golden, "fusions", "find_fusions"))
    table = pipeline.candidates()
    assert len(fusions) == pipeline.n_candidates
    names = session.fragment_names()
    offsets = table["list_offset"].astype(np.int64)
    lists = table["read_lists"]
    mine = {}
    for c in range(pipeline.n_candidates):
        flags = int(table["flags"][c])
        key = (int(table["gene1"][c]), int(table["gene2"][c]), int(table["contigs"][c]) >> 16, int(table["contigs"][c]) & 0xFFFF,
               int(table["breakpoint1"][c]), int(table["breakpoint2"][c]), flags & 1, (flags >> 1) & 1)
        mine[key] = c
    assert len(mine) == pipeline.n_candidates
    problems = []
    with_lists = None
    for f in fusions:
        key = (f["gene1"], f["gene2"], f["contig1"], f["contig2"], f["breakpoint1"], f["breakpoint2"], f["direction1"], f["direction2"])
        if key not in mine:
            problems.append(("missing", key))
            continue
        c = mine[key]
        flags = int(table["flags"][c])
        got = {"filter": int(table["filter"][c]), "split_reads1": int(table["split_reads1"][c]), "split_reads2": int(table["split_reads2"][c]),
               "discordant_mates": int(table["discordant_mates"][c]), "exonic1": (flags >> 2) & 1, "exonic2": (flags >> 3) & 1, "spliced1": (flags >> 4) & 1, "spliced2": (flags >> 5) & 1,
               "predicted_strand1": (flags >> 6) & 1, "predicted_strand2": (flags >> 7) & 1, "predicted_strands_ambiguous": (flags >> 8) & 1,
               "transcript_start": (flags >> 9) & 1, "transcript_start_ambiguous": (flags >> 10) & 1,
               "anchor_start1": int(table["anchor_start1"][c]), "anchor_start2": int(table["anchor_start2"][c])}
        for field, value in got.items():
            if value != f[field]:
                problems.append((field, key, value, f[field]))
        for k, field in enumerate(("split_read1_list", "split_read2_list", "discordant_mate_list")):
            reads = lists[offsets[3 * c + k]:offsets[3 * c + k + 1]]
            expected = f[field]
            if len(expected) == 1 and expected[0].isdigit():  # dumps written with ARRIBA_ORACLE_DUMP_LISTS=0 hold the list size only
                if len(reads) != int(expected[0]):
                    problems.append((field + ".size", key, len(reads), int(expected[0])))
            elif [names[r] for r in reads] != expected:
                problems.append((field, key))
    assert not problems, problems[:10]
    return len(fusions)


def candidate_keys(table, n):
    return [(int(table["gene1"][c]), int(table["gene2"][c]), int(table["contigs"][c]) >> 16, int(table["contigs"][c]) & 0xFFFF, int(table["breakpoint1"][c]), int(table["breakpoint2"][c]),
             int(table["flags"][c]) & 1, (int(table["flags"][c]) >> 1) & 1) for c in range(n)]


def fusion_key(f):
    return (f["gene1"], f["gene2"], f["contig1"], f["contig2"], f["breakpoint1"], f["breakpoint2"], f["direction1"], f["direction2"])


def check_evalues(session, pipeline, golden):
    """estimate_expected_fusions + filter_relative_support against the reference's dumps.  The stages between find_fusions and the e-value
    (merge_adjacent_fusions, filter_multimappers) are not part of this check: their effect on the candidate columns is taken from the dump."""
    after_evalue = golden_io.read_fusions(golden_io.find_dump(golden, "fusions", "estimate_expected_fusions"))
    table = pipeline.candidates()
    n = pipeline.n_candidates
    assert len(after_evalue) == n
    index = {key: c for c, key in enumerate(candidate_keys(table, n))}
    # hazard H2: the dump is written in the iteration order of the reference's unordered_map
    rank = pipeline.candidate_iteration_order()
    assert np.array_equal(rank, pipeline.candidate_iteration_order_on_host(table))  # literal std::unordered_map on the host
    assert [index[fusion_key(f)] for f in after_evalue] == [int(c) for c in np.argsort(rank)]
    state = {k: np.zeros(n, dtype=np.uint32) for k in ("filter", "split_reads1", "split_reads2", "discordant_mates")}
    expected = np.zeros(n, dtype=np.uint32)
    for f in after_evalue:
        c = index[fusion_key(f)]
        for k in state:
            state[k][c] = f[k]
        expected[c] = f["evalue_bits"]
        flags = int(table["flags"][c])  # the flag columns are not touched by the intermediate stages
        assert ((flags >> 4) & 1, (flags >> 5) & 1, (flags >> 2) & 1, (flags >> 3) & 1) == (f["spliced1"], f["spliced2"], f["exonic1"], f["exonic2"])
    pipeline.set_candidate_state(state["filter"].astype(np.uint8), state["split_reads1"], state["split_reads2"], state["discordant_mates"])
    evalue = pipeline.estimate_expected_fusions()
    bits = evalue.view(np.uint32)
    different = [(c, hex(int(bits[c])), hex(int(expected[c]))) for c in range(n) if bits[c] != expected[c]]
    assert not different, (len(different), different[:10])
    # the stages main() runs next (source/arriba.cpp:437-460): filter_non_coding_neighbors, filter_intragenic_both_exonic, filter_min_support,
    # filter_relative_support; the device continues from the state it has (the dump of estimate_expected_fusions)
    import re
    log = open(os.path.join(golden, "reference.log")).read()
    unfiltered = int((state["filter"] == 0).sum())
    discarded = pipeline.filter_candidate_predicates()
    for name, pattern in (("non_coding_neighbors", "adjacent non-coding"), ("intragenic_exonic", "intragenic fusions with both breakpoints in exonic"), ("min_support", r"supporting reads")):
        unfiltered -= discarded[name]
        match = re.search(r"Filtering[^\n]*%s[^\n]*\(remaining=(\d+)\)" % pattern, log)
        assert match and unfiltered == int(match.group(1)), (name, unfiltered, match and match.group(1))
    remaining = pipeline.filter_relative_support()
    after_filter = golden_io.read_fusions(golden_io.find_dump(golden, "fusions", "filter_relative_support"))
    expected_filter = np.zeros(n, dtype=np.uint8)
    for f in after_filter:
        expected_filter[index[fusion_key(f)]] = f["filter"]
        assert f["evalue_bits"] == expected[index[fusion_key(f)]]
    assert np.array_equal(pipeline.candidates()["filter"], expected_filter)
    match = re.search(r"Filtering fusions with an e-value[^\n]*\(remaining=(\d+)\)", log)
    assert match and remaining == int(match.group(1))
    return n


def check_mismappers(session, pipeline, golden):
    """make_kmer_index + filter_mismappers against the reference's dumps right before / after filter_mismappers.  The candidate and
    read state the event-level host stages produced in between is taken from the `before` dumps."""
    import re
    before = golden_io.read_fusions(golden_io.find_dump(golden, "fusions", "before_filter_mismappers"))
    after = golden_io.read_fusions(golden_io.find_dump(golden, "fusions", "filter_mismappers"))
    names, read_filters_before = golden_io.read_filters(golden_io.find_dump(golden, "filters", "before_filter_mismappers"))
    _, read_filters_after = golden_io.read_filters(golden_io.find_dump(golden, "filters", "filter_mismappers"))
    assert session.fragment_names() == names
    scalars = golden_io.read_scalars(os.path.join(golden, "scalars.tsv"))
    table = pipeline.candidates()
    n = pipeline.n_candidates
    index = {key: c for c, key in enumerate(candidate_keys(table, n))}
    state = {k: np.zeros(n, dtype=np.uint32) for k in ("filter", "split_reads1", "split_reads2", "discordant_mates")}
    for f in before:
        for k in state:
            state[k][index[fusion_key(f)]] = f[k]
    pipeline.set_candidate_state(state["filter"].astype(np.uint8), state["split_reads1"], state["split_reads2"], state["discordant_mates"])
    pipeline.set_read_filters(np.array(read_filters_before, dtype=np.uint8))
    positions = pipeline.make_kmer_index(int(scalars["kmer_index_padding"]))
    expected_positions = sum(int(value.split(",")[1]) for key, value in scalars.items() if key.startswith("kmer_index.") and key != "kmer_index_padding")
    assert positions == expected_positions, (positions, expected_positions)
    remaining, discarded = pipeline.filter_mismappers(int(scalars["max_mate_gap"]))
    mine = pipeline.filters()
    different = [(names[i], int(mine[i]), read_filters_after[i]) for i in range(len(names)) if mine[i] != read_filters_after[i]]
    assert not different, (len(different), different[:10])
    assert discarded == sum(1 for a, b in zip(read_filters_before, read_filters_after) if a != b)
    result = pipeline.candidates()
    problems = []
    for f in after:
        c = index[fusion_key(f)]
        got = (int(result["filter"][c]), int(result["split_reads1"][c]), int(result["split_reads2"][c]), int(result["discordant_mates"][c]))
        if got != (f["filter"], f["split_reads1"], f["split_reads2"], f["discordant_mates"]):
            problems.append((fusion_key(f), got, (f["filter"], f["split_reads1"], f["split_reads2"], f["discordant_mates"])))
    assert not problems, (len(problems), problems[:10])
    log = open(os.path.join(golden, "reference.log")).read()
    match = re.search(r"Re-aligning chimeric reads[^\n]*\(remaining=(\d+)\)", log)
    assert match and remaining == int(match.group(1))
    return discarded


def logged_remaining(log, pattern, which=0):
    """the "(remaining=N)" the reference's log prints for the stage whose line matches `pattern` (warnings of the stage may sit between "=" and N)"""
    import re
    return int(re.findall(pattern + r"[^\n]*\(remaining=(?:WARNING:[^\n]*\n)*(\d+)\)", log)[which])


def _inject_candidate_state(pipeline, index, fusions):
    n = pipeline.n_candidates
    state = {k: np.zeros(n, dtype=np.uint32) for k in ("filter", "split_reads1", "split_reads2", "discordant_mates")}
    for f in fusions:
        for k in state:
            state[k][index[fusion_key(f)]] = f[k]
    pipeline.set_candidate_state(state["filter"].astype(np.uint8), state["split_reads1"], state["split_reads2"], state["discordant_mates"])


def _compare_candidate_filters(pipeline, index, expected, stage):
    table = pipeline.candidates()
    problems = []
    for f in expected:
        c = index[fusion_key(f)]
        if int(table["filter"][c]) != f["filter"]:
            problems.append((stage, fusion_key(f), int(table["filter"][c]), f["filter"]))
    assert not problems, (len(problems), problems[:10])
    return sum(1 for f in expected if f["filter"] == 0)


def check_event_predicates(session, pipeline, golden):
    """filter_both_intronic from the reference's state behind recover_internal_tandem_duplication; filter_short_anchor -> filter_end_to_end ->
    filter_no_coverage chained from the state behind recover_many_spliced.  The state of the host stages in between (candidate filters and
    counters, read filters) is taken from the reference's dumps, as for filter_mismappers."""
    import re
    log = open(os.path.join(golden, "reference.log")).read()
    def logged(pattern):
        match = re.search(pattern + r"[^\n]*\(remaining=(\d+)\)", log)
        assert match, pattern
        return int(match.group(1))
    pipeline.find_fusions()
    pipeline.upload_coverage()
    table = pipeline.candidates()
    index = {key: c for c, key in enumerate(candidate_keys(table, pipeline.n_candidates))}
    # filter_both_intronic
    before = golden_io.read_fusions(golden_io.find_dump(golden, "fusions", "recover_internal_tandem_duplication"))
    names, read_filters = golden_io.read_filters(golden_io.find_dump(golden, "filters", "recover_internal_tandem_duplication"))
    assert session.fragment_names() == names
    _inject_candidate_state(pipeline, index, before)
    pipeline.set_read_filters(np.array(read_filters, dtype=np.uint8))
    remaining = pipeline.filter_both_intronic()
    after = golden_io.read_fusions(golden_io.find_dump(golden, "fusions", "filter_both_intronic"))
    _compare_candidate_filters(pipeline, index, after, "both_intronic")
    assert remaining == logged("Filtering fusions with both breakpoints in intronic/intergenic regions"), (remaining, logged("Filtering fusions with both breakpoints in intronic/intergenic regions"))
    discarded = {"both_intronic": sum(1 for f in after if f["filter"] == 13)}
    # filter_in_vitro from the state behind filter_both_intronic (read filters as above)
    try:
        after = golden_io.read_fusions(golden_io.find_dump(golden, "fusions", "filter_in_vitro"))
    except FileNotFoundError:
        after = None  # in_vitro switched off in this reference run
    if after is not None:
        _inject_candidate_state(pipeline, index, golden_io.read_fusions(golden_io.find_dump(golden, "fusions", "filter_both_intronic")))
        quantile = float(re.search(r"expression above the ([0-9.]+)% quantile", log).group(1)) / 100
        remaining = pipeline.filter_in_vitro(quantile)
        _compare_candidate_filters(pipeline, index, after, "filter_in_vitro")
        assert remaining == logged("Filtering in vitro-generated fusions"), remaining
        discarded["filter_in_vitro"] = sum(1 for f in after if f["filter"] == 22) - sum(1 for f in golden_io.read_fusions(golden_io.find_dump(golden, "fusions", "filter_both_intronic")) if f["filter"] == 22)
    # recover_both_spliced from the state behind filter_in_vitro
    try:
        state = golden_io.read_fusions(golden_io.find_dump(golden, "fusions", "filter_in_vitro"))
        after = golden_io.read_fusions(golden_io.find_dump(golden, "fusions", "recover_both_spliced"))
    except FileNotFoundError:
        after = None
    if after is not None:
        _inject_candidate_state(pipeline, index, state)
        remaining = pipeline.recover_both_spliced()
        _compare_candidate_filters(pipeline, index, after, "recover_both_spliced")
        assert remaining == logged("Searching for fusions with spliced split reads"), remaining
        before_filter = {fusion_key(f): f["filter"] for f in state}
        discarded["recover_both_spliced"] = sum(1 for f in after if f["filter"] == 0 and before_filter[fusion_key(f)] != 0)  # candidates recovered
    # select_most_supported_breakpoints (its first call, source/arriba.cpp:497-500) from the state behind recover_both_spliced
    try:
        after = golden_io.read_fusions(golden_io.find_dump(golden, "fusions", "select_most_supported_breakpoints"))
    except FileNotFoundError:
        after = None  # select_best switched off in this reference run
    if after is not None:
        _inject_candidate_state(pipeline, index, golden_io.read_fusions(golden_io.find_dump(golden, "fusions", "recover_both_spliced")))
        remaining = pipeline.select_most_supported_breakpoints()
        _compare_candidate_filters(pipeline, index, after, "select_most_supported_breakpoints")
        assert remaining == logged("Selecting best breakpoints from genes with multiple breakpoints"), remaining
        discarded["select_most_supported_breakpoints"] = sum(1 for f in after if f["filter"] == 24)
    # filter_marginal_read_through from the state behind select_most_supported_breakpoints (its first call, source/arriba.cpp:497-500)
    try:
        before = golden_io.find_dump(golden, "fusions", "select_most_supported_breakpoints")
    except FileNotFoundError:  # select_best switched off in this reference run
        before = golden_io.find_dump(golden, "fusions", "recover_both_spliced")
    _inject_candidate_state(pipeline, index, golden_io.read_fusions(before))
    remaining = pipeline.filter_marginal_read_through()
    after = golden_io.read_fusions(golden_io.find_dump(golden, "fusions", "filter_marginal_read_through"))
    _compare_candidate_filters(pipeline, index, after, "filter_marginal_read_through")
    assert remaining == logged("Filtering read-through fusions with breakpoints near the gene boundary"), remaining
    discarded["filter_marginal_read_through"] = sum(1 for f in after if f["filter"] == 25)
    # recover_many_spliced -> filter_short_anchor -> filter_end_to_end -> filter_no_coverage, chained from the state behind filter_marginal_read_through
    _inject_candidate_state(pipeline, index, golden_io.read_fusions(golden_io.find_dump(golden, "fusions", "filter_marginal_read_through")))
    before_recovery = pipeline.candidates()["filter"].copy()
    min_spliced_events = int(re.search(r"Searching for fusions with >=(\d+) spliced events", log).group(1))  # -M of the reference run
    for stage, run, filter_id, pattern in (("recover_many_spliced", lambda: pipeline.recover_many_spliced(min_spliced_events), None, "Searching for fusions with >=\\d+ spliced events"),
                                           ("filter_short_anchor", pipeline.filter_short_anchor, 26, "Filtering fusions with anchors"),
                                           ("filter_end_to_end_fusions", pipeline.filter_end_to_end, 21, "Filtering end-to-end fusions with low support"),
                                           ("filter_no_coverage", pipeline.filter_no_coverage, 27, "Filtering fusions with no coverage around the breakpoints")):
        remaining = run()
        after = golden_io.read_fusions(golden_io.find_dump(golden, "fusions", stage))
        _compare_candidate_filters(pipeline, index, after, stage)
        assert remaining == logged(pattern), (stage, remaining, logged(pattern))
        if filter_id is None:
            discarded[stage] = int(((before_recovery != 0) & (pipeline.candidates()["filter"] == 0)).sum())  # candidates recovered
        else:
            discarded[stage] = sum(1 for f in after if f["filter"] == filter_id)
    return discarded


def check_recover_itd(session, pipeline, golden):
    """recover_internal_tandem_duplication from the reference's state behind filter_relative_support: candidate filters and counters, the filter
    of every read afterwards, the remaining count.  The pipeline must have run find_fusions + merge_adjacent_fusions (merged read lists)."""
    import re
    log = open(os.path.join(golden, "reference.log")).read()
    pipeline.upload_coverage()
    table = pipeline.candidates()
    index = {key: c for c, key in enumerate(candidate_keys(table, pipeline.n_candidates))}
    before = golden_io.read_fusions(golden_io.find_dump(golden, "fusions", "filter_relative_support"))
    names, read_filters = golden_io.read_filters(golden_io.find_dump(golden, "filters", "filter_multimappers"))
    assert session.fragment_names() == names
    _inject_candidate_state(pipeline, index, before)
    pipeline.set_read_filters(np.array(read_filters, dtype=np.uint8))
    parameters = re.search(r"Searching for internal tandem duplications <=\d+bp with >=(\d+) supporting reads and >=([0-9.]+)% allele fraction[^\n]*\(remaining=(\d+)\)", log)
    remaining = pipeline.recover_internal_tandem_duplication(int(parameters.group(1)), float(parameters.group(2)) / 100)
    after = golden_io.read_fusions(golden_io.find_dump(golden, "fusions", "recover_internal_tandem_duplication"))
    result = pipeline.candidates()
    problems = []
    for f in after:
        c = index[fusion_key(f)]
        got = (int(result["filter"][c]), int(result["split_reads1"][c]), int(result["split_reads2"][c]), int(result["discordant_mates"][c]))
        if got != (f["filter"], f["split_reads1"], f["split_reads2"], f["discordant_mates"]):
            problems.append((fusion_key(f), got, (f["filter"], f["split_reads1"], f["split_reads2"], f["discordant_mates"])))
    assert not problems, (len(problems), problems[:10])
    assert remaining == int(parameters.group(3)), (remaining, parameters.group(3))
    _, read_filters_after = golden_io.read_filters(golden_io.find_dump(golden, "filters", "recover_internal_tandem_duplication"))
    mine = pipeline.filters()
    different = [(names[i], int(mine[i]), read_filters_after[i]) for i in range(len(names)) if mine[i] != read_filters_after[i]]
    assert not different, (len(different), different[:10])
    before_filter = {fusion_key(f): f["filter"] for f in before}
    return sum(1 for f in after if f["filter"] == 0 and before_filter[fusion_key(f)] != 0), sum(1 for a, b in zip(read_filters, read_filters_after) if a != b)


def check_chain_to_no_coverage(session, pipeline, golden, rules_prefix=None):
    """The reference's stages 18-35 (find_fusions ... filter_no_coverage, default filters) on the device in one go, nothing taken from the reference:
    every "(remaining=N)" of the log, and at the end the filter and the counters of every candidate and the filter of every read."""
    import re
    log = open(os.path.join(golden, "reference.log")).read()
    def logged(pattern):
        return logged_remaining(log, pattern)
    evalue_cutoff = float(re.search(r"Filtering fusions with an e-value >=([0-9.eE+-]+)", log).group(1))
    itd = re.search(r"Searching for internal tandem duplications <=\d+bp with >=(\d+) supporting reads and >=([0-9.]+)% allele fraction", log)
    quantile = float(re.search(r"expression above the ([0-9.]+)% quantile", log).group(1)) / 100
    min_spliced_events = int(re.search(r"Searching for fusions with >=(\d+) spliced events", log).group(1))
    min_anchor_length = int(re.search(r"Filtering fusions with anchors <=(\d+)nt", log).group(1))
    pipeline.find_fusions()
    pipeline.upload_coverage()
    counts = [pipeline.merge_adjacent_fusions(), pipeline.filter_multimappers()[0]]
    pipeline.estimate_expected_fusions()
    pipeline.filter_candidate_predicates()
    counts.append(pipeline.filter_relative_support())
    expected = [logged("Merging adjacent fusion breakpoints"), logged("Filtering multi-mapping fusions"), logged("Filtering fusions with an e-value")]
    stages = [(lambda: pipeline.recover_internal_tandem_duplication(int(itd.group(1)), float(itd.group(2)) / 100), "Searching for internal tandem duplications"),
              (pipeline.filter_both_intronic, "Filtering fusions with both breakpoints in intronic/intergenic regions")] + \
             ([(lambda: pipeline.recover_known_fusions(rules_prefix + ".known_fusions.tsv"), "Searching for known fusions")] if rules_prefix else []) + \
             [(lambda: pipeline.filter_in_vitro(quantile), "Filtering in vitro-generated fusions"),
              (pipeline.recover_both_spliced, "Searching for fusions with spliced split reads"),
              (pipeline.select_most_supported_breakpoints, "Selecting best breakpoints from genes with multiple breakpoints"),
              (pipeline.filter_marginal_read_through, "Filtering read-through fusions with breakpoints near the gene boundary"),
              (lambda: pipeline.recover_many_spliced(min_spliced_events), "Searching for fusions with >=\\d+ spliced events")] + \
             ([(lambda: pipeline.filter_blacklisted_ranges(rules_prefix + ".blacklist.tsv", evalue_cutoff), "Filtering blacklisted fusions")] if rules_prefix else []) + \
             [(lambda: pipeline.filter_short_anchor(min_anchor_length), "Filtering fusions with anchors"),
              (pipeline.filter_end_to_end, "Filtering end-to-end fusions with low support"),
              (pipeline.filter_no_coverage, "Filtering fusions with no coverage around the breakpoints")]
    for run, pattern in stages:
        counts.append(run())
        expected.append(logged(pattern))
    assert counts == expected, list(zip(counts, expected))
    after = golden_io.read_fusions(golden_io.find_dump(golden, "fusions", "filter_no_coverage"))
    table = pipeline.candidates()
    assert len(after) == pipeline.n_candidates
    index = {key: c for c, key in enumerate(candidate_keys(table, pipeline.n_candidates))}
    problems = []
    for f in after:
        c = index[fusion_key(f)]
        got = (int(table["filter"][c]), int(table["split_reads1"][c]), int(table["split_reads2"][c]), int(table["discordant_mates"][c]))
        if got != (f["filter"], f["split_reads1"], f["split_reads2"], f["discordant_mates"]):
            problems.append((fusion_key(f), got, (f["filter"], f["split_reads1"], f["split_reads2"], f["discordant_mates"])))
    assert not problems, (len(problems), problems[:10])
    names, read_filters = golden_io.read_filters(golden_io.find_dump(golden, "filters", "recover_internal_tandem_duplication"))
    mine = pipeline.filters()
    different = [(names[i], int(mine[i]), read_filters[i]) for i in range(len(names)) if mine[i] != read_filters[i]]
    assert not different, (len(different), different[:10])
    return counts


def check_homologs(session, pipeline, golden, multimappers=True, state_from="filter_no_coverage"):
    """make_kmer_index + filter_homologs from the reference's candidate state behind filter_no_coverage (injected; the e-values that break ties are
    the device's own) against the reference's dump behind filter_homologs.  Returns (candidates entering, candidates discarded)."""
    import re
    log = open(os.path.join(golden, "reference.log")).read()
    identity = float(re.search(r"Filtering genes with >=([0-9.]+)% identity", log).group(1)) / 100
    pipeline.find_fusions()
    pipeline.merge_adjacent_fusions()
    if multimappers:
        pipeline.filter_multimappers()
    pipeline.estimate_expected_fusions()
    table = pipeline.candidates()
    index = {key: c for c, key in enumerate(candidate_keys(table, pipeline.n_candidates))}
    before = golden_io.read_fusions(golden_io.find_dump(golden, "fusions", state_from))
    _inject_candidate_state(pipeline, index, before)
    pipeline.make_kmer_index()
    remaining = pipeline.filter_homologs(identity)
    assert remaining == int(re.search(r"Filtering genes with[^\n]*\(remaining=(\d+)\)", log).group(1))
    entering = sum(1 for f in before if f["filter"] == 0)
    assert _compare_candidate_filters(pipeline, index, golden_io.read_fusions(golden_io.find_dump(golden, "fusions", "before_filter_mismappers")), "filter_homologs") == remaining
    return entering, entering - remaining


def check_range_rules(session, pipeline, golden, rules_prefix, multimappers=True, known_from="filter_both_intronic", blacklist_from="recover_many_spliced"):
    """recover_known_fusions and filter_blacklisted_ranges, each from the reference's candidate state in front of it (injected; e-values and coverage are the
    device's own) against its dump.  Returns (known fusions recovered, candidates blacklisted)."""
    import re
    log = open(os.path.join(golden, "reference.log")).read()
    evalue_cutoff = float(re.search(r"Filtering fusions with an e-value >=([0-9.eE+-]+)", log).group(1)) if "Filtering fusions with an e-value" in log else 0.3
    pipeline.find_fusions()
    pipeline.upload_coverage()
    pipeline.merge_adjacent_fusions()
    if multimappers:
        pipeline.filter_multimappers()
    pipeline.estimate_expected_fusions()
    table = pipeline.candidates()
    index = {key: c for c, key in enumerate(candidate_keys(table, pipeline.n_candidates))}
    results = []
    for stage, state_from, run, pattern in (("recover_known_fusions", known_from, lambda: pipeline.recover_known_fusions(rules_prefix + ".known_fusions.tsv"), "Searching for known fusions"),
                                            ("filter_blacklisted_ranges", blacklist_from, lambda: pipeline.filter_blacklisted_ranges(rules_prefix + ".blacklist.tsv", evalue_cutoff), "Filtering blacklisted fusions")):
        before = golden_io.read_fusions(golden_io.find_dump(golden, "fusions", state_from))
        _inject_candidate_state(pipeline, index, before)
        remaining = run()
        assert remaining == logged_remaining(log, pattern), (stage, remaining, logged_remaining(log, pattern))
        assert _compare_candidate_filters(pipeline, index, golden_io.read_fusions(golden_io.find_dump(golden, "fusions", stage)), stage) == remaining
        results.append(abs(remaining - sum(1 for f in before if f["filter"] == 0)))
    return tuple(results)


def check_chain_to_mismappers(session, pipeline, golden, rules_prefix=None):
    """The reference's stages 18-38 (find_fusions ... filter_no_coverage -> make_kmer_index -> filter_homologs -> filter_mismappers, default filters)
    on the device in one go, nothing taken from the reference (the padding of the k-mer index and max_mate_gap come from the pipeline's own scalars)."""
    import re
    log = open(os.path.join(golden, "reference.log")).read()
    def logged(pattern):
        return int(re.search(pattern + r"[^\n]*\(remaining=(\d+)\)", log).group(1))
    counts = check_chain_to_no_coverage(session, pipeline, golden, rules_prefix)
    identity = float(re.search(r"Filtering genes with >=([0-9.]+)% identity", log).group(1)) / 100
    pipeline.make_kmer_index()
    after_homologs = pipeline.filter_homologs(identity)
    assert after_homologs == logged("Filtering genes with"), (after_homologs, logged("Filtering genes with"))
    table = pipeline.candidates()
    index = {key: c for c, key in enumerate(candidate_keys(table, pipeline.n_candidates))}
    _compare_candidate_filters(pipeline, index, golden_io.read_fusions(golden_io.find_dump(golden, "fusions", "before_filter_mismappers")), "filter_homologs")
    after_mismappers, discarded = pipeline.filter_mismappers()
    assert after_mismappers == logged("Re-aligning chimeric reads"), (after_mismappers, logged("Re-aligning chimeric reads"))
    _compare_candidate_filters(pipeline, index, golden_io.read_fusions(golden_io.find_dump(golden, "fusions", "filter_mismappers")), "filter_mismappers")
    names, read_filters = golden_io.read_filters(golden_io.find_dump(golden, "filters", "filter_mismappers"))
    mine = pipeline.filters()
    different = [(names[i], int(mine[i]), read_filters[i]) for i in range(len(names)) if mine[i] != read_filters[i]]
    assert not different, (len(different), different[:10])
    return counts + [after_homologs, after_mismappers], discarded


def check_isoforms(session, pipeline, golden, state_from="select_most_supported_breakpoints"):
    """recover_isoforms from the reference's candidate state in front of it (injected) against its dump behind it.  Returns (entering, recovered)."""
    import re
    log = open(os.path.join(golden, "reference.log")).read()
    pipeline.find_fusions()
    table = pipeline.candidates()
    index = {key: c for c, key in enumerate(candidate_keys(table, pipeline.n_candidates))}
    before = golden_io.read_fusions(golden_io.find_dump(golden, "fusions", state_from, which=-1))
    _inject_candidate_state(pipeline, index, before)
    remaining = pipeline.recover_isoforms()
    assert remaining == int(re.search(r"Searching for additional isoforms[^\n]*\(remaining=(\d+)\)", log).group(1)), remaining
    assert _compare_candidate_filters(pipeline, index, golden_io.read_fusions(golden_io.find_dump(golden, "fusions", "assign_confidence", which=-1)), "recover_isoforms") == remaining
    entering = sum(1 for f in before if f["filter"] == 0)
    return entering, remaining - entering


def check_confidence(session, pipeline, golden, multimappers=True):
    """assign_confidence from the reference's candidate state in front of it (injected; e-values and coverage are the device's own) against its dump.
    Returns the number of candidates per confidence level."""
    pipeline.find_fusions()
    pipeline.upload_coverage()
    pipeline.merge_adjacent_fusions()
    if multimappers:
        pipeline.filter_multimappers()
    pipeline.estimate_expected_fusions()
    table = pipeline.candidates()
    index = {key: c for c, key in enumerate(candidate_keys(table, pipeline.n_candidates))}
    expected = golden_io.read_fusions(golden_io.find_dump(golden, "fusions", "assign_confidence", which=-1))
    _inject_candidate_state(pipeline, index, expected)
    confidence = pipeline.assign_confidence()
    wrong = [(fusion_key(f), int(confidence[index[fusion_key(f)]]), f["confidence"]) for f in expected if int(confidence[index[fusion_key(f)]]) != f["confidence"]]
    assert not wrong, (len(wrong), wrong[:10])
    return [int((confidence == level).sum()) for level in (0, 1, 2)]


def check_chain_to_isoforms(session, pipeline, golden, rules_prefix=None):
    """The reference's candidate-level workflow to its end: stages 18-38 as in check_chain_to_mismappers, then select_most_supported_breakpoints a
    second time and recover_isoforms (source/arriba.cpp:571-584); nothing taken from the reference.  What follows in the reference
    (assign_confidence, the output writer) does not change filters or counters."""
    import re
    log = open(os.path.join(golden, "reference.log")).read()
    counts, discarded = check_chain_to_mismappers(session, pipeline, golden, rules_prefix)
    table = pipeline.candidates()
    index = {key: c for c, key in enumerate(candidate_keys(table, pipeline.n_candidates))}
    selected = pipeline.select_most_supported_breakpoints()
    assert selected == int(re.findall(r"Selecting best breakpoints[^\n]*\(remaining=(\d+)\)", log)[-1])
    recovered = pipeline.recover_isoforms()
    assert recovered == int(re.search(r"Searching for additional isoforms[^\n]*\(remaining=(\d+)\)", log).group(1)), recovered
    after = golden_io.read_fusions(golden_io.find_dump(golden, "fusions", "assign_confidence", which=-1))  # assign_confidence changes no filter: the state behind recover_isoforms
    assert _compare_candidate_filters(pipeline, index, after, "recover_isoforms") == recovered
    confidence = pipeline.assign_confidence()
    wrong = [(fusion_key(f), int(confidence[index[fusion_key(f)]]), f["confidence"]) for f in after if int(confidence[index[fusion_key(f)]]) != f["confidence"]]
    assert not wrong, (len(wrong), wrong[:10])
    result = pipeline.candidates()
    problems = [fusion_key(f) for f in after if (int(result["split_reads1"][index[fusion_key(f)]]), int(result["split_reads2"][index[fusion_key(f)]]), int(result["discordant_mates"][index[fusion_key(f)]])) != (f["split_reads1"], f["split_reads2"], f["discordant_mates"])]
    assert not problems, (len(problems), problems[:10])
    levels = [int((confidence == level).sum()) for level in (0, 1, 2)]
    return counts + [selected, recovered], discarded, levels


def check_output_files(session, pipeline, golden, directory, skip_columns=(), reference_prefix=None, rules_prefix=None):
    """After the chain: the two output files against the reference's (tests/golden/<name>/fusions.tsv.gz, discarded.tsv.gz, or the files a live run
    wrote): byte for byte, every line in the same order (columns named in skip_columns are left out of the comparison of fusions.tsv).
    Returns (lines of fusions.tsv, lines of discarded.tsv)."""
    import gzip
    results = []
    if rules_prefix:  # the reference ran with -t and -p as well
        session.load_tags(rules_prefix + ".tags.tsv")
        session.load_protein_domains(rules_prefix + ".protein_domains.gff3")
    for name, discarded in (("fusions.tsv", False), ("discarded.tsv", True)):
        path = os.path.join(directory, name)
        pipeline.write_fusions(path, discarded=discarded)
        mine = open(path).read().split("\n")
        source = reference_prefix + "." + name if reference_prefix else os.path.join(golden, name)  # a live run wrote <prefix>.fusions.tsv / <prefix>.discarded.tsv
        expected = (open(source).read() if os.path.exists(source) else gzip.open(source + ".gz", "rt").read()).split("\n")
        assert len(mine) == len(expected), (name, len(mine), len(expected))
        header = expected[0].split("\t")
        skipped = set() if discarded else {header.index(column) for column in skip_columns}
        for number, (a, b) in enumerate(zip(mine, expected)):
            if a == b:
                continue
            fields_a, fields_b = a.split("\t"), b.split("\t")
            different = [header[k] for k in range(max(len(fields_a), len(fields_b))) if k not in skipped and (fields_a[k:k + 1] != fields_b[k:k + 1])]
            assert not different, (name, number, different, a[:300], b[:300])
        results.append(len(expected) - 2)
    return tuple(results)


def check_workflow(prefix, golden, directory, api=None, rules=False, reference_prefix=None, fill_sequence_gaps=False, structural_variants=False, params=None, workflow_options=None, max_itd_length=100, external_duplicate_marking=False, device_ingest=False):
    """FASTA + GTF + BAM (+ blacklist / known fusions) -> fusions.tsv, discarded.tsv through DevicePipeline.run_workflow with the reference's default
    parameters: nothing is taken from the reference, not even the parameters its log prints.  Both files must equal the reference's byte for byte, and
    every "(remaining=N)" of its log must come out."""
    import gzip
    import re
    from arriba_amd.pipeline import DevicePipeline
    from arriba_amd.pipeline import HostSession
    session = HostSession(prefix + ".fa", prefix + ".gtf")
    if device_ingest:  # read_chimeric_alignments on the device: the host never holds the batch
        pipeline = DevicePipeline(session, params=params, api=api, bam=prefix + ".bam", external_duplicate_marking=external_duplicate_marking, max_itd_length=max_itd_length, piece_bytes=2 << 20)
    else:
        session.read_chimeric_alignments(prefix + ".bam", external_duplicate_marking=external_duplicate_marking, max_itd_length=max_itd_length)
        pipeline = DevicePipeline(session, params=params, api=api)
    stages = []
    outputs = [os.path.join(directory, "workflow.fusions.tsv"), os.path.join(directory, "workflow.discarded.tsv")]
    pipeline.run_workflow(outputs[0], outputs[1], blacklist_file=prefix + ".blacklist.tsv" if rules else None, known_fusions_file=prefix + ".known_fusions.tsv" if rules else None,
                          tags_file=prefix + ".tags.tsv" if rules else None, protein_domains_file=prefix + ".protein_domains.gff3" if rules else None,
                          genomic_breakpoints_file=prefix + ".sv.tsv" if structural_variants else None,
                          fill_sequence_gaps=fill_sequence_gaps, log=lambda stage, remaining: stages.append((stage, remaining)), **(workflow_options or {}))
    log = open(os.path.join(golden, "reference.log")).read()
    patterns = {"merge_adjacent_fusions": "Merging adjacent fusion breakpoints", "filter_multimappers": "Filtering multi-mapping fusions", "filter_relative_support": "Filtering fusions with an e-value",
                "recover_internal_tandem_duplication": "Searching for internal tandem duplications", "filter_both_intronic": "Filtering fusions with both breakpoints in intronic", "recover_known_fusions": "Searching for known fusions",
                "filter_in_vitro": "Filtering in vitro-generated fusions", "recover_both_spliced": "Searching for fusions with spliced split reads", "filter_marginal_read_through": "Filtering read-through fusions with breakpoints near",
                "recover_many_spliced": "Searching for fusions with >=\\d+ spliced events", "filter_blacklisted_ranges": "Filtering blacklisted fusions", "filter_short_anchor": "Filtering fusions with anchors",
                "filter_end_to_end": "Filtering end-to-end fusions", "filter_no_coverage": "Filtering fusions with no coverage", "filter_homologs": "Filtering genes with", "filter_mismappers": "Re-aligning chimeric reads",
                "recover_isoforms": "Searching for additional isoforms", "filter_no_genomic_support": "Filtering low-confidence events with no support from WGS", "recover_genomic_support": "Searching for fusions with support from WGS"}
    seen_select_best = 0
    for stage, remaining in stages:
        if stage == "select_most_supported_breakpoints":
            assert remaining == logged_remaining(log, "Selecting best breakpoints", seen_select_best), (stage, remaining)
            seen_select_best += 1
        elif stage == "mark_genomic_support":
            assert remaining == int(re.search(r"Marking fusions with support[^\n]*\(marked=(?:WARNING:[^\n]*\n)*(\d+)\)", log).group(1)), (stage, remaining)
        elif stage in patterns and re.search(patterns[stage], log):  # a filter switched off with -f prints no line
            assert remaining == logged_remaining(log, patterns[stage]), (stage, remaining, logged_remaining(log, patterns[stage]))
    # (the files behind the counts: a difference in a count names the stage, a difference in the bytes only the file)
    for mine, name in zip(outputs, ("fusions.tsv", "discarded.tsv")):
        source = reference_prefix + "." + name if reference_prefix else os.path.join(golden, name)
        expected = open(source).read() if os.path.exists(source) else gzip.open(source + ".gz", "rt").read()
        assert open(mine).read() == expected, name
    check_selected_candidates(pipeline)
    return stages


def check_selected_candidates(pipeline):
    """agpu_select_candidates / agpu_get_selected_candidates / agpu_get_filters_of (what the C++ workflow driver writes its files from) against the whole columns picked on the host"""
    table = pipeline.candidates(lists=False)
    whole = dict(table, evalue=pipeline.evalues(), confidence=pipeline.assign_confidence(), iteration_rank=pipeline.candidate_iteration_order())
    whole["closest_genomic_breakpoint1"], whole["closest_genomic_breakpoint2"] = pipeline.genomic_support()
    for discarded in (False, True):
        picked = pipeline.selected_candidates(discarded)
        expected = np.flatnonzero((table["filter"] != 0) if discarded else (table["filter"] == 0))
        assert np.array_equal(picked["candidate"], expected), discarded
        for key, column in picked.items():
            if key != "candidate":
                assert np.array_equal(column.view(np.uint32) if column.dtype == np.float32 else column, whole[key][expected].view(np.uint32) if column.dtype == np.float32 else whole[key][expected]), (key, discarded)
    some = np.unique(np.concatenate([np.arange(0, pipeline.n, max(1, pipeline.n // 997)), np.array([pipeline.n - 1])])).astype(np.uint32) if pipeline.n else np.zeros(0, np.uint32)
    assert np.array_equal(pipeline.filters_of(some), pipeline.filters()[some])

NON_DEFAULT_OPTIONS = {
    "reference": ["-E", "0.1", "-S", "3", "-A", "30", "-M", "2", "-L", "0.5", "-Z", "5", "-z", "0.03", "-e", "0.5", "-R", "5000", "-H", "5", "-V", "0.05", "-K", "0.5", "-m", "0.5", "-Q", "0.99", "-U", "100",
                  "-l", "60", "-D", "20000"],
    "disabled": ["homopolymer", "short_anchor"],
    "params": {"homopolymer_length": 5, "min_read_through_distance": 5000, "max_itd_length": 60, "subsampling_threshold": 100, "mismatch_pvalue_cutoff": 0.05, "max_kmer_content": 0.5, "evalue_cutoff": 0.1,
               "max_mismapper_fraction": 0.5, "exonic_fraction": 0.5, "min_support": 3, "disable_filters": ["homopolymer", "short_anchor"]},
    "workflow": {"evalue_cutoff": 0.1, "min_itd_support": 5, "min_itd_allele_fraction": 0.03, "high_expression_quantile": 0.99, "min_spliced_events": 2, "min_anchor_length": 30, "max_homolog_identity": 0.5,
                 "max_itd_length": 60, "max_genomic_breakpoint_distance": 20000},
}


def check_workflow_with_non_default_options(fragments, directory, api=None, device_ingest=False):
    """Nineteen options away from their defaults and two filters switched off, on both sides: the reference run live with the command-line options, the
    workflow with the same values through agpu_params and the stage arguments; blacklist, known fusions, tags, protein domains and structural variants given."""
    spec = {"args": ["--seed", "67", "--fragments", str(fragments), "--normal-mult", "0.4", "--contigs", "6", "--contig-len", "500000", "--junctions", "700", "--dup", "0.15", "--rule-files", "--homolog-families", "10",
                     "--itd-hotspots", "3", "--itd-hotspot-frac", "0.03"], "rule_files": True, "structural_variants": True}
    prefix = datasets.generate(spec, directory)
    dump = os.path.join(directory, "dump")
    os.makedirs(dump)
    switches = {"ARRIBA_ORACLE_DUMP_LISTS": "0", "ARRIBA_ORACLE_DUMP_READS": "0", "ARRIBA_ORACLE_DUMP_STAGES": "key"}
    os.environ.update(switches)
    try:
        log = datasets.run_reference(prefix, dump, spec, extra_args=NON_DEFAULT_OPTIONS["reference"], disable_filters=NON_DEFAULT_OPTIONS["disabled"])
    finally:
        for key in switches:
            del os.environ[key]
    with open(os.path.join(dump, "reference.log"), "w") as out:
        out.write(log)
    os.makedirs(os.path.join(directory, "mine"))
    return check_workflow(prefix, dump, os.path.join(directory, "mine"), api=api, rules=True, reference_prefix=prefix, structural_variants=True, params=NON_DEFAULT_OPTIONS["params"],
                          workflow_options=NON_DEFAULT_OPTIONS["workflow"], max_itd_length=60, device_ingest=device_ingest)


def check_read_lists(session, pipeline, golden, stage):
    """the three read lists of every candidate against the reference's dump of `stage` (contents, or sizes for dumps written without lists)"""
    fusions = golden_io.read_fusions(golden_io.find_dump(golden, "fusions", stage))
    table = pipeline.candidates()
    names = session.fragment_names()
    offsets = table["list_offset"].astype(np.int64)
    lists = table["read_lists"]
    index = {key: c for c, key in enumerate(candidate_keys(table, pipeline.n_candidates))}
    problems, appended = [], 0
    for f in fusions:
        c = index[fusion_key(f)]
        for k, field in enumerate(("split_read1_list", "split_read2_list", "discordant_mate_list")):
            reads = lists[offsets[3 * c + k]:offsets[3 * c + k + 1]]
            expected = f[field]
            if len(expected) == 1 and expected[0].isdigit():
                if len(reads) != int(expected[0]):
                    problems.append((field + ".size", fusion_key(f), len(reads), int(expected[0])))
            elif [names[r] for r in reads] != expected:
                problems.append((field, fusion_key(f), len(reads), len(expected)))
    assert not problems, (len(problems), problems[:10])
    return len(fusions)


def check_event_chain(session, pipeline, golden):
    """filter_both_intronic -> filter_in_vitro -> recover_both_spliced -> select_most_supported_breakpoints -> filter_marginal_read_through ->
    recover_many_spliced -> filter_short_anchor -> filter_end_to_end_fusions -> filter_no_coverage in one go on the device, from the reference's
    state behind recover_internal_tandem_duplication (the stage in front that is not built): every "(remaining=N)" and the final filters"""
    import re
    log = open(os.path.join(golden, "reference.log")).read()
    def logged(pattern):
        return int(re.search(pattern + r"[^\n]*\(remaining=(\d+)\)", log).group(1))
    pipeline.find_fusions()
    pipeline.upload_coverage()
    table = pipeline.candidates()
    index = {key: c for c, key in enumerate(candidate_keys(table, pipeline.n_candidates))}
    names, read_filters = golden_io.read_filters(golden_io.find_dump(golden, "filters", "recover_internal_tandem_duplication"))
    _inject_candidate_state(pipeline, index, golden_io.read_fusions(golden_io.find_dump(golden, "fusions", "recover_internal_tandem_duplication")))
    pipeline.set_read_filters(np.array(read_filters, dtype=np.uint8))
    quantile = float(re.search(r"expression above the ([0-9.]+)% quantile", log).group(1)) / 100
    min_spliced_events = int(re.search(r"Searching for fusions with >=(\d+) spliced events", log).group(1))
    min_anchor_length = int(re.search(r"Filtering fusions with anchors <=(\d+)nt", log).group(1))
    stages = [(pipeline.filter_both_intronic, "Filtering fusions with both breakpoints in intronic/intergenic regions"),
              (lambda: pipeline.filter_in_vitro(quantile), "Filtering in vitro-generated fusions"),
              (pipeline.recover_both_spliced, "Searching for fusions with spliced split reads"),
              (pipeline.select_most_supported_breakpoints, "Selecting best breakpoints from genes with multiple breakpoints"),
              (pipeline.filter_marginal_read_through, "Filtering read-through fusions with breakpoints near the gene boundary"),
              (lambda: pipeline.recover_many_spliced(min_spliced_events), "Searching for fusions with >=\\d+ spliced events"),
              (lambda: pipeline.filter_short_anchor(min_anchor_length), "Filtering fusions with anchors"),
              (pipeline.filter_end_to_end, "Filtering end-to-end fusions with low support"),
              (pipeline.filter_no_coverage, "Filtering fusions with no coverage around the breakpoints")]
    counts = []
    for run, pattern in stages:
        remaining = run()
        assert remaining == logged(pattern), (pattern, remaining, logged(pattern))
        counts.append(remaining)
    _compare_candidate_filters(pipeline, index, golden_io.read_fusions(golden_io.find_dump(golden, "fusions", "filter_no_coverage")), "chain filter_both_intronic .. filter_no_coverage")
    return counts


def check_merge_adjacent(session, pipeline, golden):
    """merge_adjacent_fusions on the device state right after find_fusions against the reference's dump of that stage"""
    import re
    after = golden_io.read_fusions(golden_io.find_dump(golden, "fusions", "merge_adjacent_fusions"))
    remaining = pipeline.merge_adjacent_fusions()
    table = pipeline.candidates()
    n = pipeline.n_candidates
    index = {key: c for c, key in enumerate(candidate_keys(table, n))}
    problems = []
    for f in after:
        c = index[fusion_key(f)]
        got = (int(table["filter"][c]), int(table["split_reads1"][c]), int(table["split_reads2"][c]), int(table["discordant_mates"][c]))
        if got != (f["filter"], f["split_reads1"], f["split_reads2"], f["discordant_mates"]):
            problems.append((fusion_key(f), got, (f["filter"], f["split_reads1"], f["split_reads2"], f["discordant_mates"])))
    assert not problems, (len(problems), problems[:10])
    log = open(os.path.join(golden, "reference.log")).read()
    match = re.search(r"Merging adjacent fusion breakpoints \(remaining=(\d+)\)", log)
    assert match and remaining == int(match.group(1)), (remaining, match and match.group(1))
    return sum(1 for f in after if f["filter"] == 23)


def check_multimappers(session, pipeline, golden):
    """find_fusions -> merge_adjacent_fusions -> filter_multimappers against the reference's dumps right after filter_multimappers
    (candidate counters and filters, and the filter id of every fragment)"""
    import re
    after = golden_io.read_fusions(golden_io.find_dump(golden, "fusions", "filter_multimappers"))
    names, read_filters_after = golden_io.read_filters(golden_io.find_dump(golden, "filters", "filter_multimappers"))
    assert session.fragment_names() == names
    pipeline.find_fusions()
    before = pipeline.filters().copy()
    pipeline.merge_adjacent_fusions()
    remaining, discarded = pipeline.filter_multimappers()
    mine = pipeline.filters()
    different = [(names[i], int(mine[i]), read_filters_after[i]) for i in range(len(names)) if mine[i] != read_filters_after[i]]
    assert not different, (len(different), different[:10])
    assert discarded == int((mine != before).sum())
    table = pipeline.candidates()
    n = pipeline.n_candidates
    assert len(after) == n
    index = {key: c for c, key in enumerate(candidate_keys(table, n))}
    problems = []
    for f in after:
        c = index[fusion_key(f)]
        got = (int(table["filter"][c]), int(table["split_reads1"][c]), int(table["split_reads2"][c]), int(table["discordant_mates"][c]))
        if got != (f["filter"], f["split_reads1"], f["split_reads2"], f["discordant_mates"]):
            problems.append((fusion_key(f), got, (f["filter"], f["split_reads1"], f["split_reads2"], f["discordant_mates"])))
    assert not problems, (len(problems), problems[:10])
    log = open(os.path.join(golden, "reference.log")).read()
    match = re.search(r"Filtering multi-mapping[^\n]*\(remaining=(\d+)\)", log)
    assert match and remaining == int(match.group(1)), (remaining, match and match.group(1))
    return discarded


def check_chain_to_relative_support(session, pipeline, golden, multimappers=False):
    """find_fusions -> merge_adjacent_fusions -> [filter_multimappers] -> e-value -> candidate predicates -> filter_relative_support on the
    device without any state taken from the reference.  multimappers=False: against a reference run with filter_multimappers switched off."""
    pipeline.find_fusions()
    pipeline.merge_adjacent_fusions()
    if multimappers:
        pipeline.filter_multimappers()
    evalue = pipeline.estimate_expected_fusions()
    pipeline.filter_candidate_predicates()
    remaining = pipeline.filter_relative_support()
    after = golden_io.read_fusions(golden_io.find_dump(golden, "fusions", "filter_relative_support"))
    table = pipeline.candidates()
    n = pipeline.n_candidates
    assert len(after) == n
    index = {key: c for c, key in enumerate(candidate_keys(table, n))}
    bits = evalue.view(np.uint32)
    problems = []
    for f in after:
        c = index[fusion_key(f)]
        got = (int(table["filter"][c]), int(table["split_reads1"][c]), int(table["split_reads2"][c]), int(table["discordant_mates"][c]), int(bits[c]))
        if got != (f["filter"], f["split_reads1"], f["split_reads2"], f["discordant_mates"], f["evalue_bits"]):
            problems.append((fusion_key(f), got, (f["filter"], f["split_reads1"], f["split_reads2"], f["discordant_mates"], f["evalue_bits"])))
    assert not problems, (len(problems), problems[:10])
    import re
    log = open(os.path.join(golden, "reference.log")).read()
    match = re.search(r"Filtering fusions with an e-value[^\n]*\(remaining=(\d+)\)", log)
    assert match and remaining == int(match.group(1))
    return n
