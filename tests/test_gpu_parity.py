"""GPU tier: the HIP kernels, called through the C ABI of libarriba_gpu.so, against the golden dumps of the reference
and (when the prebuilt oracle binary travelled with the repo) against the reference run live on a fresh dataset."""
import os
import re

import numpy as np
import pytest

import conftest
import datasets
import golden_io
import parity

# a kernel that never returns must not hold the GPU box until the harness gives up: the test process exits after 20 minutes in one test
pytestmark = [pytest.mark.gpu, pytest.mark.timeout(1200, method="thread")]


@pytest.mark.parametrize("name", ["toy3k", "shuffled2k", "mid30k", "stacked4k"])
def test_read_level_cascade_matches_reference(name, dataset_files):
    golden = conftest.golden_dir(name)
    session, pipeline = parity.run_read_level(parity.open_session, dataset_files(name))
    parity.check_gene_table(pipeline, golden)
    parity.check_read_filters(session, pipeline, golden)
    parity.check_scalars(pipeline, golden)
    if name != "mid30k":
        parity.check_annotation(session, pipeline, golden)
    if name == "stacked4k":
        counts = np.concatenate([pipeline.gene_sets(slot)[0] for slot in range(3)])
        assert (counts > 4).sum() > 500 and counts.max() >= 9  # the memory-resident tail of the gene sets is exercised
        pipeline.find_fusions()
        assert parity.check_candidates(session, pipeline, golden) > 5000
    if name == "toy3k":
        pipeline.find_fusions()
        assert parity.check_candidates(session, pipeline, golden) > 1000
        assert parity.check_evalues(session, pipeline, golden) > 1000
        assert parity.check_mismappers(session, pipeline, golden) == 2


def test_live_reference_on_larger_dataset(built, tmp_path):
    if not datasets.reference_available():
        pytest.skip("oracle/_ref/arriba_ref_dump did not travel with the repository")
    spec = {"args": ["--seed", "21", "--fragments", "120000", "--normal-mult", "0.3", "--contigs", "8", "--contig-len", "600000", "--junctions", "1500", "--dup", "0.15"]}
    prefix = datasets.generate(spec, str(tmp_path))
    dump = str(tmp_path / "dump")
    os.makedirs(dump)
    env_lists = os.environ.get("ARRIBA_ORACLE_DUMP_LISTS")
    os.environ["ARRIBA_ORACLE_DUMP_LISTS"] = "0"
    try:
        log = datasets.run_reference(prefix, dump)
    finally:
        if env_lists is None:
            del os.environ["ARRIBA_ORACLE_DUMP_LISTS"]
    with open(os.path.join(dump, "reference.log"), "w") as out:
        out.write(log)
    session, pipeline = parity.run_read_level(parity.open_session, prefix)
    parity.check_gene_table(pipeline, dump)
    parity.check_read_filters(session, pipeline, dump)
    parity.check_scalars(pipeline, dump)
    parity.check_annotation(session, pipeline, dump)
    pipeline.find_fusions()
    assert parity.check_candidates(session, pipeline, dump) > 10000
    assert parity.check_evalues(session, pipeline, dump) > 10000


def test_chain_to_relative_support_without_injected_state(built, dataset_files, tmp_path):
    """find_fusions -> merge_adjacent_fusions -> e-value -> candidate predicates -> filter_relative_support on the GPU, nothing taken from the
    reference in between: against the committed dumps of a reference run without filter_multimappers, and against the reference run live"""
    golden = conftest.golden_dir("toy3k_chain")
    session, pipeline = parity.run_read_level(parity.open_session, dataset_files("toy3k_chain"))
    pipeline.find_fusions()
    assert parity.check_merge_adjacent(session, pipeline, golden) >= 0
    session, pipeline = parity.run_read_level(parity.open_session, dataset_files("toy3k_chain"))
    assert parity.check_chain_to_relative_support(session, pipeline, golden) > 1000
    golden = conftest.golden_dir("toy3k")  # default filters: filter_multimappers in the chain
    session, pipeline = parity.run_read_level(parity.open_session, dataset_files("toy3k"))
    assert parity.check_multimappers(session, pipeline, golden) > 50
    session, pipeline = parity.run_read_level(parity.open_session, dataset_files("toy3k"))
    assert parity.check_chain_to_relative_support(session, pipeline, golden, multimappers=True) > 1000
    if not datasets.reference_available():
        return
    spec = {"args": ["--seed", "23", "--fragments", "120000", "--normal-mult", "0.3", "--contigs", "8", "--contig-len", "600000", "--junctions", "1500", "--dup", "0.15"]}
    prefix = datasets.generate(spec, str(tmp_path))
    dump = str(tmp_path / "dump")
    os.makedirs(dump)
    os.environ["ARRIBA_ORACLE_DUMP_LISTS"] = "0"
    try:
        log = datasets.run_reference(prefix, dump, disable_filters=["multimappers"])
    finally:
        del os.environ["ARRIBA_ORACLE_DUMP_LISTS"]
    with open(os.path.join(dump, "reference.log"), "w") as out:
        out.write(log)
    session, pipeline = parity.run_read_level(parity.open_session, prefix)
    pipeline.find_fusions()
    assert parity.check_merge_adjacent(session, pipeline, dump) >= 0
    session, pipeline = parity.run_read_level(parity.open_session, prefix)
    assert parity.check_chain_to_relative_support(session, pipeline, dump) > 10000
    # the same sample with the reference's default filters
    dump = str(tmp_path / "dump_default")
    os.makedirs(dump)
    os.environ["ARRIBA_ORACLE_DUMP_LISTS"] = "0"
    try:
        log = datasets.run_reference(prefix, dump)
    finally:
        del os.environ["ARRIBA_ORACLE_DUMP_LISTS"]
    with open(os.path.join(dump, "reference.log"), "w") as out:
        out.write(log)
    session, pipeline = parity.run_read_level(parity.open_session, prefix)
    assert parity.check_multimappers(session, pipeline, dump) > 100
    session, pipeline = parity.run_read_level(parity.open_session, prefix)
    assert parity.check_chain_to_relative_support(session, pipeline, dump, multimappers=True) > 10000


def test_live_reference_above_launch_grid_caps(built, tmp_path):
    """700 k fragments: more than the 2048 x 256 threads of the capped launch grids (counting kernels with grid-stride loops), so a kernel that
    is launched on a capped grid without looping over its items fails here; the smaller datasets cannot see that."""
    if not datasets.reference_available():
        pytest.skip("oracle/_ref/arriba_ref_dump did not travel with the repository")
    spec = {"args": ["--seed", "51", "--fragments", "700000", "--normal-mult", "0.05", "--contigs", "8", "--contig-len", "1500000", "--junctions", "8000", "--dup", "0.2"]}
    prefix = datasets.generate(spec, str(tmp_path))
    dump = str(tmp_path / "dump")
    os.makedirs(dump)
    switches = {"ARRIBA_ORACLE_DUMP_LISTS": "0", "ARRIBA_ORACLE_DUMP_READS": "0", "ARRIBA_ORACLE_DUMP_STAGES": "key"}  # compact dumps: ~45 s of reference
    os.environ.update(switches)
    try:
        log = datasets.run_reference(prefix, dump)
    finally:
        for key in switches:
            del os.environ[key]
    with open(os.path.join(dump, "reference.log"), "w") as out:
        out.write(log)
    session, pipeline = parity.run_read_level(parity.open_session, prefix)
    assert pipeline.n > 2048 * 256
    parity.check_read_filters(session, pipeline, dump)
    parity.check_scalars(pipeline, dump)
    assert parity.check_chain_to_relative_support(session, pipeline, dump, multimappers=True) > 100000


def test_live_reference_mismapper_stress(built, tmp_path):
    """make_kmer_index + filter_mismappers on a dataset whose clipped segments stem from the split read's own gene, with the
    event-level filters in front switched off in the reference so that every candidate's reads are re-aligned."""
    if not datasets.reference_available():
        pytest.skip("oracle/_ref/arriba_ref_dump did not travel with the repository")
    spec = {"args": ["--seed", "32", "--fragments", "60000", "--normal-mult", "0.3", "--contigs", "6", "--contig-len", "500000", "--junctions", "600", "--dup", "0.1",
                     "--partner-clip", "0.5", "--clip-min", "40", "--clip-max", "70"]}
    prefix = datasets.generate(spec, str(tmp_path))
    dump = str(tmp_path / "dump")
    os.makedirs(dump)
    os.environ["ARRIBA_ORACLE_DUMP_LISTS"] = "0"
    try:
        log = datasets.run_reference(prefix, dump, disable_filters=["relative_support", "min_support", "intronic", "non_coding_neighbors", "intragenic_exonic", "in_vitro", "no_coverage",
                                                                    "end_to_end", "short_anchor", "select_best", "marginal_read_through", "homologs", "merge_adjacent", "multimappers"])
    finally:
        del os.environ["ARRIBA_ORACLE_DUMP_LISTS"]
    with open(os.path.join(dump, "reference.log"), "w") as out:
        out.write(log)
    session, pipeline = parity.run_read_level(parity.open_session, prefix)
    pipeline.find_fusions()
    assert parity.check_candidates(session, pipeline, dump) > 10000
    assert parity.check_mismappers(session, pipeline, dump) > 20


def test_hip_path_matches_oracle_restatement_on_fresh_inputs(built, tmp_path):
    """Inputs without a committed dump: the HIP kernels against the independent CPU restatement (oracle/liboracle.so)."""
    import oracle_lib
    for seed, extra in ((101, []), (102, ["--shuffle", "--separate-mates"]), (103, ["--stranded", "--dup", "0.5", "--noise", "0.7"])):
        spec = {"args": ["--seed", str(seed), "--fragments", "40000", "--normal-mult", "0.2", "--contigs", "5", "--contig-len", "500000", "--junctions", "400"] + extra}
        prefix = datasets.generate(spec, str(tmp_path), name="fresh%d" % seed)
        session, pipeline = parity.run_read_level(parity.open_session, prefix)
        pipeline.find_fusions()
        oracle = oracle_lib.OraclePipeline(session)
        oracle.run_read_level(pipeline.scalars["strandedness"])
        assert oracle.remaining == pipeline.remaining
        assert np.array_equal(oracle.filters(), pipeline.filters())
        assert np.array_equal(oracle.fragment_bits(), pipeline.fragment_bits())
        for slot in range(3):
            count_a, genes_a = oracle.gene_sets(slot)
            count_b, genes_b = pipeline.gene_sets(slot)
            assert np.array_equal(count_a, count_b) and np.array_equal(genes_a, genes_b)
        oracle.find_fusions(pipeline.scalars["max_mate_gap"])
        mine, theirs = pipeline.candidates(), oracle.candidates()
        assert pipeline.n_candidates == oracle.n_candidates
        for key in ("gene1", "gene2", "contigs", "breakpoint1", "breakpoint2", "flags", "filter", "split_reads1", "split_reads2", "discordant_mates", "anchor_start1", "anchor_start2", "list_offset", "read_lists"):
            assert np.array_equal(np.asarray(mine[key], dtype=np.int64), np.asarray(theirs[key], dtype=np.int64)), (seed, key)
        assert np.array_equal(pipeline.discordant_swapped(), oracle.oracle.discordant_swapped())
        # strands after annotation (slot order as uploaded)
        for slot in range(3):
            normalise = lambda bits: np.where(bits & 32, bits & 0x2F, bits & 0x3F)  # the predicted strand is meaningless while it is ambiguous
            assert np.array_equal(normalise(oracle.alignment_bits(slot)), normalise(pipeline.alignment_bits(slot))), (seed, slot)


def test_cascade_properties_at_scale(built, tmp_path):
    """Size-independent properties on a dataset too large for the oracle in a unit test: stage counts are monotone,
    a duplicate key survives exactly once, and rerunning the cascade on the same batch is idempotent."""
    from arriba_amd.pipeline import DevicePipeline
    spec = {"args": ["--seed", "33", "--fragments", "400000", "--normal-mult", "0.1", "--contigs", "8", "--contig-len", "800000", "--junctions", "4000"]}
    prefix = datasets.generate(spec, str(tmp_path))
    session = parity.open_session(prefix)
    first = DevicePipeline(session)
    remaining = first.run_read_level()
    order = ["duplicates", "uninteresting_contigs", "viral_contigs", "top_expressed_viral_contigs", "low_coverage_viral_contigs", "read_through", "inconsistently_clipped",
             "homopolymer", "small_insert_size", "long_gap", "same_gene", "hairpin", "mismatches", "low_entropy"]
    counts = [remaining[name] for name in order]
    assert all(a >= b for a, b in zip(counts, counts[1:])) and counts[0] <= first.n
    filters = first.filters()
    assert int((filters == 0).sum()) == remaining["low_entropy"]
    first.find_fusions()
    table = first.candidates()
    # every read is emitted once per gene pair; the candidate table partitions the emissions
    assert int(table["list_offset"][-1]) == len(table["read_lists"])
    assert (table["split_reads1"] <= 300).all() and (table["split_reads2"] <= 300).all() and (table["discordant_mates"] <= 300).all()
    unfiltered = table["filter"] == 0
    assert ((table["split_reads1"] + table["split_reads2"] + table["discordant_mates"])[unfiltered] >= 1).all()
    second = DevicePipeline(session)
    second.run_read_level()
    second.find_fusions()
    again = second.candidates()
    for key in table:
        assert np.array_equal(table[key], again[key]), key
    assert np.array_equal(filters, second.filters())
    for slot in range(3):
        count1, genes1 = first.gene_sets(slot)
        count2, genes2 = second.gene_sets(slot)
        assert np.array_equal(count1, count2) and np.array_equal(genes1, genes2)


def test_sharded_pipeline_two_ranks_on_one_device(built, dataset_files, tmp_path):
    """The sharded entry points of the C ABI on the GPU: two processes (gloo for the exchanges) drive one shard each on cuda:0;
    the merged result must equal the single-process pipeline (the RCCL transport itself needs >= 2 GPUs and is exercised by bench.py --gpus N)."""
    import test_sharded
    reports = test_sharded.run_sharded(dataset_files("mid30k"), 2, "gpu", str(tmp_path / "report"), 29655)
    assert reports[0]["problems"] == [], reports[0]["problems"]
    assert sum(r["owned_candidates"] for r in reports) == reports[0]["candidates"]


def test_one_sample_over_two_ranks_on_one_device(built, dataset_files, tmp_path):
    """One sample over several ranks (arriba_amd/one_sample.py) with the kernels of the GPU: two processes on cuda:0 ingest their halves of the file, the all-gather
    of the parts and the all-reduce of the mis-mapper verdicts go through gloo (RCCL needs a GPU per rank: bench.py --gpus N); batch, stage counts and both output
    files must be those of the single-process run over the whole file."""
    import test_one_sample
    report = test_one_sample.check_reports(test_one_sample.run_one_sample(dataset_files("mid30k"), 2, "gpu", str(tmp_path / "report"), 29755))
    assert report["mismapper_jobs"] > 0 and report["fusions"] > 0


def test_workflow_library_over_two_ranks_on_one_device(built, dataset_files, tmp_path):
    """The C++ driver as a collective call (include/arriba_workflow.h: arriba_workflow_set_communicator) with the kernels of the GPU: two processes on cuda:0, each with a session
    of libarriba_workflow.so, feed their halves of the file; the parts, the verdicts of filter_mismappers and the row texts go through the host collectives of the communicator
    (gloo; RCCL needs a GPU per rank: the next test).  fusions.tsv, discarded.tsv and every stage count equal those of one rank without a communicator; two samples in a queue."""
    import test_one_sample
    test_one_sample.check_workflow_over_ranks("product", dataset_files("mid30k"), 2, tmp_path, 29715, samples=2)


@pytest.mark.parametrize("world,name", [(4, "homologs8k"), (3, "itd6k"), (8, "toy3k")])
def test_read_sharded_sample_over_ranks_on_one_device(world, name, built, dataset_files, tmp_path):
    """The read-sharded split with the kernels of the GPU at more ranks and on the datasets that decide by re-alignments (homologs8k) and internal tandem duplications (itd6k):
    `world` processes on cuda:0, every one holding its share of the fragments through the whole sample (tests/test_one_sample.py: check_workflow_over_ranks asserts that), the
    sharded forms of filter_multimappers / filter_in_vitro / filter_mismappers / recover_internal_tandem_duplication on the device: both files and every count of one rank."""
    import test_one_sample
    test_one_sample.check_workflow_over_ranks("product", dataset_files(name), world, tmp_path, 29600 + 8 * world)


def test_workflow_library_over_one_rccl_rank(built, dataset_files, tmp_path):
    """arriba_workflow_join_rccl -- the communicator of RCCL alone that `bench.py --gpus N` gives the C++ driver -- with ONE rank (the GPU box has one GPU): ncclGetUniqueId /
    ncclCommInitRank by dlopen, the part of the batch through agpu_shard_merge_rccl (ncclAllGather in device memory), the verdicts through agpu_filter_mismappers_rccl, sizes,
    status words and row texts bounced through the device (agpu_rccl_all_gather_host / _all_reduce_host): the files and counts of the plain session."""
    from arriba_amd.pipeline import WorkflowSession
    prefix = dataset_files("mid30k")
    plain = WorkflowSession(prefix + ".fa", prefix + ".gtf")
    expected = plain.sample(prefix + ".bam", str(tmp_path / "plain.tsv"), str(tmp_path / "plain.discarded.tsv"))
    plain.close()
    session = WorkflowSession(prefix + ".fa", prefix + ".gtf")
    session.join_rccl(session.rccl_unique_id(), 0, 1)
    session.submit(prefix + ".bam")
    for k in range(2):
        if k == 0:
            session.submit(prefix + ".bam")
        report = session.sample(prefix + ".bam", str(tmp_path / ("rccl%d.tsv" % k)), str(tmp_path / ("rccl%d.discarded.tsv" % k)))
        assert report == expected and session.timing["exchange_parts"] > 0
        assert session.timing["shard_fragments"] == dict(report)["read_chimeric_alignments"]  # (the read-sharded split, its one rank holding every fragment)
        for name in (".tsv", ".discarded.tsv"):
            assert open(str(tmp_path / ("rccl%d" % k)) + name, "rb").read() == open(str(tmp_path / "plain") + name, "rb").read(), (k, name)
    session.close()
    assert len(open(str(tmp_path / "plain.tsv")).read().splitlines()) > 3


def test_device_ingest_in_parts_on_the_gpu(built, dataset_files, tmp_path):
    """agpu_shard_export / agpu_shard_merge on the GPU: 3 and 7 parts of a stored-BGZF file and of one with 997-byte blocks, merged from blocks in host memory
    (as a gloo all-gather leaves them) and compared column by column with the ingest of the whole file"""
    import test_host_and_device_logic as host_tests
    from arriba_amd import _capi
    from arriba_amd.pipeline import DevicePipeline, HostSession
    api = _capi.bind_device_api(_capi.device_library(), "agpu_")
    prefix = dataset_files("itd6k")
    session = HostSession(prefix + ".fa", prefix + ".gtf")
    expected = host_tests._device_batch_columns(session, DevicePipeline(session, bam=prefix + ".bam"))
    small = str(tmp_path / "small.bam")
    host_tests._write_bgzf(small, host_tests._bam_payload(prefix + ".bam"), 1, block=997)
    for path, parts in ((prefix + ".bam", 3), (small, 7)):
        merged_session, merged, _ = host_tests._ingest_in_parts(prefix, path, parts, api)
        assert host_tests._device_batch_columns(merged_session, merged) == expected, (path, parts)
    # names in the order of a FASTQ file: the merged batch is sorted on the device
    scrambled = dataset_files("scrambled3k")
    session = HostSession(scrambled + ".fa", scrambled + ".gtf")
    expected = host_tests._device_batch_columns(session, DevicePipeline(session, bam=scrambled + ".bam"))
    merged_session, merged, _ = host_tests._ingest_in_parts(scrambled, scrambled + ".bam", 4, api)
    assert merged.ingest_result.names_were_sorted == 0
    assert host_tests._device_batch_columns(merged_session, merged) == expected


def test_rccl_compositions_with_one_rank(built, dataset_files, tmp_path):
    """agpu_shard_merge_rccl and agpu_filter_mismappers_rccl -- the two exchanges of one sample over N GPUs issued under the C ABI (include/arriba_gpu.h; SURVEY.md 8b:
    agpu_shard_merge(ctx, ncclComm_t, hipStream_t)) -- on a communicator of ONE rank (the GPU box has one GPU): librccl is found at run time, ncclAllReduce / ncclAllGather
    run on the context's stream, and the batch, the candidates, every count and both output files are those of the plain path."""
    import ctypes
    from ctypes import byref, c_uint64
    import test_host_and_device_logic as host_tests
    from arriba_amd import _capi
    from arriba_amd.pipeline import DevicePipeline, HostSession
    rccl = ctypes.CDLL("librccl.so")

    class UniqueId(ctypes.Structure):
        _fields_ = [("internal", ctypes.c_char * 128)]
    rccl.ncclGetUniqueId.argtypes = [ctypes.POINTER(UniqueId)]; rccl.ncclGetUniqueId.restype = ctypes.c_int
    rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]; rccl.ncclCommInitRank.restype = ctypes.c_int
    rccl.ncclCommDestroy.argtypes = [ctypes.c_void_p]; rccl.ncclCommDestroy.restype = ctypes.c_int
    unique, communicator = UniqueId(), ctypes.c_void_p()
    assert rccl.ncclGetUniqueId(byref(unique)) == 0
    assert rccl.ncclCommInitRank(byref(communicator), 1, unique, 0) == 0 and communicator.value
    try:
        api = _capi.bind_device_api(_capi.device_library(), "agpu_")
        prefix = dataset_files("homologs8k")  # (hundreds of re-alignments decide in this dataset)
        session = HostSession(prefix + ".fa", prefix + ".gtf")
        plain = DevicePipeline(session, bam=prefix + ".bam")
        expected_batch = host_tests._device_batch_columns(session, plain)

        class OneRank(DevicePipeline):
            def read_chimeric_alignments(self, bam, external_duplicate_marking=False, max_itd_length=100, piece_bytes=1 << 20):
                config, _, _ = self._ingest_records(bam, external_duplicate_marking, max_itd_length, piece_bytes, part=0, parts=1)  # (a part of a sample: the batch is handed on by agpu_shard_export)
                result = _capi.IngestResult()
                self._check(self.api.shard_merge_rccl(self.ctx, communicator, 1, byref(result)))
                self._adopt_ingest(config, result)
                return self.n

            def filter_mismappers(self, max_mate_gap=None):
                remaining, discarded = c_uint64(), c_uint64()
                self._check(self.api.filter_mismappers_rccl(self.ctx, communicator, self.scalars["max_mate_gap"] if max_mate_gap is None else max_mate_gap, 0, 1, byref(remaining), byref(discarded)))
                self._record("filter_mismappers")
                return remaining.value, discarded.value
        rccl_session = HostSession(prefix + ".fa", prefix + ".gtf")
        through_rccl = OneRank(rccl_session, api=api, bam=prefix + ".bam")
        assert host_tests._device_batch_columns(rccl_session, through_rccl) == expected_batch
        outputs = {}
        for name, pipeline in (("plain", plain), ("rccl", through_rccl)):
            stages = []
            files = [str(tmp_path / (name + ".fusions.tsv")), str(tmp_path / (name + ".discarded.tsv"))]
            pipeline.run_workflow(files[0], files[1], log=lambda stage, remaining: stages.append((stage, remaining)))
            outputs[name] = (stages, open(files[0]).read(), open(files[1]).read())
        assert outputs["rccl"] == outputs["plain"]
        assert "filter_mismappers" in through_rccl.timings and dict(outputs["rccl"][0])["filter_mismappers"] > 0
    finally:
        rccl.ncclCommDestroy(communicator)


def test_gene_set_capacity_is_reported_not_truncated(built, tmp_path):
    from arriba_amd.pipeline import ArribaError
    prefix = datasets.generate({"args": ["--seed", "13", "--fragments", "3000", "--contigs", "3", "--contig-len", "300000", "--junctions", "80", "--genes-per-mb", "40", "--gene-stack", "24"]}, str(tmp_path))
    with pytest.raises(ArribaError, match="gene set exceeded the device capacity"):
        parity.run_read_level(parity.open_session, prefix)


def test_event_level_predicates_match_reference(built, dataset_files, tmp_path):
    """filter_both_intronic, filter_short_anchor, filter_end_to_end_fusions, filter_no_coverage on the GPU: against the committed dumps, and against
    the reference run live with the filters in front of them switched off, so that thousands of candidates reach every predicate"""
    session, pipeline = parity.run_read_level(parity.open_session, dataset_files("toy3k"))
    assert parity.check_event_predicates(session, pipeline, conftest.golden_dir("toy3k"))["both_intronic"] > 20
    session, pipeline = parity.run_read_level(parity.open_session, dataset_files("toy3k"))
    assert parity.check_event_chain(session, pipeline, conftest.golden_dir("toy3k"))[-1] > 0
    if not datasets.reference_available():
        return
    spec = {"args": ["--seed", "33", "--fragments", "60000", "--normal-mult", "0.3", "--contigs", "6", "--contig-len", "500000", "--junctions", "600", "--dup", "0.1"]}
    prefix = datasets.generate(spec, str(tmp_path))
    dump = str(tmp_path / "dump")
    os.makedirs(dump)
    os.environ["ARRIBA_ORACLE_DUMP_LISTS"] = "0"
    try:
        log = datasets.run_reference(prefix, dump, disable_filters=["relative_support", "min_support", "non_coding_neighbors", "intragenic_exonic", "homologs", "merge_adjacent", "multimappers"],
                                     extra_args=["-M", "1"])  # -M 1: recover_many_spliced has something to recover
    finally:
        del os.environ["ARRIBA_ORACLE_DUMP_LISTS"]
    with open(os.path.join(dump, "reference.log"), "w") as out:
        out.write(log)
    session, pipeline = parity.run_read_level(parity.open_session, prefix)
    discarded = parity.check_event_predicates(session, pipeline, dump)
    assert discarded["filter_in_vitro"] > 3000 and discarded["recover_both_spliced"] >= 0 and discarded["select_most_supported_breakpoints"] > 5000 and discarded["recover_many_spliced"] > 0 and discarded["filter_marginal_read_through"] > 0, discarded
    assert min(discarded[stage] for stage in ("both_intronic", "filter_short_anchor", "filter_end_to_end_fusions", "filter_no_coverage")) > 1000, discarded
    session, pipeline = parity.run_read_level(parity.open_session, prefix)
    assert parity.check_event_chain(session, pipeline, dump)[-1] > 1000


def test_chain_to_no_coverage_without_injected_state(built, dataset_files, tmp_path):
    """find_fusions ... filter_no_coverage (the reference's stages 18-35, default filters) on the GPU with nothing taken from the reference: golden
    datasets (one with recurrent internal tandem duplications: appended read lists, recovered candidates, un-filtered reads) and a live 120 k run"""
    golden = conftest.golden_dir("itd6k")
    session, pipeline = parity.run_read_level(parity.open_session, dataset_files("itd6k"))
    pipeline.find_fusions()
    pipeline.merge_adjacent_fusions()
    assert parity.check_read_lists(session, pipeline, golden, "merge_adjacent_fusions") > 2000
    assert parity.check_recover_itd(session, pipeline, golden)[0] >= 3
    for name in ("toy3k", "itd6k"):
        session, pipeline = parity.run_read_level(parity.open_session, dataset_files(name))
        assert parity.check_chain_to_no_coverage(session, pipeline, conftest.golden_dir(name))[-1] > 0
    if not datasets.reference_available():
        return
    spec = {"args": ["--seed", "23", "--fragments", "120000", "--normal-mult", "0.3", "--contigs", "8", "--contig-len", "600000", "--junctions", "1500", "--dup", "0.15"]}
    prefix = datasets.generate(spec, str(tmp_path))
    dump = str(tmp_path / "dump")
    os.makedirs(dump)
    os.environ["ARRIBA_ORACLE_DUMP_LISTS"] = "0"
    try:
        log = datasets.run_reference(prefix, dump)
    finally:
        del os.environ["ARRIBA_ORACLE_DUMP_LISTS"]
    with open(os.path.join(dump, "reference.log"), "w") as out:
        out.write(log)
    session, pipeline = parity.run_read_level(parity.open_session, prefix)
    assert parity.check_chain_to_no_coverage(session, pipeline, dump)[-1] > 100


def test_homologs_and_chain_to_the_last_filter(built, dataset_files, tmp_path):
    """filter_homologs on the GPU (verdicts: one thread per gene pair; elimination on the host): a sample with families of homologous genes with the
    reference's filters in front switched off (thousands of candidates), then the reference's stages 18-41 (to recover_isoforms, the last filter, and assign_confidence) as one
    chain with nothing taken from the reference, on the golden datasets and on a live run"""
    session, pipeline = parity.run_read_level(parity.open_session, dataset_files("homologs8k_open"))
    entering, discarded = parity.check_homologs(session, pipeline, conftest.golden_dir("homologs8k_open"), state_from="recover_many_spliced")
    assert entering > 3000 and discarded > 400
    for name in ("homologs8k", "toy3k"):
        session, pipeline = parity.run_read_level(parity.open_session, dataset_files(name))
        counts, reads_discarded, confidence_levels = parity.check_chain_to_isoforms(session, pipeline, conftest.golden_dir(name))
        assert counts[0] > counts[-1] > 0
    session, pipeline = parity.run_read_level(parity.open_session, dataset_files("toy3k"))
    assert parity.check_isoforms(session, pipeline, conftest.golden_dir("toy3k")) == (46, 2)
    session, pipeline = parity.run_read_level(parity.open_session, dataset_files("toy3k"))
    assert parity.check_confidence(session, pipeline, conftest.golden_dir("toy3k")) == confidence_levels
    if not datasets.reference_available():
        return
    spec = {"args": ["--seed", "41", "--fragments", "120000", "--normal-mult", "0.3", "--contigs", "8", "--contig-len", "600000", "--junctions", "1500", "--dup", "0.15", "--homolog-families", "60"]}
    prefix = datasets.generate(spec, str(tmp_path))
    dump = str(tmp_path / "dump")
    os.makedirs(dump)
    os.environ["ARRIBA_ORACLE_DUMP_LISTS"] = "0"
    try:
        log = datasets.run_reference(prefix, dump)
    finally:
        del os.environ["ARRIBA_ORACLE_DUMP_LISTS"]
    with open(os.path.join(dump, "reference.log"), "w") as out:
        out.write(log)
    session, pipeline = parity.run_read_level(parity.open_session, prefix)
    counts, reads_discarded, confidence_levels = parity.check_chain_to_isoforms(session, pipeline, dump)
    assert counts[-5] > counts[-4] > 100 and counts[-1] > counts[-2], counts


def test_blacklist_and_known_fusions(built, dataset_files, tmp_path):
    """filter_blacklisted_ranges and recover_known_fusions on the GPU (one thread per candidate through the genome-bin index of the rules): from injected
    state and inside the chain find_fusions ... assign_confidence, on the golden dataset and on a live run with 2000 junctions"""
    prefix = dataset_files("rules8k")
    golden = conftest.golden_dir("rules8k")
    session, pipeline = parity.run_read_level(parity.open_session, prefix)
    recovered, blacklisted = parity.check_range_rules(session, pipeline, golden, prefix)
    assert recovered > 50 and blacklisted > 10
    session, pipeline = parity.run_read_level(parity.open_session, prefix)
    counts, reads_discarded, confidence_levels = parity.check_chain_to_isoforms(session, pipeline, golden, rules_prefix=prefix)
    assert len(counts) == 19 and counts[-1] > 0
    if not datasets.reference_available():
        return
    spec = {"args": ["--seed", "47", "--fragments", "100000", "--normal-mult", "0.3", "--contigs", "6", "--contig-len", "500000", "--junctions", "2000", "--dup", "0.1", "--rule-files"], "rule_files": True}
    prefix = datasets.generate(spec, str(tmp_path))
    dump = str(tmp_path / "dump")
    os.makedirs(dump)
    os.environ["ARRIBA_ORACLE_DUMP_LISTS"] = "0"
    try:
        log = datasets.run_reference(prefix, dump, spec)
    finally:
        del os.environ["ARRIBA_ORACLE_DUMP_LISTS"]
    with open(os.path.join(dump, "reference.log"), "w") as out:
        out.write(log)
    session, pipeline = parity.run_read_level(parity.open_session, prefix)
    counts, reads_discarded, confidence_levels = parity.check_chain_to_isoforms(session, pipeline, dump, rules_prefix=prefix)
    assert counts[5] > counts[4] + 1000 and counts[11] < counts[10] - 100, counts  # known fusions recovered, candidates blacklisted


def test_output_files_equal_the_reference(built, dataset_files, tmp_path):
    """After the whole chain on the GPU both output files equal the reference's byte for byte (fusions.tsv with the fusion transcript assembled from
    the read pileups, best transcripts, peptide and reading frame); golden datasets and a live run (tens of thousands of discarded candidates)"""
    for name in ("toy3k", "rules8k"):
        prefix = dataset_files(name)
        session, pipeline = parity.run_read_level(parity.open_session, prefix)
        parity.check_chain_to_isoforms(session, pipeline, conftest.golden_dir(name), rules_prefix=prefix if name == "rules8k" else None)
        os.makedirs(str(tmp_path / name))
        fusions, discarded = parity.check_output_files(session, pipeline, conftest.golden_dir(name), str(tmp_path / name), rules_prefix=prefix if name == "rules8k" else None)
        assert fusions > 40 and discarded > 1500
    if not datasets.reference_available():
        return
    spec = {"args": ["--seed", "53", "--fragments", "120000", "--normal-mult", "0.3", "--contigs", "8", "--contig-len", "600000", "--junctions", "1500", "--dup", "0.15"]}
    prefix = datasets.generate(spec, str(tmp_path))
    dump = str(tmp_path / "dump")
    os.makedirs(dump)
    os.environ["ARRIBA_ORACLE_DUMP_LISTS"] = "0"
    try:
        log = datasets.run_reference(prefix, dump)
    finally:
        del os.environ["ARRIBA_ORACLE_DUMP_LISTS"]
    with open(os.path.join(dump, "reference.log"), "w") as out:
        out.write(log)
    session, pipeline = parity.run_read_level(parity.open_session, prefix)
    parity.check_chain_to_isoforms(session, pipeline, dump)
    os.makedirs(str(tmp_path / "mine"))
    fusions, discarded = parity.check_output_files(session, pipeline, dump, str(tmp_path / "mine"), reference_prefix=prefix)
    assert fusions > 100 and discarded > 30000


def test_workflow_from_input_files_to_output_files(built, dataset_files, tmp_path):
    """FASTA + GTF + BAM (+ blacklist and known fusions) -> fusions.tsv + discarded.tsv on the GPU through DevicePipeline.run_workflow with the reference's
    default parameters, nothing taken from the reference: both files byte-identical to the reference's (golden datasets; a live run of 150 k fragments)"""
    for name in ("toy3k", "rules8k", "homologs8k", "toy3k_fill", "wgs8k"):
        os.makedirs(str(tmp_path / name))
        stages = parity.check_workflow(dataset_files(name), conftest.golden_dir(name), str(tmp_path / name), rules=name in ("rules8k", "wgs8k"), fill_sequence_gaps=name == "toy3k_fill", structural_variants=name == "wgs8k")
        assert stages[-1][0] == "recover_isoforms" and stages[-1][1] > 40
    if not datasets.reference_available():
        return
    spec = {"args": ["--seed", "59", "--fragments", "150000", "--normal-mult", "0.3", "--contigs", "8", "--contig-len", "600000", "--junctions", "1500", "--dup", "0.15", "--rule-files", "--homolog-families", "20",
                     "--itd-hotspots", "3", "--itd-hotspot-frac", "0.02"], "rule_files": True, "structural_variants": True}
    prefix = datasets.generate(spec, str(tmp_path))
    dump = str(tmp_path / "dump")
    os.makedirs(dump)
    switches = {"ARRIBA_ORACLE_DUMP_LISTS": "0", "ARRIBA_ORACLE_DUMP_READS": "0", "ARRIBA_ORACLE_DUMP_STAGES": "key"}
    os.environ.update(switches)
    try:
        log = datasets.run_reference(prefix, dump, spec)
    finally:
        for key in switches:
            del os.environ[key]
    with open(os.path.join(dump, "reference.log"), "w") as out:
        out.write(log)
    os.makedirs(str(tmp_path / "mine"))
    stages = parity.check_workflow(prefix, dump, str(tmp_path / "mine"), rules=True, reference_prefix=prefix, structural_variants=True)
    assert stages[-1][1] > 300 and dict(stages)["mark_genomic_support"] > 10000


def _with_implicit_lists(window_entries):
    """context manager: the discordant-mate lists of the candidates implicit (fusion_core.hpp: CandidateTable::discordant_before), expanded in windows of so many entries"""
    import contextlib
    @contextlib.contextmanager
    def manager():
        os.environ["ARRIBA_IMPLICIT_LISTS"] = "1"
        os.environ["ARRIBA_LIST_WINDOW_ENTRIES"] = str(window_entries)
        try:
            yield
        finally:
            del os.environ["ARRIBA_IMPLICIT_LISTS"], os.environ["ARRIBA_LIST_WINDOW_ENTRIES"]
    return manager()


def test_kernels_of_a_wavefront_per_item_are_launched_in_chunks(built, dataset_files, tmp_path):
    """review of round 5, item 8 (the bug of round 5: 22.7 M workgroups x 256 lanes are more work-items than a launch takes, the runtime refused, nobody asked).  Every kernel that
    gives an ITEM a wavefront -- a candidate with its read lists, a queued bucket of discordant mates, a long list, a large group of select_best, a pair of homologous genes --
    is launched in chunks of < 2^26 items with the index of its first item (device_utils.hpp: for_each_wave_chunk; tests/test_host_and_device_logic.py audits the sources for
    launches that do not).  Here the chunk is SIXTEEN items (the smallest: chunks are multiples of 16, a workgroup's worth of wavefronts), every list of more than 4 entries is "long", every group of select_best of two goes to the wavefront's fold, the discordant
    lists are implicit and expanded in windows of 1 024 entries: each of those kernels runs in many chunks on the golden samples, and counts and both files must be the reference's."""
    knobs = {"ARRIBA_WAVE_CHUNK": "16", "ARRIBA_LONG_LIST": "4", "ARRIBA_SELECT_BEST_SMALL": "1"}
    os.environ.update(knobs)
    try:
        for implicit in (False, True):
            for name in ("toy3k", "homologs8k", "itd6k"):
                directory = str(tmp_path / ("%s_%d" % (name, implicit)))
                os.makedirs(directory)
                if implicit:
                    with _with_implicit_lists(1024):
                        stages = parity.check_workflow(dataset_files(name), conftest.golden_dir(name), directory)
                else:
                    stages = parity.check_workflow(dataset_files(name), conftest.golden_dir(name), directory)
                assert stages[-1][0] == "recover_isoforms" and stages[-1][1] > 40
    finally:
        for key in knobs:
            del os.environ[key]


def test_implicit_discordant_lists_give_the_files_of_the_reference(built, dataset_files, tmp_path):
    """review of round 4, item 1: a candidate's discordant mates as (bucket range, predicate, cut-off at -U) instead of a list -- what BASELINE.json's config 3 needs at its stated
    size (10^8 fragments with -U 32767 would list 92.7 G reads).  The lists are made implicit by force on the golden datasets and expanded in windows so small (1 024 entries) that a
    toy sample has dozens of them: every stage that walks read lists -- filter_multimappers, filter_both_intronic, filter_in_vitro, recover_both_spliced, the jobs and the recount
    of filter_mismappers, recover_internal_tandem_duplication and merge_adjacent_fusions on a sample with ITD hot spots (split-read lists appended, the lists rebuilt), the lists the
    writer fetches -- must give the reference's counts and both output files byte for byte; and the three lists of every candidate, fetched window by window, must be the
    reference's (parity.check_read_lists)."""
    with _with_implicit_lists(1024):
        for name in ("toy3k", "rules8k", "homologs8k", "wgs8k", "itd6k"):
            os.makedirs(str(tmp_path / name))
            stages = parity.check_workflow(dataset_files(name), conftest.golden_dir(name), str(tmp_path / name), rules=name in ("rules8k", "wgs8k"), structural_variants=name == "wgs8k")
            assert stages[-1][0] == "recover_isoforms" and stages[-1][1] > 40
        for name in ("toy3k", "itd6k"):
            session, pipeline = parity.run_read_level(parity.open_session, dataset_files(name))
            pipeline.find_fusions()
            assert pipeline.fusion_stats()["list_entries"] > 1024
            parity.check_candidates(session, pipeline, conftest.golden_dir(name))
            pipeline.merge_adjacent_fusions()
            parity.check_read_lists(session, pipeline, conftest.golden_dir(name), "merge_adjacent_fusions")
            pipeline.close(); session.close()
    if not datasets.reference_available():
        return
    spec = {"args": ["--seed", "59", "--fragments", "150000", "--normal-mult", "0.3", "--contigs", "8", "--contig-len", "600000", "--junctions", "1500", "--dup", "0.15", "--rule-files", "--homolog-families", "20",
                     "--itd-hotspots", "3", "--itd-hotspot-frac", "0.02"], "rule_files": True, "structural_variants": True}
    prefix = datasets.generate(spec, str(tmp_path))
    dump = str(tmp_path / "dump")
    os.makedirs(dump)
    switches = {"ARRIBA_ORACLE_DUMP_LISTS": "0", "ARRIBA_ORACLE_DUMP_READS": "0", "ARRIBA_ORACLE_DUMP_STAGES": "key"}
    os.environ.update(switches)
    try:
        log = datasets.run_reference(prefix, dump, spec)
    finally:
        for key in switches:
            del os.environ[key]
    with open(os.path.join(dump, "reference.log"), "w") as out:
        out.write(log)
    os.makedirs(str(tmp_path / "mine"))
    with _with_implicit_lists(20000):
        stages = parity.check_workflow(prefix, dump, str(tmp_path / "mine"), rules=True, reference_prefix=prefix, structural_variants=True, device_ingest=True)
    assert stages[-1][1] > 300


def test_workflow_with_non_default_options_against_the_live_reference(built, tmp_path):
    """the whole workflow on the GPU with 19 options away from their defaults, two filters off and every optional input file given, against the reference run live"""
    if not datasets.reference_available():
        pytest.skip("needs the oracle build of the reference (oracle/_ref)")
    stages = parity.check_workflow_with_non_default_options(60000, str(tmp_path))
    assert dict(stages)["mark_genomic_support"] > 1000 and stages[-1][1] > 200


@pytest.mark.parametrize("kind", ["indels_and_non_template_bases", "single_end", "long_reads_multimappers", "short_stranded_single_end", "soft_clips_and_n_bases", "missing_hi_tags"])
def test_workflow_on_other_kinds_of_libraries_against_the_live_reference(kind, built, tmp_path):
    """CIGAR operations I and D, non-template bases, single-end libraries, reads of 60 and 150 nt, soft-clipped supplementary alignments, N bases (the golden datasets have
    none of these) through the kernels of the GPU,
    against the reference run live: every count and both output files"""
    if not datasets.reference_available():
        pytest.skip("needs the oracle build of the reference (oracle/_ref)")
    import test_host_and_device_logic as cpu_tier
    cpu_tier.check_library_against_the_live_reference(kind, str(tmp_path))


def test_cpp_workflow_driver_over_the_c_abis(built, dataset_files, tmp_path):
    """arriba_amd/lib/arriba_gpu_workflow: the reference's main() behind its option parser as C++ over the two C ABIs (no Python in the loop), on the GPU:
    both output files equal the reference's byte for byte, without and with every optional input file"""
    import gzip
    import subprocess
    driver = os.path.join(conftest.ROOT, "arriba_amd", "lib", "arriba_gpu_workflow")
    os.chmod(driver, 0o755)  # the snapshot that travels to the GPU box may drop the mode
    for name in ("toy3k", "wgs8k"):
        prefix = dataset_files(name)
        os.makedirs(str(tmp_path / name))
        outputs = [str(tmp_path / name / "fusions.tsv"), str(tmp_path / name / "discarded.tsv")]
        optional = [prefix + suffix for suffix in (".blacklist.tsv", ".known_fusions.tsv", ".tags.tsv", ".protein_domains.gff3", ".sv.tsv")] if name == "wgs8k" else []
        result = subprocess.run([driver, prefix + ".fa", prefix + ".gtf", prefix + ".bam"] + outputs + optional,
                                stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True)
        assert result.returncode == 0, result.stderr[-2000:]
        for mine, reference in zip(outputs, ("fusions.tsv.gz", "discarded.tsv.gz")):
            assert open(mine).read() == gzip.open(os.path.join(conftest.golden_dir(name), reference), "rt").read(), (name, reference)


def _batch_columns_of_both(prefix, piece_bytes=8 << 20):
    """(host ingest's batch, device ingest's batch) of one BAM file as comparable dictionaries"""
    from arriba_amd.pipeline import DevicePipeline, HostSession
    import test_host_and_device_logic as cpu_tier
    host = parity.open_session(prefix)
    expected = cpu_tier._batch_columns(host)
    expected["coverage"] = int(host._lib.ahost_coverage_checksum(host._session))
    session = HostSession(prefix + ".fa", prefix + ".gtf")
    pipeline = DevicePipeline(session, bam=prefix + ".bam", piece_bytes=piece_bytes)
    return host, expected, session, pipeline, cpu_tier._device_batch_columns(session, pipeline)


@pytest.mark.parametrize("name", ["toy3k", "shuffled2k", "itd6k", "shuffled_dups_40k", "stranded_multimappers_20k"])
def test_device_ingest_builds_the_batch_of_the_host_ingest(name, built, tmp_path):
    """read_chimeric_alignments in HBM (agpu_ingest.hip): the record chain cut by speculative segments, records collated with the radix sort, one thread per read
    name replaying the reference's loop, the name order (first-occurrence fast path and the chunked string sort), pack: every column, pool and name of the batch,
    coverage_t, counters and the strandedness vote equal the host ingest (which is byte-identical to the reference's read table)"""
    import test_host_and_device_logic as cpu_tier
    prefix = datasets.generate({"args": cpu_tier.DEVICE_INGEST_DATASETS[name]}, str(tmp_path))
    host, expected, session, pipeline, columns = _batch_columns_of_both(prefix, piece_bytes=1 << 20)
    different = [key for key in expected if expected[key] != columns[key]]
    assert not different, different
    assert pipeline.ingest_result.names_were_sorted == (0 if "--shuffle" in cpu_tier.DEVICE_INGEST_DATASETS[name] else 1)
    assert pipeline.detect_strandedness() == host.detect_strandedness()


@pytest.mark.parametrize("name", ["mid30k", "stranded_multimappers_20k", "shuffled_dups_40k"])
def test_front_of_the_ingest_in_windows_and_behind_the_last_piece(name, built, tmp_path, monkeypatch):
    """The front of the ingest (record chain, record keys, runs of one name, the loop body per name) runs in windows of the stream while the pieces arrive, and behind the last
    piece only when it has to: windows of 1 MB that end 64 KB in front of the bytes that have arrived (dozens of them over these files, records and runs of a name across their
    borders), the default (one window: the files are smaller than 128 MB) and ARRIBA_INGEST_WINDOWS=0 (the way of round 2) give the batch of the host ingest; a file whose
    mates lie apart (--separate-mates: a name in two places of the stream) is noticed by the keys of the runs and done the other way"""
    import test_host_and_device_logic as cpu_tier
    from arriba_amd.pipeline import DevicePipeline, HostSession
    prefix = datasets.generate({"args": cpu_tier.DEVICE_INGEST_DATASETS[name]}, str(tmp_path))
    host = parity.open_session(prefix)
    expected = cpu_tier._batch_columns(host)
    expected["coverage"] = int(host._lib.ahost_coverage_checksum(host._session))
    apart = "--separate-mates" in cpu_tier.DEVICE_INGEST_DATASETS[name]
    for knob, windows in (("1048576,65536", None), (None, 1), ("0", 0), ("1048576,1", None)):
        if knob is None:
            monkeypatch.delenv("ARRIBA_INGEST_WINDOWS", raising=False)
        else:
            monkeypatch.setenv("ARRIBA_INGEST_WINDOWS", knob)
        session = HostSession(prefix + ".fa", prefix + ".gtf")
        pipeline = DevicePipeline(session, bam=prefix + ".bam", piece_bytes=1 << 20)
        columns = cpu_tier._device_batch_columns(session, pipeline)
        assert [key for key in expected if expected[key] != columns[key]] == [], knob
        made = pipeline.ingest_result.windows
        if apart:
            assert made == 0, (knob, made)
        elif windows is None:
            assert made >= os.path.getsize(prefix + ".bam") // (2 << 20) and made >= 3, (knob, made)  # (a piece of 1 MB of the file holds a little less than 1 MB of the stream: a window every second piece)
        else:
            assert made == windows, (knob, made)
        assert pipeline.detect_strandedness() == host.detect_strandedness()


def test_device_ingest_of_a_file_without_records(built, tmp_path):
    """a BAM file that is a header and nothing else (raw, and as one stored BGZF block + end-of-file marker): the last -- and only -- window of the ingest has no segment, no
    record and no run; the workflow ends with the reference's 'no normal reads found' (source/read_chimeric_alignments.cpp:759), as the host ingest does"""
    import struct
    import test_host_and_device_logic as cpu_tier
    from arriba_amd.pipeline import ArribaError, DevicePipeline, HostSession
    prefix = datasets.generate({"args": ["--seed", "2", "--fragments", "10", "--contigs", "2", "--contig-len", "150000", "--junctions", "5", "--reference-only"]}, str(tmp_path))
    names = [b"1", b"2"]
    header = b"BAM\x01" + struct.pack("<i", 0) + struct.pack("<i", len(names))
    for name in names:
        header += struct.pack("<i", len(name) + 1) + name + b"\x00" + struct.pack("<i", 150000)
    open(str(tmp_path / "raw.bam"), "wb").write(header)
    cpu_tier._write_bgzf(str(tmp_path / "stored.bam"), header, 0)
    for variant in ("raw.bam", "stored.bam"):
        with pytest.raises(ArribaError, match="no normal reads found"):
            DevicePipeline(HostSession(prefix + ".fa", prefix + ".gtf"), bam=str(tmp_path / variant), piece_bytes=1 << 20)


def test_device_ingest_at_scale_and_every_container(built, tmp_path):
    """1.2 M fragments + 0.6 M ordinary pairs (more segments, groups and fragments than any launch grid cap), shuffled names: the batch equals the host ingest's;
    the same stream as deflated BGZF and as raw BAM gives the same batch"""
    import gzip
    import test_host_and_device_logic as cpu_tier
    from arriba_amd.pipeline import DevicePipeline, HostSession
    spec = {"args": ["--seed", "41", "--fragments", "1200000", "--normal-mult", "0.5", "--contigs", "8", "--contig-len", "2000000", "--junctions", "20000", "--dup", "0.2", "--shuffle"]}
    prefix = datasets.generate(spec, str(tmp_path))
    host, expected, session, pipeline, columns = _batch_columns_of_both(prefix, piece_bytes=64 << 20)
    different = [key for key in expected if expected[key] != columns[key]]
    assert not different, different
    assert expected["n"] > 1100000 and pipeline.ingest_result.names_were_sorted == 0
    assert pipeline.ingest_result.windows >= 2  # (64 MB pieces of a file of ~0.8 GB: a window when 512 MB have arrived, and the last one)
    payload = gzip.open(prefix + ".bam", "rb").read()
    open(str(tmp_path / "raw.bam"), "wb").write(payload)
    cpu_tier._write_bgzf(str(tmp_path / "deflated.bam"), payload, 1)
    for variant in ("raw.bam", "deflated.bam"):
        other = HostSession(prefix + ".fa", prefix + ".gtf")
        rows = cpu_tier._device_batch_columns(other, DevicePipeline(other, bam=str(tmp_path / variant), piece_bytes=32 << 20))
        assert [key for key in expected if expected[key] != rows[key]] == [], variant


def test_workflow_from_the_bam_file_through_the_device_ingest(built, dataset_files, tmp_path):
    """BAM bytes -> fusions.tsv + discarded.tsv with read_chimeric_alignments on the GPU and nothing of the batch on the host: byte-identical to the reference's files"""
    for name in ("toy3k", "rules8k", "homologs8k", "toy3k_fill", "wgs8k"):
        os.makedirs(str(tmp_path / name))
        stages = parity.check_workflow(dataset_files(name), conftest.golden_dir(name), str(tmp_path / name), rules=name in ("rules8k", "wgs8k"), fill_sequence_gaps=name == "toy3k_fill",
                                       structural_variants=name == "wgs8k", device_ingest=True)
        assert stages[-1][0] == "recover_isoforms" and stages[-1][1] > 40
    if datasets.reference_available():
        os.makedirs(str(tmp_path / "options"))
        stages = parity.check_workflow_with_non_default_options(60000, str(tmp_path / "options"), device_ingest=True)
        assert dict(stages)["mark_genomic_support"] > 1000 and stages[-1][1] > 200


def _run_plain_reference(prefix, directory, extra=()):
    """the oracle build of the reference without the dump hooks (oracle/_ref/arriba_ref): log + both output files"""
    import re
    import subprocess
    os.makedirs(directory, exist_ok=True)
    command = [datasets.ARRIBA_REF, "-x", prefix + ".bam", "-g", prefix + ".gtf", "-a", prefix + ".fa", "-o", prefix + ".fusions.tsv", "-O", prefix + ".discarded.tsv", "-f", "blacklist"] + list(extra)
    import time
    started = time.time()
    result = subprocess.run(command, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True)
    assert result.returncode == 0, result.stdout[-2000:]
    try:  # how long the reference takes on the host cores of the GPU box, by sample size: kept for the fit of its run time (profiles/)
        total = re.search(r"Reading chimeric alignments from .*\(total=(\d+)\)", result.stdout.replace("\n", " "))
        with open(os.path.join(conftest.ROOT, "gpurun_out", "reference_times.txt"), "a") as times:
            times.write("%s fragments (total=%s) %s: %.1f s\n" % (os.path.basename(prefix), total.group(1) if total else "?", " ".join(extra), time.time() - started))
    except OSError:
        pass
    with open(os.path.join(directory, "reference.log"), "w") as out:
        out.write(re.sub(r"\[\d{4}-\d\d-\d\dT\d\d:\d\d:\d\d\] ", "", result.stdout))


def test_workflow_at_config_scale_against_the_live_reference(built, tmp_path):
    """BASELINE.json config 2 at half size -- 5 M chimeric fragments of exactly the workload bench.py times (13 M BAM records, 2.7 GB) -- through the whole workflow
    on the GPU, read_chimeric_alignments included, against the unmodified reference run on the same files: every "(remaining=N)" of its log, fusions.tsv and
    discarded.tsv byte for byte.  (A launch-grid bug once skipped 95 % of a 10 M batch while every smaller test was green: profiles/README.md.)"""
    import subprocess
    import bench
    if not os.path.exists(datasets.ARRIBA_REF):
        pytest.skip("oracle/_ref/arriba_ref did not travel with the repository")
    fragments = int(os.environ.get("ARRIBA_SCALE_TEST_FRAGMENTS", "5000000"))
    prefix = str(tmp_path / "scale")
    subprocess.run([datasets.GEN_SYNTH, "--out", prefix, "--threads", "32"] + bench.workload_args(fragments, 1000), check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    _run_plain_reference(prefix, str(tmp_path / "reference"))
    os.makedirs(str(tmp_path / "mine"))
    stages = parity.check_workflow(prefix, str(tmp_path / "reference"), str(tmp_path / "mine"), reference_prefix=prefix, device_ingest=True)
    assert dict(stages)["find_fusions"] > fragments // 5 and stages[-1][1] > 100


def test_reference_data_at_the_scale_of_hg38_against_the_live_reference(built, tmp_path):
    """review of round 5 (missing, item 4; BASELINE.json config 2 says "hg38 exon index"): every other test and the bench use a genome of 288 Mb with 5.8 k genes.  Here the reference
    data have the SIZE of hg38 / GENCODE -- 24 contigs of 130 Mb = 3.12 Gb of bases, 6.5 x 10^4 genes, 1.4 x 10^6 exons (tools/gen_synth --contig-len 130000000 --genes-per-mb 20; no
    download: there is no network) -- so that the flattened interval index (source/annotation.t.hpp:25-45) holds 3 x 10^6 keys per kind, coverage_t (source/read_stats.hpp:17-27)
    1.56 x 10^8 windows, the genome 3.1 GB of HBM, and make_kmer_index (source/filter_mismappers.cpp:47-84) marks its windows in a bitmap over 3.1 G positions: 1 M chimeric
    fragments (4.7 M records) against the unmodified reference run on the same files -- every "(remaining=N)" of its log, fusions.tsv and discarded.tsv byte for byte.  What the
    device holds for the reference data and what the kernels that depend on their size take is written to gpurun_out/hg38_scale.json (profiles/r06*_hg38_scale.json)."""
    import json
    import subprocess
    import time
    import ctypes
    from arriba_amd.pipeline import WorkflowSession

    def device_memory():
        # (through the HIP runtime the device library is linked against.  Until round 6 this test asked torch for the free memory: torch brings a ROCm stack of its own, and the
        # interpreter that had BOTH initialised died in its teardown -- "double free or corruption" inside the destructor of torch/lib/libcaffe2_nvrtc.so, after "60 passed":
        # profiles/r06p_fini.txt.  The tests of this tier keep torch in processes of their own.)
        hip = ctypes.CDLL("/opt/rocm/lib/libamdhip64.so")
        free_bytes, total_bytes = ctypes.c_size_t(), ctypes.c_size_t()
        assert hip.hipMemGetInfo(ctypes.byref(free_bytes), ctypes.byref(total_bytes)) == 0
        return free_bytes.value, total_bytes.value
    if not os.path.exists(datasets.ARRIBA_REF):
        pytest.skip("oracle/_ref/arriba_ref did not travel with the repository")
    fragments = int(os.environ.get("ARRIBA_HG38_TEST_FRAGMENTS", "1000000"))
    prefix = str(tmp_path / "hg38")
    started = time.time()
    subprocess.run([datasets.GEN_SYNTH, "--out", prefix, "--threads", "32", "--seed", "3", "--fragments", str(fragments), "--contigs", "24", "--contig-len", "130000000", "--genes-per-mb", "20", "--junctions", "20000"],
                   check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    generated = time.time() - started
    started = time.time()
    _run_plain_reference(prefix, str(tmp_path / "reference"))
    reference_seconds = time.time() - started
    os.makedirs(str(tmp_path / "mine"))
    stages = parity.check_workflow(prefix, str(tmp_path / "reference"), str(tmp_path / "mine"), reference_prefix=prefix, device_ingest=True)
    assert dict(stages)["find_fusions"] > fragments // 5 and stages[-1][1] > 100
    # ... and the product path (the C++ driver, a resident session) on the same files, for the numbers: the same files again, HBM of the reference data, the kernels that scale with them
    free_before, total = device_memory()
    started = time.time()
    session = WorkflowSession(prefix + ".fa", prefix + ".gtf")
    opened = time.time() - started
    free_open, _ = device_memory()
    session.set_profiling(True)
    started = time.time()
    report = session.sample(prefix + ".bam", str(tmp_path / "session.tsv"), str(tmp_path / "session.discarded.tsv"))
    sample_seconds = time.time() - started
    free_after, _ = device_memory()
    kernels = {}
    for name, ms, size in session.kernel_profile():
        kernels[name] = kernels.get(name, 0.0) + ms
    session.close()
    for mine, theirs in (("session.tsv", ".fusions.tsv"), ("session.discarded.tsv", ".discarded.tsv")):
        assert open(str(tmp_path / mine), "rb").read() == open(prefix + theirs, "rb").read(), mine
    genome_bytes = os.path.getsize(prefix + ".fa")
    record = {"what": "synthetic reference data of the size of hg38 / GENCODE: 24 x 130 Mb, %d fragments" % dict(report)["read_chimeric_alignments"], "fasta_GB": round(genome_bytes / 1e9, 2),
              "gtf_MB": round(os.path.getsize(prefix + ".gtf") / 1e6, 1), "generate_seconds": round(generated, 1), "reference_seconds_on_this_box": round(reference_seconds, 1),
              "session_open_seconds": round(opened, 1), "sample_seconds": round(sample_seconds, 2), "hbm_GB_behind_open (annotation + index)": round((free_before - free_open) / 1e9, 2),
              "hbm_GB_behind_the_sample (genome, coverage_t, batch, candidates, k-mer index, scratch)": round((free_before - free_after) / 1e9, 2),
              "coverage_windows": int(genome_bytes / 20), "stages": report,
              "kernel_ms": {name: round(ms, 3) for name, ms in sorted(kernels.items(), key=lambda item: -item[1])[:32]}}
    try:
        with open(os.path.join(conftest.ROOT, "gpurun_out", "hg38_scale.json"), "w") as out:
            json.dump(record, out, indent=1)
    except OSError:
        pass


@pytest.mark.parametrize("name,fragments,normal_mult", [("bench10m", 10000000, None), ("bench20m", 20000000, None), ("normal5m", 5000000, 4)])
def test_bench_sample_of_config_2_against_the_reference(name, fragments, normal_mult, built, tmp_path):
    """BASELINE.json config 2 at FULL size, exactly the sample bench.py times at 10 M (10 444 615 fragments, 26.6 M BAM records) and the same workload at 20 M (20 906 803 fragments): the
    product path -- arriba_workflow_sample of the C++ workflow library, resident session -- against the unmodified reference, which was run once on these very samples where the
    repository is built (tools/make_bench_golden.py; 10 M: 7 min 40 s, 15.9 GB; 20 M: 16 min 38 s, 29 GB -- tests/golden/bench10m, bench20m): the generated BAM file is the one the
    reference read (SHA-256), fusions.tsv is the one it wrote (SHA-256), and the counts of its log are met.  The second sample of every session goes through the queue
    (arriba_workflow_submit: its file fed while the stages of the first run).
    normal5m (round 5; review of round 4, item 7d): the composition SURVEY.md section 8(d)-2 specifies -- 4 N ordinary proper pairs beside the N chimeric fragments, 5.2 M + 20 M here
    (53.5 M BAM records): the ingest skips four records in five, coverage_t and mapped_reads are those of a whole Aligned.out.bam; the reference took 5 min on it."""
    import hashlib
    import json
    import subprocess
    import bench
    from arriba_amd.pipeline import WorkflowSession
    golden = conftest.golden_dir(name)
    meta = json.load(open(os.path.join(golden, "meta.json")))
    prefix = str(tmp_path / "bench")
    arguments = bench.workload_args(fragments, 1000)
    if normal_mult is not None:
        arguments[arguments.index("--normal-mult") + 1] = str(normal_mult)
    subprocess.run([datasets.GEN_SYNTH, "--out", prefix, "--threads", str(bench.cpu_budget())] + arguments, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)

    def sha256(path):
        digest = hashlib.sha256()
        with open(path, "rb") as stream:
            for piece in iter(lambda: stream.read(1 << 24), b""):
                digest.update(piece)
        return digest.hexdigest()
    assert sha256(prefix + ".bam") == meta["bam_sha256"], "the generator drifted from the sample the reference was run on (tests/golden/%s/meta.json)" % name
    session = WorkflowSession(prefix + ".fa", prefix + ".gtf")
    session.submit(prefix + ".bam")
    for repeat in range(2):  # (a resident session: the second sample through the other lane, fed under the stages of the first)
        output = str(tmp_path / ("fusions%d.tsv" % repeat))
        if repeat == 0:
            session.submit(prefix + ".bam")
        discarded = str(tmp_path / "discarded.tsv") if repeat == 1 and "discarded_tsv_sha256" in meta else None  # (review of round 4, item 7c: -O pinned as well, 3.9 M / 7.9 M rows)
        counts = dict(session.sample(prefix + ".bam", output, discarded))
        assert sha256(output) == meta["fusions_tsv_sha256"], repeat
        if discarded:
            assert sha256(discarded) == meta["discarded_tsv_sha256"], "discarded.tsv"
            os.remove(discarded)
    log = open(os.path.join(golden, "reference.log")).read()
    assert counts["read_chimeric_alignments"] == int(re.search(r"Reading chimeric alignments[^\n]*\(total=(\d+)\)", log).group(1))
    for stage, pattern in (("filter_duplicates", "Filtering duplicates"), ("filter_mismatches", "Filtering reads with a mismatch"), ("filter_low_entropy", "Filtering reads with low entropy"), ("merge_adjacent_fusions", "Merging adjacent fusion breakpoints"),
                           ("filter_relative_support", "Filtering fusions with an e-value"), ("filter_in_vitro", "Filtering in vitro-generated fusions"), ("filter_homologs", "Filtering genes with"),
                           ("filter_mismappers", "Re-aligning chimeric reads"), ("recover_isoforms", "Searching for additional isoforms")):
        assert counts[stage] == parity.logged_remaining(log, pattern), stage
    assert counts["recover_isoforms"] == meta["fusions"]
    session.close()


@pytest.mark.parametrize("lists", ["explicit", "implicit"])
def test_mismapper_stress_at_scale_against_the_live_reference(lists, built, tmp_path):
    """(lists: the read lists in memory, or the discordant ones implicit and expanded in windows of 50 M entries -- the way a sample of the stated size of config 3 must go)
    BASELINE.json config 3 at 0.3 M fragments: clipped segments of 40-70 nt copied from the partner gene, -U 32767 (no read is subsampled away before
    filter_mismappers): log counts and both output files against the unmodified reference run here (the reference needs 5 1/2 minutes for 1 M fragments of this workload, and
    24 for the 3 M of the test below, which was run once where the repository is built)"""
    import subprocess
    import bench
    if not os.path.exists(datasets.ARRIBA_REF):
        pytest.skip("oracle/_ref/arriba_ref did not travel with the repository")
    fragments = int(os.environ.get("ARRIBA_STRESS_TEST_FRAGMENTS", "300000"))
    prefix = str(tmp_path / "stress")
    subprocess.run([datasets.GEN_SYNTH, "--out", prefix, "--threads", "32"] + bench.workload_args(fragments, 1000, stress=True), check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    _run_plain_reference(prefix, str(tmp_path / "reference"), extra=["-U", "32767"])
    os.makedirs(str(tmp_path / "mine"))
    import contextlib
    with (_with_implicit_lists(50000000) if lists == "implicit" else contextlib.nullcontext()):
        stages = parity.check_workflow(prefix, str(tmp_path / "reference"), str(tmp_path / "mine"), reference_prefix=prefix, device_ingest=True, params={"subsampling_threshold": 32767})
    assert dict(stages)["filter_mismappers"] > 300


def test_implicit_lists_equal_materialised_lists_at_10m_stress(built, tmp_path):
    """review of round 4, item 1 ("a new GPU test asserts implicit == materialised lists (candidate counters, fusions.tsv) at 10 M stress"): BASELINE.json config 3 at 10 M
    fragments with -U 32767 -- ~9 G list entries, 36 GB: the largest size at which both ways run -- through one resident session, once with the lists in memory and once with the
    discordant lists implicit (windows of what the device has room for): every count of the log and the bytes of fusions.tsv must be the same."""
    import hashlib
    import subprocess
    import bench
    from arriba_amd.pipeline import WorkflowSession
    fragments = int(os.environ.get("ARRIBA_IMPLICIT_TEST_FRAGMENTS", "10000000"))
    prefix = str(tmp_path / "stress")
    subprocess.run([datasets.GEN_SYNTH, "--out", prefix, "--threads", str(bench.cpu_budget())] + bench.workload_args(fragments, 1000, stress=True), check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    sha256 = lambda path: hashlib.sha256(open(path, "rb").read()).hexdigest()
    session = WorkflowSession(prefix + ".fa", prefix + ".gtf", params={"subsampling_threshold": 32767})
    explicit = list(session.sample(prefix + ".bam", str(tmp_path / "explicit.tsv")))
    os.environ["ARRIBA_IMPLICIT_LISTS"] = "1"
    try:
        implicit = list(session.sample(prefix + ".bam", str(tmp_path / "implicit.tsv")))
    finally:
        del os.environ["ARRIBA_IMPLICIT_LISTS"]
    session.close()
    assert dict(explicit)["recover_isoforms"] > 1000
    different = [(a, b) for a, b in zip(explicit, implicit) if a != b]
    assert len(explicit) == len(implicit) and not different, different
    assert sha256(str(tmp_path / "explicit.tsv")) == sha256(str(tmp_path / "implicit.tsv"))


def test_mismapper_stress_of_config_3_against_the_reference(built, tmp_path):
    """BASELINE.json config 3 (mismapper stress, -U 32767) at the largest size the reference finishes in the build container: 3 M fragments -- 3 131 k chimeric fragments, 24 minutes,
    38 GB there (tests/golden/stress3m, tools/make_bench_golden.py --stress); here the candidates of the sample list 3.2 G supporting reads (64-bit list offsets).  The generated
    BAM file is the one the reference read (SHA-256), fusions.tsv the one it wrote (SHA-256), the counts of its log are met."""
    import hashlib
    import json
    import subprocess
    import bench
    from arriba_amd.pipeline import WorkflowSession
    golden = conftest.golden_dir("stress3m")
    meta = json.load(open(os.path.join(golden, "meta.json")))
    prefix = str(tmp_path / "stress")
    subprocess.run([datasets.GEN_SYNTH, "--out", prefix, "--threads", str(bench.cpu_budget())] + bench.workload_args(3000000, 1000, stress=True), check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    sha256 = lambda path: hashlib.sha256(open(path, "rb").read()).hexdigest()
    assert sha256(prefix + ".bam") == meta["bam_sha256"], "the generator drifted from the sample the reference was run on (tests/golden/stress3m/meta.json)"
    session = WorkflowSession(prefix + ".fa", prefix + ".gtf", params={"subsampling_threshold": 32767})
    output = str(tmp_path / "fusions.tsv")
    discarded = str(tmp_path / "discarded.tsv") if "discarded_tsv_sha256" in meta else None
    counts = dict(session.sample(prefix + ".bam", output, discarded))
    assert sha256(output) == meta["fusions_tsv_sha256"]
    if discarded:
        assert hashlib.sha256(open(discarded, "rb").read()).hexdigest() == meta["discarded_tsv_sha256"], "discarded.tsv"
    log = open(os.path.join(golden, "reference.log")).read()
    assert counts["read_chimeric_alignments"] == meta["chimeric_fragments"]
    for stage, pattern in (("filter_duplicates", "Filtering duplicates"), ("merge_adjacent_fusions", "Merging adjacent fusion breakpoints"), ("filter_relative_support", "Filtering fusions with an e-value"),
                           ("filter_in_vitro", "Filtering in vitro-generated fusions"), ("filter_mismappers", "Re-aligning chimeric reads"), ("recover_isoforms", "Searching for additional isoforms")):
        assert counts[stage] == parity.logged_remaining(log, pattern), stage
    assert counts["recover_isoforms"] == meta["fusions"]
    session.close()


def test_filter_mismappers_gives_the_same_verdicts_under_every_schedule(built, tmp_path):
    """review of round 4, item 3: the verdict of a read is a function of the read (source/filter_mismappers.cpp:86-187), but mismapper_heavy_kernel shares a memo table and task
    lists among the 64 lanes of a wavefront and hands the reads of a queue to 5 120 persistent workgroups -- and one GPU box of round 4 once gave 3 reads another verdict
    (DESIGN.md section 2).  The stage is run again and again on two samples -- the mismapper stress of config 3 (clips of 40-70 nt copied from the partner gene, -U 32767: long
    searches, hardly a mis-mapper among them) and a sample with families of homologous genes (hundreds of reads that ARE mis-mappers) --: ten times with the workgroups of the
    product, three times with 7, once with a single one (every read behind the other, one memo table for all of them), once with four wavefronts per SIMD and once with the jobs
    in the other order: the verdict bytes of all runs must be the same bytes."""
    import subprocess
    import bench
    from ctypes import byref, c_uint64
    from arriba_amd.pipeline import DevicePipeline, HostSession
    fragments = int(os.environ.get("ARRIBA_DETERMINISM_FRAGMENTS", "100000"))
    samples = [("stress", bench.workload_args(fragments, 1000, stress=True), {"subsampling_threshold": 32767}, 0),
               ("homologs", ["--seed", "29", "--fragments", str(2 * fragments), "--contigs", "4", "--contig-len", "300000", "--junctions", "80", "--homolog-families", "4"], None, 20)]
    knobs = ("ARRIBA_HEAVY_WORKGROUPS", "ARRIBA_MISMAPPER_JOB_ORDER", "ARRIBA_HEAVY_WAVES", "ARRIBA_STRANDS_TOGETHER", "ARRIBA_MEMO_FRONT", "ARRIBA_MEMO_SLOTS_LOG2", "ARRIBA_TASK_CAPACITY_LOG2")
    for name, arguments, params, least_positive in samples:
        prefix = str(tmp_path / name)
        subprocess.run([datasets.GEN_SYNTH, "--out", prefix, "--threads", "32"] + arguments, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        session = HostSession(prefix + ".fa", prefix + ".gtf")
        pipeline = DevicePipeline(session, params=params, bam=prefix + ".bam", piece_bytes=8 << 20)
        pipeline.run_read_level()
        pipeline.find_fusions()
        pipeline.upload_coverage()
        pipeline.merge_adjacent_fusions(); pipeline.filter_multimappers(); pipeline.estimate_expected_fusions(); pipeline.filter_candidate_predicates(); pipeline.filter_relative_support()
        pipeline.recover_internal_tandem_duplication(); pipeline.filter_both_intronic(); pipeline.filter_in_vitro(); pipeline.recover_both_spliced(); pipeline.select_most_supported_breakpoints()
        pipeline.filter_marginal_read_through(); pipeline.recover_many_spliced(); pipeline.filter_short_anchor(); pipeline.filter_end_to_end(); pipeline.filter_no_coverage()
        pipeline.make_kmer_index()
        pipeline.filter_homologs()
        filters_before = pipeline.filters().copy()
        def verdicts(environment):
            for key in knobs:
                os.environ.pop(key, None)
            os.environ.update(environment)
            try:
                pipeline.set_read_filters(filters_before)  # (a read that a run marked stays marked: every run starts from the filters in front of the stage)
                n_jobs = c_uint64()
                pipeline._check(pipeline.api.mismapper_jobs(pipeline.ctx, byref(n_jobs)))
                out = np.zeros(max(n_jobs.value, 1), dtype=np.uint8)
                pipeline._check(pipeline.api.mismapper_verdicts(pipeline.ctx, int(pipeline.scalars["max_mate_gap"]), 0, 1, out.ctypes.data))
                return n_jobs.value, out, pipeline.filters() == 11  # FILTER_mismappers (source/common.hpp:29-67): the jobs are the same reads whatever their order -- the verdicts per read
            finally:
                for key in knobs:
                    os.environ.pop(key, None)
        n_jobs, first, first_per_read = verdicts({})
        assert n_jobs > fragments // 50 and first.sum() >= least_positive, (name, n_jobs, int(first.sum()))
        assert int(first.sum()) == int(first_per_read.sum()) - int((filters_before == 11).sum()), name
        schedules = [("5120 workgroups, run %d" % k, {}) for k in range(2, 11)] + [("7 workgroups, run %d" % k, {"ARRIBA_HEAVY_WORKGROUPS": "7"}) for k in range(1, 4)] + \
                    [("one workgroup", {"ARRIBA_HEAVY_WORKGROUPS": "1"}), ("four wavefronts per SIMD", {"ARRIBA_HEAVY_WAVES": "4", "ARRIBA_HEAVY_WORKGROUPS": "4096"}),
                     # (round 6) the strands of a segment one after the other instead of in one sweep; no front of the memo in LDS; a memo of 2^10 slots and lists of 64 calls per workgroup:
                     # the tables full and the lists running over, the searches the sweep gives up done by the kernel that holds the recursion
                     ("strand by strand", {"ARRIBA_STRANDS_TOGETHER": "0"}), ("memo and list in HBM only", {"ARRIBA_MEMO_FRONT": "0"}),
                     ("small tables, short lists", {"ARRIBA_MEMO_SLOTS_LOG2": "10", "ARRIBA_TASK_CAPACITY_LOG2": "6"})]
        for label, environment in schedules:
            count, again, _ = verdicts(environment)
            assert count == n_jobs, (name, label)
            different = np.flatnonzero(again != first)
            assert different.size == 0, (name, label, int(different.size), different[:8].tolist())
        count, _, per_read = verdicts({"ARRIBA_MISMAPPER_JOB_ORDER": "candidate"})  # (another order of the jobs: compared per read)
        assert count == n_jobs and np.array_equal(per_read, first_per_read), name
        print("%s: %d jobs, %d mis-mappers, %d runs identical" % (name, n_jobs, int(first.sum()), len(schedules) + 2))
        pipeline.close()
        session.close()


def test_samples_in_a_queue_through_one_session_on_the_gpu(built, tmp_path):
    """arriba_workflow_submit on the device: the file of the next sample is fed through the sibling context while the stages of the current one run -- two different samples,
    the files of every one equal to those it gives alone (tests/test_host_and_device_logic.py: the same check on the stepping harness)"""
    import test_host_and_device_logic as host_tests
    host_tests.check_samples_in_a_queue("product", tmp_path)


def test_an_allocation_that_fails_inside_ingest_finish_leaves_the_ingest_alone(built, tmp_path):
    """advisor, round 3: an allocation that fails inside agpu_ingest_finish makes the contexts give back what they merely keep (DeviceBuffer::release_idle_buffers) and is tried
    again -- the stream and the tables of the very ingest that is finishing are not among that.  A resident pipeline works through a small sample (the scratch of its stages stays
    behind), then ingests a larger one of the same genome with the first allocation of agpu_ingest_finish failing once (agpu_debug_fail_allocation_in_finish; a second failure would find nothing left to give back and be reported, as it should): the
    batch is the one a fresh pipeline builds."""
    import subprocess
    import test_host_and_device_logic as cpu_tier
    from arriba_amd.pipeline import DevicePipeline, HostSession
    arguments = ["--seed", "11", "--contigs", "4", "--contig-len", "300000", "--junctions", "60"]
    for name, fragments, read_seed in (("small", "3000", "0"), ("large", "60000", "5")):
        subprocess.run([datasets.GEN_SYNTH, "--out", str(tmp_path / name), "--fragments", fragments, "--read-seed", read_seed] + arguments, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    small, large = str(tmp_path / "small"), str(tmp_path / "large")
    fresh_session = HostSession(large + ".fa", large + ".gtf")
    expected = cpu_tier._device_batch_columns(fresh_session, DevicePipeline(fresh_session, bam=large + ".bam", piece_bytes=1 << 20))
    session = HostSession(small + ".fa", small + ".gtf")
    pipeline = DevicePipeline(session, bam=small + ".bam", piece_bytes=1 << 20)
    os.makedirs(str(tmp_path / "out"))
    pipeline.run_workflow(str(tmp_path / "out" / "fusions.tsv"), None)
    pipeline.api.debug_fail_allocation_in_finish(1)
    pipeline.read_chimeric_alignments(large + ".bam", piece_bytes=1 << 20)
    pipeline.api.debug_fail_allocation_in_finish(0)
    columns = cpu_tier._device_batch_columns(session, pipeline)
    assert [key for key in expected if expected[key] != columns[key]] == []
    assert expected["n"] > 50000


def test_select_best_by_a_wavefront_per_group(built, dataset_files, tmp_path, monkeypatch):
    """select_most_supported_breakpoints: a gene pair with more than 48 unfiltered candidates is folded by a wavefront (the maximum of (rank, supporting reads) over the group, then the
    reference's order-dependent fold over the candidates that have it); ARRIBA_SELECT_BEST_SMALL=1 sends every group of two and more that way: the reference's verdicts on the golden
    dump and on a live run with the filters in front switched off (thousands of candidates per stage)"""
    monkeypatch.setenv("ARRIBA_SELECT_BEST_SMALL", "1")
    test_event_level_predicates_match_reference(built, dataset_files, tmp_path)
