"""ctypes bindings of the two C ABIs (include/arriba_gpu.h, include/arriba_host.h).

The device library is mandatory: there is no CPU fallback for the hot path.  Loading fails loudly if
``arriba_amd/lib/libarriba_gpu.so`` is missing (build it with ``__graft_entry__.build()``).
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int32, c_size_t, c_uint8, c_uint16, c_uint32, c_uint64, c_void_p

LIB_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib")
FILTER_COUNT = 38
KERNEL_NAME_LENGTH = 48
FILTER_NAMES = [
    "", "duplicates", "inconsistently_clipped", "homopolymer", "read_through", "same_gene", "small_insert_size", "long_gap", "hairpin",
    "multimappers", "mismatches", "mismappers", "relative_support", "intronic", "non_coding_neighbors", "intragenic_exonic",
    "internal_tandem_duplication", "min_support", "known_fusions", "spliced", "blacklist", "end_to_end", "in_vitro", "merge_adjacent",
    "select_best", "marginal_read_through", "short_anchor", "no_coverage", "many_spliced", "no_genomic_support", "uninteresting_contigs",
    "viral_contigs", "top_expressed_viral_contigs", "low_coverage_viral_contigs", "genomic_support", "isoforms", "low_entropy", "homologs",
]


class FlatIndex(ctypes.Structure):
    _fields_ = [("n_contigs", c_uint32), ("contig_offset", POINTER(c_uint32)), ("n_keys", c_uint32), ("keys", POINTER(c_int32)),
                ("member_offset", POINTER(c_uint32)), ("n_members", c_uint32), ("members", POINTER(c_uint32))]


class AnnotationView(ctypes.Structure):
    _fields_ = [("n_genes", c_uint32), ("gene_contig", POINTER(c_uint16)), ("gene_start", POINTER(c_int32)), ("gene_end", POINTER(c_int32)),
                ("gene_bits", POINTER(c_uint8)), ("gene_exonic_length", POINTER(c_int32)),
                ("n_exons", c_uint32), ("exon_start", POINTER(c_int32)), ("exon_end", POINTER(c_int32)), ("exon_gene", POINTER(c_uint32)),
                ("exon_previous", POINTER(c_int32)), ("exon_next", POINTER(c_int32)), ("exon_cds_start", POINTER(c_int32)), ("exon_cds_end", POINTER(c_int32)),
                ("exon_index", FlatIndex), ("gene_index", FlatIndex)]


class GenomeView(ctypes.Structure):
    _fields_ = [("n_contigs", c_uint32), ("contig_offset", POINTER(c_uint64)), ("contig_bits", POINTER(c_uint8)), ("bases", c_void_p)]


class CoverageView(ctypes.Structure):
    _fields_ = [("n_contigs", c_uint32), ("window_offset", POINTER(c_uint64)), ("coverage", POINTER(c_uint16)), ("fragment_starts", POINTER(c_uint8)), ("fragment_ends", POINTER(c_uint8))]


class RangeItem(ctypes.Structure):
    _fields_ = [("type", c_uint8), ("strand_defined", c_uint8), ("strand", c_uint8), ("reserved", c_uint8), ("contig", c_uint32), ("start", c_int32), ("end", c_int32), ("gene", c_uint32)]


class RangeRule(ctypes.Structure):
    _fields_ = [("first", RangeItem), ("second", RangeItem)]


class GenomicBreakpoint(ctypes.Structure):
    _fields_ = [("contig1", c_uint32), ("contig2", c_uint32), ("position1", c_int32), ("position2", c_int32), ("upstream1", c_uint8), ("upstream2", c_uint8), ("reserved", c_uint8 * 2)]


class FusionTable(ctypes.Structure):
    _fields_ = [("n_candidates", c_uint32), ("gene1", c_void_p), ("gene2", c_void_p), ("contigs", c_void_p), ("breakpoint1", c_void_p), ("breakpoint2", c_void_p), ("flags", c_void_p), ("filter", c_void_p),
                ("split_reads1", c_void_p), ("split_reads2", c_void_p), ("discordant_mates", c_void_p), ("list_offset", c_void_p), ("read_lists", c_void_p), ("evalue", c_void_p), ("confidence", c_void_p),
                ("iteration_rank", c_void_p), ("read_filter", c_void_p), ("closest_genomic_breakpoint1", c_void_p), ("closest_genomic_breakpoint2", c_void_p), ("n_genes", c_uint32), ("gene_contig", c_void_p), ("gene_start", c_void_p), ("gene_end", c_void_p), ("read_filter_of_rows", c_void_p)]


class SelectedCandidates(ctypes.Structure):
    _fields_ = [(name, c_void_p) for name in ("candidate", "gene1", "gene2", "contigs", "breakpoint1", "breakpoint2", "flags", "filter", "split_reads1", "split_reads2", "discordant_mates", "evalue", "confidence",
                                              "iteration_rank", "closest_genomic_breakpoint1", "closest_genomic_breakpoint2")]


class BatchView(ctypes.Structure):
    _fields_ = [("n", c_uint64), ("n_aln", POINTER(c_uint8)), ("fbits", POINTER(c_uint8)), ("group", POINTER(c_uint32)),
                ("contig", POINTER(c_uint16) * 3), ("start", POINTER(c_int32) * 3), ("end", POINTER(c_int32) * 3), ("abits", POINTER(c_uint8) * 3),
                ("cigar_offset", POINTER(c_uint32) * 3), ("cigar_count", POINTER(c_uint16) * 3),
                ("cigar_pool_size", c_uint64), ("cigar_pool", POINTER(c_uint32)),
                ("seq_offset", POINTER(c_uint32) * 2), ("seq_length", POINTER(c_uint32) * 2),
                ("seq_pool_size", c_uint64), ("seq_pool", POINTER(c_uint8))]


class IngestConfig(ctypes.Structure):
    _fields_ = [("n_targets", c_uint32), ("tid_to_contig", POINTER(c_uint32)), ("first_record_offset", c_uint64), ("stream_size_hint", c_uint64), ("n_contigs", c_uint32),
                ("coverage_window_offset", POINTER(c_uint64)), ("external_duplicate_marking", c_uint8), ("max_itd_length", c_uint32), ("part_of_sample", c_uint8), ("host_buffers", c_uint8)]


class BgzfBlock(ctypes.Structure):
    _fields_ = [("raw_offset", c_uint64), ("payload_offset", c_uint32), ("payload_size", c_uint32), ("stream_offset", c_uint64), ("crc32", c_uint32), ("isize", c_uint32), ("skip", c_uint32), ("keep", c_uint32)]


class IngestResult(ctypes.Structure):
    _fields_ = [("records", c_uint64), ("fragments", c_uint64), ("mapped_reads", c_uint64), ("malformed_count", c_uint64), ("missing_hi_tag", c_uint64), ("no_chimeric_reads", c_uint8),
                ("names_were_sorted", c_uint8), ("windows", ctypes.c_uint16), ("reserved", c_uint8 * 4), ("stream_bytes", c_uint64)]


class BamPiece(ctypes.Structure):
    _fields_ = [("stored_bgzf", c_int), ("bytes", c_size_t), ("stream_bytes", c_size_t), ("n_blocks", c_uint32)]


class BatchRows(ctypes.Structure):
    _fields_ = [("n", c_uint64), ("n_aln", c_void_p), ("fbits", c_void_p), ("group", c_void_p),
                ("contig", c_void_p * 3), ("start", c_void_p * 3), ("end", c_void_p * 3), ("abits", c_void_p * 3), ("cigar_offset", c_void_p * 3), ("cigar_count", c_void_p * 3),
                ("cigar_pool_size", c_uint64), ("cigar_pool", c_void_p), ("seq_offset", c_void_p * 2), ("seq_length", c_void_p * 2), ("seq_pool_size", c_uint64), ("seq_pool", c_void_p),
                ("name_offset", c_void_p), ("names_size", c_uint64), ("names", c_void_p)]


class Params(ctypes.Structure):
    _fields_ = [("homopolymer_length", c_uint32), ("min_read_through_distance", c_uint32), ("max_itd_length", c_uint32), ("subsampling_threshold", c_uint32),
                ("mismatch_pvalue_cutoff", c_float), ("max_kmer_content", c_float), ("evalue_cutoff", c_float), ("max_mismapper_fraction", c_float),
                ("fragment_length", c_uint32), ("external_duplicate_marking", c_uint8), ("strandedness", c_uint8), ("filter_enabled", c_uint8 * FILTER_COUNT),
                ("exonic_fraction", c_float), ("min_support", c_uint32)]


def _load(path):
    if not os.path.exists(path):
        raise ImportError(
            "%s is missing: the MI355X hot path has no CPU fallback. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950)." % path)
    return ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)


def bind_device_api(lib, prefix="agpu_"):
    """Declare the argument types of the device C ABI on `lib` and return a namespace of its functions."""
    class Api(object):
        pass
    api = Api()
    ctx = c_void_p
    signatures = {
        "last_error": (c_char_p, []),
        "api_version": (c_int, []),
        "device_count": (c_int, []),
        "default_params": (None, [POINTER(Params)]),
        "create": (ctx, [c_int, POINTER(Params)]),
        "destroy": (None, [ctx]),
        "set_params": (c_int, [ctx, POINTER(Params)]),
        "upload_annotation": (c_int, [ctx, POINTER(AnnotationView)]),
        "upload_genome": (c_int, [ctx, POINTER(GenomeView)]),
        "upload_batch": (c_int, [ctx, POINTER(BatchView)]),
        "reset": (c_int, [ctx]),
        "mark_multimappers": (c_int, [ctx, POINTER(c_uint64)]),
        "annotate": (c_int, [ctx, POINTER(c_uint32)]),
        "read_filters_stage1": (c_int, [ctx, c_void_p, c_void_p]),
        "get_viral_integration_sites": (c_int, [ctx, c_void_p, c_uint64, POINTER(c_uint64)]),
        "fragment_length_samples": (c_int, [ctx, c_void_p, POINTER(c_uint32), POINTER(c_uint64)]),
        "read_filters_stage2": (c_int, [ctx, c_void_p]),
        "find_fusions": (c_int, [ctx, c_int32, POINTER(c_uint64)]),
        "get_candidates": (c_int, [ctx] + [c_void_p] * 13),
        "get_candidate_read_lists": (c_int, [ctx, c_void_p, c_uint64, POINTER(c_uint64)]),
        "get_candidate_read_lists_of": (c_int, [ctx, c_void_p, c_uint64, c_void_p, c_void_p, c_uint64, POINTER(c_uint64)]),
        "get_discordant_swapped": (c_int, [ctx, c_void_p]),
        "shard_merge_rccl": (c_int, [ctx, c_void_p, c_uint32, POINTER(IngestResult)]),
        "filter_mismappers_rccl": (c_int, [ctx, c_void_p, c_int32, c_uint32, c_uint32, POINTER(c_uint64), POINTER(c_uint64)]),
        "rccl_unique_id": (c_int, [c_void_p]),
        "rccl_join": (c_int, [ctx, c_void_p, c_uint32, c_uint32, POINTER(c_void_p)]),
        "rccl_leave": (c_int, [c_void_p]),
        "rccl_all_gather_host": (c_int, [ctx, c_void_p, c_uint32, c_void_p, c_void_p, c_uint64]),
        "rccl_all_reduce_host": (c_int, [ctx, c_void_p, c_void_p, c_uint64, c_int]),
        "get_filters": (c_int, [ctx, c_void_p]),
        "get_filters_of": (c_int, [ctx, c_void_p, c_uint64, c_void_p]),
        "select_candidates": (c_int, [ctx, c_int, POINTER(c_uint64)]),
        "get_selected_candidates": (c_int, [ctx, POINTER(SelectedCandidates)]),
        "get_alignment_bits": (c_int, [ctx, c_int, c_void_p]),
        "get_fragment_bits": (c_int, [ctx, c_void_p]),
        "get_gene_sets": (c_int, [ctx, c_int, c_void_p, c_void_p, c_uint64, POINTER(c_uint64)]),
        "get_gene_table": (c_int, [ctx, c_uint32, c_uint32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
        "get_fusion_stats": (c_int, [ctx, c_void_p]),
        "set_shard": (c_int, [ctx, c_uint64, c_uint64]),
        "annotate_begin": (c_int, [ctx, POINTER(c_uint64)]),
        "copy_unmapped_positions": (c_int, [ctx, c_void_p]),
        "annotate_finish": (c_int, [ctx, c_void_p, c_uint64, POINTER(c_uint32)]),
        "duplicates_begin": (c_int, [ctx, POINTER(c_uint64)]),
        "copy_duplicate_entries": (c_int, [ctx, c_void_p]),
        "read_filters_stage1_global": (c_int, [ctx, c_void_p, c_uint64, c_void_p, c_void_p]),
        "fragment_length_samples_limited": (c_int, [ctx, c_uint32, c_void_p, POINTER(c_uint32), POINTER(c_uint64)]),
        "build_emissions": (c_int, [ctx, c_uint32, c_void_p]),
        "copy_emissions": (c_int, [ctx, c_void_p]),
        "find_fusions_from_emissions": (c_int, [ctx, c_void_p, c_uint64, c_int32, POINTER(c_uint64)]),
        "get_candidate_first_occurrence": (c_int, [ctx, c_void_p]),
        "import_candidates": (c_int, [ctx, c_uint64] + [c_void_p] * 12),
        "set_read_filters": (c_int, [ctx, c_void_p]),
        "make_kmer_index": (c_int, [ctx, c_int32, POINTER(c_uint64)]),
        "filter_mismappers": (c_int, [ctx, c_int32, POINTER(c_uint64), POINTER(c_uint64)]),
        "candidate_iteration_order": (c_int, [ctx, c_void_p]),
        "merge_adjacent_fusions": (c_int, [ctx, c_int32, POINTER(c_uint64)]),
        "filter_multimappers": (c_int, [ctx, POINTER(c_uint64), POINTER(c_uint64)]),
        "upload_coverage": (c_int, [ctx, POINTER(CoverageView)]),
        "recover_internal_tandem_duplication": (c_int, [ctx, c_uint32, c_float, POINTER(c_uint64)]),
        "filter_both_intronic": (c_int, [ctx, POINTER(c_uint64)]),
        "filter_short_anchor": (c_int, [ctx, c_uint32, POINTER(c_uint64)]),
        "filter_end_to_end": (c_int, [ctx, POINTER(c_uint64)]),
        "filter_no_coverage": (c_int, [ctx, POINTER(c_uint64)]),
        "filter_marginal_read_through": (c_int, [ctx, POINTER(c_uint64)]),
        "select_most_supported_breakpoints": (c_int, [ctx, POINTER(c_uint64)]),
        "recover_many_spliced": (c_int, [ctx, c_uint32, POINTER(c_uint64)]),
        "filter_in_vitro": (c_int, [ctx, c_float, POINTER(c_uint64)]),
        "filter_homologs": (c_int, [ctx, c_float, POINTER(c_uint64)]),
        "recover_isoforms": (c_int, [ctx, POINTER(c_uint64)]),
        "mark_genomic_support": (c_int, [ctx, POINTER(GenomicBreakpoint), c_uint32, c_int32, POINTER(c_uint64)]),
        "get_genomic_support": (c_int, [ctx, c_void_p, c_void_p]),
        "filter_no_genomic_support": (c_int, [ctx, POINTER(c_uint64)]),
        "recover_genomic_support": (c_int, [ctx, POINTER(c_uint64)]),
        "filter_blacklisted_ranges": (c_int, [ctx, POINTER(RangeRule), c_uint32, c_float, c_int32, POINTER(c_uint64)]),
        "recover_known_fusions": (c_int, [ctx, POINTER(RangeRule), c_uint32, c_int32, POINTER(c_uint64)]),
        "assign_confidence": (c_int, [ctx, c_void_p]),
        "recover_both_spliced": (c_int, [ctx, c_uint32, c_float, c_int32, c_uint32, POINTER(c_uint64)]),
        "set_owned_candidates": (c_int, [ctx, c_void_p, c_uint64]),
        "copy_multimapper_flags": (c_int, [ctx, c_void_p]),
        "multimappers_begin": (c_int, [ctx, c_void_p, POINTER(c_uint64)]),
        "multimappers_partial_best": (c_int, [ctx, c_void_p]),
        "multimappers_resolve": (c_int, [ctx, c_void_p, c_void_p, POINTER(c_uint64)]),
        "multimappers_recount": (c_int, [ctx, c_void_p, c_void_p]),
        "multimappers_finish": (c_int, [ctx, c_void_p, POINTER(c_uint64)]),
        "set_candidate_state": (c_int, [ctx, c_void_p, c_void_p, c_void_p, c_void_p]),
        "estimate_expected_fusions": (c_int, [ctx, c_uint64, c_void_p]),
        "get_evalues": (c_int, [ctx, c_void_p]),
        "filter_candidate_predicates": (c_int, [ctx, c_void_p]),
        "filter_relative_support": (c_int, [ctx, POINTER(c_uint64)]),
        "host_alloc": (c_void_p, [c_size_t]),
        "host_free": (None, [c_void_p]),
        "ingest_begin": (c_int, [ctx, POINTER(IngestConfig)]),
        "ingest_push": (c_int, [ctx, c_void_p, c_size_t]),
        "ingest_push_bgzf": (c_int, [ctx, c_void_p, c_size_t, POINTER(BgzfBlock), c_uint32, c_size_t]),
        "ingest_finish": (c_int, [ctx, POINTER(IngestResult)]),
        "shard_export_size": (c_int, [ctx, POINTER(c_uint64)]),
        "shard_export": (c_int, [ctx, c_void_p, c_uint64]),
        "shard_merge": (c_int, [ctx, c_void_p, c_uint64, c_uint32, POINTER(IngestResult)]),
        "mismapper_jobs": (c_int, [ctx, POINTER(c_uint64)]),
        "mismapper_verdicts": (c_int, [ctx, c_int32, c_uint32, c_uint32, c_void_p]),
        "filter_mismappers_apply": (c_int, [ctx, c_void_p, POINTER(c_uint64), POINTER(c_uint64)]),
        "get_viral_read_counts": (c_int, [ctx, c_void_p]),
        "get_coverage": (c_int, [ctx, c_void_p, c_void_p, c_void_p]),
        "detect_strandedness": (c_int, [ctx, POINTER(c_int)]),
        "get_read_lengths": (c_int, [ctx, c_uint64, c_uint64, c_void_p, c_void_p]),
        "gather_rows_begin": (c_int, [ctx, c_void_p, c_uint64, POINTER(c_uint64), POINTER(c_uint64), POINTER(c_uint64)]),
        "gather_rows_copy": (c_int, [ctx, POINTER(BatchRows)]),
        "set_profiling": (c_int, [ctx, c_int]),
        "debug_fail_allocation_in_finish": (None, [c_int]),
        "keep_batch_buffers": (c_int, [ctx, c_int]),
        "get_kernel_profile": (c_int, [ctx, c_void_p, c_void_p, c_void_p, c_uint32, POINTER(c_uint32)]),
        "last_kernel_ms": (c_int, [ctx, POINTER(c_float)]),
        "last_kernel_bytes": (c_int, [ctx, POINTER(c_uint64)]),
    }
    for name, (restype, argtypes) in signatures.items():
        function = getattr(lib, prefix + name, None)
        if function is None:
            continue
        function.restype = restype
        function.argtypes = argtypes
        setattr(api, name, function)
    api.exported = sorted(name for name in signatures if hasattr(api, name))
    api.declared = sorted(signatures)
    return api


def bind_host_api(lib):
    session = c_void_p
    signatures = {
        "ahost_last_error": (c_char_p, []),
        "ahost_open": (session, [c_char_p, c_char_p, c_char_p, c_char_p, c_char_p]),
        "ahost_close": (None, [session]),
        "ahost_ingest_bam_file": (c_int, [session, c_char_p, c_int, ctypes.c_uint]),
        "ahost_ingest_bam_memory": (c_int, [session, c_void_p, c_size_t, c_int, ctypes.c_uint]),
        "ahost_save_ingest": (c_int, [session, c_char_p]),
        "ahost_load_ingest": (c_int, [session, c_char_p]),
        "ahost_annotation_view": (POINTER(AnnotationView), [session]),
        "ahost_genome_view": (POINTER(GenomeView), [session]),
        "ahost_batch_view": (POINTER(BatchView), [session]),
        "ahost_fragment_count": (c_uint64, [session]),
        "ahost_mapped_reads": (c_uint64, [session]),
        "ahost_coverage_checksum": (c_uint64, [session]),
        "ahost_coverage_view": (POINTER(CoverageView), [session]),
        "ahost_write_fusions": (c_int, [session, POINTER(FusionTable), c_char_p, c_int, c_int, c_uint32, c_int, c_int]),
        "ahost_detach_sample": (c_void_p, [session]),
        "ahost_write_fusions_of": (c_int, [c_void_p, POINTER(FusionTable), c_char_p, c_int, c_int, c_uint32, c_int, c_int]),
        "ahost_release_sample": (None, [c_void_p]),
        "ahost_format_fusions": (c_int, [session, POINTER(FusionTable), c_int, c_int, c_uint32, c_int, c_int, c_uint32, c_uint32, POINTER(c_void_p), POINTER(c_uint64)]),
        "ahost_load_tags": (c_int, [session, c_char_p]),
        "ahost_load_genomic_breakpoints": (c_int, [session, c_char_p, POINTER(POINTER(GenomicBreakpoint)), POINTER(c_uint32)]),
        "ahost_load_protein_domains": (c_int, [session, c_char_p]),
        "ahost_load_range_rules": (c_int, [session, c_char_p, c_int, POINTER(POINTER(RangeRule)), POINTER(c_uint32)]),
        "ahost_contig_count": (c_uint32, [session]),
        "ahost_contig_name": (c_char_p, [session, c_uint32]),
        "ahost_fragment_name": (c_void_p, [session, c_uint64, POINTER(c_uint32)]),
        "ahost_detect_strandedness": (c_int, [session]),
        "ahost_viral_verdicts": (c_int, [session, c_void_p, c_uint64, c_void_p, c_uint32, ctypes.c_uint, c_float, c_void_p, c_void_p]),
        "ahost_read_length_sum": (c_float, [session, c_float, c_uint64, c_uint64]),
        "ahost_shard_boundary": (c_uint64, [session, c_uint64]),
        "ahost_batch_slice_view": (POINTER(BatchView), [session, c_uint64, c_uint64]),
        "ahost_estimate_fragment_length_from_sums": (c_int, [c_void_p, c_uint32, c_float, c_uint64, ctypes.c_uint, POINTER(c_float), POINTER(c_float), POINTER(c_float), POINTER(c_int32)]),
        "ahost_candidate_iteration_order": (c_int, [c_uint64] + [c_void_p] * 7),
        "ahost_bam_open": (c_int, [session, c_char_p, c_int, ctypes.c_uint, POINTER(IngestConfig)]),
        "ahost_bam_open_part": (c_int, [session, c_char_p, c_int, ctypes.c_uint, ctypes.c_uint, ctypes.c_uint, POINTER(IngestConfig)]),
        "ahost_bam_next": (c_int, [session, c_void_p, c_size_t, POINTER(BgzfBlock), c_uint32, POINTER(BamPiece)]),
        "ahost_bam_close": (None, [session]),
        "ahost_adopt_device_ingest": (c_int, [session, POINTER(IngestResult), c_void_p, c_void_p, c_void_p, c_void_p]),
        "ahost_set_batch_rows": (c_int, [session, POINTER(BatchRows), c_void_p]),
        "ahost_fusion_table_reads": (c_int, [POINTER(FusionTable), c_int, c_void_p, c_uint64, POINTER(c_uint64)]),
        "ahost_read_length_sum_of": (c_float, [c_float, c_void_p, c_void_p, c_uint64]),
        "ahost_estimate_fragment_length": (c_int, [session, c_void_p, c_uint32, c_uint64, ctypes.c_uint, POINTER(c_float), POINTER(c_float), POINTER(c_float), POINTER(c_int32)]),
    }
    for name, (restype, argtypes) in signatures.items():
        function = getattr(lib, name)
        function.restype = restype
        function.argtypes = argtypes
    return lib


_device_lib = None
_host_lib = None


class WorkflowOptions(ctypes.Structure):
    """arriba_workflow_options (include/arriba_workflow.h): options_t of the reference (source/options.hpp)"""
    _fields_ = [(name, c_char_p) for name in ("assembly_file", "gene_annotation_file", "chimeric_bam_file", "output_file", "discarded_output_file", "blacklist_file", "known_fusions_file", "tags_file",
                                              "protein_domains_file", "genomic_breakpoints_file", "interesting_contigs", "viral_contigs", "gtf_features")] + [
        ("device", Params), ("min_itd_support", c_uint32), ("min_itd_allele_fraction", c_float), ("high_expression_quantile", c_float), ("min_spliced_events", c_uint32), ("min_anchor_length", c_uint32),
        ("max_homolog_identity", c_float), ("top_viral_contigs", c_uint32), ("viral_contig_min_covered_fraction", c_float), ("max_genomic_breakpoint_distance", c_int32),
        ("print_extra_info_for_discarded_fusions", c_uint8), ("fill_sequence_gaps", c_uint8), ("device_index", c_int), ("log_to_stdout", c_uint8), ("host_ingest", c_uint8)]


class WorkflowStage(ctypes.Structure):
    _fields_ = [("stage", ctypes.c_char * 48), ("count", c_uint64)]


class WorkflowReport(ctypes.Structure):
    _fields_ = [("n_stages", c_uint32), ("stages", WorkflowStage * 64)]


class WorkflowTiming(ctypes.Structure):
    _fields_ = [(name, ctypes.c_double) for name in ("total", "feed", "ingest", "adopt", "stages", "filter_mismappers", "output", "output_results", "output_rows", "output_format", "feed_read", "feed_push", "feed_total", "exchange_parts", "exchange_verdicts", "exchange_rows", "shard_fragments", "exchanged_bytes")]


WORKFLOW_MAX, WORKFLOW_MIN, WORKFLOW_SUM = 0, 1, 2
RCCL_ID_BYTES = 128
ALL_GATHER = ctypes.CFUNCTYPE(c_int, c_void_p, c_void_p, c_void_p, c_uint64)
ALL_REDUCE_INT64 = ctypes.CFUNCTYPE(c_int, c_void_p, c_void_p, c_uint64, c_int)
ALL_REDUCE_MAX_BYTES = ctypes.CFUNCTYPE(c_int, c_void_p, c_void_p, c_uint64)


class WorkflowCommunicator(ctypes.Structure):
    """arriba_workflow_communicator (include/arriba_workflow.h): the ranks of one sample and the three collectives over host memory the driver asks its caller for"""
    _fields_ = [("rank", c_uint32), ("size", c_uint32), ("state", c_void_p), ("all_gather", ALL_GATHER), ("all_reduce_int64", ALL_REDUCE_INT64), ("all_reduce_max_bytes", ALL_REDUCE_MAX_BYTES),
                ("rccl_communicator", c_void_p)]


_workflow_lib = None


def workflow_library():
    """libarriba_workflow.so: the reference's main() over the two C ABIs (include/arriba_workflow.h); tests replace it with the build against the host stepping harness"""
    global _workflow_lib
    if _workflow_lib is None:
        device_library(); host_library()
        lib = _load(os.environ.get("ARRIBA_WORKFLOW_LIBRARY", os.path.join(LIB_DIR, "libarriba_workflow.so")))
        lib.arriba_workflow_default_options.argtypes = [POINTER(WorkflowOptions)]; lib.arriba_workflow_default_options.restype = None
        lib.arriba_workflow_last_error.restype = c_char_p
        lib.arriba_workflow_run.argtypes = [POINTER(WorkflowOptions), POINTER(WorkflowReport)]; lib.arriba_workflow_run.restype = c_int
        lib.arriba_workflow_open.argtypes = [POINTER(WorkflowOptions)]; lib.arriba_workflow_open.restype = c_void_p
        lib.arriba_workflow_sample.argtypes = [c_void_p, c_char_p, c_char_p, c_char_p, POINTER(WorkflowReport), POINTER(WorkflowTiming)]; lib.arriba_workflow_sample.restype = c_int
        lib.arriba_workflow_submit.argtypes = [c_void_p, c_char_p]; lib.arriba_workflow_submit.restype = c_int
        lib.arriba_workflow_cancel.argtypes = [c_void_p]; lib.arriba_workflow_cancel.restype = c_int
        lib.arriba_workflow_defer_output.argtypes = [c_void_p, c_int]; lib.arriba_workflow_defer_output.restype = c_int
        lib.arriba_workflow_finish_ahead.argtypes = [c_void_p, c_int]; lib.arriba_workflow_finish_ahead.restype = c_int
        lib.arriba_workflow_flush.argtypes = [c_void_p, POINTER(ctypes.c_double)]; lib.arriba_workflow_flush.restype = c_int
        lib.arriba_workflow_set_communicator.argtypes = [c_void_p, POINTER(WorkflowCommunicator)]; lib.arriba_workflow_set_communicator.restype = c_int
        lib.arriba_workflow_rccl_unique_id.argtypes = [c_void_p]; lib.arriba_workflow_rccl_unique_id.restype = c_int
        lib.arriba_workflow_join_rccl.argtypes = [c_void_p, c_void_p, c_uint32, c_uint32]; lib.arriba_workflow_join_rccl.restype = c_int
        lib.arriba_workflow_device.argtypes = [c_void_p]; lib.arriba_workflow_device.restype = c_void_p
        lib.arriba_workflow_lane_device.argtypes = [c_void_p, c_int]; lib.arriba_workflow_lane_device.restype = c_void_p
        lib.arriba_workflow_host.argtypes = [c_void_p]; lib.arriba_workflow_host.restype = c_void_p
        lib.arriba_workflow_close.argtypes = [c_void_p]; lib.arriba_workflow_close.restype = None
        _workflow_lib = lib
    return _workflow_lib


def device_library():
    global _device_lib
    if _device_lib is None:
        _device_lib = _load(os.path.join(LIB_DIR, "libarriba_gpu.so"))
    return _device_lib


def host_library():
    global _host_lib
    if _host_lib is None:
        _host_lib = bind_host_api(_load(os.environ.get("ARRIBA_HOST_LIBRARY", os.path.join(LIB_DIR, "libarriba_host.so"))))  # the override: a sanitizer build (tools/)
    return _host_lib
