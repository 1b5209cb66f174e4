/*
 * include/arriba_host.h -- C ABI of the host driver library (libarriba_host.so).
 *
 * The host side reads the reference data and the BAM once and produces the structure-of-arrays views
 * of include/arriba_gpu.h; it also holds the few inherently sequential scalar stages of the path
 * (strandedness vote, fragment-length estimate, per-contig viral verdicts) that the reference
 * computes between its per-read stages.  No per-read filtering or clustering happens here.
 * Each entry point cites the reference code it restates.
 */
#ifndef ARRIBA_HOST_C_H
#define ARRIBA_HOST_C_H 1

#include "arriba_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ahost_session ahost_session;

const char* ahost_last_error(void);

/* load_assembly + read_annotation_gtf + make_annotation_index (source/arriba.cpp:91-113).
 * NULL strings select the reference defaults (source/options.cpp:74-75, source/annotation.hpp:23). */
ahost_session* ahost_open(const char* fasta_path, const char* gtf_path, const char* interesting_contigs, const char* viral_contigs, const char* gtf_features);
void ahost_close(ahost_session* session);

/* read_chimeric_alignments for -x (source/read_chimeric_alignments.cpp:560-773); data = raw (inflated) BAM stream */
int ahost_ingest_bam_file(ahost_session* session, const char* bam_path, int external_duplicate_marking, unsigned int max_itd_length);
int ahost_ingest_bam_memory(ahost_session* session, const uint8_t* data, size_t size, int external_duplicate_marking, unsigned int max_itd_length);
/* The result of an ingest (batch, counters, coverage) as a file, and back into a session opened on the same assembly and annotation: repeated
 * benchmark and profiling runs over one batch skip the parse.  Tooling; not a step of the reference. */
int ahost_save_ingest(ahost_session* session, const char* path);
int ahost_load_ingest(ahost_session* session, const char* path);

/* ---- read_chimeric_alignments on the device (agpu_ingest_*, include/arriba_gpu.h): the host side --------------------------------------------------
 * The host opens the file (BGZF, gzip or raw BAM; a path, a pipe, /dev/stdin or "-": sam_open, source/read_chimeric_alignments.cpp:563), parses the BAM
 * header -- the contigs of the run must be known before the records are classified (:566-582) -- and hands the bytes on in pieces.
 *   ahost_bam_open     opens and parses the header; fills `config` for agpu_ingest_begin (its pointers stay valid until ahost_bam_close).  The genome
 *                      view of the session then holds the contigs of the header, too: upload it (again) before agpu_ingest_begin.
 *   ahost_bam_next     the next piece into `buffer` (>= 1 MiB; pinned memory from agpu_host_alloc): stored_bgzf = 1: raw BGZF bytes whose blocks are
 *                      all stored, with the table of the blocks -> agpu_ingest_push_bgzf; 0: bytes of the uncompressed stream (deflated blocks inflated
 *                      by all cores, CRC-checked) -> agpu_ingest_push.  Returns 1, 0 at the end of the file, -1 on error.
 *   ahost_adopt_device_ingest   what the device found: the reference's checks and warnings behind its loop (:759-771: "no normal reads found", malformed
 *                      records, no chimeric reads, missing HI tags), the counters, and coverage_t (flat arrays in the order of config.coverage_window_offset)
 *   ahost_set_batch_rows   the rows of the batch the host itself works on (the output writer: names, CIGARs and sequences of the supporting reads), fetched
 *                      with agpu_gather_rows_*; fragment indices in an ahost_fusion_table then refer to these rows */
typedef struct { int stored_bgzf; /* 0: the buffer holds bytes of the stream; 1: raw BGZF bytes whose blocks are stored, with their table; 2: raw BGZF bytes whose blocks are deflated, with their table (both: agpu_ingest_push_bgzf) */ size_t bytes; size_t stream_bytes; uint32_t n_blocks; } ahost_bam_piece;
/* The processors this process may use at once (affinity, CPU quota of the cgroup), and a limit for every decision about a number of threads that is made on the calling thread
 * from now on (0: none): a session that feeds the next sample beside the stages of the current one and writes the last file beside both gives each of the three its share --
 * more busy threads than the quota has CPUs are all stopped together for the rest of the scheduler's period. */
unsigned int ahost_cpu_budget(void);
void ahost_limit_threads_of_this_thread(unsigned int n);
int ahost_bam_open(ahost_session* session, const char* bam_path, int external_duplicate_marking, unsigned int max_itd_length, agpu_ingest_config* config);
/* One sample over several GPUs: as ahost_bam_open, but the pieces that follow hold only part `part` of `parts` of the records -- the file (BGZF or
 * uncompressed BAM on disk; the alignments of a read name next to each other, as STAR writes them) is cut between read names near the byte offsets
 * size * k / parts, at places every rank finds by itself from the bytes of the file.  config->part_of_sample is set: agpu_shard_export / agpu_shard_merge
 * (include/arriba_gpu.h) put the parts together, and ahost_adopt_device_ingest then takes the result of the merge. */
int ahost_bam_open_part(ahost_session* session, const char* bam_path, int external_duplicate_marking, unsigned int max_itd_length, unsigned int part, unsigned int parts, agpu_ingest_config* config);
int ahost_bam_next(ahost_session* session, void* buffer, size_t capacity, agpu_bgzf_block* blocks, uint32_t block_capacity, ahost_bam_piece* piece);
void ahost_bam_close(ahost_session* session);
int ahost_adopt_device_ingest(ahost_session* session, const agpu_ingest_result* result, const uint64_t* viral_read_counts, const uint16_t* coverage, const uint8_t* fragment_starts, const uint8_t* fragment_ends);
int ahost_set_batch_rows(ahost_session* session, const agpu_batch_rows* rows, const uint32_t* fragments /* [rows->n] ascending: the fragment every row holds; NULL: row k holds the fragment
                         of entry k of the read lists of the table the next ahost_write_fusions writes (the reads of a candidate next to each other; the table then carries read_filter_of_rows) */);

/* A blacklist (allow_keywords = 1: the second column may hold a keyword, source/filter_blacklisted_ranges.cpp:91-102) or a known-fusions file
 * (allow_keywords = 0) parsed into rules for agpu_filter_blacklisted_ranges / agpu_recover_known_fusions (parse_blacklist_item, parse_range:
 * source/filter_blacklisted_ranges.cpp:17-118; plain or gzip).  Malformed lines are skipped with the reference's warning.  The rules stay
 * valid until the next call with the same allow_keywords or ahost_close. */
int ahost_load_range_rules(ahost_session* session, const char* path, int allow_keywords, const agpu_range_rule** rules, uint32_t* n_rules);

/* The output files (write_fusions_to_file, source/output_fusions.cpp:1043-1261, called at source/arriba.cpp:604-610).  The candidate table is what the
 * device hands back at the end of the workflow: agpu_get_candidates + agpu_get_candidate_read_lists, agpu_get_evalues, agpu_assign_confidence,
 * agpu_candidate_iteration_order (the discarded candidates are written in the iteration order of the reference's fusions_t), agpu_get_filters
 * (filter of every fragment) and agpu_get_gene_table (GTF genes + the dummy genes of the sample).  write_discarded = 0: the candidates that passed
 * all filters, sorted by support; 1: the discarded ones.  print_extra_info: read identifiers (and, once built, transcript / peptide columns). */
typedef struct {
	uint32_t n_candidates;
	const uint32_t* gene1; const uint32_t* gene2; const uint32_t* contigs; const int32_t* breakpoint1; const int32_t* breakpoint2; const uint32_t* flags; const uint8_t* filter;
	const uint32_t* split_reads1; const uint32_t* split_reads2; const uint32_t* discordant_mates;
	const uint64_t* list_offset;   /* [3 * n_candidates + 1]; 64-bit: with -U 32767 the lists of the candidates of a sample pass 2^32 entries */
	const uint32_t* read_lists;
	const float* evalue; const uint8_t* confidence; const uint32_t* iteration_rank;
	const uint8_t* read_filter;    /* [fragments of the session's batch] */
	const int32_t* closest_genomic_breakpoint1; const int32_t* closest_genomic_breakpoint2; /* agpu_get_genomic_support; NULL = no structural variants given */
	uint32_t n_genes; const uint16_t* gene_contig; const int32_t* gene_start; const int32_t* gene_end;
	const uint8_t* read_filter_of_rows; /* instead of read_filter (then NULL): the filters of the fragments handed over with ahost_set_batch_rows, in the order of the rows (agpu_get_filters_of) */
} ahost_fusion_table;
int ahost_write_fusions(ahost_session* session, const ahost_fusion_table* table, const char* path, int write_discarded, int print_extra_info, unsigned int max_itd_length, int max_mate_gap,
                        int fill_sequence_gaps /* -I: complete the fusion transcript from the assembly along the chosen transcripts */);
/* The last output file of a sample written while the session reads the next one (arriba_workflow_defer_output): ahost_detach_sample moves what the writer reads of the SAMPLE --
 * coverage_t, the rows of ahost_set_batch_rows, the contig names as its header left them -- out of the session (which then holds no sample until its next ingest);
 * ahost_write_fusions_of writes from there on any thread, beside ahost_bam_open / ahost_bam_next / ahost_adopt_device_ingest of the next sample on the same session, with the
 * reference data of the session (annotation, assembly, tags, protein domains), which no sample changes; ahost_release_sample gives the vectors of the rows back to the session.
 * ahost_close waits for the samples that are still detached.  (reference: the same call as ahost_write_fusions, source/output_fusions.cpp:791-1007) */
typedef struct ahost_detached_sample ahost_detached_sample;
ahost_detached_sample* ahost_detach_sample(ahost_session* session);
int ahost_write_fusions_of(ahost_detached_sample* sample, const ahost_fusion_table* table, const char* path, int write_discarded, int print_extra_info, unsigned int max_itd_length, int max_mate_gap,
                           int fill_sequence_gaps);
void ahost_release_sample(ahost_detached_sample* sample);
/* One sample over several ranks that hold the same candidates: the rows part, part + parts, part + 2 parts, ... of the file ahost_write_fusions would write, as text
 * (the header line in front of the rows of part 0); *text stays valid until the next call.  Rows are independent of each other: the rank that gathers the texts of all
 * parts writes row k of the file from the text of part k % parts. */
int ahost_format_fusions(ahost_session* session, const ahost_fusion_table* table, int write_discarded, int print_extra_info, unsigned int max_itd_length, int max_mate_gap, int fill_sequence_gaps,
                         unsigned int part, unsigned int parts, const char** text, uint64_t* bytes);
/* the fragments whose rows ahost_write_fusions(table, write_discarded, print_extra_info = 1) reads: the supporting reads of the candidates it writes, ascending, unique.
 * Call with fragments == NULL to get the number in *count.  (Without print_extra_info the writer needs no rows.) */
int ahost_fusion_table_reads(const ahost_fusion_table* table, int write_discarded, uint32_t* fragments, uint64_t capacity, uint64_t* count);
/* estimate_fragment_length's float sum of the read lengths (source/read_stats.cpp:30-37, hazard H4: sequential, in name order) over lengths fetched from the
 * device (agpu_get_read_lengths); continues `running_sum` */
float ahost_read_length_sum_of(float running_sum, const uint32_t* mate1_lengths, const uint32_t* mate2_lengths, uint64_t count);
/* Optional inputs of the output files: a tags file (-t; load_tags, source/annotate_tags.cpp:11-44) and protein domains in GFF3 (-p; load_protein_domains,
 * source/annotate_protein_domains.cpp:33-121).  Loaded into the session; ahost_write_fusions fills the columns `tags` and `retained_protein_domains` from them. */
int ahost_load_tags(ahost_session* session, const char* path);
/* Structural variants from WGS (-d) for agpu_mark_genomic_support: Arriba's four-column format or VCF (source/filter_genomic_support.cpp:15-165).  The
 * variants stay valid until the next call or ahost_close. */
int ahost_load_genomic_breakpoints(ahost_session* session, const char* path, const agpu_genomic_breakpoint** variants, uint32_t* n_variants);
int ahost_load_protein_domains(ahost_session* session, const char* path);

const agpu_annotation_view* ahost_annotation_view(ahost_session* session);
const agpu_genome_view* ahost_genome_view(ahost_session* session);
const agpu_coverage_view* ahost_coverage_view(ahost_session* session);  /* coverage_t of the ingest for agpu_upload_coverage; NULL before an ingest */
const agpu_batch_view* ahost_batch_view(ahost_session* session);
/* Shards of the batch for one context per GPU: contiguous ranges of fragments in name order.  ahost_shard_boundary moves a cut forward
 * until it does not separate fragments of one read name (mark_multimappers compares neighbours, source/read_chimeric_alignments.cpp:792-802);
 * ahost_batch_slice_view returns the view of fragments [first, first + count) (valid until the next call). */
uint64_t ahost_shard_boundary(ahost_session* session, uint64_t target);
const agpu_batch_view* ahost_batch_slice_view(ahost_session* session, uint64_t first, uint64_t count);

uint64_t ahost_fragment_count(ahost_session* session);
uint64_t ahost_mapped_reads(ahost_session* session);
/* FNV-1a over coverage_t as the ingest built it (coverage windows, fragment start/end flags of every contig; reference: coverage_t,
 * source/read_stats.hpp:17-27) -- a fingerprint for tests and for comparing runs, e.g. with different numbers of ingest threads. */
uint64_t ahost_coverage_checksum(ahost_session* session);
uint32_t ahost_contig_count(ahost_session* session);
const char* ahost_contig_name(ahost_session* session, uint32_t contig);
/* "QNAME,HI" of fragment i (the reference's map key); valid until the session is closed */
const char* ahost_fragment_name(ahost_session* session, uint64_t i, uint32_t* length);

/* detect_strandedness (source/read_stats.cpp:94-143): 0 no, 1 yes, 2 reverse */
int ahost_detect_strandedness(ahost_session* session);

/* per-contig verdicts of filter_top_expressed_viral_contigs (source/filter_top_expressed_viral_contigs.cpp:51-127) and
 * filter_low_coverage_viral_contigs (source/filter_low_coverage_viral_contigs.cpp:11-27); pairs = (viral contig, gene id)
 * from agpu_get_viral_integration_sites, gene_bits = AGPU_GBIT_* for every gene id incl. dummy genes */
int ahost_viral_verdicts(ahost_session* session, const uint32_t* pairs, uint64_t n_pairs, const uint8_t* gene_bits, uint32_t n_genes,
                         unsigned int top_count, float min_covered_fraction, uint8_t* top_expressed_verdict, uint8_t* low_coverage_verdict);

/* estimate_fragment_length (source/read_stats.cpp:11-92) + source/arriba.cpp:352-364.  mate_gaps / fragments_visited come from
 * agpu_fragment_length_samples.  Returns 1 if estimated, 0 if the defaults were used. */
int ahost_estimate_fragment_length(ahost_session* session, const int32_t* mate_gaps, uint32_t n_samples, uint64_t fragments_visited, unsigned int default_fragment_length,
                                   float* mate_gap_mean, float* mate_gap_stddev, float* read_length_mean, int32_t* max_mate_gap);

/* The same for a sample whose fragments are spread over several sessions (shards in name order): the reference's float sum of the read
 * lengths is sequential over the whole sample (hazard H4), so each shard continues the running sum of the shards before it
 * (ahost_read_length_sum) and the estimate is made from the mate gaps, the final sum and the number of fragments visited. */
float ahost_read_length_sum(ahost_session* session, float running_sum, uint64_t first, uint64_t count);
int ahost_estimate_fragment_length_from_sums(const int32_t* mate_gaps, uint32_t n_samples, float read_length_sum, uint64_t fragments_visited, unsigned int default_fragment_length,
                                             float* mate_gap_mean, float* mate_gap_stddev, float* read_length_mean, int32_t* max_mate_gap);

/* Position of every candidate in the iteration order of the reference's fusions_t (std::unordered_map with the tuple hash of
 * source/common.hpp:286-314), given the candidates in insertion order as agpu_get_candidates returns them.  Stages whose result
 * depends on that order (partner dedup in estimate_expected_fusions, source/filter_relative_support.cpp:22-29) take it as input. */
int ahost_candidate_iteration_order(uint64_t n, const uint32_t* gene1, const uint32_t* gene2, const uint32_t* contigs, const int32_t* breakpoint1, const int32_t* breakpoint2, const uint32_t* flags, uint32_t* iteration_rank);

#ifdef __cplusplus
}
#endif

#endif
