/*
 * include/arriba_gpu.h -- C ABI of the MI355X-native hot path of the fusion caller.
 *
 * The reference (suhrig/arriba v2.5.1) has no plugin/FFI interface; its de-facto operator API is the
 * set of free stage functions that main() calls in a fixed order, each mutating a container in place
 * and returning the number of still-unfiltered entries (source/arriba.cpp:119-565).  This header is
 * the boundary a maintainer would bind instead of those calls: plain pointers and sizes, no C++ or
 * torch types, int status codes (0 = ok, negative = error, text via agpu_last_error()).
 *
 * Data handed over is structure-of-arrays, produced once by the host driver
 * (arriba_amd/csrc/host/): the flattened gene/exon interval index, the genome, and the packed
 * table of chimeric fragments in read-name order (index == name rank).  All results stay resident
 * in HBM between calls and are fetched with the agpu_get_* functions.
 *
 * Each entry point cites the reference interface it replaces.
 */
#ifndef ARRIBA_GPU_H
#define ARRIBA_GPU_H 1

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AGPU_API_VERSION 1

/* status codes */
#define AGPU_OK 0
#define AGPU_ERR_INVALID (-1)       /* bad argument / call order */
#define AGPU_ERR_DEVICE (-2)        /* HIP runtime error */
#define AGPU_ERR_CAPACITY (-3)      /* an internal device pool overflowed (retry after agpu_set_capacity) */
#define AGPU_ERR_NO_DEVICE (-4)     /* no gfx950 device visible */
#define AGPU_ERR_NO_MEMORY (-5)     /* a device allocation failed (also after the contexts gave back what they merely keep for their next sample): the text names the buffer, the
                                       size asked for and what the device has free.  A session of two lanes retries the sample alone on this code (include/arriba_workflow.h) */

/* filter ids == position in the reference's registry (source/common.hpp:29-67) */
#define AGPU_FILTER_COUNT 38

/* bits of agpu_batch_view.abits / alignment bits returned by agpu_get_alignment_bits (source/common.hpp:191-204) */
#define AGPU_ABIT_STRAND 1u                        /* 1 = forward */
#define AGPU_ABIT_FIRST_IN_PAIR 2u
#define AGPU_ABIT_SUPPLEMENTARY 4u
#define AGPU_ABIT_EXONIC 8u
#define AGPU_ABIT_PREDICTED_STRAND 16u             /* 1 = forward; meaningless while AMBIGUOUS is set */
#define AGPU_ABIT_PREDICTED_STRAND_AMBIGUOUS 32u
/* bits of agpu_batch_view.fbits (source/common.hpp:212-219) */
#define AGPU_FBIT_SINGLE_END 1u
#define AGPU_FBIT_MULTIMAPPER 2u
#define AGPU_FBIT_DUPLICATE 4u
/* bits of agpu_annotation_view.gene_bits */
#define AGPU_GBIT_STRAND 1u
#define AGPU_GBIT_DUMMY 2u
#define AGPU_GBIT_PROTEIN_CODING 4u
/* bits of agpu_genome_view.contig_bits */
#define AGPU_CBIT_INTERESTING 1u
#define AGPU_CBIT_VIRAL 2u

/* bits of the candidate flags returned by agpu_get_candidates (source/common.hpp:237-250) */
#define AGPU_CFLAG_UPSTREAM1 1u                    /* direction1 == UPSTREAM */
#define AGPU_CFLAG_UPSTREAM2 2u
#define AGPU_CFLAG_EXONIC1 4u
#define AGPU_CFLAG_EXONIC2 8u
#define AGPU_CFLAG_SPLICED1 16u
#define AGPU_CFLAG_SPLICED2 32u
#define AGPU_CFLAG_PREDICTED_STRAND1 64u           /* 1 = forward */
#define AGPU_CFLAG_PREDICTED_STRAND2 128u
#define AGPU_CFLAG_PREDICTED_STRANDS_AMBIGUOUS 256u
#define AGPU_CFLAG_TRANSCRIPT_START_GENE1 512u
#define AGPU_CFLAG_TRANSCRIPT_START_AMBIGUOUS 1024u

typedef struct agpu_ctx agpu_ctx;

/* Flattened interval index (source/annotation.t.hpp:25-45): per contig a sorted array of boundary keys;
 * bucket k lists the ids of the features containing position keys[k], ascending. */
typedef struct {
	uint32_t n_contigs;
	const uint32_t* contig_offset;   /* [n_contigs+1] into keys */
	uint32_t n_keys;
	const int32_t* keys;
	const uint32_t* member_offset;   /* [n_keys+1] into members */
	uint32_t n_members;
	const uint32_t* members;
} agpu_flat_index;

/* Gene and exon tables (source/common.hpp:148-183); ids are the reference's gene->id and the exon's
 * rank in allocation order (the canonical order inside exon sets). */
typedef struct {
	uint32_t n_genes;
	const uint16_t* gene_contig;
	const int32_t* gene_start;
	const int32_t* gene_end;
	const uint8_t* gene_bits;
	const int32_t* gene_exonic_length;
	uint32_t n_exons;
	const int32_t* exon_start;
	const int32_t* exon_end;
	const uint32_t* exon_gene;
	const int32_t* exon_previous;    /* exon id or -1 */
	const int32_t* exon_next;
	const int32_t* exon_cds_start;   /* -1 = non-coding */
	const int32_t* exon_cds_end;
	agpu_flat_index exon_index;
	agpu_flat_index gene_index;      /* GTF genes only; dummy genes are created on the device */
} agpu_annotation_view;

/* coverage_t (source/read_stats.hpp:17-27) flattened: the 20 bp windows of contig c are [window_offset[c], window_offset[c+1]); contigs without a
 * loaded sequence have no windows.  Built by the ingest (ahost_coverage_view). */
typedef struct {
	uint32_t n_contigs;
	const uint64_t* window_offset;   /* [n_contigs+1] */
	const uint16_t* coverage;        /* saturating fragment count per window */
	const uint8_t* fragment_starts;  /* 1 = a non-chimeric fragment starts in the window */
	const uint8_t* fragment_ends;
} agpu_coverage_view;

/* Genome as upper-case ASCII, contigs concatenated (source/assembly.cpp:28-58). */
typedef struct {
	uint32_t n_contigs;
	const uint64_t* contig_offset;   /* [n_contigs+1] into bases; empty range = sequence not loaded */
	const uint8_t* contig_bits;      /* [n_contigs] AGPU_CBIT_* */
	const char* bases;
} agpu_genome_view;

/* Packed chimeric fragments in name order (source/common.hpp:191-220).  Slot-major columns: slot 0 =
 * MATE1, slot 1 = MATE2 / SPLIT_READ, slot 2 = SUPPLEMENTARY.  Sequences are BAM 4-bit codes (two
 * bases per byte, high nibble first), each starting on a 4-byte boundary; slots 0 and 1 only. */
typedef struct {
	uint64_t n;
	const uint8_t* n_aln;            /* 2 or 3 */
	const uint8_t* fbits;
	const uint32_t* group;           /* fragments of one read name (all HI values) share a group id; ascending */
	const uint16_t* contig[3];
	const int32_t* start[3];
	const int32_t* end[3];
	const uint8_t* abits[3];
	const uint32_t* cigar_offset[3];
	const uint16_t* cigar_count[3];
	uint64_t cigar_pool_size;
	const uint32_t* cigar_pool;
	const uint32_t* seq_offset[2];   /* in units of 4 bytes */
	const uint32_t* seq_length[2];   /* in bases */
	uint64_t seq_pool_size;          /* bytes */
	const uint8_t* seq_pool;
} agpu_batch_view;

/* Parameters (defaults of source/options.cpp:71-107) */
typedef struct {
	uint32_t homopolymer_length;         /* -H 6 */
	uint32_t min_read_through_distance;  /* -R 10000 */
	uint32_t max_itd_length;             /* -l 100 */
	uint32_t subsampling_threshold;      /* -U 300 */
	float mismatch_pvalue_cutoff;        /* -V 0.01 */
	float max_kmer_content;              /* -K 0.6 */
	float evalue_cutoff;                 /* -E 0.3 */
	float max_mismapper_fraction;        /* -m 0.8 */
	uint32_t fragment_length;            /* -F 200 */
	uint8_t external_duplicate_marking;  /* -u */
	uint8_t strandedness;                /* 0 no, 1 yes, 2 reverse (already resolved; 3 = auto is a host decision) */
	uint8_t filter_enabled[AGPU_FILTER_COUNT]; /* -f */
	float exonic_fraction;               /* -e 0.33 */
	uint32_t min_support;                /* -S 2 */
} agpu_params;

void agpu_default_params(agpu_params* params);

const char* agpu_last_error(void);
int agpu_api_version(void);
int agpu_device_count(void);

agpu_ctx* agpu_create(int device, const agpu_params* params);
/* A second context on the device of `of` that shares its scratch buffers (the stream and the tables of the ingest, the buffers of the stages; everything that is loaded or
 * built per context -- annotation, genome, batch, candidates -- is its own).  For a resident session that feeds the file of the next sample through one context while the
 * stages of the current sample run in the other (include/arriba_workflow.h: arriba_workflow_submit): the caller must not let both be between agpu_ingest_begin and
 * agpu_ingest_finish at once, nor both in their stages.  Destroy both with agpu_destroy, in any order. */
agpu_ctx* agpu_create_sibling(agpu_ctx* of);
/* test hook: the next `count` device allocations made inside agpu_ingest_finish on the calling thread are treated as failed once each -- the contexts then give back what they
 * merely keep for their next sample and the allocation is tried again (tests/test_gpu_parity.py: the stream and the tables of the ingest that is finishing must survive that) */
void agpu_debug_fail_allocation_in_finish(int count);
/* test hook: the next `finishes` calls of agpu_ingest_finish on the calling thread by a context that has a sibling return AGPU_ERR_NO_MEMORY ("hipMalloc failed ..."), as they do when
 * two samples in flight do not fit the device together; finishes < 0: every call, sibling or not (a device too small for the sample alone); 0: off.  What a session of two lanes does
 * with that code -- drain, close the second lane, run the sample again alone -- is exercised on the GPU with it (tests/workflow_session_worker.py) */
void agpu_debug_exhaust_memory_in_finish(int finishes);
void agpu_destroy(agpu_ctx* ctx);
int agpu_set_params(agpu_ctx* ctx, const agpu_params* params);

/* uploads (host pointers; copied to HBM) -- replace the in-memory containers every reference stage receives */
int agpu_upload_annotation(agpu_ctx* ctx, const agpu_annotation_view* annotation); /* gene/exon annotation + index: source/arriba.cpp:100-113 */
int agpu_upload_genome(agpu_ctx* ctx, const agpu_genome_view* genome);               /* assembly_t: source/arriba.cpp:97-98 */
int agpu_upload_batch(agpu_ctx* ctx, const agpu_batch_view* batch);                  /* chimeric_alignments_t: source/arriba.cpp:119-130 */

/* ---- read_chimeric_alignments on the device (source/read_chimeric_alignments.cpp:560-773) --------------------------------
 * Instead of agpu_upload_batch: the uncompressed BAM stream goes to HBM as it comes out of the container, the records are cut, classified, collated
 * by read name, sanity-checked (remove_malformed_alignments, :377-506), sorted by name (the order of the reference's std::map) and packed into the
 * batch on the device; coverage_t (source/read_stats.cpp:161-266) is built there, too.  The host only feeds bytes:
 *   agpu_ingest_begin        the BAM header as the host parsed it (reference id -> contig id), the windows of coverage_t
 *   agpu_ingest_push         the next piece of the uncompressed stream (host memory; pinned memory from agpu_host_alloc makes the copy asynchronous)
 *   agpu_ingest_push_bgzf    the next piece of a BGZF file whose blocks are stored (STAR --outBAMcompression 0, run_arriba.sh:34): the raw bytes and
 *                            the table of the blocks inside them; the payloads are moved into the stream on the device
 *   agpu_ingest_finish       everything else; the batch is then resident as after agpu_upload_batch
 * A push returns when the piece pushed BEFORE it has left its buffer: the caller alternates between two buffers (host_buffers = n: takes n buffers in turn, and a push
 * returns when the piece pushed n - 1 pushes ago has left its buffer -- with three the reader of the file never waits for a copy; at most 4).
 * Needs agpu_upload_annotation and agpu_upload_genome (gene index for the read-through extraction, assembly for the tandem-duplication probe). */
typedef struct {
	uint32_t n_targets;              /* reference sequences of the BAM header */
	const uint32_t* tid_to_contig;   /* [n_targets] contig id of every reference id (contigs_t of the run) */
	uint64_t first_record_offset;    /* size of the BAM header in the uncompressed stream */
	uint64_t stream_size_hint;       /* expected size of the uncompressed stream, 0 = unknown (the buffer grows) */
	uint32_t n_contigs;
	const uint64_t* coverage_window_offset; /* [n_contigs + 1] windows of coverage_t per contig (assembly size / 20 + 2; none without sequence) */
	uint8_t external_duplicate_marking;     /* -u */
	uint32_t max_itd_length;                /* -l */
	uint8_t part_of_sample;                 /* the stream holds a part of the sample's records, other contexts hold the rest: agpu_shard_export / agpu_shard_merge follow */
	uint8_t host_buffers;                   /* buffers the caller pushes from in turn: 0 or 2 = two (see above) */
} agpu_ingest_config;
/* A BGZF block of a pushed piece (offsets inside the piece / inside the piece's part of the stream).  isize == 0: a stored block -- payload_offset / payload_size are its data
 * as they are (already cut to the records of a part of the file where the block is the first or last of a part), crc32 the CRC-32 of the whole data, 0 = not to be checked.
 * isize != 0: a deflated block -- payload_offset / payload_size are its DEFLATE stream, isize what it inflates to (the ISIZE of the gzip trailer), and of that the bytes
 * [skip, skip + keep) go to the stream (keep == isize but for the first / last block of a part of the file); crc32 as above.  All blocks of a piece are of one kind. */
typedef struct { uint64_t raw_offset; uint32_t payload_offset, payload_size; uint64_t stream_offset; uint32_t crc32; uint32_t isize; uint32_t skip; uint32_t keep; } agpu_bgzf_block;
typedef struct {
	uint64_t records;                /* alignment records in the stream */
	uint64_t fragments;              /* chimeric fragments in the batch (chimeric_alignments.size()) */
	uint64_t mapped_reads;           /* source/read_chimeric_alignments.cpp:653-654 */
	uint64_t malformed_count;        /* "WARNING: N SAM records were malformed and ignored" (:761-763) */
	uint64_t missing_hi_tag;         /* (:622-625) */
	uint8_t no_chimeric_reads;       /* (:767-770) */
	uint8_t names_were_sorted;       /* 1 = the order of first occurrence was the name order (no string sort needed) */
	uint16_t windows;                /* windows of the stream that went through the front of the ingest while the pieces arrived; 0 = everything was done behind the last piece */
	uint8_t reserved[4];
	uint64_t stream_bytes;
} agpu_ingest_result;
void* agpu_host_alloc(size_t bytes);  /* pinned host memory for the pieces (NULL on failure) */
void agpu_host_free(void* pointer);
/* A session of two lanes (agpu_create_sibling) hands the buffers of a sample from one lane to the other inside agpu_ingest_finish, when the sibling's sample is done on the device.  With
 * agpu_keep_batch_buffers(ctx, 1) (both lanes) every lane keeps the columns and pools of its own batch (~25 GB more at 10^8 fragments) and only the buffers of the stages change hands, when
 * the stages begin (agpu_mark_multimappers): agpu_ingest_finish of the next sample may then run -- on another thread -- while the stages of the current one still do
 * (arriba_workflow_finish_ahead).  source/arriba.cpp has no counterpart: a resident service is not what the reference is. */
int agpu_keep_batch_buffers(agpu_ctx* ctx, int on);
int agpu_ingest_begin(agpu_ctx* ctx, const agpu_ingest_config* config);
int agpu_ingest_push(agpu_ctx* ctx, const void* bytes, size_t size);
int agpu_ingest_push_bgzf(agpu_ctx* ctx, const void* raw, size_t raw_size, const agpu_bgzf_block* blocks, uint32_t n_blocks, size_t stream_bytes);
int agpu_ingest_finish(agpu_ctx* ctx, agpu_ingest_result* result);
/* what the host's sequential stages and its output writer need from a batch that lives on the device:
 *   agpu_get_viral_read_counts   mapped_viral_reads_by_contig (source/read_chimeric_alignments.cpp:735-739)
 *   agpu_get_coverage            coverage_t as the reference holds it (16-bit saturating windows, start/end flags); sizes by coverage_window_offset
 *   agpu_detect_strandedness     detect_strandedness (source/read_stats.cpp:94-143): 0 no, 1 yes, 2 reverse
 *   agpu_get_read_lengths        sequence lengths of MATE1 / MATE2 of fragments [first, first + count) (the float sum of estimate_fragment_length is the host's)
 *   agpu_gather_rows_begin/copy  the rows of the given fragments as a small batch of the same layout plus their names ("QNAME,HI"), alignment and
 *                                fragment bits as ingested; begin returns the pool sizes, copy fills the caller's arrays */
int agpu_get_viral_read_counts(agpu_ctx* ctx, uint64_t* counts /* [n_contigs] */);
int agpu_get_coverage(agpu_ctx* ctx, uint16_t* coverage, uint8_t* fragment_starts, uint8_t* fragment_ends);
int agpu_detect_strandedness(agpu_ctx* ctx, int* strandedness);
int agpu_get_read_lengths(agpu_ctx* ctx, uint64_t first, uint64_t count, uint32_t* mate1, uint32_t* mate2);
typedef struct {
	uint64_t n;
	uint8_t* n_aln; uint8_t* fbits; uint32_t* group;
	uint16_t* contig[3]; int32_t* start[3]; int32_t* end[3]; uint8_t* abits[3]; uint32_t* cigar_offset[3]; uint16_t* cigar_count[3];
	uint64_t cigar_pool_size; uint32_t* cigar_pool;
	uint32_t* seq_offset[2]; uint32_t* seq_length[2];
	uint64_t seq_pool_size; uint8_t* seq_pool;
	uint32_t* name_offset;           /* [n + 1] */
	uint64_t names_size; char* names;
} agpu_batch_rows;
int agpu_gather_rows_begin(agpu_ctx* ctx, const uint32_t* fragments /* NULL = all, in order */, uint64_t n, uint64_t* cigar_pool_size, uint64_t* seq_pool_size, uint64_t* names_size);
int agpu_gather_rows_copy(agpu_ctx* ctx, agpu_batch_rows* rows);

/* ---- one sample over the GPUs of a node (SURVEY.md section 8 row e; BASELINE.json config 4) ----------------------------------------------------------
 * Every rank runs the ingest above over its part of the records (agpu_ingest_config.part_of_sample = 1; no read name in two parts).  What the reference's
 * loop (source/read_chimeric_alignments.cpp:560-773) leaves behind is additive over such parts: the fragments concatenate (each part is in name order and
 * the parts follow each other in name order -- checked), mapped_reads / malformed / missing-HI counters and mapped_viral_reads_by_contig add up, coverage_t
 * adds up before its 16-bit saturation (source/read_stats.cpp:161-266).  The exchange is ONE collective over device memory:
 *   agpu_shard_export_size / agpu_shard_export   the part of this context as one block of bytes (header + columns; csrc/device/shard_host.hpp), written to
 *                                                 device (or host) memory of the caller -- e.g. straight into the send buffer of an all-gather
 *   agpu_shard_merge                              the blocks of all parts, `stride` bytes apart in rank order (the receive buffer of the all-gather): the
 *                                                 context then holds the batch of the whole sample exactly as a single ingest of all records leaves it
 * after which every rank runs the stages on the whole batch (they take ~1 % of the time of a sample) and shares out filter_mismappers (below). */
int agpu_shard_export_size(agpu_ctx* ctx, uint64_t* bytes);
int agpu_shard_export(agpu_ctx* ctx, void* block, uint64_t capacity);
int agpu_shard_merge(agpu_ctx* ctx, const void* blocks, uint64_t stride, uint32_t n_parts, agpu_ingest_result* result /* sums over the parts; may be NULL */);

/* The two exchanges issued from inside the C ABI for hosts that hold an RCCL communicator (ncclComm_t passed as void*): export +
 * ncclAllReduce(max size) + ncclAllGather on the context's stream + merge, and jobs + verdicts + ncclAllReduce(max) + apply.  librccl is looked up at run
 * time (dlopen).  Compositions of the entry points above; not exercised in round 2 (RCCL needs one GPU per rank). */
int agpu_shard_merge_rccl(agpu_ctx* ctx, void* nccl_comm, uint32_t n_ranks, agpu_ingest_result* result);
int agpu_filter_mismappers_rccl(agpu_ctx* ctx, void* nccl_comm, int32_t max_mate_gap, uint32_t rank, uint32_t n_ranks, uint64_t* remaining, uint64_t* discarded_reads);
/* ... and a communicator of RCCL alone, for a host that has none (the C++ driver: arriba_workflow_join_rccl, include/arriba_workflow.h).  agpu_rccl_unique_id on one rank
 * (ncclGetUniqueId: AGPU_RCCL_ID_BYTES bytes that reach the other ranks by whatever started them), agpu_rccl_join on every rank (ncclCommInitRank on the device of the
 * context; *nccl_comm is what the two calls above take), agpu_rccl_leave at the end.  The small exchanges of a driver -- sizes, status words, the texts of the rows of the
 * output files -- are host memory: agpu_rccl_all_gather_host (`bytes` of every rank, in rank order, into all[n_ranks * bytes]) and agpu_rccl_all_reduce_host (in place;
 * kind: AGPU_REDUCE_*) bounce them through a buffer of the context on its stream. */
#define AGPU_RCCL_ID_BYTES 128
enum { AGPU_REDUCE_MAX_INT64 = 0, AGPU_REDUCE_MIN_INT64 = 1, AGPU_REDUCE_SUM_INT64 = 2, AGPU_REDUCE_MAX_BYTES = 3, AGPU_REDUCE_SUM_UINT32 = 4 /* device collectives only */ };
int agpu_rccl_unique_id(uint8_t* id /* [AGPU_RCCL_ID_BYTES] */);
int agpu_rccl_join(agpu_ctx* ctx, const uint8_t* id, uint32_t rank, uint32_t n_ranks, void** nccl_comm);
int agpu_rccl_leave(void* nccl_comm);
int agpu_rccl_all_gather_host(agpu_ctx* ctx, void* nccl_comm, uint32_t n_ranks, const void* mine, void* all, uint64_t bytes);
int agpu_rccl_all_reduce_host(agpu_ctx* ctx, void* nccl_comm, void* values, uint64_t count, int kind);
/* ... and the same two collectives over DEVICE memory, for the exchanges of the read-sharded split below (emissions, read states, duplicate keys, coverage_t: from the kernel that wrote
 * them to the kernel that reads them without leaving HBM).  agpu_scratch_buffer: a device buffer of the context by name (grow-only, kept for the next sample; not one of the names the
 * stages use: "exchange.*"); agpu_device_copy: `bytes` from source to destination, either in host or device memory. */
int agpu_rccl_all_gather_device(agpu_ctx* ctx, void* nccl_comm, const void* mine, void* all /* [n_ranks * bytes] */, uint64_t bytes);
int agpu_rccl_all_reduce_device(agpu_ctx* ctx, void* nccl_comm, void* values, uint64_t count, int kind);
int agpu_scratch_buffer(agpu_ctx* ctx, const char* name, uint64_t bytes, void** pointer);
int agpu_device_copy(agpu_ctx* ctx, void* destination, const void* source, uint64_t bytes);

/* restore the batch to its state right after agpu_upload_batch (filters, strands and gene sets cleared) so that the stages can be run again */
int agpu_reset(agpu_ctx* ctx);

/* mark_multimappers (source/read_chimeric_alignments.cpp:792-802); returns the reference's "marked" count in *marked */
int agpu_mark_multimappers(agpu_ctx* ctx, uint64_t* marked);

/* assign_strands_from_strandedness + annotate_alignments + gene fallback + dummy genes + gene ids
 * (source/arriba.cpp:160-325, source/annotation.cpp:431-555).  n_dummy_genes receives the number of
 * synthesized intergenic genes (ids n_genes .. n_genes+n_dummy-1). */
int agpu_annotate(agpu_ctx* ctx, uint32_t* n_dummy_genes);

/* Read-level filter cascade, stage group 1 (source/arriba.cpp:327-350): duplicates, uninteresting_contigs,
 * viral_contigs, top_expressed_viral_contigs, low_coverage_viral_contigs.  The two viral filters take
 * per-contig verdict tables that the host derives from coverage / expression scalars
 * (source/filter_top_expressed_viral_contigs.cpp:51-127, source/filter_low_coverage_viral_contigs.cpp:11-27);
 * pass NULL to treat no contig as filtered. */
int agpu_read_filters_stage1(agpu_ctx* ctx, const uint8_t* top_expressed_viral_verdict, const uint8_t* low_coverage_viral_verdict);

/* Host genes hit by virus-host chimeric fragments (input of filter_top_expressed_viral_contigs,
 * source/filter_top_expressed_viral_contigs.cpp:95-112): pairs (viral contig, gene id). */
int agpu_get_viral_integration_sites(agpu_ctx* ctx, uint32_t* pairs /* [2*capacity] */, uint64_t capacity, uint64_t* count);

/* Mate-gap samples for estimate_fragment_length (source/read_stats.cpp:11-44): the spliced distances of
 * the first <=100001 unfiltered paired split reads in name order, and the number of fragments the
 * reference's loop visits before it stops (for its sequential float read-length mean). */
int agpu_fragment_length_samples(agpu_ctx* ctx, int32_t* mate_gaps /* [100001] */, uint32_t* n_samples, uint64_t* fragments_visited);

/* Read-level filter cascade, stage group 2 (source/arriba.cpp:366-409): read_through, inconsistently_clipped,
 * homopolymer, small_insert_size, long_gap, same_gene, hairpin, mismatches, low_entropy.
 * remaining[AGPU_FILTER_COUNT] receives, for each read-level filter id, the reference's "(remaining=N)" count. */
int agpu_read_filters_stage2(agpu_ctx* ctx, uint64_t* remaining);

/* find_fusions (source/fusions.cpp:203-473).  Returns the number of candidates in *n_candidates. */
int agpu_find_fusions(agpu_ctx* ctx, int32_t max_mate_gap, uint64_t* n_candidates);

/* Candidate table after find_fusions, in order of first occurrence in name order (== the order in which the reference inserts
 * candidates into fusions_t, source/fusions.cpp:252).  Any pointer may be NULL.  flags = AGPU_CFLAG_*; contigs = contig1 << 16 | contig2;
 * list_offset[3*n+1] indexes the concatenated read lists (split_read1_list, split_read2_list, discordant_mate_list per candidate). */
int agpu_get_candidates(agpu_ctx* ctx, uint32_t* gene1, uint32_t* gene2, uint32_t* contigs, int32_t* breakpoint1, int32_t* breakpoint2, uint32_t* flags, uint8_t* filter,
                        uint32_t* split_reads1, uint32_t* split_reads2, uint32_t* discordant_mates, int32_t* anchor_start1, int32_t* anchor_start2, uint64_t* list_offset);
/* Round 5: the discordant_mate_lists may be IMPLICIT on the device -- every candidate of a gene pair lists the discordant mates of the pair near its breakpoints up to -U
 * (source/fusions.cpp:379-437): with -U 32767 a 10^8-fragment sample would hold 92.7 G entries.  list_offset is the same either way; agpu_get_candidate_read_lists then expands the
 * lists window by window into `reads` (capacity must hold all of them), agpu_get_candidate_read_lists_of expands those of the given candidates ("Read lists"). */
int agpu_get_candidate_read_lists(agpu_ctx* ctx, uint32_t* reads, uint64_t capacity, uint64_t* total);
/* the read lists of the given candidates only, packed: list_offset[3*n+1] starts at 0 (the output writer wants those of the candidates it prints -- a few
 * thousand of millions); call with reads == NULL to get *total and list_offset first */
int agpu_get_candidate_read_lists_of(agpu_ctx* ctx, const uint32_t* candidates, uint64_t n, uint64_t* list_offset, uint32_t* reads, uint64_t capacity, uint64_t* total);
/* The candidates one output file holds, picked on the device: write_fusions_to_file (source/output_fusions.cpp:1083-1089) writes the candidates with filter == FILTER_none
 * to -o (discarded = 0) and the others to -O (1) -- a few thousand of millions for -o, so only their columns travel.  agpu_select_candidates says how many there are,
 * agpu_get_selected_candidates fills arrays of that many entries (candidate = index in the table, ascending; NULL pointers are skipped).  evalue, confidence,
 * iteration_rank and the closest genomic breakpoints are those of agpu_get_evalues, agpu_assign_confidence (which must have run behind the last stage that changed a
 * filter), agpu_candidate_iteration_order and agpu_get_genomic_support. */
typedef struct {
	uint32_t* candidate;
	uint32_t* gene1; uint32_t* gene2; uint32_t* contigs; int32_t* breakpoint1; int32_t* breakpoint2; uint32_t* flags; uint8_t* filter;
	uint32_t* split_reads1; uint32_t* split_reads2; uint32_t* discordant_mates;
	float* evalue; uint8_t* confidence; uint32_t* iteration_rank;
	int32_t* closest_genomic_breakpoint1; int32_t* closest_genomic_breakpoint2;
} agpu_selected_candidates;
int agpu_select_candidates(agpu_ctx* ctx, int discarded, uint64_t* n);
int agpu_get_selected_candidates(agpu_ctx* ctx, const agpu_selected_candidates* out);
/* sizes of the last agpu_find_fusions: stats[0] gene-pair emissions, [1] candidates, [2] read-list entries, [3] discordant emissions,
 * [4] candidates whose discordant bucket was scanned by a whole wavefront */
int agpu_get_fusion_stats(agpu_ctx* ctx, uint64_t* stats /* [5] */);
/* 1 for every discordant fragment whose MATE1/MATE2 the reference swaps in place while attaching it (source/fusions.cpp:414-421) */
int agpu_get_discordant_swapped(agpu_ctx* ctx, uint8_t* swapped /* [n] */);

/* merge_adjacent_fusions (source/merge_adjacent_fusions.cpp:19-108, called at source/arriba.cpp:420-423 with max_distance 5):
 * a candidate absorbs the split reads of the candidates of its gene pair whose breakpoints are shifted by <= max_distance bp along the
 * same diagonal if it has the most support; the absorbed ones get the filter `merge_adjacent`.  *remaining = "(remaining=N)".
 * (For internal tandem duplications the reference also appends the read lists of the absorbed candidates; here only their sizes are kept.) */
int agpu_merge_adjacent_fusions(agpu_ctx* ctx, int32_t max_distance, uint64_t* remaining);

/* filter_multimappers (source/filter_multimappers.cpp:109-221, called at source/arriba.cpp:426-429): of the alignments of one read name the
 * one with the highest alignment score stays (ties: the one whose best candidate has more support), the others get the filter id
 * `multimappers`; candidates lose those reads from their counters and are discarded when none is left.
 * *remaining = "(remaining=N)", *discarded_reads = fragments newly filtered. */
int agpu_filter_multimappers(agpu_ctx* ctx, uint64_t* remaining, uint64_t* discarded_reads);

/* filter_multimappers when this context holds one shard of the sample.  After agpu_import_candidates the candidate table is
 * replicated, the read lists are with the rank that built a candidate, the reads with the rank that holds their shard.  Every rank calls, in order:
 *   agpu_set_owned_candidates     position in the replicated table of every candidate this rank built
 *   agpu_copy_multimapper_flags   flags[n]: 1 = the fragment is one of several alignments of a read name      -> all-gather (rank order)
 *   agpu_multimappers_begin       ranks the candidates (fusion_has_more_support), numbers the multi-mapping reads of the sample
 *   agpu_multimappers_partial_best  best[n_multimappers] (int32, 0x7FFFFFFF = none): best rank among the owned candidates -> all-reduce MIN
 *   agpu_multimappers_resolve     keeps the best alignment of every group of this shard; discarded_flags[n]  -> all-gather (rank order)
 *   agpu_multimappers_recount     owners lower their candidates' counters; counters[3 * n_candidates] (int32)   -> all-reduce MIN
 *   agpu_multimappers_finish      takes the reduced counters, filters the candidates left without support, *remaining = "(remaining=N)"
 * Pointers may be host or device memory. */
int agpu_set_owned_candidates(agpu_ctx* ctx, const uint32_t* global_index, uint64_t n_owned);
int agpu_copy_multimapper_flags(agpu_ctx* ctx, uint8_t* flags);
int agpu_multimappers_begin(agpu_ctx* ctx, const uint8_t* global_flags, uint64_t* n_multimappers);
int agpu_multimappers_partial_best(agpu_ctx* ctx, int32_t* best);
int agpu_multimappers_resolve(agpu_ctx* ctx, const int32_t* best, uint8_t* discarded_flags, uint64_t* discarded);
int agpu_multimappers_recount(agpu_ctx* ctx, const uint8_t* global_discarded, int32_t* counters);
int agpu_multimappers_finish(agpu_ctx* ctx, const int32_t* counters, uint64_t* remaining);

/* Candidate state as changed by the event-level stages that run on the host between find_fusions and the e-value
 * (merge_adjacent_fusions, filter_multimappers: source/arriba.cpp:420-430).  NULL = leave the column as it is. */
int agpu_set_candidate_state(agpu_ctx* ctx, const uint8_t* filter, const uint32_t* split_reads1, const uint32_t* split_reads2, const uint32_t* discordant_mates);

/* Position of every candidate in the iteration order of the reference's fusions_t (std::unordered_map with the tuple hash of
 * source/common.hpp:286-314, hazard H2), computed on the device and kept there for the stages that need it; iteration_rank may be NULL. */
int agpu_candidate_iteration_order(agpu_ctx* ctx, uint32_t* iteration_rank /* [n_candidates] */);

/* estimate_expected_fusions (source/filter_relative_support.cpp:17-207).  iteration_rank[c] = position of candidate c in the iteration
 * order of the reference's fusions_t: the partner dedup of :22-29 keeps the first event per key.  NULL = use the order computed by
 * agpu_candidate_iteration_order. */
int agpu_estimate_expected_fusions(agpu_ctx* ctx, uint64_t mapped_reads, const uint32_t* iteration_rank);
int agpu_get_evalues(agpu_ctx* ctx, float* evalue /* [n_candidates] */);
/* The three candidate predicates main() runs between the e-value and filter_relative_support (source/arriba.cpp:437-455):
 * filter_non_coding_neighbors, filter_intragenic_both_exonic, filter_min_support.  discarded[k] = candidates discarded by stage k. */
int agpu_filter_candidate_predicates(agpu_ctx* ctx, uint64_t* discarded /* [3] */);
/* filter_relative_support (source/filter_relative_support.cpp:209-224); *remaining = the reference's "(remaining=N)" */
int agpu_filter_relative_support(agpu_ctx* ctx, uint64_t* remaining);

/* Event-level predicates behind filter_relative_support, each a pure function of one candidate; candidates that already
 * have a filter are skipped, *remaining = "(remaining=N)" as the reference counts it (filter_both_intronic and filter_end_to_end_fusions leave the
 * candidates on viral contigs out of their count).  The state the stages in between produce on the host (recover_*, select_best, ...)
 * goes in with agpu_set_candidate_state / agpu_set_read_filters.
 *   agpu_upload_coverage       coverage_t of the ingest (source/read_stats.cpp:161-266), needed by filter_no_coverage
 *   agpu_filter_both_intronic  source/filter_both_intronic.cpp:18-36, called at source/arriba.cpp:469-472
 *   agpu_filter_short_anchor   source/filter_short_anchor.cpp:7-24,  source/arriba.cpp:531-534 (min_length = -A, default 23)
 *   agpu_filter_end_to_end     source/filter_end_to_end.cpp:28-78,   source/arriba.cpp:536-539
 *   agpu_filter_no_coverage    source/filter_no_coverage.cpp:9-103,  source/arriba.cpp:541-544
 *   agpu_filter_marginal_read_through  source/filter_marginal_read_through.cpp:8-46, source/arriba.cpp:503-506 */
int agpu_upload_coverage(agpu_ctx* ctx, const agpu_coverage_view* coverage);
/* recover_internal_tandem_duplication (source/recover_internal_tandem_duplication.cpp:11-85, called at source/arriba.cpp:463-466 with -Z min_itd_support,
 * default 10, and -z min_itd_allele_fraction, default 0.07; max_itd_length and the subsampling threshold come from agpu_params).  Also un-filters reads. */
int agpu_recover_internal_tandem_duplication(agpu_ctx* ctx, uint32_t min_supporting_reads, float min_fraction_of_coverage, uint64_t* remaining);
int agpu_filter_both_intronic(agpu_ctx* ctx, uint64_t* remaining);
int agpu_filter_short_anchor(agpu_ctx* ctx, uint32_t min_length, uint64_t* remaining);
int agpu_filter_end_to_end(agpu_ctx* ctx, uint64_t* remaining);
int agpu_filter_no_coverage(agpu_ctx* ctx, uint64_t* remaining);
int agpu_filter_marginal_read_through(agpu_ctx* ctx, uint64_t* remaining);
/* select_most_supported_breakpoints (source/select_best.cpp:21-80, called at source/arriba.cpp:497-500 and :567-570): of the unfiltered candidates of
 * one gene pair and direction pair only the best stays; the reference's fold runs in the iteration order of fusions_t (hazard H2), which the device
 * computes (agpu_candidate_iteration_order) if it has not been computed yet. */
int agpu_select_most_supported_breakpoints(agpu_ctx* ctx, uint64_t* remaining);
/* recover_many_spliced (source/recover_many_spliced.cpp:8-51, called at source/arriba.cpp:511-514; min_spliced_events = -M, default 4) */
int agpu_recover_many_spliced(agpu_ctx* ctx, uint32_t min_spliced_events, uint64_t* remaining);
/* filter_in_vitro (source/filter_in_vitro.cpp:82-228, called at source/arriba.cpp:483-486; high_expression_quantile = -Q, default 0.998): chimeric
 * fragments per gene as the expression proxy, breakpoints inside exons per gene pair, the verdict per candidate. */
int agpu_filter_in_vitro(agpu_ctx* ctx, float high_expression_quantile, uint64_t* remaining);
/* recover_both_spliced (source/recover_both_spliced.cpp:72-182, called at source/arriba.cpp:489-492 as recover_both_spliced(..., 200, 0.998, 1000, 1000)) */
int agpu_recover_both_spliced(agpu_ctx* ctx, uint32_t max_fusions_to_recover, float high_expression_quantile, int32_t max_exon_size, uint32_t max_coverage, uint64_t* remaining);

/* Read-level filter state as changed by stages that run on the host after the read-level cascade (filter_multimappers,
 * source/arriba.cpp:427-430): replaces the filter id of every fragment. */
int agpu_set_read_filters(agpu_ctx* ctx, const uint8_t* filter /* [n] */);

/* make_kmer_index (source/filter_mismappers.cpp:47-84, called at source/arriba.cpp:547-553): 8-mer positions of the genes of all
 * unfiltered candidates with gene1 != gene2, padded by `padding` = max_mate_gap + 2 * read_length_mean (as int). */
/* A line of a blacklist or known-fusions file (source/filter_blacklisted_ranges.hpp:13-22): two items, each a gene, a position, a range or -- second
 * column of a blacklist -- a keyword.  ahost_load_range_rules (arriba_host.h) parses the files. */
enum { AGPU_RULE_RANGE = 0, AGPU_RULE_POSITION = 1, AGPU_RULE_GENE = 2, AGPU_RULE_ANY = 3, AGPU_RULE_SPLIT_READ_DONOR = 4, AGPU_RULE_SPLIT_READ_ACCEPTOR = 5, AGPU_RULE_SPLIT_READ_ANY = 6,
       AGPU_RULE_DISCORDANT_MATES = 7, AGPU_RULE_READ_THROUGH = 8, AGPU_RULE_LOW_SUPPORT = 9, AGPU_RULE_FILTER_SPLICED = 10, AGPU_RULE_NOT_BOTH_SPLICED = 11 };
typedef struct {
	uint8_t type;            /* AGPU_RULE_* */
	uint8_t strand_defined;  /* position / range given as +contig:... or -contig:... */
	uint8_t strand;          /* 1 = forward */
	uint8_t reserved;
	uint32_t contig;         /* position / range / gene */
	int32_t start, end;      /* 0-based inclusive; a gene: its start and end */
	uint32_t gene;           /* gene id (AGPU_RULE_GENE) */
} agpu_range_item;
typedef struct { agpu_range_item first, second; } agpu_range_rule;
/* filter_blacklisted_ranges (source/filter_blacklisted_ranges.cpp:227-301, called at source/arriba.cpp:527-530 with -E evalue_cutoff and max_mate_gap):
 * an unfiltered candidate that a line matches gets the filter `blacklist`.  Needs the e-values. */
int agpu_filter_blacklisted_ranges(agpu_ctx* ctx, const agpu_range_rule* rules, uint32_t n_rules, float evalue_cutoff, int32_t max_mate_gap, uint64_t* remaining);
/* recover_known_fusions (source/recover_known_fusions.cpp:14-100, called at source/arriba.cpp:475-478): a candidate discarded for low support
 * (relative_support, min_support) between different genes that a line matches comes back.  Needs the coverage. */
int agpu_recover_known_fusions(agpu_ctx* ctx, const agpu_range_rule* rules, uint32_t n_rules, int32_t max_mate_gap, uint64_t* remaining);
/* Structural variants from whole-genome sequencing (-d; source/filter_genomic_support.cpp).  ahost_load_genomic_breakpoints (arriba_host.h) parses the file.
 * agpu_mark_genomic_support (mark_genomic_support, :81-219, called at source/arriba.cpp:415-418 behind find_fusions; max_distance = -D, default 100000):
 *   the closest pair of genomic breakpoints that can explain a candidate; used by the blacklist, assign_confidence, the two filters below and the output.
 * agpu_filter_no_genomic_support (:401-417, source/arriba.cpp:516-523; behind agpu_assign_confidence), agpu_recover_genomic_support (:419-444, source/arriba.cpp:568-571). */
typedef struct { uint32_t contig1, contig2; int32_t position1, position2; uint8_t upstream1, upstream2, reserved[2]; } agpu_genomic_breakpoint; /* contig1/position1 = the smaller coordinate */
int agpu_mark_genomic_support(agpu_ctx* ctx, const agpu_genomic_breakpoint* variants, uint32_t n_variants, int32_t max_distance, uint64_t* marked);
int agpu_get_genomic_support(agpu_ctx* ctx, int32_t* closest_genomic_breakpoint1 /* [n_candidates], -1 = none */, int32_t* closest_genomic_breakpoint2);
int agpu_filter_no_genomic_support(agpu_ctx* ctx, uint64_t* remaining);
int agpu_recover_genomic_support(agpu_ctx* ctx, uint64_t* remaining);
/* assign_confidence (source/filter_genomic_support.cpp:222-399, called at source/arriba.cpp:587-589; without structural variants from WGS).
 * confidence: [n_candidates] 0 = low, 1 = medium, 2 = high (source/common.hpp:224-227); may be NULL */
int agpu_assign_confidence(agpu_ctx* ctx, uint8_t* confidence);
/* recover_isoforms (source/recover_isoforms.cpp:10-47, called at source/arriba.cpp:580-584): the last of the candidate-level filters */
int agpu_recover_isoforms(agpu_ctx* ctx, uint64_t* remaining);
/* filter_homologs (source/filter_homologs.cpp:68-141, called at source/arriba.cpp:556-560 behind make_kmer_index; max_identity_fraction = -L, default 0.3).
 * Needs the k-mer index and the e-values.  The elimination only compares candidates that share a gene (the reference compares all pairs). */
int agpu_filter_homologs(agpu_ctx* ctx, float max_identity_fraction, uint64_t* remaining);
int agpu_make_kmer_index(agpu_ctx* ctx, int32_t padding, uint64_t* n_positions);
/* filter_mismappers (source/filter_mismappers.cpp:272-359): re-aligns the reads of every unfiltered candidate to the other gene
 * (mis-mapped reads get the filter id `mismappers`), then discards candidates that consist mostly of mis-mappers and lowers
 * their counters.  *remaining = the reference's "(remaining=N)"; *discarded_reads = reads newly marked as mis-mappers. */
int agpu_filter_mismappers(agpu_ctx* ctx, int32_t max_mate_gap, uint64_t* remaining, uint64_t* discarded_reads);
/* The same stage shared out over the ranks that hold the same batch (one sample over several GPUs, after agpu_shard_merge): the re-alignments are
 * independent per read and make up ~80 % of the time of a large sample, everything around them is the same on every rank.
 *   agpu_mismapper_jobs            the supporting reads of the unfiltered candidates, ordered by candidate (the same list on every rank): their number
 *   agpu_mismapper_verdicts        re-aligns the jobs part, part + parts, part + 2 parts, ...; verdicts (n_jobs bytes, device or host memory) receives 1 where
 *                                  a job of this part is a mis-mapper and 0 everywhere else -> one all-reduce (max) over the ranks
 *   agpu_filter_mismappers_apply   the verdicts of all ranks: the reads get the filter id, then the candidates are judged as in agpu_filter_mismappers */
int agpu_mismapper_jobs(agpu_ctx* ctx, uint64_t* n_jobs);
int agpu_mismapper_verdicts(agpu_ctx* ctx, int32_t max_mate_gap, uint32_t part, uint32_t parts, uint8_t* verdicts);
int agpu_filter_mismappers_apply(agpu_ctx* ctx, const uint8_t* verdicts, uint64_t* remaining, uint64_t* discarded_reads);

/* ---- Sharded samples: one context per GPU holds a contiguous range of the fragments in name order.
 * The per-fragment stages run on the shard; the entry points below expose the three places where the reference's result depends on
 * fragments of other shards, so that the host can exchange the state between the ranks (RCCL / any transport) in between.
 * Pointers named `destination` / `positions` / `entries` / `emissions` may be host or device memory. */
/* global name rank of fragment 0 of this context and the number of fragments of the whole sample */
int agpu_set_shard(agpu_ctx* ctx, uint64_t first_rank, uint64_t global_n);
/* agpu_annotate in two steps: the dummy genes (source/arriba.cpp:207-260) are cut from the unmapped positions of ALL shards */
int agpu_annotate_begin(agpu_ctx* ctx, uint64_t* n_unmapped);
int agpu_copy_unmapped_positions(agpu_ctx* ctx, uint64_t* destination /* [n_unmapped] contig << 32 | position */);
int agpu_annotate_finish(agpu_ctx* ctx, const uint64_t* positions /* of all shards, any order; NULL = own */, uint64_t n_positions, uint32_t* n_dummy_genes);
/* filter_duplicates keeps the first fragment of a key in the name order of the whole sample (source/filter_duplicates.cpp:8-55):
 * the local winners (16 bytes each: contigs u32, position1 i32, position2 i32, global name rank u32; ascending rank) are exchanged,
 * then every shard filters against the concatenation of all shards' entries in rank order */
#define AGPU_DUPLICATE_ENTRY_BYTES 16
int agpu_duplicates_begin(agpu_ctx* ctx, uint64_t* n_entries);
int agpu_copy_duplicate_entries(agpu_ctx* ctx, void* destination);
int agpu_read_filters_stage1_global(agpu_ctx* ctx, const void* entries, uint64_t n_entries, const uint8_t* top_expressed_viral_verdict, const uint8_t* low_coverage_viral_verdict);
/* agpu_fragment_length_samples stopping after `limit` samples (the shards contribute to the first 100001 samples in name order) */
int agpu_fragment_length_samples_limited(agpu_ctx* ctx, uint32_t limit, int32_t* mate_gaps, uint32_t* n_samples, uint64_t* fragments_visited);
/* find_fusions over shards: the emissions (one 36-byte record per read x gene1 x gene2) are partitioned by the owner of their gene
 * pair, name order kept inside a partition; the owner builds the candidates of its gene pairs from the emissions of all shards
 * concatenated in shard order.  counts[p] = emissions for partition p. */
#define AGPU_EMISSION_BYTES 36
int agpu_build_emissions(agpu_ctx* ctx, uint32_t n_partitions, uint64_t* counts /* [n_partitions] */);
int agpu_copy_emissions(agpu_ctx* ctx, void* destination /* all partitions, in partition order */);
int agpu_find_fusions_from_emissions(agpu_ctx* ctx, const void* emissions, uint64_t n_emissions, int32_t max_mate_gap, uint64_t* n_candidates);
/* first occurrence of every candidate in the name order of the whole sample (name rank of the read << 8 | ordinal of the emission):
 * sorting the candidates of all owners by it gives the reference's insertion order */
int agpu_get_candidate_first_occurrence(agpu_ctx* ctx, uint64_t* first_occurrence /* [n_candidates] */);

/* candidate columns of the whole sample (gathered from the owners, in insertion order) as this context's candidate table, so that the
 * candidate-level stages can run replicated on every rank; the read lists stay with the owners */
int agpu_import_candidates(agpu_ctx* ctx, uint64_t n_candidates, const uint32_t* gene1, const uint32_t* gene2, const uint32_t* contigs, const int32_t* breakpoint1, const int32_t* breakpoint2,
                           const uint32_t* flags, const uint8_t* filter, const uint32_t* split_reads1, const uint32_t* split_reads2, const uint32_t* discordant_mates, const int32_t* anchor_start1, const int32_t* anchor_start2);

/* ---- One sample over the GPUs of a node, the READS SHARDED (round 6; SURVEY.md section 8 row e, BASELINE.json north_star: "BAM records shard naturally by read ... a single
 * RCCL all-gather ... to merge breakpoint histograms before candidate scoring").  Rank r ingests part r of the records (agpu_ingest_config.part_of_sample) and KEEPS its
 * fragments: the batch -- columns, CIGAR / sequence / name pools, gene sets -- exists once over the ranks, not once per rank.  What the reference's result depends on across
 * reads travels, and nothing else:
 *   coverage_t, mapped_viral_reads_by_contig       sums / ORs over the parts (agpu_coverage_partial -> all-reduce -> agpu_coverage_total)            source/read_stats.cpp:161-266
 *   detect_strandedness                            the first 100 informative fragments of the sample in name order (agpu_strandedness_votes)         source/read_stats.cpp:94-143
 *   dummy genes, duplicates, mate gaps             agpu_annotate_begin/finish, agpu_duplicates_begin / agpu_read_filters_stage1_global, agpu_fragment_length_samples_limited (above)
 *   find_fusions                                   agpu_build_emissions(1 partition) -> ONE all-gather of the emissions (36 bytes per read x gene pair) -> agpu_find_fusions_from_emissions:
 *                                                  every rank builds every candidate and every read list (global name ranks), identically               source/fusions.cpp:203-473
 *   what a walk over read lists asks of a read     one byte per fragment -- its filter, is it a multi-mapper, is an alignment exonic -- replicated whenever the filters have changed
 *                                                  (agpu_read_state_export -> all-gather -> agpu_read_state_import: 10^8 bytes at 10^8 fragments, 1/220 of the batch); the stages that
 *                                                  judge candidates by their lists (filter_both_intronic, recover_both_spliced, recover_internal_tandem_duplication, the recounts of
 *                                                  filter_multimappers and filter_mismappers, the filters of the rows of the output files) read that byte on every rank
 *   what needs the alignments of a read            is computed where the read lives: the scores of filter_multimappers (agpu_filter_multimappers_resolve), the re-alignments of
 *                                                  filter_mismappers (agpu_filter_mismappers_search), the clipped discordant mates of filter_in_vitro as partial counts per candidate
 *                                                  over the reads a rank holds (agpu_in_vitro_clipped_mates -> all-gather -> agpu_filter_in_vitro_sharded), chimeric fragments per gene
 *                                                  (agpu_gene_read_counts -> all-reduce -> agpu_set_gene_read_counts), the rows of the supporting reads of the written candidates
 *                                                  (agpu_gather_rows_* with this rank's fragments)
 * Requires that the parts follow each other in name order (agpu_shard_boundary_names: the caller checks last name of part r < first name of part r + 1 -- every part is sorted, so
 * no read name is in two parts); a file whose names are in another order goes through agpu_shard_export / agpu_shard_merge above.  Pointers may be host or device memory. */
/* "QNAME,HI" of the first and of the last fragment of this context's batch (empty strings when it holds none); capacity = bytes of each of the two buffers */
int agpu_shard_boundary_names(agpu_ctx* ctx, char* first_name, char* last_name, uint32_t capacity);
/* the batch of the part this context ingested stays its own: fragment i has the global name rank first_rank + i, the sample holds global_n fragments (agpu_set_shard); the stages
 * behind find_fusions switch to their sharded form (below) */
int agpu_shard_keep(agpu_ctx* ctx, uint64_t first_rank, uint64_t global_n);
/* coverage_t of this part before its 16-bit saturation (windows[W] saturated at 65535 as 32-bit words: min(sum of min(x_r, 65535), 65535) == min(sum x_r, 65535)), its start / end
 * flags and viral read counts[n_contigs]; agpu_coverage_total takes the sums / ORs over the parts and clamps: coverage_t of the sample on this context */
int agpu_coverage_partial(agpu_ctx* ctx, uint32_t* windows, uint8_t* fragment_starts, uint8_t* fragment_ends, uint64_t* viral_counts);
int agpu_coverage_total(agpu_ctx* ctx, const uint32_t* windows, const uint8_t* fragment_starts, const uint8_t* fragment_ends, const uint64_t* viral_counts);
/* the votes of detect_strandedness among the first `wanted` informative fragments of this context: how many there are (<= wanted) and how many of them match */
int agpu_strandedness_votes(agpu_ctx* ctx, uint32_t wanted, uint32_t* informative, uint32_t* matching);
/* one byte per fragment: filter id (bits 0-5) | multi-mapper (bit 6) | an alignment is exonic (bit 7); export: the n fragments of this context, import: the global_n of the sample in
 * name order (the own ones among them must be the exported ones) */
#define AGPU_READ_STATE_MULTIMAPPER 0x40u
#define AGPU_READ_STATE_EXONIC 0x80u
int agpu_read_state_export(agpu_ctx* ctx, uint8_t* state /* [n] */);
int agpu_read_state_import(agpu_ctx* ctx, const uint8_t* state /* [global_n] */);
/* filter_multimappers (source/filter_multimappers.cpp:109-221) in two halves around an exchange of the read states: _resolve ranks the candidates, finds the best candidate of every
 * multi-mapping read of the sample (all lists are here), scores and resolves the groups of alignments of THIS context's reads (*discarded_reads = newly filtered here);
 * _recount (behind agpu_read_state_import) lowers the counters of the candidates and discards those left without support; *remaining = "(remaining=N)" */
int agpu_filter_multimappers_resolve(agpu_ctx* ctx, uint64_t* discarded_reads);
int agpu_filter_multimappers_recount(agpu_ctx* ctx, uint64_t* remaining);
/* chimeric fragments per gene (source/filter_in_vitro.cpp:49-58) of this context's reads; counts[n_genes + n_dummy_genes]; agpu_set_gene_read_counts: those of the sample */
int agpu_gene_read_counts(agpu_ctx* ctx, uint32_t* counts);
int agpu_set_gene_read_counts(agpu_ctx* ctx, const uint32_t* counts);
/* filter_in_vitro (source/filter_in_vitro.cpp:82-228): the discordant mates of a candidate that are clipped at one of its breakpoints (:133-158), counted over the reads this context
 * holds -- entries of 12 bytes (candidate, count at breakpoint1, count at breakpoint2; candidates without any are left out); the entries of all ranks, in any order, give the verdicts */
#define AGPU_CLIPPED_MATES_ENTRY_BYTES 12
int agpu_in_vitro_clipped_mates(agpu_ctx* ctx, uint64_t* n_entries);
int agpu_copy_in_vitro_clipped_mates(agpu_ctx* ctx, void* destination);
int agpu_filter_in_vitro_sharded(agpu_ctx* ctx, float high_expression_quantile, const void* entries, uint64_t n_entries, uint64_t* remaining);
/* filter_mismappers (source/filter_mismappers.cpp:272-359): _search re-aligns the reads of this context that the unfiltered candidates list and discards the mis-mappers among them
 * (*discarded_reads: here); _judge (behind agpu_read_state_import) counts the mis-mappers of every candidate and discards the candidates; *remaining = "(remaining=N)" */
int agpu_filter_mismappers_search(agpu_ctx* ctx, int32_t max_mate_gap, uint64_t* discarded_reads);
int agpu_filter_mismappers_judge(agpu_ctx* ctx, uint64_t* remaining);

/* result access (device -> host copies) */
int agpu_get_filters(agpu_ctx* ctx, uint8_t* filter /* [n] */);
/* ... of the given fragments only (the writer looks at the filters of the supporting reads of the candidates it writes) */
int agpu_get_filters_of(agpu_ctx* ctx, const uint32_t* fragments, uint64_t n, uint8_t* filter /* [n] */);
int agpu_get_alignment_bits(agpu_ctx* ctx, int slot, uint8_t* abits /* [n] */);
int agpu_get_fragment_bits(agpu_ctx* ctx, uint8_t* fbits /* [n] */);
/* gene sets as CSR: count[n] then the concatenated ids; call with genes == NULL to get the total in *total */
int agpu_get_gene_sets(agpu_ctx* ctx, int slot, uint8_t* count /* [n] */, uint32_t* genes, uint64_t capacity, uint64_t* total);
/* gene table including dummy genes */
int agpu_get_gene_table(agpu_ctx* ctx, uint32_t first, uint32_t count, uint16_t* contig, int32_t* start, int32_t* end, uint8_t* bits, int32_t* exonic_length);

/* Per-kernel timing: while enabled, every kernel launch of the stage calls is bracketed by HIP events on the launch stream.
 * agpu_get_kernel_profile returns one sample per launch since profiling was switched on (names: [capacity][AGPU_KERNEL_NAME_LENGTH],
 * ms: duration, bytes: algorithmic bytes of that launch, 0 = not modelled); *count receives the number of samples available. */
#define AGPU_KERNEL_NAME_LENGTH 48
int agpu_set_profiling(agpu_ctx* ctx, int enabled);
int agpu_get_kernel_profile(agpu_ctx* ctx, char* names, float* ms, uint64_t* bytes, uint32_t capacity, uint32_t* count);

/* timing of the kernels launched by the last call, measured with HIP events on the launch stream (ms) */
int agpu_last_kernel_ms(agpu_ctx* ctx, float* ms);
/* algorithmic bytes (inputs read + outputs written once) of the kernels launched by the last call */
int agpu_last_kernel_bytes(agpu_ctx* ctx, uint64_t* bytes);

#ifdef __cplusplus
}
#endif

#endif /* ARRIBA_GPU_H */
