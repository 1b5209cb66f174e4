/* include/arriba_workflow.h -- the reference's main() behind its option parser (source/arriba.cpp:84-615) as one call over the two C ABIs
 * (arriba_host.h: loaders, ingest, sequential scalar stages, output writer; arriba_gpu.h: the stages on the MI355X).  The structure mirrors
 * options_t (source/options.hpp); arriba_workflow_default_options fills in the defaults of source/options.cpp:71-107.  The command line of the
 * reference (source/options.cpp:270-481) in front of it is arriba_amd/csrc/workflow/main.cpp -> arriba_amd/lib/arriba_gpu_workflow. */
#ifndef ARRIBA_WORKFLOW_H
#define ARRIBA_WORKFLOW_H 1

#include <stdint.h>
#include "arriba_gpu.h"
#include "arriba_host.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
	/* input / output files; NULL = not given */
	const char* assembly_file;            /* -a */
	const char* gene_annotation_file;     /* -g */
	const char* chimeric_bam_file;        /* -x (or -c) */
	const char* output_file;              /* -o */
	const char* discarded_output_file;    /* -O */
	const char* blacklist_file;           /* -b */
	const char* known_fusions_file;       /* -k */
	const char* tags_file;                /* -t */
	const char* protein_domains_file;     /* -p */
	const char* genomic_breakpoints_file; /* -d */
	const char* interesting_contigs;      /* -i; NULL = the reference's default */
	const char* viral_contigs;            /* -v */
	const char* gtf_features;             /* -G */
	agpu_params device;                   /* -H -R -l -U -V -K -E -m -F -u -e -S -f; strandedness: 0 no, 1 yes, 2 reverse, 3 auto (-s) */
	uint32_t min_itd_support;             /* -Z 10 */
	float min_itd_allele_fraction;        /* -z 0.07 */
	float high_expression_quantile;       /* -Q 0.998 */
	uint32_t min_spliced_events;          /* -M 4 */
	uint32_t min_anchor_length;           /* -A 23 */
	float max_homolog_identity;           /* -L 0.3 */
	uint32_t top_viral_contigs;           /* -T 5 */
	float viral_contig_min_covered_fraction; /* -C 0.05 */
	int32_t max_genomic_breakpoint_distance; /* -D 100000 */
	uint8_t print_extra_info_for_discarded_fusions; /* -X */
	uint8_t fill_sequence_gaps;           /* -I */
	int device_index;                     /* which GPU */
	uint8_t log_to_stdout;                /* 1: the progress lines of the reference's main() ("[time] Filtering duplicates (remaining=N)", source/arriba.cpp:96-610) go to stdout */
	uint8_t host_ingest;                  /* 0 (default): read_chimeric_alignments runs on the device (agpu_ingest_*), the host feeds the bytes of the file;
	                                         1: the multi-threaded host ingest builds the batch and uploads it */
} arriba_workflow_options;

/* what the reference prints as "(remaining=N)" / "(total=N)" / "(marked=N)", in the order of the stages; stage names as in the reference's source */
typedef struct { char stage[48]; uint64_t count; } arriba_workflow_stage;
typedef struct { uint32_t n_stages; arriba_workflow_stage stages[64]; } arriba_workflow_report;

void arriba_workflow_default_options(arriba_workflow_options* options);
/* returns 0, or a negative number with the text in arriba_workflow_last_error() */
int arriba_workflow_run(const arriba_workflow_options* options, arriba_workflow_report* report /* may be NULL */);
const char* arriba_workflow_last_error(void);

/* The same workflow as a resident service: what main() does once per process -- load_assembly, read_annotation_gtf, make_annotation_index, the -t / -p files
 * (source/arriba.cpp:97-113,586-598) and, here, the device context with the annotation in HBM -- happens in arriba_workflow_open; arriba_workflow_sample is
 * everything main() does per sample: read_chimeric_alignments ... the output file(s) (source/arriba.cpp:119-610).  The buffers of a sample (the pinned pieces the file
 * is fed through, the HBM of the stages) stay with the session for the next one.  arriba_workflow_run == open + sample + close.  bench.py times arriba_workflow_sample. */
typedef struct arriba_workflow_session arriba_workflow_session;
typedef struct { /* seconds of one sample, by part (wall clock of the calling thread) */
	double total;
	double feed;             /* the bytes of the file on their way to HBM (ahost_bam_next + agpu_ingest_push*), up to the last piece */
	double ingest;           /* agpu_ingest_finish: read_chimeric_alignments on the device behind the last piece */
	double adopt;            /* counters, coverage_t and viral read counts of the ingest back to the host session */
	double stages;           /* mark_multimappers ... filter_homologs and what follows filter_mismappers up to assign_confidence (every stage on the device but the next line) */
	double filter_mismappers;
	double output;           /* write_fusions_to_file for -o (and -O): results back, rows of the supporting reads, formatting, the file */
	double output_results;   /*   of it: the candidate table and read lists of the written candidates back to the host */
	double output_rows;      /*   of it: the rows (names, CIGARs, sequences) of their supporting reads */
	double output_format;    /*   of it: ahost_write_fusions */
	double feed_read;        /* of feed: inside ahost_bam_next (the bytes of the file into the pinned pieces) */
	double feed_push;        /* of feed: inside agpu_ingest_push* (enqueue the copy, wait for the copy before it, move the windows of the ingest on) */
	double feed_total;       /* the feed from the moment the file was opened to the last piece, wherever it ran: a sample submitted ahead is fed under the stages of the sample before,
	                            and `feed` is then only what arriba_workflow_sample still had to wait for */
	double exchange_parts;   /* one sample over several ranks (arriba_workflow_set_communicator): the parts of the batch exported, all-gathered and merged */
	double exchange_verdicts;/*   ... the all-reduce of the verdicts of filter_mismappers (inside filter_mismappers above) */
	double exchange_rows;    /*   ... the texts of the rows of the output file(s) gathered (inside output_format above) */
	double shard_fragments;  /* one sample over several ranks, the reads sharded (below): the fragments THIS rank held through the sample (of the `read_chimeric_alignments` of the report); 0 = every
	                            rank held the whole batch (one rank, or the split by an all-gather of the batch) */
	double exchanged_bytes;  /*   ... and the bytes this rank received from the others in the exchanges of the sample (host collectives) */
} arriba_workflow_timing;
/* options->chimeric_bam_file, output_file and discarded_output_file are not used by open (they belong to a sample); NULL + arriba_workflow_last_error() on failure */
arriba_workflow_session* arriba_workflow_open(const arriba_workflow_options* options);
int arriba_workflow_sample(arriba_workflow_session* session, const char* chimeric_bam_file, const char* output_file, const char* discarded_output_file /* may be NULL */,
                           arriba_workflow_report* report /* may be NULL */, arriba_workflow_timing* timing /* may be NULL */);
/* Samples one after the other through a resident session: arriba_workflow_submit(next) before arriba_workflow_sample(current) lets the file of the next sample cross PCIe and go
 * through the front of read_chimeric_alignments (the windows of the ingest) while the stages, filter_mismappers and the writer of the current one run -- a second lane (host
 * session + sibling device context, agpu_create_sibling) is made for it at the first call.  Samples are worked on in the order they were submitted; at most one can be submitted
 * ahead of the one at work.  arriba_workflow_sample(bam) with nothing submitted submits bam itself (the behaviour without this call).  A sample that was submitted and never
 * asked for is thrown away by arriba_workflow_close.  Returns 0, or a negative number with the text in arriba_workflow_last_error(). */
int arriba_workflow_submit(arriba_workflow_session* session, const char* chimeric_bam_file);
/* (If the device runs out of memory while two samples are in flight, arriba_workflow_sample throws away what was fed ahead, closes the second lane, runs its sample again with the device
 * to itself and submits the other sample again behind it -- once; a sample that does not fit the device alone fails the call.  INTEGRATION.md, "Memory".) */
/* on: arriba_workflow_sample returns when the last output file of the sample (-O if given, else -o) has everything it needs off the device; the file is formatted and written
 * by a thread of the session while the next sample is worked on.  It is complete when the next-but-one arriba_workflow_sample of the session has returned (the writer works from a sample detached from its host session --
 * ahost_detach_sample -- so the feed of the lane's next sample does not wait for it), or
 * arriba_workflow_flush, or arriba_workflow_close; a failure to write it is reported by the next arriba_workflow_sample or by arriba_workflow_flush.  Off by default. */
int arriba_workflow_defer_output(arriba_workflow_session* session, int on);
/* round 5: the ingest of a submitted sample is FINISHED by the thread that feeds its file (agpu_ingest_finish + the counters and coverage_t back to the host), beside the stages of the
 * sample in front, instead of by the arriba_workflow_sample that asks for it: a step of a queue is then max(feed + finish, stages + filter_mismappers + results) instead of
 * finish + max(feed, ...).  The lanes keep their batch buffers for it (agpu_keep_batch_buffers: ~25 GB more HBM at 10^8 fragments).  Before the first submit; off by default. */
int arriba_workflow_finish_ahead(arriba_workflow_session* session, int on);
int arriba_workflow_flush(arriba_workflow_session* session, double* seconds_of_last_writer /* may be NULL */);

/* ---- one sample over the ranks of a job: one process per GPU, every rank with a session of its own over the same reference files (SURVEY.md section 8 row e; BASELINE.json
 * config 4).  With a communicator set, arriba_workflow_sample is a collective call -- every rank calls it with the same file names -- and does what one rank does, shared out:
 *   read_chimeric_alignments   rank r feeds and ingests part r of the records of the file (ahost_bam_open_part: cut between read names at places every rank finds by itself);
 *                              ONE all-gather of the parts (agpu_shard_export -> all-gather -> agpu_shard_merge): every rank then holds the batch of the whole sample, bit for bit
 *                              what a single ingest builds (fragments concatenate in name order, counters and coverage_t add up before saturation)
 *   the stages                 on every rank over the whole batch: identical inputs, identical kernels, no exchange
 *   filter_mismappers          rank r re-aligns the jobs r, r + size, ... (agpu_mismapper_verdicts), ONE all-reduce (max) of the verdict bytes, agpu_filter_mismappers_apply
 *   the output files           rank r formats the rows r, r + size, ... (ahost_format_fusions), the texts are gathered and rank 0 writes the file(s)
 * A failure on one rank (a damaged block in its part of the file, no memory) ends the call on all of them: in front of every collective the ranks tell each other how they fared.
 * The transport is the caller's: three collectives over HOST memory as callbacks (torch.distributed with gloo in the CPU tests, MPI, ...), each returning 0 or non-zero; and,
 * if the ranks hold an RCCL communicator over their GPUs, `rccl_communicator` (ncclComm_t): the two large exchanges -- the parts of the batch, 22 GB at 10^8 fragments, and the
 * verdicts -- then stay in device memory (agpu_shard_merge_rccl, agpu_filter_mismappers_rccl) and only sizes, status words and row texts go through the callbacks.
 * arriba_workflow_join_rccl makes such a communicator out of RCCL alone (callbacks included: host bytes bounced through the device): rank 0 calls arriba_workflow_rccl_unique_id, the
 * id reaches the other ranks by whatever started them, every rank joins.  With a communicator the session works on one sample at a time on the calling thread's schedule:
 * arriba_workflow_submit still feeds the next sample's part ahead, arriba_workflow_defer_output and arriba_workflow_finish_ahead are ignored (their threads would issue collectives
 * in an order of their own).  Before the first submit / sample; NULL takes the communicator away again. */
enum { ARRIBA_WORKFLOW_MAX = 0, ARRIBA_WORKFLOW_MIN = 1, ARRIBA_WORKFLOW_SUM = 2 };
typedef struct {
	uint32_t rank, size;
	void* state;                                                                                   /* handed to the callbacks */
	int (*all_gather)(void* state, const void* mine, void* all /* [size * bytes] */, uint64_t bytes);   /* `bytes` of every rank, in rank order */
	int (*all_reduce_int64)(void* state, int64_t* values, uint64_t count, int operation);          /* element-wise over the ranks, in place; ARRIBA_WORKFLOW_MAX / MIN / SUM */
	int (*all_reduce_max_bytes)(void* state, uint8_t* values, uint64_t count);                     /* element-wise maximum over the ranks, in place */
	void* rccl_communicator;                                                                       /* ncclComm_t of the ranks' GPUs, or NULL */
} arriba_workflow_communicator;
int arriba_workflow_set_communicator(arriba_workflow_session* session, const arriba_workflow_communicator* communicator);
int arriba_workflow_rccl_unique_id(unsigned char* id /* [AGPU_RCCL_ID_BYTES] */);
int arriba_workflow_join_rccl(arriba_workflow_session* session, const unsigned char* id, uint32_t rank, uint32_t size);
int arriba_workflow_cancel(arriba_workflow_session* session); /* what was submitted and not yet worked on is thrown away (its feed is waited for first) */
agpu_ctx* arriba_workflow_device(arriba_workflow_session* session);      /* the device context of the lane that worked on the last sample, e.g. for agpu_get_kernel_profile */
agpu_ctx* arriba_workflow_lane_device(arriba_workflow_session* session, int lane); /* lane 0 or 1; NULL if the lane does not exist (yet) */
ahost_session* arriba_workflow_host(arriba_workflow_session* session);
void arriba_workflow_close(arriba_workflow_session* session);

#ifdef __cplusplus
}
#endif
#endif
