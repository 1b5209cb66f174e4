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

#ifdef __cplusplus
}
#endif
#endif
