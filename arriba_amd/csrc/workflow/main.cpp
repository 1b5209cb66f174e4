// arriba_amd/csrc/workflow/main.cpp -- minimal driver of arriba_workflow_run: input files in, output files out, every option at the reference's default.
// (Not the reference's command line; its option parser, source/options.cpp, is out of scope.)
#include "../../../include/arriba_workflow.h"
#include <cstdio>
#include <cstring>
int main(int argc, char** argv) {
	arriba_workflow_options options;
	arriba_workflow_default_options(&options);
	const char** slots[] = { &options.assembly_file, &options.gene_annotation_file, &options.chimeric_bam_file, &options.output_file, &options.discarded_output_file, &options.blacklist_file, &options.known_fusions_file,
	                         &options.tags_file, &options.protein_domains_file, &options.genomic_breakpoints_file };
	if (argc < 5) { fprintf(stderr, "usage: %s assembly.fa annotation.gtf chimeric.bam fusions.tsv [discarded.tsv [blacklist [known_fusions [tags [protein_domains.gff3 [structural_variants]]]]]]  (\"-\" = not given)\n", argv[0]); return 2; }
	for (int a = 1; a < argc && a <= 10; ++a) *slots[a - 1] = strcmp(argv[a], "-") == 0 ? NULL : argv[a];
	arriba_workflow_report report;
	if (arriba_workflow_run(&options, &report) != 0) { fprintf(stderr, "%s\n", arriba_workflow_last_error()); return 1; }
	for (uint32_t s = 0; s < report.n_stages; ++s) printf("%s\t%llu\n", report.stages[s].stage, (unsigned long long) report.stages[s].count);
	return 0;
}
