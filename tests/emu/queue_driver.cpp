// tests/emu/queue_driver.cpp -- TEST ONLY: samples through one resident session in a queue with their last files deferred (arriba_workflow_submit / _defer_output / _flush), for
// tools/sanitize_workflow_threads.sh: the feeder of a sample, the writer of the sample two in front of it (from a detached sample of the same host session) and the thread that
// runs the stages of the sample between them, under ThreadSanitizer.  queue_driver GTF FASTA OUT_PREFIX BAM [BAM ...]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/arriba_workflow.h"

int main(int argc, char** argv) {
	if (argc < 5) { fprintf(stderr, "usage: queue_driver GTF FASTA OUT_PREFIX BAM [BAM ...]\n"); return 2; }
	arriba_workflow_options options;
	arriba_workflow_default_options(&options);
	options.gene_annotation_file = argv[1]; options.assembly_file = argv[2];
	options.print_extra_info_for_discarded_fusions = 1;
	arriba_workflow_session* session = arriba_workflow_open(&options);
	if (!session) { fprintf(stderr, "%s\n", arriba_workflow_last_error()); return 1; }
	std::vector<std::string> samples(argv + 4, argv + argc);
	arriba_workflow_defer_output(session, 1);
	if (getenv("QUEUE_FINISH_AHEAD") != nullptr && arriba_workflow_finish_ahead(session, 1) != 0) { fprintf(stderr, "%s\n", arriba_workflow_last_error()); return 1; } // (the feeder of a sample also finishes its ingest: one more thing that runs beside the stages)
	int status = 0;
	if (arriba_workflow_submit(session, samples[0].c_str()) != 0) status = 1;
	for (size_t k = 0; k < samples.size() && status == 0; ++k) {
		if (k + 1 < samples.size() && arriba_workflow_submit(session, samples[k + 1].c_str()) != 0) { status = 1; break; }
		const std::string out = std::string(argv[3]) + std::to_string(k) + ".tsv", discarded = std::string(argv[3]) + std::to_string(k) + ".discarded.tsv";
		if (arriba_workflow_sample(session, samples[k].c_str(), out.c_str(), discarded.c_str(), NULL, NULL) != 0) status = 1;
	}
	if (status == 0 && arriba_workflow_flush(session, NULL) != 0) status = 1;
	if (status != 0) fprintf(stderr, "%s\n", arriba_workflow_last_error());
	arriba_workflow_close(session);
	return status;
}
