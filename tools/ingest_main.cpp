// tools/ingest_main.cpp -- minimal driver of the host library (open + ingest one BAM) for tools/sanitize_host.sh and for profiling the ingest
#include "arriba_host.h"
#include <cstdio>
int main(int argc, char** argv) {
	ahost_session* s = ahost_open(argv[1], argv[2], NULL, NULL, NULL);
	if (!s) { printf("open failed: %s\n", ahost_last_error()); return 1; }
	int status = ahost_ingest_bam_file(s, argv[3], 0, 100);
	printf("status %d %s\n", status, status ? ahost_last_error() : "");
	for (int a = 4; a < argc; ++a) { // blacklist / known-fusions files: the parser (malformed lines included)
		const agpu_range_rule* rules = NULL; uint32_t n_rules = 0;
		const int parsed = ahost_load_range_rules(s, argv[a], a == 4, &rules, &n_rules);
		printf("rules %d %u %s\n", parsed, n_rules, parsed ? ahost_last_error() : "");
	}
	return 0;
}
