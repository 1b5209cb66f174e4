// tools/ingest_main.cpp -- minimal driver of the host library (open + ingest one BAM) for tools/sanitize_host.sh and for profiling the ingest
#include "arriba_host.h"
#include <cstdio>
int main(int argc, char** argv) {
	ahost_session* s = ahost_open(argv[1], argv[2], NULL, NULL, NULL);
	if (!s) { printf("open failed: %s\n", ahost_last_error()); return 1; }
	int status = ahost_ingest_bam_file(s, argv[3], 0, 100);
	printf("status %d %s\n", status, status ? ahost_last_error() : "");
	return 0;
}
