#!/bin/bash
# tools/sanitize_host.sh -- the host library under ThreadSanitizer and AddressSanitizer/UBSan (test tooling; the reference has no
# sanitizer or race-detection setup).  Builds a small driver (tools/ingest_main.cpp) with the host sources, generates a
# dataset with shuffled names and separated mates (first mates wait parked) and ingests it with 1 and 4 worker threads.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
WORK=$(mktemp -d /tmp/sanitize_XXXXXX)
GEN=$ROOT/arriba_amd/lib/gen_synth
$GEN --out $WORK/data --seed 17 --fragments 30000 --normal-mult 0.5 --contigs 5 --contig-len 400000 --junctions 400 --dup 0.2 --shuffle --separate-mates --rule-files > /dev/null 2>&1
for MODE in thread address,undefined; do
	g++ -std=c++17 -O1 -g -pthread -fsanitize=$MODE -fno-omit-frame-pointer -I$ROOT/include -I$ROOT/arriba_amd/csrc/host -o $WORK/ingest_$MODE $ROOT/tools/ingest_main.cpp $ROOT/arriba_amd/csrc/host/*.cpp -lz
	for THREADS in 1 4; do
		echo "== -fsanitize=$MODE, $THREADS worker thread(s)"
		ARRIBA_INGEST_THREADS=$THREADS ASAN_OPTIONS=detect_leaks=0 $WORK/ingest_$MODE $WORK/data.fa $WORK/data.gtf $WORK/data.bam $WORK/data.blacklist.tsv $WORK/data.known_fusions.tsv 2>&1 | grep -v "SAM records were malformed\|^WARNING" || true
	done
done
rm -rf $WORK
