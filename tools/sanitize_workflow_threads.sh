#!/bin/bash
# tools/sanitize_workflow_threads.sh -- the C++ workflow driver (arriba_gpu_workflow's code) on the host stepping harness, host library and harness built with
# ThreadSanitizer: the threads of the file feed (pread in ranges), of the host side of the device ingest and of the output writer (rows formatted in parallel) on a
# sample with insertions, deletions and non-template bases, -X (fusion transcripts for the discarded candidates, too); the file in pieces of 1 MB, so that the reader thread and
# the pusher of the driver (four pinned buffers in turn) hand ~20 pieces to each other.  Test tooling.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
WORK=$(mktemp -d /tmp/tsan_XXXXXX)
cd $ROOT/tests/emu
g++ -std=c++17 -O1 -g -fPIC -shared -pthread -fsanitize=thread -Wno-parentheses -Wno-sign-compare -Wno-unused-variable -o $WORK/libemu.so emu.cpp
(cd $ROOT/arriba_amd/csrc && g++ -std=c++17 -O1 -g -fPIC -shared -pthread -fsanitize=thread -Wno-parentheses -Wno-sign-compare -o $WORK/libarriba_host.so host/*.cpp -lz)
g++ -std=c++17 -O1 -g -pthread -fsanitize=thread -include emu_names.h -o $WORK/workflow_tsan $ROOT/arriba_amd/csrc/workflow/workflow.cpp $ROOT/arriba_amd/csrc/workflow/main.cpp -L$WORK -lemu -larriba_host -Wl,-rpath,$WORK
$ROOT/arriba_amd/lib/gen_synth --out $WORK/d --seed 303 --fragments 30000 --normal-mult 0.4 --contigs 5 --contig-len 400000 --junctions 200 --dup 0.1 --indels 1.0 --non-template 0.5 > /dev/null 2>&1
cd $WORK
ARRIBA_FEED_PIECE_MB=1 ARRIBA_WRITER_THREADS=8 ARRIBA_INGEST_THREADS=4 ./workflow_tsan -x d.bam -g d.gtf -a d.fa -o f.tsv -O disc.tsv -X -f blacklist > log 2>&1 || true
echo "ThreadSanitizer warnings: $(grep -c 'WARNING: ThreadSanitizer' log || true); $(wc -l < f.tsv) lines in fusions.tsv"
# ... and four samples in a queue through one resident session, their last files deferred: the feeder of sample k + 1, the writer of sample k - 1 (from a sample detached from the
# host session that feeder has just opened its file on) and the stages of sample k at once; every file must equal the file of the sample run alone
g++ -std=c++17 -O1 -g -pthread -fsanitize=thread -include $ROOT/tests/emu/emu_names.h -o $WORK/queue_tsan $ROOT/arriba_amd/csrc/workflow/workflow.cpp $ROOT/tests/emu/queue_driver.cpp -L$WORK -lemu -larriba_host -Wl,-rpath,$WORK
for k in 1 2 3; do $ROOT/arriba_amd/lib/gen_synth --out $WORK/q$k --seed 303 --read-seed $k --fragments $((8000 + 4000 * k)) --normal-mult 0.4 --contigs 5 --contig-len 400000 --junctions 200 --dup 0.1 --indels 1.0 --non-template 0.5 > /dev/null 2>&1; done
ARRIBA_FEED_PIECE_MB=1 ARRIBA_WRITER_THREADS=8 ARRIBA_INGEST_THREADS=4 ./queue_tsan d.gtf d.fa $WORK/queued d.bam q1.bam q2.bam q3.bam > queue_log 2>&1 || echo "queue_driver failed: $(tail -2 queue_log)"
SAME=0; for k in 1 2 3; do ARRIBA_FEED_PIECE_MB=1 ./workflow_tsan -x q$k.bam -g d.gtf -a d.fa -o alone$k.tsv -O alone$k.discarded.tsv -X -f blacklist > /dev/null 2>&1 || true; cmp -s alone$k.tsv queued$k.tsv && cmp -s alone$k.discarded.tsv queued$k.discarded.tsv && SAME=$((SAME + 1)); done
cmp -s f.tsv queued0.tsv && cmp -s disc.tsv queued0.discarded.tsv && SAME=$((SAME + 1))
echo "queue of 4 samples: ThreadSanitizer warnings: $(grep -c 'WARNING: ThreadSanitizer' queue_log || true); $SAME of 4 samples wrote the files of the sample alone"
# ... and the same queue with arriba_workflow_finish_ahead (round 5): the feeder of sample k + 1 also finishes its ingest and hands it to the host session while the stages of sample k run
QUEUE_FINISH_AHEAD=1 ARRIBA_FEED_PIECE_MB=1 ARRIBA_WRITER_THREADS=8 ARRIBA_INGEST_THREADS=4 ./queue_tsan d.gtf d.fa $WORK/ahead d.bam q1.bam q2.bam q3.bam > ahead_log 2>&1 || echo "queue_driver (finish ahead) failed: $(tail -2 ahead_log)"
SAME=0; for k in 1 2 3; do cmp -s alone$k.tsv ahead$k.tsv && cmp -s alone$k.discarded.tsv ahead$k.discarded.tsv && SAME=$((SAME + 1)); done
cmp -s f.tsv ahead0.tsv && cmp -s disc.tsv ahead0.discarded.tsv && SAME=$((SAME + 1))
echo "queue of 4 samples, ingest finished ahead: ThreadSanitizer warnings: $(grep -c 'WARNING: ThreadSanitizer' ahead_log || true); $SAME of 4 samples wrote the files of the sample alone"
rm -rf $WORK
