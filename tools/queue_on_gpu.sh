#!/bin/bash
# tools/queue_on_gpu.sh -- four samples through one resident session in a queue on the device, last files deferred (tests/emu/queue_driver.cpp linked against the real libraries:
# arriba_amd/lib/queue_driver), against the command line driver run on each sample alone; no Python: 20 s of a GPU box.  Test tooling.
set -e
W=/tmp/qg; mkdir -p $W; L=arriba_amd/lib
for k in 0 1 2 3; do $L/gen_synth --out $W/q$k --seed 303 --read-seed $((k+1)) --fragments $((200000 + 100000 * k)) --normal-mult 0.4 --contigs 5 --contig-len 400000 --junctions 200 --dup 0.1 --indels 1.0 --non-template 0.5 > /dev/null 2>&1; done
$L/queue_driver $W/q0.gtf $W/q0.fa $W/queued $W/q0.bam $W/q1.bam $W/q2.bam $W/q3.bam > $W/log 2>&1 || { tail -3 $W/log; exit 1; }
SAME=0
for k in 0 1 2 3; do $L/arriba_gpu_workflow -x $W/q$k.bam -g $W/q0.gtf -a $W/q0.fa -o $W/alone$k.tsv -O $W/alone$k.discarded.tsv -X -f blacklist > /dev/null 2>&1; cmp -s $W/alone$k.tsv $W/queued$k.tsv && cmp -s $W/alone$k.discarded.tsv $W/queued$k.discarded.tsv && SAME=$((SAME+1)); done
echo "queue on the GPU: $SAME of 4 samples wrote the files of the sample alone; $(wc -l < $W/queued3.tsv) rows in the last"
