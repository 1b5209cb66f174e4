#!/usr/bin/env python3
"""tools/diff_discarded.py -- which rows of a large discarded.tsv differ from the reference's?  The reference's file of a 20 M-fragment sample (1.5 GB) exists only where the
repository is built, the device only on the GPU box, and only 64 MB come back from there: so the rows travel as 64-bit hashes.
    hash FILE OUT.bin                      sorted hashes of the rows of FILE (here, over the reference's file)
    compare FRAGMENTS REF.bin OUT_PREFIX   (GPU box) generates bench.py's sample, runs two samples through a resident session with -O, and writes the rows of each discarded.tsv the reference
                                           does not have (OUT_PREFIX.sampleK.only_mine.tsv) and the hashes of the reference's rows it lacks (OUT_PREFIX.sampleK.only_reference.bin)
    lookup FILE HASHES.bin                 (here) the rows of FILE with these hashes
Test tooling."""
import hashlib
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def row_hashes(path):
    rows = open(path, "rb").read().split(b"\n")
    return rows, np.array([int.from_bytes(hashlib.blake2b(row, digest_size=8).digest(), "little") for row in rows], dtype=np.uint64)


def main():
    mode = sys.argv[1]
    if mode == "hash":
        _, hashes = row_hashes(sys.argv[2])
        np.unique(hashes).tofile(sys.argv[3])
    elif mode == "lookup":
        rows, hashes = row_hashes(sys.argv[2])
        wanted = np.fromfile(sys.argv[3], dtype=np.uint64)
        for k in np.flatnonzero(np.isin(hashes, wanted)):
            print(rows[k].decode(errors="replace"))
    elif mode == "compare":
        import bench
        import datasets
        from arriba_amd.pipeline import WorkflowSession
        fragments, reference, out = int(sys.argv[2]), np.fromfile(sys.argv[3], dtype=np.uint64), sys.argv[4]
        prefix = "/tmp/diff_discarded/s"
        os.makedirs(os.path.dirname(prefix), exist_ok=True)
        subprocess.run([datasets.GEN_SYNTH, "--out", prefix, "--threads", str(bench.cpu_budget())] + bench.workload_args(fragments, 1000), check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        session = WorkflowSession(prefix + ".fa", prefix + ".gtf")
        session.submit(prefix + ".bam")
        for k in range(2):
            if k == 0:
                session.submit(prefix + ".bam")
            session.sample(prefix + ".bam", prefix + ".fusions%d.tsv" % k, prefix + ".discarded%d.tsv" % k)
            rows, hashes = row_hashes(prefix + ".discarded%d.tsv" % k)
            only_mine = np.flatnonzero(~np.isin(hashes, reference))
            only_reference = reference[~np.isin(reference, hashes)]
            print("sample %d: %d rows, %d only here, %d only in the reference's file" % (k, len(rows), only_mine.size, only_reference.size))
            with open("%s.sample%d.only_mine.tsv" % (out, k), "wb") as stream:
                for index in only_mine[:2000]:
                    stream.write(rows[index] + b"\n")
            only_reference[:100000].tofile("%s.sample%d.only_reference.bin" % (out, k))
            os.remove(prefix + ".discarded%d.tsv" % k)
        session.close()


if __name__ == "__main__":
    main()
