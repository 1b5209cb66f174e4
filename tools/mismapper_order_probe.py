#!/usr/bin/env python3
"""tools/mismapper_order_probe.py -- does the verdict of filter_mismappers depend on how the search is scheduled?  It must not: the verdict of a read is a function of the read.
Runs make_kmer_index + filter_mismappers of a golden dataset (state right before the stage taken from the reference's dump, as parity.check_mismappers does) under the knobs of
agpu_mismappers.hip -- order of the jobs, number of persistent workgroups, sweep / task list on and off -- and prints how many reads every variant judges differently from the
reference's dump.  Needs a GPU.  usage: python tools/mismapper_order_probe.py [dataset ...]"""
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import datasets  # noqa: E402
import golden_io  # noqa: E402
import parity  # noqa: E402

VARIANTS = [("default", {}), ("order=candidate", {"ARRIBA_MISMAPPER_JOB_ORDER": "candidate"}), ("workgroups=1", {"ARRIBA_HEAVY_WORKGROUPS": "1"}), ("workgroups=7", {"ARRIBA_HEAVY_WORKGROUPS": "7"}),
            ("order=candidate workgroups=1", {"ARRIBA_MISMAPPER_JOB_ORDER": "candidate", "ARRIBA_HEAVY_WORKGROUPS": "1"}),
            ("order=candidate sweep=0", {"ARRIBA_MISMAPPER_JOB_ORDER": "candidate", "ARRIBA_MISMAPPER_SWEEP": "0"}),
            ("order=candidate worklist=0", {"ARRIBA_MISMAPPER_JOB_ORDER": "candidate", "ARRIBA_MISMAPPER_WORKLIST": "0"}),
            ("order=candidate waves=4", {"ARRIBA_MISMAPPER_JOB_ORDER": "candidate", "ARRIBA_HEAVY_WAVES": "4"}),
            ("order=candidate first_pass", {"ARRIBA_MISMAPPER_JOB_ORDER": "candidate", "ARRIBA_MISMAPPER_FIRST_PASS": "1"})]
KNOBS = sorted(set(key for _, environment in VARIANTS for key in environment))


def main():
    names = sys.argv[1:] or ["toy3k", "homologs8k"]
    for name in names:
        directory = tempfile.mkdtemp(prefix="order_probe_")
        prefix = datasets.generate(datasets.DATASETS[name], directory)
        golden = os.path.join(ROOT, "tests", "golden", name)
        before = golden_io.read_fusions(golden_io.find_dump(golden, "fusions", "before_filter_mismappers"))
        _, filters_before = golden_io.read_filters(golden_io.find_dump(golden, "filters", "before_filter_mismappers"))
        _, filters_after = golden_io.read_filters(golden_io.find_dump(golden, "filters", "filter_mismappers"))
        scalars = golden_io.read_scalars(os.path.join(golden, "scalars.tsv"))
        session, pipeline = parity.run_read_level(parity.open_session, prefix)
        pipeline.find_fusions()
        table = pipeline.candidates()
        n = pipeline.n_candidates
        index = {key: c for c, key in enumerate(parity.candidate_keys(table, n))}
        state = {k: np.zeros(n, dtype=np.uint32) for k in ("filter", "split_reads1", "split_reads2", "discordant_mates")}
        for f in before:
            for k in state:
                state[k][index[parity.fusion_key(f)]] = f[k]
        expected = np.array(filters_after, dtype=np.uint8)
        for label, environment in VARIANTS:
            for key in KNOBS:
                os.environ.pop(key, None)
            os.environ.update(environment)
            pipeline.set_candidate_state(state["filter"].astype(np.uint8), state["split_reads1"], state["split_reads2"], state["discordant_mates"])
            pipeline.set_read_filters(np.array(filters_before, dtype=np.uint8))
            pipeline.make_kmer_index(int(scalars["kmer_index_padding"]))
            remaining, discarded = pipeline.filter_mismappers(int(scalars["max_mate_gap"]))
            mine = pipeline.filters()
            different = np.flatnonzero(mine != expected)
            print(json.dumps({"dataset": name, "variant": label, "reads_judged_differently": int(different.size), "first": [(int(i), int(mine[i]), int(expected[i])) for i in different[:6]], "discarded": int(discarded), "remaining": int(remaining)}))
            sys.stdout.flush()


if __name__ == "__main__":
    main()
