"""CPU tier: pin the oracle restatement (oracle/restate.cpp) against the golden dumps of the real reference.
The oracle is what the GPU tier compares the HIP kernels with on inputs that have no committed dump."""
import numpy as np
import pytest

import conftest
import golden_io
import oracle_lib
import parity


@pytest.mark.parametrize("name", ["toy3k", "shuffled2k", "mid30k", "stacked4k"])
def test_oracle_restatement_matches_reference(name, dataset_files):
    golden = conftest.golden_dir(name)
    session = parity.open_session(dataset_files(name))
    oracle = oracle_lib.OraclePipeline(session)
    oracle.run_read_level()
    scalars = golden_io.read_scalars(golden + "/scalars.tsv")
    assert oracle.scalars["marked_multimappers"] == int(scalars["marked_multimappers"])
    parity.check_read_filters(session, oracle, golden)
    genes = golden_io.read_genes(golden + "/genes.tsv")
    table = oracle.gene_table()
    assert len(genes) == len(table["start"])
    for g in genes:
        assert (table["contig"][g["id"]], table["start"][g["id"]], table["end"][g["id"]], table["bits"][g["id"]] & 1, (table["bits"][g["id"]] >> 1) & 1) == (g["contig"], g["start"], g["end"], g["strand"], g["is_dummy"])
    import datasets
    assert oracle.remaining == datasets.parse_remaining(open(golden + "/reference.log").read())
    if name != "mid30k":
        parity.check_annotation(session, oracle, golden)
    if name == "toy3k":
        oracle.find_fusions(int(scalars["max_mate_gap"]))
        assert parity.check_candidates(session, oracle, golden) > 1000
