"""Multi-process tier (gloo, CPU): the sharded pipeline of arriba_amd/sharded.py -- per-shard stages + the four exchanges -- must give
exactly the result of the single-process pipeline over the whole sample.  The device stages are stepped on the host (tests/emu)."""
import json
import os
import subprocess
import sys

import pytest

import conftest
import datasets

ROOT = conftest.ROOT


def run_sharded(prefix, world, api, out_path, port):
    command = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port", str(port),
               os.path.join(ROOT, "tests", "sharded_worker.py"), prefix, api, out_path]
    result = subprocess.run(command, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, timeout=900)
    assert result.returncode == 0, result.stdout[-3000:]
    return [json.load(open("%s.rank%d.json" % (out_path, rank))) for rank in range(world)]


@pytest.mark.parametrize("world,name", [(2, "toy3k"), (3, "mid30k")])
def test_sharded_pipeline_equals_single_process(world, name, dataset_files, emu_api, tmp_path):
    reports = run_sharded(dataset_files(name), world, "emu", str(tmp_path / "report"), 29600 + world)
    assert reports[0]["problems"] == [], reports[0]["problems"]
    assert sum(r["count"] for r in reports) == reports[0]["fragments"]
    assert sum(r["owned_candidates"] for r in reports) == reports[0]["candidates"]
    assert all(r["exchange"]["emissions_sent"] > 0 for r in reports)
    assert reports[0]["multimappers"][1] > 0  # fragments discarded by filter_multimappers across the shards
