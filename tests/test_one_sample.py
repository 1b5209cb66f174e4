"""Multi-process tier (gloo, CPU): one sample over several ranks (arriba_amd/one_sample.py) -- every rank ingests its part of the records of the file, one
all-gather puts the batch together, filter_mismappers is shared out -- must give exactly the batch, the stage counts and the output files of the
single-process pipeline over the whole file.  The device stages are stepped on the host (tests/emu)."""
import json
import os
import subprocess
import sys

import pytest

import conftest
import datasets

ROOT = conftest.ROOT


def run_one_sample(prefix, world, api, out_path, port, bam=None):
    command = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port", str(port),
               os.path.join(ROOT, "tests", "one_sample_worker.py"), prefix, api, out_path, bam or prefix + ".bam"]
    result = subprocess.run(command, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, timeout=1500)
    assert result.returncode == 0, result.stdout[-3000:]
    return [json.load(open("%s.rank%d.json" % (out_path, rank))) for rank in range(world)]


def check_reports(reports):
    for report in reports:
        assert "error" not in report, report["error"]
    assert reports[0]["problems"] == [], reports[0]["problems"]
    for report in reports[1:]:  # every rank holds the same batch and reaches the same verdicts
        for key in ("fragments", "records", "mapped_reads", "batch", "log", "filters", "mismapper_jobs"):
            assert report[key] == reports[0][key], key
    return reports[0]


@pytest.mark.parametrize("world,name", [(2, "toy3k"), (3, "homologs8k"), (4, "mid30k"), (3, "scrambled3k")])
def test_one_sample_over_ranks_equals_single_process(world, name, dataset_files, emu_api, tmp_path):
    report = check_reports(run_one_sample(dataset_files(name), world, "emu", str(tmp_path / "report"), 29700 + world + (10 if name == "scrambled3k" else 0)))
    assert sum(1 for size in report["part_bytes"] if size > 4096) == world  # every rank read a part of the file
    assert report["mismapper_jobs"] > 0 and report["fusions"] > 0


@pytest.mark.parametrize("flags", [[], ["--per-rank-samples"]])
def test_bench_with_two_ranks_prints_one_line(flags, built, emu_api, tmp_path):
    """bench.py --gpus 2 as the driver launches it (torch.distributed.run, one rank per GPU): rank 0 prints ONE JSON line with the contract's fields; by default the
    ranks work on one sample (scaling: strong), with --per-rank-samples on a sample each (weak).  Run on the stepping harness with torch.cuda stubbed
    (tests/bench_on_harness.py): only the control flow is under test."""
    command = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(29741 + len(flags)),
               os.path.join(ROOT, "tests", "bench_on_harness.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--fragments", "20000"] + flags
    result = subprocess.run(command, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=1500)
    assert result.returncode == 0, result.stderr[-3000:]
    lines = [line for line in result.stdout.splitlines() if line.startswith("{")]
    assert len(lines) == 1, result.stdout[-2000:]
    line = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["n_gpus"] == 2 and line["steps"] == 1 and line["value"] > 0 and line["vs_baseline"] is None
    assert line["scaling"] == ("weak" if flags else "strong")
    assert ("one sample over 2 GPUs" in line["config"]["parallelism"]) == (not flags)
    assert set(line["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}


def test_bench_launches_its_own_ranks_and_refuses_a_wrong_world_size(built, emu_api, tmp_path):
    """`python bench.py --gpus 2` without a launcher (the way the driver calls it at N = 1) starts the two ranks itself and prints the line of two GPUs; under a launcher
    whose WORLD_SIZE is not --gpus it refuses instead of reporting GPUs that did not take part."""
    harness = os.path.join(ROOT, "tests", "bench_on_harness.py")
    environment = {key: value for key, value in os.environ.items() if key not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    result = subprocess.run([sys.executable, harness, "--gpus", "2", "--steps", "1", "--warmup", "0", "--fragments", "20000"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=1500, env=environment)
    assert result.returncode == 0, result.stderr[-3000:]
    lines = [line for line in result.stdout.splitlines() if line.startswith("{")]
    assert len(lines) == 1, result.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and "one sample over 2 GPUs" in line["config"]["parallelism"]
    result = subprocess.run([sys.executable, harness, "--gpus", "4", "--steps", "1", "--warmup", "0", "--fragments", "20000"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=600,
                            env=dict(environment, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert result.returncode != 0 and "WORLD_SIZE=1" in result.stderr and not [line for line in result.stdout.splitlines() if line.startswith("{")]


@pytest.mark.parametrize("flags", [[], ["--python-stages"]])
def test_bench_with_one_rank_times_the_workflow_library(flags, built, emu_api, tmp_path):
    """bench.py as the driver runs it at N = 1: the timed call is arriba_workflow_sample of the product library (here its build over the stepping harness,
    tests/emu/libworkflow_on_harness.so), with --python-stages the ctypes mirror of the stage order; both end with the same counts and one line of the contract."""
    command = [sys.executable, os.path.join(ROOT, "tests", "bench_on_harness.py"), "--steps", "2", "--warmup", "1", "--fragments", "20000", "--no-cpu-baseline"] + flags
    result = subprocess.run(command, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=1500)
    assert result.returncode == 0, result.stderr[-3000:]
    lines = [line for line in result.stdout.splitlines() if line.startswith("{")]
    assert len(lines) == 1, result.stdout[-2000:]
    line = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["n_gpus"] == 1 and line["steps"] == 2 and line["value"] > 0
    assert ("arriba_workflow_sample" in line["config"]["timed_call"]) == (not flags)
    assert line["config"]["fragments_per_gpu"] == 20886 and line["config"]["candidates"] == 10278 and line["config"]["fusions"] == 343  # (the same sample through both)
    assert [stage for stage, _, _ in line["stages"]][-1] == "recover_isoforms"
    if not flags:  # the leg with 4 N ordinary proper pairs beside the chimeric fragments: five times the records, more chimeric fragments (read-through pairs among the ordinary ones)
        leg = line["value_with_normal_pairs"]
        assert leg["chimeric_reads_per_s"] > 0 and leg["bam_records_per_s"] > 4 * leg["chimeric_reads_per_s"] and "4 N ordinary proper pairs" in leg["what"]


def test_an_error_on_one_rank_ends_the_run_on_all(dataset_files, emu_api, tmp_path):
    """A damaged BGZF block in the last third of the file: the rank that reads it fails with the reference's message, the others are told before they enter the
    all-gather of the parts -- nobody is left waiting in a collective."""
    import test_host_and_device_logic as host_tests
    prefix = dataset_files("mid30k")
    damaged = str(tmp_path / "damaged.bam")
    host_tests._write_bgzf(damaged, host_tests._bam_payload(prefix + ".bam"), 6)
    raw = bytearray(open(damaged, "rb").read())
    raw[len(raw) * 5 // 6] ^= 0x5A
    open(damaged, "wb").write(bytes(raw))
    reports = run_one_sample(prefix, 3, "emu", str(tmp_path / "report"), 29731, bam=damaged)
    assert "failed to load alignments" in reports[2]["error"], reports[2]  # (the rank in front of it may meet the block, too, when it looks for the end of its part)
    assert "another rank of the sample failed" in reports[0]["error"], reports[0]
    assert all("error" in report for report in reports)


def run_workflow_over_ranks(mode, prefix, world, out, port, bams=None, plain=False, environment=None):
    """tests/workflow_ranks_worker.py under torch.distributed.run: arriba_workflow_sample of the C++ driver as a collective call of `world` ranks"""
    os.makedirs(out, exist_ok=True)
    command = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port", str(port),
               os.path.join(ROOT, "tests", "workflow_ranks_worker.py"), mode, prefix + ".fa", prefix + ".gtf", out] + (bams or [prefix + ".bam"])
    result = subprocess.run(command, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, timeout=1500,
                            env=dict(os.environ, **(dict({"WORKFLOW_RANKS_PLAIN": "1"} if plain else {}, **(environment or {})))))
    assert result.returncode == 0, result.stdout[-3000:]
    reports = [json.load(open(os.path.join(out, "rank%d.json" % rank))) for rank in range(world)]
    for report in reports:
        assert "error" not in report, report["error"]
    return reports


def check_workflow_over_ranks(mode, prefix, world, tmp_path, port, samples=1, split="sharded"):
    """split: what the ranks are expected to do with the reads -- "sharded": every rank keeps the fragments of its part (the default whenever the parts follow each other in name order);
    "replicated": one all-gather of the batch, the stages on every rank (names in another order; or asked for with ARRIBA_RANKS_SPLIT=replicated)"""
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu"), "libworkflow_on_harness.so"], check=True)
    bams = [prefix + ".bam"] * samples
    alone = run_workflow_over_ranks(mode, prefix, 1, str(tmp_path / "alone"), port, bams[:1], plain=True)[0]
    reports = run_workflow_over_ranks(mode, prefix, world, str(tmp_path / "ranks"), port + 1, bams, environment={"ARRIBA_RANKS_SPLIT": "replicated"} if split == "forced_replicated" else None)
    read = lambda directory, name: open(str(tmp_path / directory / name), "rb").read()
    assert len(read("alone", "sample0.tsv").splitlines()) > 3 and len(read("alone", "sample0.discarded.tsv").splitlines()) > 100
    for report in reports:
        assert len(report["samples"]) == samples
        for sample in report["samples"]:
            assert "error" not in sample, sample["error"]
            assert sample["report"] == alone["samples"][0]["report"]  # every "(remaining=N)" of the log, on every rank
            assert sample["exchange_parts"] > 0
    for k in range(samples):  # rank 0 wrote the files: byte for byte those of one rank
        assert read("ranks", "sample%d.tsv" % k) == read("alone", "sample0.tsv")
        assert read("ranks", "sample%d.discarded.tsv" % k) == read("alone", "sample0.discarded.tsv")
        held = [report["samples"][k]["shard_fragments"] for report in reports]
        if split == "sharded":  # every rank held its share of the fragments through the whole sample, and together they held each once
            assert sum(held) == reports[0]["samples"][k]["fragments"] and max(held) < 0.75 * sum(held) * (2.0 / world if world > 2 else 1.0), held
        else:
            assert held == [0] * world, held
    return reports


@pytest.mark.parametrize("world,name,samples", [(2, "toy3k", 3), (3, "homologs8k", 1), (4, "mid30k", 1), (8, "itd6k", 1), (3, "scrambled3k", 2)])
def test_workflow_library_over_ranks_writes_the_files_of_one_rank(world, name, samples, dataset_files, built, emu_api, tmp_path):
    """The C++ driver as a collective call (include/arriba_workflow.h: arriba_workflow_set_communicator; the collectives are torch.distributed's, gloo), THE READS SHARDED: every rank
    feeds and ingests its part of the records and keeps its fragments through the whole sample -- the read-level cascade on the shard with its small exchanges (dummy genes, duplicate
    keys, mate gaps, strandedness votes, coverage), ONE all-gather of the emissions of find_fusions, the candidates and their read lists built on every rank, one byte of state per
    fragment replicated whenever the filters have changed, the stages that need alignments (scores of filter_multimappers, filter_mismappers, the clipped mates of filter_in_vitro)
    where the reads are, the rows of the written candidates gathered from their ranks -- fusions.tsv and discarded.tsv of rank 0 and the counts of every stage on every rank equal
    those of one rank without a communicator; with several samples in a queue (the part of the next file fed beside the stages of the current one).  scrambled3k: the names of the
    file are not in order, so the parts do not follow each other: the ranks notice and put the batch together on every rank instead (the split of rounds 2-5)."""
    check_workflow_over_ranks("harness", dataset_files(name), world, tmp_path, 29760 + 4 * world + (2 if name == "scrambled3k" else 0), samples, split="replicated" if name == "scrambled3k" else "sharded")


@pytest.mark.parametrize("world,name", [(8, "toy3k"), (8, "homologs8k"), (8, "mid30k"), (2, "itd6k"), (3, "rules8k"), (3, "stacked4k")])
def test_read_sharded_sample_over_more_ranks_and_datasets(world, name, dataset_files, built, emu_api, tmp_path):
    """... the same at 8 ranks (parts of a few hundred fragments: ranks without a candidate's reads, without multi-mappers, without clipped mates) and on the datasets whose
    files exercise the blacklist / known fusions (rules8k), sets of nine genes per alignment (stacked4k) and internal tandem duplications (itd6k)"""
    check_workflow_over_ranks("harness", dataset_files(name), world, tmp_path, 29900 + 8 * world + {"toy3k": 0, "homologs8k": 2, "mid30k": 4, "itd6k": 0, "rules8k": 2, "stacked4k": 4}[name])


def test_the_split_of_the_batch_can_still_be_asked_for(dataset_files, built, emu_api, tmp_path):
    """ARRIBA_RANKS_SPLIT=replicated: one all-gather of the batch and the stages on every rank, for comparisons with the read-sharded split"""
    check_workflow_over_ranks("harness", dataset_files("toy3k"), 2, tmp_path, 29990, split="forced_replicated")


def test_a_failure_on_one_rank_ends_the_sample_on_all_ranks(dataset_files, built, emu_api, tmp_path):
    """rank 1 cannot read its part of the file (a path that does not exist there): its message on rank 1, "another rank of the sample failed" on rank 0, nobody left waiting in a collective"""
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu"), "libworkflow_on_harness.so"], check=True)
    prefix = dataset_files("toy3k")
    reports = run_workflow_over_ranks("harness", prefix, 2, str(tmp_path / "ranks"), 29795, [prefix + ".bam"], environment={"WORKFLOW_RANKS_BREAK_RANK": "1"})
    assert "another rank of the sample failed" in reports[0]["samples"][0]["error"]
    assert "error" in reports[1]["samples"][0] and "another rank" not in reports[1]["samples"][0]["error"]


def test_workflow_library_over_ranks_reads_deflated_parts_and_refuses_what_one_rank_refuses(dataset_files, built, emu_api, tmp_path):
    """The parts of a DEFLATED file (small blocks that straddle the cuts) through the C++ driver over three ranks: the files of one rank on the stored file.  And a file no rank may
    cut -- plain gzip, not seekable by blocks -- is refused on every rank with the message of the part reader, nobody left in a collective."""
    import gzip
    import test_host_and_device_logic as host_tests
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu"), "libworkflow_on_harness.so"], check=True)
    prefix = dataset_files("toy3k")
    payload = host_tests._bam_payload(prefix + ".bam")
    deflated, plain = str(tmp_path / "deflated.bam"), str(tmp_path / "plain.bam")
    host_tests._write_bgzf(deflated, payload, 6, block=4093)
    with gzip.open(plain, "wb") as out:
        out.write(payload)
    alone = run_workflow_over_ranks("harness", prefix, 1, str(tmp_path / "alone"), 29801, [prefix + ".bam"], plain=True)[0]
    reports = run_workflow_over_ranks("harness", prefix, 3, str(tmp_path / "ranks"), 29802, [deflated, plain, deflated])
    read = lambda directory, name: open(str(tmp_path / directory / name), "rb").read()
    for report in reports:
        first, refused, third = report["samples"]
        assert "error" not in first and first["report"] == alone["samples"][0]["report"]
        assert "error" in refused and "another rank" not in refused["error"], refused  # (every rank hears it from its own part reader)
        assert "error" not in third and third["report"] == alone["samples"][0]["report"]  # (the session goes on behind a refused sample)
    for k in (0, 2):
        assert read("ranks", "sample%d.tsv" % k) == read("alone", "sample0.tsv") and read("ranks", "sample%d.discarded.tsv" % k) == read("alone", "sample0.discarded.tsv")
