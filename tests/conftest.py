import ctypes
import hashlib
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import datasets  # noqa: E402


@pytest.fixture(autouse=True)
def scratch_files_of_a_test_are_freed(tmp_path):
    """The samples of the tests at scale are 5-11 GB each and pytest keeps the directories of a session to its end: on the GPU box the generator of the next sample then
    fails for lack of room now and then (profiles/r06h_pair_3.log: gen_synth exit 1 behind the hg38-size reference data).  What a test wrote goes when it ends."""
    yield
    import shutil
    shutil.rmtree(str(tmp_path), ignore_errors=True)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (run on the GPU box through gpurun)")


def _ensure_built():
    lib = os.path.join(ROOT, "arriba_amd", "lib")
    needed = [os.path.join(lib, f) for f in ("libarriba_host.so", "libarriba_gpu.so", "libarriba_workflow.so", "arriba_gpu_workflow", "gen_synth")]
    if not all(os.path.exists(p) for p in needed):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def built():
    _ensure_built()
    return True


@pytest.fixture(scope="session")
def emu_api(built):
    """The TEST-ONLY host stepping harness for the device code (tests/emu); never used by the product."""
    directory = os.path.join(ROOT, "tests", "emu")
    subprocess.run(["make", "-s", "-C", directory], check=True)
    from arriba_amd import _capi
    return _capi.bind_device_api(ctypes.CDLL(os.path.join(directory, "libemu.so")), "emu_")


@pytest.fixture(scope="session")
def dataset_files(built, tmp_path_factory):
    """name -> prefix of generated FASTA/GTF/BAM; generated once per session and checked against the golden checksum."""
    cache = {}

    def get(name):
        if name not in cache:
            directory = tmp_path_factory.mktemp("data_" + name)
            prefix = datasets.generate(datasets.DATASETS[name], str(directory))
            meta_path = os.path.join(ROOT, "tests", "golden", name, "meta.json")
            if os.path.exists(meta_path):
                meta = json.load(open(meta_path))
                digest = hashlib.sha256(open(prefix + ".bam", "rb").read()).hexdigest()
                assert digest == meta["bam_sha256"], "generator drifted from the golden fixtures of %s; rerun tools/make_golden.py" % name
            cache[name] = prefix
        return cache[name]
    return get


def golden_dir(name):
    return os.path.join(ROOT, "tests", "golden", name)
