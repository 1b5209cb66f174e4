"""the last seconds of GPU time of round 2: the task list of the second mismapper pass with the real kernel -- step budget 64, so nearly every read goes through it --
on the golden dataset homologs8k: stage counts and both output files must equal the reference's"""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["ARRIBA_FIRST_PASS_STEPS"] = "64"
started = time.time()
import datasets, parity
directory = tempfile.mkdtemp(prefix="r02p_")
prefix = datasets.generate(datasets.DATASETS["homologs8k"], directory)
os.makedirs(os.path.join(directory, "mine"))
stages = parity.check_workflow(prefix, os.path.join(ROOT, "tests", "golden", "homologs8k"), os.path.join(directory, "mine"))
print("task list on the GPU: homologs8k identical to the reference,", dict(stages)["filter_mismappers"], "candidates behind filter_mismappers,", round(time.time() - started, 1), "s", flush=True)
