#!/usr/bin/env python3
"""tools/pmc_summary.py -- per-kernel sums of rocprofv3 --pmc counters (from *_counter_collection.csv files) as JSON:
{kernel: {"dispatches": n, counter: total, ...}}.  Usage: pmc_summary.py out.json dir_or_csv [dir_or_csv ...]"""
import csv
import glob
import json
import os
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"\(.*$", "", name)
    name = re.sub(r"<.*$", "", name)
    return name[:60]


def main():
    out_path, sources = sys.argv[1], sys.argv[2:]
    files = []
    for source in sources:
        if os.path.isdir(source):
            files += glob.glob(os.path.join(source, "**", "*counter_collection.csv"), recursive=True)
        else:
            files.append(source)
    kernels = {}
    for path in files:
        with open(path, newline="") as handle:
            reader = csv.DictReader(handle)
            seen = set()
            for row in reader:
                kernel = short(row.get("Kernel_Name") or row.get("kernel_name") or "?")
                counter = row.get("Counter_Name") or row.get("counter_name")
                value = float(row.get("Counter_Value") or row.get("counter_value") or 0)
                dispatch = row.get("Dispatch_Id") or row.get("dispatch_id")
                entry = kernels.setdefault(kernel, {"dispatches": {}})
                entry[counter] = entry.get(counter, 0.0) + value
                entry["dispatches"].setdefault(counter, set()).add(dispatch)
    for kernel, entry in kernels.items():
        entry["dispatches"] = max(len(v) for v in entry["dispatches"].values())
    json.dump(kernels, open(out_path, "w"), indent=1, sort_keys=True)
    for kernel, entry in sorted(kernels.items(), key=lambda item: -sum(v for k, v in item[1].items() if k != "dispatches"))[:25]:
        print(kernel, entry)


if __name__ == "__main__":
    main()
