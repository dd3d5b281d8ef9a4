#!/usr/bin/env python3
"""Runs on the GPU box after a `rocprofv3 --pmc ...` pass: folds the pass's counter_collection CSVs (too large to copy
back) into {dispatch id: {counter: value}} for the engine kernel, the form tools/pmc_summary.py accepts as a pass.

  python3 tools/pmc_reduce.py gpurun_out/pmc_fetch gpurun_out/pmc_fetch.json && rm -rf gpurun_out/pmc_fetch"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    src, dst = sys.argv[1], sys.argv[2]
    kernel = sys.argv[3] if len(sys.argv) > 3 else "fcu_ctu_engine"
    per = defaultdict(lambda: defaultdict(float))
    for f in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if kernel in r["Kernel_Name"]:
                    per[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    with open(dst, "w") as fh:
        json.dump({str(k): dict(v) for k, v in sorted(per.items())}, fh)
    print(dst, {k: len(v) for k, v in per.items()})


if __name__ == "__main__":
    main()
