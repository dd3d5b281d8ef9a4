#!/usr/bin/env python3
"""Folds rocprofv3 counter-collection passes of `bench.py` into profiles/<round>_pmc_summary.json.

Each pass is its own run of the same command with ONE --pmc set (the guide: FETCH_SIZE and WRITE_SIZE do not fit one
pass; never together with a trace flag), e.g.

  rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_fetch --output-format csv -- python3 bench.py --no-cpu-baseline --no-sweep
  rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_write --output-format csv -- python3 bench.py --no-cpu-baseline --no-sweep
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAVES -d gpurun_out/pmc_sq --output-format csv -- python3 bench.py ...
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d gpurun_out/pmc_tcc --output-format csv -- python3 bench.py ...

  python tools/pmc_summary.py --config intra --chains 8160 --ctus 30 --timed 3 --out profiles/r02_pmc_summary.json \
         gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_sq gpurun_out/pmc_tcc

A pass may also be given as a JSON file already reduced on the GPU box ({dispatch id: {counter: value}} of the engine
kernel; the raw CSVs of a full bench run exceed what gpurun copies back).  Per counter: the mean over the last `--timed` dispatches of the engine kernel (the timed steps of bench.py).  FETCH_SIZE
and WRITE_SIZE are in KiB; FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950."""
import argparse
import csv
import glob
import json
import os
import subprocess
from collections import defaultdict


def read_pass(d, kernel):
    per = defaultdict(lambda: defaultdict(float))          # dispatch -> counter -> value
    if d.endswith(".json"):                                # a pass already reduced on the GPU box: {dispatch: {counter: value}} of the engine kernel
        with open(d) as fh:
            for k, v in json.load(fh).items():
                for n, x in v.items():
                    per[int(k)][n] += float(x)
        return per
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if kernel in r["Kernel_Name"]:
                    per[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    return per


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dirs", nargs="+")
    ap.add_argument("--kernel", default="fcu_ctu_engine")
    ap.add_argument("--config", default="intra")
    ap.add_argument("--chains", type=int, required=True)
    ap.add_argument("--ctus", type=int, required=True)
    ap.add_argument("--timed", type=int, default=3)
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    counters, launches = {}, {}
    for d in a.dirs:
        per = read_pass(d, a.kernel)
        ids = sorted(per)[-a.timed:]
        names = set().union(*[set(per[i]) for i in ids]) if ids else set()
        for n in names:
            counters[n] = sum(per[i][n] for i in ids) / len(ids)
            launches[n] = len(ids)
    out = {"config": a.config, "kernel": a.kernel, "chains_per_launch": a.chains, "ctus_per_chain_per_launch": a.ctus,
           "launches_averaged": launches, "counters_per_launch": counters}
    if "FETCH_SIZE" in counters and "WRITE_SIZE" in counters:
        out["fetch_bytes_per_launch_corrected"] = counters["FETCH_SIZE"] * 1024 * 2
        out["write_bytes_per_launch"] = counters["WRITE_SIZE"] * 1024
        out["traffic_bytes_per_launch"] = out["fetch_bytes_per_launch_corrected"] + out["write_bytes_per_launch"]
        out["traffic_bytes_per_ctu"] = out["traffic_bytes_per_launch"] / (a.chains * a.ctus)
    if "SQ_WAIT_ANY" in counters and "SQ_WAVE_CYCLES" in counters:
        out["wait_share_of_wave_cycles"] = counters["SQ_WAIT_ANY"] / counters["SQ_WAVE_CYCLES"]
    if "TCC_HIT_sum" in counters and "TCC_MISS_sum" in counters:
        out["l2_hit_rate"] = counters["TCC_HIT_sum"] / (counters["TCC_HIT_sum"] + counters["TCC_MISS_sum"])
    try:
        out["commit"] = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], cwd=os.path.dirname(os.path.abspath(__file__))).decode().strip()
    except Exception:
        out["commit"] = None
    with open(a.out, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
