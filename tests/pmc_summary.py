"""Reduce rocprofv3 --pmc counter_collection CSVs (gpurun_out/pmc*/) to one small JSON:
per-counter sums over the fcu_ctu_engine dispatches.  Run on the GPU box after the PMC passes."""
import csv, glob, json, os, sys, collections
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
res = {}
for f in sorted(glob.glob(os.path.join(root, "pmc*", "*counter_collection.csv"))):
    acc = collections.defaultdict(float)
    disp = set()
    for r in csv.DictReader(open(f)):
        if "fcu_ctu_engine" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
            disp.add(r["Dispatch_Id"])
    res[os.path.basename(os.path.dirname(f))] = {"launches": len(disp), "counters": dict(acc)}
    os.remove(f)
# roofline.traffic for bench.py: per launch, FETCH_SIZE/WRITE_SIZE are in KiB; gfx950 FETCH_SIZE counts 128-B requests as 64 B
fetch = sum(v["counters"].get("FETCH_SIZE", 0.0) / max(1, v["launches"]) for v in res.values())
write = sum(v["counters"].get("WRITE_SIZE", 0.0) / max(1, v["launches"]) for v in res.values())
if fetch and write:
    res["fetch_bytes_per_launch_corrected"] = 2.0 * fetch * 1024.0
    res["write_bytes_per_launch"] = write * 1024.0
    res["traffic_bytes_per_launch"] = 2.0 * fetch * 1024.0 + write * 1024.0
    res["chains_per_launch"] = int(os.environ.get("FCU_PMC_CHAINS", "4096"))
    res["ctus_per_chain_per_launch"] = int(os.environ.get("FCU_PMC_CTUS", "1"))
json.dump(res, open(os.path.join(root, "pmc_summary.json"), "w"), indent=1)
print(json.dumps(res))
