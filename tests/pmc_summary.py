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
json.dump(res, open(os.path.join(root, "pmc_summary.json"), "w"), indent=1)
print(json.dumps(res))
