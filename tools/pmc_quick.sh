#!/bin/bash
# usage (GPU box): tools/pmc_quick.sh <tag> <counters...> -- <python args of tools/sweep.py>
# one rocprofv3 --pmc pass of the sweep tool; prints per-counter sums over the engine kernel's dispatches
tag=$1; shift
ctrs=""
while [ "$1" != "--" ]; do ctrs="$ctrs $1"; shift; done; shift
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
rm -rf $out
rocprofv3 --output-format csv --pmc $ctrs --kernel-include-regex "fcu_ctu_engine" -d $out -o p -- python3 $GRAFT_REPO_ROOT/tools/sweep.py "$@" > $out.log 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys, os, collections
acc = collections.defaultdict(float)
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        acc[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
by = collections.defaultdict(dict)
for (d, c), v in acc.items(): by[d][c] = v
for d in sorted(by, key=int): print("dispatch", d, {k: int(v) for k, v in sorted(by[d].items())})
PY
rm -rf $out
