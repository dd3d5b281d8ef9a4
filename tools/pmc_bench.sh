#!/bin/bash
# GPU box: the rocprofv3 --pmc passes of one bench.py command, each reduced to {dispatch: {counter: value}} of the engine kernel
# (tools/pmc_reduce.py) under gpurun_out/<tag>_<pass>.json -- the inputs of tools/pmc_summary.py.  One counter set per pass,
# never with a trace flag (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass).
#   tools/pmc_bench.sh r03_intra                       (default intra bench)
#   tools/pmc_bench.sh r03_ldp --config ldp
set -e
tag=$1; shift
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run_pass() {   # name, counters...
  local name=$1; shift
  local out=$R/gpurun_out/${tag}_$name
  rm -rf "$out"
  echo "pass $name: $*"
  rocprofv3 --output-format csv --pmc "$@" --kernel-include-regex fcu_ctu_engine -d "$out" -o p -- python3 $R/bench.py --no-cpu-baseline --no-sweep --no-ldp-leg $EXTRA > "$out.log" 2>&1
  python3 $R/tools/pmc_reduce.py "$out" "$out.json" > /dev/null
  grep -a '"value"' "$out.log" | cut -c1-160 || true
  rm -rf "$out"
}
EXTRA="$*"
run_pass fetch FETCH_SIZE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run_pass write WRITE_SIZE
run_pass sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES
run_pass tcc TCC_HIT_sum TCC_MISS_sum
