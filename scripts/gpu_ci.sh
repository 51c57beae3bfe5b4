#!/bin/bash
# One GPU-box session: parity tests, smoke, bench, rocprof summary.  Every stage runs under its own timeout; a
# stage that times out or is killed aborts the session (no further GPU step after a hang).
#   usage: scripts/gpu_ci.sh <tag> [stages...]      stages: tests smoke bench prof dp1 pmc pmc2 pmc3 pmc5 full infer census nodes bw
set -u
TAG=${1:-run}; shift || true
STAGES=${*:-"tests smoke bench prof"}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
cd "$(dirname "$0")/.."
export TMPDIR=/tmp

run_stage() {  # name timeout cmd...
  local name=$1 tmo=$2; shift 2
  echo "== stage $name: $*" | tee -a "$OUT/summary.txt"
  timeout -k 10 "$tmo" "$@" > "$OUT/$name.log" 2>&1
  local rc=$?
  echo "== stage $name exit $rc" | tee -a "$OUT/summary.txt"
  tail -n 6 "$OUT/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then
    echo "== stage $name timed out / was killed: aborting session" | tee -a "$OUT/summary.txt"
    exit $rc
  fi
  return 0
}

for s in $STAGES; do
  case $s in
    tests) run_stage tests 900 python -m pytest tests -m gpu -q -p no:cacheprovider ;;
    smoke) run_stage smoke 300 python __graft_entry__.py smoke ;;
    bench) run_stage bench 600 python bench.py --steps 6 --warmup 2 ;;
    prof)  run_stage prof 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/rocprof" -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline
           # keep only the small summaries
           find "$OUT/rocprof" -name '*kernel_stats*.csv' -exec cp {} "$OUT/kernel_stats.csv" \; 2>/dev/null
           find "$OUT/rocprof" -name '*kernel_trace*.csv' -size +20M -delete 2>/dev/null ;;
    dp1)   run_stage dp1 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --force-dp ;;
    pmc)   # HBM traffic of the dilated-conv kernels: separate passes (FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2)
           export ONLY_DILATED=1 ITERS=1   # every launch of the set runs exactly twice (warm-up + 1)
           run_stage pmc_fetch 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 scripts/dilated_bench.py
           run_stage pmc_write 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 scripts/dilated_bench.py
           unset ONLY_DILATED ITERS ;;
    pmc2)  # matrix-pipe and LDS counters of the same kernels (separate passes; SQ counters are per-SE sums)
           export ONLY_DILATED=1 ITERS=1
           run_stage pmc_mfma 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_mfma" -- python3 scripts/dilated_bench.py
           run_stage pmc_lds 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d "$OUT/pmc_lds" -- python3 scripts/dilated_bench.py
           unset ONLY_DILATED ITERS ;;
    pmc3)  # the same counters over the LDS-patch kernel (scripts/patch_bench.py, one timed iteration per launch)
           export ITERS=1
           run_stage pmc_mfma 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_mfma" -- python3 scripts/patch_bench.py
           run_stage pmc_lds 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d "$OUT/pmc_lds" -- python3 scripts/patch_bench.py
           unset ITERS ;;
    pmc5)  # round 5: the roofline set as the step runs it (shared activation planes, planes-in filter gradients): four separate passes
           export ITERS=1
           run_stage pmc_fetch 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 scripts/dilated_step.py
           run_stage pmc_write 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 scripts/dilated_step.py
           run_stage pmc_mfma 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_mfma" -- python3 scripts/dilated_step.py
           run_stage pmc_lds 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d "$OUT/pmc_lds" -- python3 scripts/dilated_step.py
           find "$OUT" -name '*kernel_trace*.csv' -size +20M -delete 2>/dev/null
           unset ITERS ;;
    full)  run_stage full 900 python -m pytest tests/test_fullsize_gpu.py -m gpu -q -s -p no:cacheprovider ;;
    infer) run_stage infer 600 python scripts/bench_infer.py ;;
    census) run_stage census 600 python scripts/conv_census.py ;;
    nodes) run_stage nodes 600 python scripts/node_census.py ;;
    bw) run_stage bw 600 python scripts/bw_census.py ;;
    *) echo "unknown stage $s" ;;
  esac
done
echo "== done" | tee -a "$OUT/summary.txt"
