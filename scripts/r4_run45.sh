#!/bin/bash
# round-4 GPU session 45: does the host's thread pool size matter to the eager step's enqueue time?
set -u
OUT=gpurun_out/r4S; mkdir -p $OUT
python -c "import os,torch; print('cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)), 'torch threads', torch.get_num_threads(), 'interop', torch.get_num_interop_threads())" | tee -a $OUT/summary.txt
BB="timeout -k 10 400 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-bf16-leg --no-jit"
run() { name=$1; shift; env "$@" $BB > $OUT/bench_$name.json 2> $OUT/bench_$name.err; echo "bench $name rc=$?" | tee -a $OUT/summary.txt; }
for rep in 1 2; do
  run default_$rep A=1
  run omp16_$rep OMP_NUM_THREADS=16 MKL_NUM_THREADS=16
  run omp1_$rep OMP_NUM_THREADS=1 MKL_NUM_THREADS=1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4S/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], d["ms_per_step"], "host", d["config"]["host_enqueue_ms_per_step"], "probe", d["roofline"]["ms_per_step"])
    except Exception as e: print(f, "unreadable", e)
PY
