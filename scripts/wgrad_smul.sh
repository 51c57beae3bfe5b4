mkdir -p gpurun_out/r5q
for m in 1 2 3; do
  echo "== SG_WGRAD_PIN_SMUL=$m" >> gpurun_out/r5q/smul.txt
  SG_WGRAD_PIN_SMUL=$m python scripts/wgrad_planes_bench.py 2>&1 | grep -E "2048   256 3|64x64    512|sum" >> gpurun_out/r5q/smul.txt
done
cat gpurun_out/r5q/smul.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for m in 2 3; do
ITERS=1 SG_WGRAD_PIN_SMUL=$m rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/r5q/f$m -- python3 scripts/dilated_step.py > /dev/null 2>&1
ITERS=1 SG_WGRAD_PIN_SMUL=$m rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/r5q/w$m -- python3 scripts/dilated_step.py > /dev/null 2>&1
done
find gpurun_out/r5q -name '*kernel_trace*' -delete
