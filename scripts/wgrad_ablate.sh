# timing ablations of the filter-gradient kernels (SG_X6_ABLATE: 1 no loads, 2 no store / barriers, 4 no fragment reads + MFMAs,
# 8 no fragment reads: MFMAs on stale registers)
mkdir -p gpurun_out/$1
for a in 0 3 11; do
  echo "== SG_X6_ABLATE=$a SG_WGRAD_PIN_PF=1" >> gpurun_out/$1/ablate.txt
  SG_X6_ABLATE=$a SG_WGRAD_PIN_PF=1 python scripts/wgrad_planes_bench.py 2>&1 | grep -E "HxW|2048   256 3|64x64    512|128x128   256" >> gpurun_out/$1/ablate.txt
done
cat gpurun_out/$1/ablate.txt
