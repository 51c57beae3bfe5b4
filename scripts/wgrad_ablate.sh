# timing ablations of the filter-gradient kernels (SG_X6_ABLATE: 1 no loads, 2 no store / barriers, 4 no fragment reads + MFMAs)
mkdir -p gpurun_out/$1
for a in 0 3; do
  echo "== SG_X6_ABLATE=$a SG_WGRAD_PIN_PF=1" >> gpurun_out/$1/ablate.txt
  SG_X6_ABLATE=$a SG_WGRAD_PIN_PF=1 python scripts/wgrad_planes_bench.py 2>&1 | grep -E "HxW|x" >> gpurun_out/$1/ablate.txt
done
echo "== SG_WGRAD_PIN_PF=2" >> gpurun_out/$1/ablate.txt
SG_WGRAD_PIN_PF=2 python scripts/wgrad_planes_bench.py 2>&1 | grep -E "HxW|x" >> gpurun_out/$1/ablate.txt
cat gpurun_out/$1/ablate.txt
