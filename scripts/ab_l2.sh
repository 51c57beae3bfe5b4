#!/bin/bash
# A/B of the L2-locality switches (SG_CONV_L2: bit 0 grouped tile order, bit 1 channel-block K order, bit 2 wgrad
# channel-block-major order) on the dilated-conv micro-benchmark: timing interleaved over two rounds with 30
# iterations per launch, then one FETCH_SIZE pass per variant.
OUT=gpurun_out/${1:-ab}
mkdir -p $OUT
export ONLY_DILATED=1
for round in 1 2; do
  for v in ${VARIANTS:-2 3 6 7}; do
    echo "== round $round SG_CONV_L2=$v" >> $OUT/ab.log
    ITERS=30 SG_CONV_L2=$v timeout -k 10 300 python scripts/dilated_bench.py >> $OUT/ab.log 2>&1 || exit 1
  done
done
grep -v amdgpu.ids $OUT/ab.log | grep "==\|dilated set\|aspp"
export TMPDIR=/tmp
for v in ${VARIANTS:-2 3 6 7}; do
  SG_CONV_L2=$v ITERS=1 timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_l2_$v -- python3 scripts/dilated_bench.py > $OUT/pmc_l2_$v.log 2>&1 || exit 1
done
echo done
