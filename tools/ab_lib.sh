#!/usr/bin/env bash
# Same-box A/B of a variant library against the product build (run through gpurun):
#   tools/ab_lib.sh <variant> [sizes] [fits]     libgapro_hip_<variant>.so against libgapro_hip.so
# 1. outputs of a fixed mixed batch of fits, bit for bit (tools/ab_bitwise.py); 2. per-size rates, alternating;
# 3. the headline bench, alternating (new / base / new / base).  Results under gpurun_out/ab_<variant>/.
set -u
V=$1
SIZES=${2:-160,200,256,320,384,448}
FITS=${3:-512}
O=gpurun_out/ab_$V
mkdir -p $O
python tools/ab_bitwise.py --out $O/base.npz > $O/bitwise.log 2>&1
python tools/ab_bitwise.py --lib libgapro_hip_$V.so --out $O/var.npz >> $O/bitwise.log 2>&1
python tools/ab_bitwise.py --compare $O/base.npz $O/var.npz | tee -a $O/bitwise.log
rm -f $O/base.npz $O/var.npz
for rep in 1 2; do
  python tools/bench_fit.py --sizes $SIZES --fits $FITS --reps 2 2>&1 | grep "^M=" | sed "s/^/base $rep: /" | tee -a $O/sizes.log
  python tools/bench_fit.py --sizes $SIZES --fits $FITS --reps 2 --lib libgapro_hip_$V.so 2>&1 | grep "^M=" | sed "s/^/$V $rep: /" | tee -a $O/sizes.log
done
if [[ "${AB_BENCH:-1}" == "1" ]]; then
  cp gapro_amd/libgapro_hip.so /tmp/base.so
  cp gapro_amd/libgapro_hip_$V.so /tmp/var.so
  L="--no-cpu-baseline --no-extra-lines --no-fixed-line --no-driver-line"
  for rep in 1 2; do
    for v in var base; do
      cp /tmp/$v.so gapro_amd/libgapro_hip.so
      python bench.py $L --steps ${AB_STEPS:-8} > $O/${v}_$rep.json 2> $O/${v}_$rep.err
      python tools/show_bench.py $O/${v}_$rep.json | head -4 | sed "s/^/$v $rep: /" | tee -a $O/bench.log
    done
  done
  cp /tmp/base.so gapro_amd/libgapro_hip.so
fi
