#!/bin/bash
# routing audit, second pass (kernel-only warm times on a -DGGQ_TUNING build): Q2_K's small batches on the dot4 / LDS-tile kernels
# against the streamed kernel; units x K-slices of the streamed kernel at batch 128 and 48 on the few-row shapes
export GGQ_LIB=scripts/_variants/libggq_tuning.so
run() { timeout -k 10 90 python scripts/sweep_mmq.py "$@" 2>&1 | grep "^type" | sed 's/ *(.*//'; }
for shape in "11008 4096" "3584 8192" "4096 11008"; do set -- $shape; N=$1; export K=$2
  for b in 4 8 16 32; do
    echo -n "dot4/LDS-tile: "; run 10 $b $N
    echo -n "stream       : "; TILED=1 run 10 $b $N
  done
  [ $N = 11008 ] && continue
  for tb in 1 2; do for ks in 4 8; do
    echo -n "b128 TB=$tb KS=$ks: "; TILED=1 GGQ_MMQ_TB=$tb GGQ_MMQ_KS=$ks run 12 128 $N
    echo -n "b48  TB=$tb KS=$ks: "; TILED=1 GGQ_MMQ_TB=$tb GGQ_MMQ_KS=$ks run 12 48 $N
  done; done
done
