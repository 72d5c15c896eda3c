#!/bin/bash
# routing audit, seventh pass: four against eight K-slices per unit of the streamed kernel at batch 32 (32-token units), by row count (kernel-only, warm / cold)
export GGQ_LIB=scripts/_variants/libggq_tuning.so
run() { timeout -k 10 90 python scripts/sweep_mmq.py "$@" 2>&1 | grep "^type" | sed 's/ *(.*//'; }
for t in 2 12; do
for shape in "6144 4096" "8192 4096" "11008 4096" "14336 4096" "16384 4096" "20480 4096" "8192 8192"; do set -- $shape; N=$1; export K=$2
  for ks in 4 8; do
    echo -n "KS=$ks: "; TILED=1 GGQ_MMQ_KS=$ks run $t 32 $N
    echo -n "KS=$ks: "; COLD=1 TILED=1 GGQ_MMQ_KS=$ks run $t 32 $N
  done
done; done
