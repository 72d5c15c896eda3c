#!/bin/bash
# routing audit, fourth pass: 32- against 64-token units of the streamed kernel at batch 48 over matrix shapes (Q4_K, Q4_1; kernel-only warm + cold)
export GGQ_LIB=scripts/_variants/libggq_tuning.so
run() { timeout -k 10 90 python scripts/sweep_mmq.py "$@" 2>&1 | grep "^type" | sed 's/ *(.*//'; }
for t in 12 3; do
for shape in "2048 4096" "4096 4096" "6144 4096" "8192 4096" "14336 4096" "16384 4096" "28672 4096" "4096 8192" "8192 8192" "28672 8192"; do set -- $shape; N=$1; export K=$2
  for tb in 1 2; do
    echo -n "TB=$tb: "; TILED=1 GGQ_MMQ_TB=$tb run $t 48 $N
    echo -n "TB=$tb: "; COLD=1 TILED=1 GGQ_MMQ_TB=$tb run $t 48 $N
  done
done; done
