#!/bin/bash
# routing audit, sixth pass: 32- against 64-token units beyond batch 64 on matrices with very few rows (kernel-only, warm and cold)
export GGQ_LIB=scripts/_variants/libggq_tuning.so
run() { timeout -k 10 90 python scripts/sweep_mmq.py "$@" 2>&1 | grep "^type" | sed 's/ *(.*//'; }
for t in 12 2; do
for shape in "1024 4096" "2048 4096" "2048 8192" "3072 4096" "4096 4096"; do set -- $shape; N=$1; export K=$2
  for b in 96 128; do for tb in 1 2; do
    echo -n "TB=$tb: "; TILED=1 GGQ_MMQ_TB=$tb run $t $b $N
    echo -n "TB=$tb: "; COLD=1 TILED=1 GGQ_MMQ_TB=$tb run $t $b $N
  done; done
done; done
