#!/bin/bash
# routing audit, fifth pass: 32- against 64-token units at batch 48 with few rows, for the formats that preferred 64-token units at 11008 x 4096
export GGQ_LIB=scripts/_variants/libggq_tuning.so
run() { timeout -k 10 90 python scripts/sweep_mmq.py "$@" 2>&1 | grep "^type" | sed 's/ *(.*//'; }
for t in 2 6 11 14 13 7; do
for shape in "4096 4096" "3584 8192" "4096 11008" "8192 4096"; do set -- $shape; N=$1; export K=$2
  for tb in 1 2; do
    echo -n "TB=$tb: "; TILED=1 GGQ_MMQ_TB=$tb run $t 48 $N
    echo -n "TB=$tb: "; COLD=1 TILED=1 GGQ_MMQ_TB=$tb run $t 48 $N
  done
done; done
