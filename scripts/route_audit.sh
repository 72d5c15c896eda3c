#!/bin/bash
# routing audit: kernel-only warm times, LDS-tile (prequant) vs streamed 32- / 64-token units
export GGQ_LIB=scripts/_variants/libggq_tuning.so
run() { timeout -k 10 90 python scripts/sweep_mmq.py "$@" 2>&1 | grep "^type" | sed 's/ *(.*//'; }
for shape in "11008 4096" "3584 8192" "4096 11008"; do set -- $shape; N=$1; export K=$2
  for t in 8 14; do
    for b in 17 32 48 64; do
      [ $t = 14 ] && [ $b -gt 32 ] && continue
      echo -n "LDS-tile   : "; run $t $b $N
      echo -n "stream TB=1: "; TILED=1 GGQ_MMQ_TB=1 run $t $b $N
      [ $b -gt 32 ] && { echo -n "stream TB=2: "; TILED=1 GGQ_MMQ_TB=2 run $t $b $N; }
    done
  done
  for t in 12 13 3 7 2; do
    for b in 80 96; do
      echo -n "stream TB=1: "; TILED=1 GGQ_MMQ_TB=1 run $t $b $N
      echo -n "stream TB=2: "; TILED=1 GGQ_MMQ_TB=2 run $t $b $N
    done
  done
done
