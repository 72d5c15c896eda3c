#!/bin/bash
# routing audit, third pass (kernel-only warm times): LDS-tile (or dot4) kernel against the streamed kernel for the mid batches of the other formats
export GGQ_LIB=scripts/_variants/libggq_tuning.so
run() { timeout -k 10 90 python scripts/sweep_mmq.py "$@" 2>&1 | grep "^type" | sed 's/ *(.*//'; }
for shape in "11008 4096" "3584 8192" "4096 11008"; do set -- $shape; N=$1; export K=$2
  for cfg in "10 2" "10 24" "10 33" "10 48" "10 64" "11 17" "11 32" "11 48" "11 64" "13 17" "13 32" "2 17" "2 32" "6 17" "6 32" "3 32" "7 32"; do set -- $cfg
    echo -n "dot4/LDS-tile: "; run $1 $2 $N
    echo -n "stream       : "; TILED=1 run $1 $2 $N
  done
done
