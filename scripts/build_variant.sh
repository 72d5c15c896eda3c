#!/bin/bash
# usage: scripts/build_variant.sh NAME "-DGGQ_ABL=1 ..." — links scripts/_variants/libggq_NAME.so with mmq.hip rebuilt
set -e
cd "$(dirname "$0")/.."
mkdir -p scripts/_variants /tmp/ggqvar
B=ggml-libtorch_amd/_build
hipcc -ffp-contract=off -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-gpu-rdc -Wno-unused-result $2 -c ggml-libtorch_amd/csrc/hip/${3:-mmq}.hip -o /tmp/ggqvar/$1.o
objs=""
for n in dequant quantize mmvq mmq mmq_t16 mmq_x64 peer traits; do
  if [ "$n" == "${3:-mmq}" ]; then objs="$objs /tmp/ggqvar/$1.o"; else objs="$objs $B/$n.o"; fi
done
hipcc -shared -fPIC --offload-arch=gfx950 -fno-gpu-rdc -o scripts/_variants/libggq_$1.so $objs
echo built scripts/_variants/libggq_$1.so
