#!/bin/bash
# usage: scripts/x64_variant.sh NAME "X64_INPLACE=0 X64_NOPS=4"  -> scripts/_variants/libggq_NAME.so with the K loop regenerated under those switches
set -e
cd "$(dirname "$0")/.."
mkdir -p /tmp/ggqvar
env $2 X64_OUT=/tmp/ggqvar/x64_$1.inc python3 scripts/gen_mmq_x64.py
scripts/build_variant.sh $1 "-DGGQ_X64_LOOPS_INC=\"/tmp/ggqvar/x64_$1.inc\"" mmq_x64
