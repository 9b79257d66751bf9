#!/bin/bash
# tools/heapguard.c under the randomised soak (3 processes sharing the GPU): does the HIP runtime, or
# anything else in the process, touch freed host memory or write past the end of a block?
# usage: soak_guard_matrix.sh [name:ENV=..;ENV=..]...   (default: every algorithm, library defaults)
mkdir -p gpurun_out/guard
HG=$PWD/tools/bin/libheapguard.so
[ -f $HG ] || { mkdir -p tools/bin && gcc -O2 -g -fPIC -shared -o $HG tools/heapguard.c -ldl -lpthread; }
S=${SOAK_SECONDS:-60}
run() { name=$1; shift; for sd in 101 102 103; do ( f=gpurun_out/guard/$name.$sd.log; timeout $((S + 90)) env "$@" LD_PRELOAD=$HG SOAK_HEAPGUARD=$HG SECONDS=$S SEED=$sd python tests/soak_samplers.py > $f 2>&1; if grep -q "heapguard: invalid access" $f; then echo "[$name $sd] $(grep 'heapguard: invalid access' $f | cut -c1-150)"; else echo "[$name $sd] $(tail -1 $f | cut -c1-150)"; fi ) & done; wait; }
[ $# -eq 0 ] && set -- all:X=1
for v in "$@"; do run "${v%%:*}" $(echo "${v#*:}" | tr ';' ' '); done   # name:ENV=a,b;ENV2=c
