#!/bin/bash
# soak variants x seeds, 3 processes at a time (sharing the GPU makes stream races far more likely);
# full output of each under gpurun_out/soak/
mkdir -p gpurun_out/soak
HG=$PWD/tools/bin/libheapguard.so
run() { name=$1; shift; for sd in ${SOAK_SEEDS:-101 102 103}; do ( timeout $(( ${SOAK_SECONDS:-140} + 90 )) env "$@" SECONDS=${SOAK_SECONDS:-140} SEED=$sd python tests/soak_samplers.py > gpurun_out/soak/$name.$sd.log 2>&1; tail -1 gpurun_out/soak/$name.$sd.log | cut -c1-100 | sed "s/^/[$name seed $sd] /" ) & done; wait; }
for v in "$@"; do
  case $v in
    all_on) run all_on X=1;;
    no_two_pass) run no_two_pass SOAK_NO_TWO_PASS=1;;
    no_fused_draw) run no_fused_draw SOAK_NO_FUSED_DRAW=1;;
    no_fused_zt) run no_fused_zt SOAK_NO_FUSED_ZT=1;;
    no_graph) run no_graph SOAK_NO_GRAPH=1;;
    no_prefetch) run no_prefetch SOAK_NO_PREFETCH=1;;
    no_graph_no_prefetch) run no_graph_no_prefetch SOAK_NO_GRAPH=1 SOAK_NO_PREFETCH=1;;
    sentinels) run sentinels SOAK_SENTINELS=1;;
    one_queue) run one_queue GPU_MAX_HW_QUEUES=1;;
    sync_drop) run sync_drop SOAK_SYNC_BEFORE_DROP=1;;
    pair_g1p0) run pair_g1p0 ALGS=mala,drghmc SOAK_MALA_GRAPH=1 SOAK_MALA_PREFETCH=0;;
    pair_g1p1) run pair_g1p1 ALGS=mala,drghmc SOAK_MALA_GRAPH=1 SOAK_MALA_PREFETCH=1;;
    pair_g0p1) run pair_g0p1 ALGS=mala,drghmc SOAK_MALA_GRAPH=0 SOAK_MALA_PREFETCH=1;;
    pair_g0p0) run pair_g0p0 ALGS=mala,drghmc SOAK_MALA_GRAPH=0 SOAK_MALA_PREFETCH=0;;
    guard=*) run "guard_${v#guard=}" ALGS="${v#guard=}" LD_PRELOAD=$HG SOAK_HEAPGUARD=$HG;;
    algs=*) run "${v#algs=}" ALGS="${v#algs=}";;
    hmc_only) run hmc_only ALGS=hmc;;
    hmc_metro) run hmc_metro ALGS=hmc,metropolis;;
  esac
done
