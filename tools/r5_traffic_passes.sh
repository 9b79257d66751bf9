#!/bin/bash
# HBM traffic (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes) of the two HBM-bound kernels added late in round 5: the
# streaming one-launch leapfrog step of a separable density (bke::k_step, tools/step_stream_bench.py) and MALA's step kernel with
# the density inlined (bkm::k_mala_step_sep, tools/mala_bench.py).  Summaries by profiles/summarize_rocpd.py.
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5; W=/tmp/pmc_r5t; rm -rf $W; mkdir -p $W
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $W/step_$c -o p -- python3 tools/step_stream_bench.py > /dev/null 2>&1
  rocprofv3 --pmc $c --kernel-trace -d $W/mala_$c -o p -- python3 tools/mala_bench.py > /dev/null 2>&1
done
rocprofv3 --kernel-trace --stats -d $W/step_trace -o p -- python3 tools/step_stream_bench.py > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $W/mala_trace -o p -- python3 tools/mala_bench.py > /dev/null 2>&1
for w in step mala; do
  python3 profiles/summarize_rocpd.py $W/${w}_trace/p_results.db $W/${w}_FETCH_SIZE/p_results.db $W/${w}_WRITE_SIZE/p_results.db > gpurun_out/r5/traffic_$w.md 2>&1
done
wc -l gpurun_out/r5/traffic_*.md
