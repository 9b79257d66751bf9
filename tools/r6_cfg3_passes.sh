#!/bin/bash
# Round 6: the rocprofv3 evidence for the headline (config 3) with the library AS SHIPPED: a --kernel-trace --stats pass and
# separate --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py` itself (headline loop only), summarised by
# profiles/summarize_rocpd.py; the per-launch HBM traffic of k_kick_drift_v2 goes to profiles/cfg3_traffic.json stamped with
# sha256(libbkhip.so) and the kernel source hash (tools/stamp_traffic.py).  Run on the GPU box from the repository root.
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6; W=/tmp/r6_cfg3; rm -rf $W; mkdir -p $W
ARGS="--steps 10 --warmup 2 --no-cpu-baseline --no-secondary --no-fused-extra --no-strong-shard --ess-draws 0"
rocprofv3 --kernel-trace --stats -d $W/trace -o p -- python3 bench.py $ARGS > gpurun_out/r6/cfg3_bench_under_rocprof.json 2> /dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $W/$c -o p -- python3 bench.py $ARGS --no-kernel-events > /dev/null 2>&1
done
python3 profiles/summarize_rocpd.py $W/trace/p_results.db $W/FETCH_SIZE/p_results.db $W/WRITE_SIZE/p_results.db \
  --json gpurun_out/r6/cfg3_traffic.json > gpurun_out/r6/cfg3_rocprof.md 2>&1
python3 tools/stamp_traffic.py gpurun_out/r6/cfg3_traffic.json "round 6 (profiles/r6_cfg3_rocprof.md)"
head -12 gpurun_out/r6/cfg3_rocprof.md; cat gpurun_out/r6/cfg3_traffic.json
