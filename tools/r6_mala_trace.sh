#!/bin/bash
# Round 6: kernel table of steady-state MALA draws (model-opaque and density-inlined), and every schedule the class exposes.
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6; W=/tmp/r6_mala; rm -rf $W; mkdir -p $W
for inl in 0 1; do
  INLINED=$inl rocprofv3 --kernel-trace --stats -d $W/t$inl -o p -- python3 tools/mala_bench.py > gpurun_out/r6/mala_run_inl$inl.txt 2>/dev/null
  python3 profiles/summarize_rocpd.py $W/t$inl/p_results.db > gpurun_out/r6/mala_kernels_inl$inl.md 2>&1
done
for inl in 0 1; do for gw in grad step; do for ser in 0 1; do
  echo "INLINED=$inl generate_with=$gw serialize_step=$ser: $(INLINED=$inl generate_with=$gw serialize_step=$ser python3 tools/mala_bench.py 2>/dev/null | tail -1 | cut -c1-120)"
done; done; done > gpurun_out/r6/mala_schedules.txt
head -12 gpurun_out/r6/mala_kernels_inl0.md; head -12 gpurun_out/r6/mala_kernels_inl1.md; cat gpurun_out/r6/mala_schedules.txt
