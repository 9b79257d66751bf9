#!/bin/bash
# Round 6: timeline of steady-state whole-draw HMC draws at config-3 shape (the generator of draw n+1 on the side stream).
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6; W=/tmp/r6_fused; rm -rf $W; mkdir -p $W
rocprofv3 --kernel-trace --stats --output-format csv -d $W/t -- python3 tools/fused_hmc_profile_run.py > gpurun_out/r6/fused_run.txt 2>/dev/null
python3 tools/attic/trace_timeline.py $W/t 16 > gpurun_out/r6/fused_timeline.txt 2>&1
python3 tools/kernel_stats_table.py $(find $W/t -name "*kernel_stats.csv" | head -1) > gpurun_out/r6/fused_kernels.md 2>&1
PREFETCH=0 python3 tools/fused_hmc_profile_run.py 2>/dev/null | tail -1 >> gpurun_out/r6/fused_run.txt
cat gpurun_out/r6/fused_run.txt; cat gpurun_out/r6/fused_timeline.txt; head -12 gpurun_out/r6/fused_kernels.md
