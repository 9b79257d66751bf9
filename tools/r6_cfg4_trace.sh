#!/bin/bash
# Round 6: kernel table of stationary config-4 draws (one-launch proposals, one hipGraph per draw).
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6; W=/tmp/r6_cfg4; rm -rf $W; mkdir -p $W
OPAQUE=0 WARM=150 N=100 rocprofv3 --kernel-trace --stats -d $W/t -o p -- python3 tools/cfg4_profile_run.py > gpurun_out/r6/cfg4_run.txt 2>/dev/null
python3 profiles/summarize_rocpd.py $W/t/p_results.db > gpurun_out/r6/cfg4_kernels.md 2>&1
tail -1 gpurun_out/r6/cfg4_run.txt | cut -c1-200; head -24 gpurun_out/r6/cfg4_kernels.md
