#!/bin/bash
# Round 6: timeline (start / end of every kernel) of the last draws of a config-4 run, one hipGraph per draw.
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6; W=/tmp/r6_cfg4tl; rm -rf $W; mkdir -p $W
OPAQUE=0 WARM=${WARM:-150} N=40 rocprofv3 --kernel-trace --output-format csv -d $W/t -- python3 tools/cfg4_profile_run.py > gpurun_out/r6/cfg4_tl_run.txt 2>/dev/null
python3 tools/attic/trace_timeline.py $W/t 40 > gpurun_out/r6/cfg4_timeline.txt 2>&1
tail -1 gpurun_out/r6/cfg4_tl_run.txt | cut -c1-400; cat gpurun_out/r6/cfg4_timeline.txt
