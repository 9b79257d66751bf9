#!/bin/bash
# Round 6 (late): vector-instruction counters of the one-launch proposal kernel inside config-4 draws with the end-of-round
# library (the funnel's exp as a fused-multiply-add sequence), one rocprofv3 --pmc pass per counter (tools/r5_pmc_passes.sh's
# config-4 half; compare profiles/r5_valu_counters.md).  Databases stay in /tmp; the summary comes back.
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6; W=/tmp/pmc_r6c4; rm -rf $W; mkdir -p $W
CTRS="SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU"
for c in $CTRS; do
  OPAQUE=0 WARM=100 N=20 rocprofv3 --pmc $c --kernel-trace -d $W/cfg4_$c -o p -- python3 tools/cfg4_profile_run.py > /dev/null 2>&1
done
OPAQUE=0 WARM=100 N=20 rocprofv3 --kernel-trace --stats -d $W/cfg4_trace -o p -- python3 tools/cfg4_profile_run.py 2>&1 | tail -1 > gpurun_out/r6/valu_cfg4_run.txt
python3 profiles/summarize_rocpd.py $W/cfg4_trace/p_results.db $(for c in $CTRS; do echo $W/cfg4_$c/p_results.db; done) > gpurun_out/r6/valu_counters_cfg4.md 2>&1
cut -c1-300 gpurun_out/r6/valu_cfg4_run.txt; grep -n "k_lane_traj" gpurun_out/r6/valu_counters_cfg4.md | cut -c1-200
