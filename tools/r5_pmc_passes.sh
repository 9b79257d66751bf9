#!/bin/bash
# The separate rocprofv3 --pmc passes behind profiles/r5_valu_counters.md: the whole-draw HMC kernel alone
# (tools/traj_q_bench.py, ONLY_L=64: no generator sharing the ALUs) and the one-launch proposal kernel inside config-4 draws
# (tools/cfg4_profile_run.py).  One counter per pass (the databases stay in /tmp; only the summaries come back).
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5; W=/tmp/pmc_r5; rm -rf $W; mkdir -p $W
CTRS="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAVES GRBM_GUI_ACTIVE"
for c in $CTRS; do
  ONLY_L=64 rocprofv3 --pmc $c --kernel-trace -d $W/trajq_$c -o p -- python3 tools/traj_q_bench.py > /dev/null 2>&1
  OPAQUE=0 WARM=100 N=20 rocprofv3 --pmc $c --kernel-trace -d $W/cfg4_$c -o p -- python3 tools/cfg4_profile_run.py > /dev/null 2>&1
done
ONLY_L=64 rocprofv3 --kernel-trace --stats -d $W/trajq_trace -o p -- python3 tools/traj_q_bench.py 2>&1 | grep momentum > gpurun_out/r5/valu_trajq_run.txt
OPAQUE=0 WARM=100 N=20 rocprofv3 --kernel-trace --stats -d $W/cfg4_trace -o p -- python3 tools/cfg4_profile_run.py 2>&1 | tail -1 > gpurun_out/r5/valu_cfg4_run.txt
for w in trajq cfg4; do
  python3 profiles/summarize_rocpd.py $W/${w}_trace/p_results.db $(for c in $CTRS; do echo $W/${w}_$c/p_results.db; done) > gpurun_out/r5/valu_counters_$w.md 2>&1
done
wc -l gpurun_out/r5/valu_counters_*.md
