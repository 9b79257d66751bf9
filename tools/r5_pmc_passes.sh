cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5/pmc
# whole-draw HMC kernel alone (ONLY_L=64: no generator sharing the ALUs) and the funnel trajectory kernel (config-4 draws)
for c in SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE; do
  ONLY_L=64 rocprofv3 --pmc $c --kernel-trace -d gpurun_out/r5/pmc/trajq_$c -o p -- python3 tools/traj_q_bench.py > /dev/null 2>&1
  OPAQUE=0 WARM=100 N=20 rocprofv3 --pmc $c --kernel-trace -d gpurun_out/r5/pmc/cfg4_$c -o p -- python3 tools/cfg4_profile_run.py > /dev/null 2>&1
done
ONLY_L=64 rocprofv3 --kernel-trace --stats -d gpurun_out/r5/pmc/trajq_trace -o p -- python3 tools/traj_q_bench.py 2>&1 | grep momentum
OPAQUE=0 WARM=100 N=20 rocprofv3 --kernel-trace --stats -d gpurun_out/r5/pmc/cfg4_trace -o p -- python3 tools/cfg4_profile_run.py 2>&1 | tail -1 | cut -c1-200
ls gpurun_out/r5/pmc | head -40; du -sh gpurun_out/r5/pmc
