#!/bin/bash
# Separate rocprofv3 --pmc passes for the generator alone (tools/zig_profile_run.py); usage: r6_zig_pmc.sh TAG [LIB]
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6; TAG=$1; W=/tmp/pmc_r6_$TAG; rm -rf $W; mkdir -p $W
[ -n "$2" ] && export LIB="$GRAFT_REPO_ROOT/$2"
CTRS=${CTRS:-"SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"}
for c in $CTRS; do
  rocprofv3 --pmc $c --kernel-trace -d $W/$c -o p -- python3 tools/zig_profile_run.py > /dev/null 2>&1
done
rocprofv3 --kernel-trace --stats -d $W/trace -o p -- python3 tools/zig_profile_run.py 2>&1 | grep "generator alone" > gpurun_out/r6/zig_pmc_$TAG.md
python3 profiles/summarize_rocpd.py $W/trace/p_results.db $(for c in $CTRS; do echo $W/$c/p_results.db; done) 2>&1 | grep -v "k_init_philox\|elementwise\|^$" >> gpurun_out/r6/zig_pmc_$TAG.md
