#!/bin/bash
# Smoke run of the regression tools (tools/README.md): every script once with small settings; "ok" = exit code 0.
# usage (GPU box, repository root): bash tools/smoke_tools.sh > gpurun_out/tools_smoke.txt
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
run() { name="$1"; shift; out=$("$@" 2>&1); rc=$?; echo "$([ $rc -eq 0 ] && echo ok || echo FAIL"($rc)") | $name | $(echo "$out" | grep -v amdgpu.ids | tail -1 | cut -c1-160)"; }
run "cfg4_profile_run.py OPAQUE=0" env OPAQUE=0 WARM=5 N=5 python3 tools/cfg4_profile_run.py
run "cfg4_profile_run.py OPAQUE=1" env OPAQUE=1 WARM=3 N=3 python3 tools/cfg4_profile_run.py
run "cfg4_profile_run.py OPAQUE=plugin" env OPAQUE=plugin WARM=3 N=3 python3 tools/cfg4_profile_run.py
run "cfg4_profile_run.py OPAQUE=lanes" env OPAQUE=lanes WARM=3 N=3 python3 tools/cfg4_profile_run.py
run "cfg4_profile_run.py OPAQUE=lanes_fused" env OPAQUE=lanes_fused WARM=3 N=3 python3 tools/cfg4_profile_run.py
run "cfg4_profile_run.py OPAQUE=source" env OPAQUE=source WARM=3 N=3 python3 tools/cfg4_profile_run.py
run "cfg4_profile_run.py STATIONARY ATTACH PER=10" env STATIONARY=1 ATTACH=1 ADVANCE=1 PER=10 WARM=5 N=20 python3 tools/cfg4_profile_run.py
run "cfg4_geometry_scan.py" env WARM=5 python3 tools/cfg4_geometry_scan.py 4608:12288
run "zig_welford_overlap.py" python3 tools/zig_welford_overlap.py
run "funnel_traj_bench.py" python3 tools/funnel_traj_bench.py
run "traj_q_bench.py ONLY_L=16" env ONLY_L=16 python3 tools/traj_q_bench.py
run "counted_step_bench.py" env REPS=20 LANES=64,2228 python3 tools/counted_step_bench.py
run "counted_step_bench.py SOURCE=lanes" env REPS=20 LANES=64,2228 SOURCE=lanes python3 tools/counted_step_bench.py
run "counted_step_bench.py SOURCE=chain" env REPS=20 LANES=64,2228 SOURCE=chain python3 tools/counted_step_bench.py
run "counted_step_bench.py PLUGIN=1" env REPS=20 LANES=64,2228 PLUGIN=1 python3 tools/counted_step_bench.py
run "fused_hmc_profile_run.py" env N=3 python3 tools/fused_hmc_profile_run.py
run "mala_bench.py" python3 tools/mala_bench.py
run "mala_bench.py INLINED=0" env INLINED=0 python3 tools/mala_bench.py
run "mala_inlined_schedules.py" python3 tools/mala_inlined_schedules.py
run "step_stream_bench.py" python3 tools/step_stream_bench.py
run "mala_two_pass/probe.py" python3 tools/mala_two_pass/probe.py
run "cfg3_trajectory_length.py" env C=4096 DRAWS=10 python3 tools/cfg3_trajectory_length.py
run "cfg4_damping_scan.py" env C=4096 DRAWS=50 python3 tools/cfg4_damping_scan.py
run "funnel_parity_report.py" python3 tools/funnel_parity_report.py
run "soak_rng.py" env SECONDS=5 python3 tools/soak_rng.py
run "config5_ladder.py" python3 tools/config5_ladder.py
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/smoke_trace -o p -- python3 tools/fused_hmc_profile_run.py > /dev/null 2>&1
run "kernel_stats_table.py" python3 tools/kernel_stats_table.py /tmp/smoke_trace
run "draw_gap_scan.py" python3 tools/draw_gap_scan.py /tmp/smoke_trace k_quarter_sums
OPAQUE=0 WARM=5 N=5 rocprofv3 --kernel-trace --output-format csv -d /tmp/smoke_cfg4 -o p -- python3 tools/cfg4_profile_run.py > /dev/null 2>&1
run "cfg4_stage_table.py" python3 tools/cfg4_stage_table.py "$(find /tmp/smoke_cfg4 -name '*kernel_trace.csv' | head -1)"
