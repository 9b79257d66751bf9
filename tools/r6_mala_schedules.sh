#!/bin/bash
# Round 6: every MALA schedule with the round-6 generator (93 VGPRs: it now fits beside a step workgroup), both step kernels.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6
for inl in 0 1; do for gw in grad step; do for ser in 0 1; do for wg in 0 256 512 1024; do
  r=$(INLINED=$inl generate_with=$gw serialize_step=$ser generator_workgroups=$wg python3 tools/mala_bench.py 2>/dev/null | tail -1 | python3 -c "import sys,json; print('%.3f' % json.loads(sys.stdin.read())['ms_per_draw'])")
  echo "inlined=$inl generate_with=$gw serialize_step=$ser generator_workgroups=$wg: $r ms"
done; done; done; done > gpurun_out/r6/mala_schedules2.txt
cat gpurun_out/r6/mala_schedules2.txt
