#!/bin/bash
# Same-box A/B of MALA's serialize_step (alternating, 3 rounds), both step kernels, + the generator alone on this box.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6
python3 tools/zig_profile_run.py 2>/dev/null | tail -1
for r in 1 2 3; do for inl in 0 1; do for ser in 1 0; do
  v=$(INLINED=$inl serialize_step=$ser python3 tools/mala_bench.py 2>/dev/null | tail -1 | python3 -c "import sys,json; print('%.3f' % json.loads(sys.stdin.read())['ms_per_draw'])")
  echo "round $r inlined=$inl serialize_step=$ser: $v ms"
done; done; done
