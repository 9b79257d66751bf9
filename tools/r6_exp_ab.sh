#!/bin/bash
# Round 6: the one-launch proposal kernel with three definitions of bk_exp, same box, alternating:
#   expF = fdlibm e_exp.c (branches + a division), expA = the fma sequence as the compiler schedules it,
#   (none) = the library as built (fma sequence, three-operand fma written out, clamps instead of selects).
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6
for i in 1 2; do for v in expF expA ""; do
  if [ -n "$v" ]; then export BK_LIB=tools/bin/libbkhip_$v.so; else unset BK_LIB; fi
  PAD=72 python3 tools/funnel_traj_bench.py 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['lib'] or 'library', ' '.join('%s %.1f (%.3f)' % (t['traj'], t['us'], t['us_per_extra_step']) for t in d['trajectories']), '| sum', d['sum_us'])"
done; done > gpurun_out/r6/exp_ab.txt
cat gpurun_out/r6/exp_ab.txt
