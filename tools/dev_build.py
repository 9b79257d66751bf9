"""Development helper: build a variant of libbkhip.so that differs in ONE translation unit (extra hipcc flags) into
tools/bin/libbkhip_<name>.so, for the A/B tools that take BK_LIB (tools/funnel_traj_bench.py ...).
usage: dev_build.py NAME --unit bk_targets.hip --flags=-DBKHIP_X=1"""
import argparse, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bayes-kit_amd"))
import build as B
ap = argparse.ArgumentParser()
ap.add_argument("name"); ap.add_argument("--unit", default="bk_targets.hip"); ap.add_argument("--flags", default="")
a = ap.parse_args()
obj = f"/tmp/_dev_{a.name}.o"
subprocess.check_call([B._hipcc()] + B.FLAGS + B.FILE_FLAGS.get(a.unit, []) + a.flags.split() + ["-c", os.path.join(B.CSRC, a.unit), "-o", obj])
objs = [os.path.join(B.OBJ_DIR, s.replace(".hip", ".o")) for s in B.SOURCES if s != a.unit] + [obj]
out = os.path.join(ROOT, "tools", "bin", f"libbkhip_{a.name}.so")
subprocess.check_call([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
print(out)
