"""Development helper: build a variant of libbkhip.so that differs in bk_rng.hip only (extra hipcc flags and / or a
__launch_bounds__ waves-per-EU bound on k_zig_parallel) into tools/bin/libbkhip_<name>.so, for tools/attic/zig_bench.py.
usage: zig_dev_build.py NAME [--wpe N] [--flags "..."] [--src other_bk_rng.hip]"""
import argparse, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "bayes-kit_amd", "csrc")
OBJ = os.path.join(ROOT, "bayes-kit_amd", "build")
ap = argparse.ArgumentParser()
ap.add_argument("name"); ap.add_argument("--wpe", type=int, default=0); ap.add_argument("--flags", default=""); ap.add_argument("--src", default=None)
a = ap.parse_args()
src = open(a.src or os.path.join(CSRC, "bk_rng.hip")).read()
if a.wpe:
    src = src.replace("__global__ __launch_bounds__(ZP_WAVES* BK_WAVE) void k_zig_parallel", f"__global__ __launch_bounds__(ZP_WAVES* BK_WAVE, {a.wpe}) void k_zig_parallel")
tmp = os.path.join(CSRC, f"_dev_{a.name}.hip")
open(tmp, "w").write(src)
obj = f"/tmp/_dev_{a.name}.o"
try:
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math"] + a.flags.split() + ["-c", tmp, "-o", obj]
    subprocess.check_call(cmd)
finally:
    os.remove(tmp)
objs = [os.path.join(OBJ, f) for f in os.listdir(OBJ) if f.endswith(".o") and f != "bk_rng.o"] + [obj]
out = os.path.join(ROOT, "tools", "bin", f"libbkhip_{a.name}.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
print(out)
