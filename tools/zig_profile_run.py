"""The generator alone (bk_normals_chain_major at 65,536 x 1024), N launches, for rocprofv3 passes.  LIB= another build."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tools", "attic")]
import torch
import zig_bench as zb
lib = zb.load(os.environ.get("LIB", os.path.join(ROOT, "bayes-kit_amd", "bayes_kit_amd", "lib", "libbkhip.so")))
C, D = int(os.environ.get("C", 65536)), int(os.environ.get("D", 1024))
_, _, ms = zb.run(lib, C, D, int(os.environ.get("N", 5)), True)
print(f"generator alone {C} x {D}: {ms*1e3:.1f} us per launch")
