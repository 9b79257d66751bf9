"""Times bk_normals_chain_major at 65,536 x 1024 with BK_ZIG_PROBE (development probes of k_zig_parallel<16>: 1 = Philox
alone, 2 = + staging / fast test / stores, 3 = + ordering without the slow path, 0 = the kernel).  Each probe in its own
process (the library reads the variable once)."""
import os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tools", "attic")]
    import zig_bench as zb
    lib = zb.load(os.path.join(ROOT, "bayes-kit_amd", "bayes_kit_amd", "lib", "libbkhip.so"))
    for C, D in ((65536, 1024), (32768, 101)):
        _, _, ms = zb.run(lib, C, D, 3, True)
        print(f"probe {os.environ.get('BK_ZIG_PROBE', '0')}: C={C} D={D}: {ms*1e3:8.1f} us", flush=True)
        if os.environ.get("BK_ZIG_PROBE") == "9":
            import ctypes
            import torch
            buf = (ctypes.c_ulonglong * 8)()
            torch.cuda.synchronize()
            lib.bk_debug_zig_cycles(buf, 1)
            tot = sum(buf)
            names = ["philox", "stage+fast", "slow loop", "cover logic", "scan+stores", "-", "-", "loop head"]
            print("   region share of wave time:", ", ".join(f"{n} {100.0*b/tot:.1f}%" for n, b in zip(names, buf) if b))
else:
    for probe in sys.argv[1:] or ["1", "2", "3", "0"]:
        env = dict(os.environ, BK_ZIG_PROBE=probe)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, check=False)
