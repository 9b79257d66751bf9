"""Config-4 draws (stationary start, advance(): no returned copies) at several geometry thresholds of the device-counted
lane sets (BK_LANES_AUTO_MID / BK_LANES_AUTO_WIDE, csrc/bk_lanes.hpp): ms per draw, each setting in a process of its own.
usage: cfg4_geometry_scan.py [mid:wide ...]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pairs = sys.argv[1:] or ["4608:12288", "4096:8192", "4352:8704", "4608:9216", "4096:12288", "5120:10240", "4608:16384"]
for rep in range(2):
    for p in pairs:
        mid, wide = p.split(":")
        env = dict(os.environ, BK_LANES_AUTO_MID=mid, BK_LANES_AUTO_WIDE=wide, STATIONARY=os.environ.get("STATIONARY", "1"), ADVANCE="1",
                   WARM=os.environ.get("WARM", "300"), N="400")
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "cfg4_profile_run.py")], env=env, capture_output=True, text=True).stdout
        line = [l for l in out.splitlines() if l.startswith("{")][-1]
        d = eval(line)
        print(json.dumps({"mid": int(mid), "wide": int(wide), "ms_per_draw": round(d["ms_per_draw"], 4), "stages": [n for _, n in d["stages_last"]]}), flush=True)
