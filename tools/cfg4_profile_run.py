"""Config-4 draws only (no diagnostics), for rocprofv3: construction + warm-up + capture, then N replays."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bayes-kit_amd")]
import torch
import bayes_kit_amd as bk
if os.environ.get("BK_LIB"):  # A/B against another build of the library
    bk._lib._LIB_PATH = os.path.abspath(os.environ["BK_LIB"])
C, D, N = int(os.environ.get("C", 32768)), 101, int(os.environ.get("N", 200))
# OPAQUE=1: the gradient as a separate (counted) op per leapfrog step; OPAQUE=plugin: the same through the user plugin
opaque = os.environ.get("OPAQUE", "0")
model = bk.Funnel(D)
kw = {}
if opaque == "plugin":
    model = bk.CTarget(os.path.join(ROOT, "examples", "plugin_target", "libfunnel_target.so"), "funnel_target", D,
                       counted_symbol="funnel_target_n")
elif opaque == "source":   # the funnel as a per-chain function handed to CTarget.from_source (one lane per chain)
    model = bk.CTarget.from_source("""
__device__ double bk_chain(const BkTheta& th, const BkGrad& g, i64 D, const double*) {
  const double v = th[0];
  double cs[16];
  for (int c = 0; c < 16; ++c) { double a = 0.0; for (i64 d = 1 + c; d < D; d += 16) { const double x = th[d]; a = a + x * x; } cs[c] = a; }
  double q[4];
  for (int k = 0; k < 4; ++k) q[k] = ((cs[k] + cs[k + 4]) + cs[k + 8]) + cs[k + 12];
  const double s = ((q[0] + q[1]) + q[2]) + q[3];
  const double ev = bk_exp(-v), hn = 0.5 * (double)(D - 1), he = 0.5 * ev;
  if (g.wanted()) { g.set(0, ((-v / 9.0) - hn) + he * s); for (i64 d = 1; d < D; ++d) g.set(d, -(ev * th[d])); }
  return ((-(v * v) / 18.0) - hn * v) - he * s;
}""", D, form="chain")
elif opaque in ("lanes", "lanes_fused"):  # the funnel as a lane-spread density from source (bk_lanes.hpp); lanes: counted steps
    sys.path.insert(0, ROOT)
    import bench_secondary as bs
    model = bk.CTarget.from_source(bs.FUNNEL_LANES_SRC, D, form="lanes", head=1)
    if opaque == "lanes":
        kw["path"] = "step"
elif opaque != "0":
    kw["path"] = "step"
init = None
if os.environ.get("STATIONARY") == "1":   # exact draws of the funnel, as bench_secondary's spec_length_stationary_start
    g = torch.Generator().manual_seed(5)
    v0 = 3.0 * torch.randn(C, generator=g, dtype=torch.float64)
    init = torch.cat([v0[:, None], torch.exp(0.5 * v0)[:, None] * torch.randn((C, D - 1), generator=g, dtype=torch.float64)], dim=1)
    kw["init"] = init
s = bk.DrGhmcDiag(model, 3, [0.2, 0.05, 0.0125], [10, 40, 160], 0.1, chains=C, seed=20242,
                  device_counts={"0": False, "1": True}.get(os.environ.get("DEVCOUNTS", ""), None),
                  fuse_first_ghost=os.environ.get("FUSE_GHOST", "1") == "1", recompute_gradient=os.environ.get("REGRAD", "1") == "1", **kw)
if os.environ.get("DEFER") == "0":
    bk.DrGhmcDiag.DEFER_MOMENTS = False
if os.environ.get("ATTACH") == "1":   # the diagnostics bench_secondary feeds from inside the draw's graph
    s.attach(moments=bk.RunningMoments(D, C), recorder=bk.DrawRecorder([0, 1, D - 1], 4000, C))
draw = s.advance if os.environ.get("ADVANCE") == "1" else s.sample
if os.environ.get("PER"):   # advance(n): graphs of n consecutive draws
    per = int(os.environ["PER"])
    N = N // per * per
    draw = lambda: s.advance(per)   # noqa: E731
    reps = N // per   # advance(): a draw without returned copies
for _ in range(int(os.environ.get("WARM", 100))):
    draw()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(reps if os.environ.get("PER") else N):
    draw()
torch.cuda.synchronize()
el = time.perf_counter() - t0
print({"ms_per_draw": 1e3 * el / N, "opaque": opaque, "one_launch": s._one_launch, "fuse_first_ghost": s._fuse_first_ghost, "device_counts": s._dev_counts, "graph": s._use_graph, "lane_steps_last": s.last_lane_steps,
       "stages_last": s.last_stage_lanes})
