"""One leapfrog step of the counted (lane count on the device) model-opaque path -- gradient op + kick+drift --
as a chain of REPS steps inside one hipGraph, over lane sets of several sizes inside a bound of C chains:
microseconds per launch pair.  LANES=0 measures what a launch costs when every workgroup is surplus.
usage: [C=32768] [D=101] [REPS=200] [PLUGIN=1 | SOURCE=lanes|chain] python tools/counted_step_bench.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bayes-kit_amd")]
import torch
import bayes_kit_amd as bk

C, D, REPS = int(os.environ.get("C", 32768)), int(os.environ.get("D", 101)), int(os.environ.get("REPS", 200))
ops = bk._lib.default_ops()
dev = ops.device
model = bk.Funnel(D)
if os.environ.get("PLUGIN"):
    model = bk.CTarget(os.path.join(ROOT, "examples", "plugin_target", "libfunnel_target.so"), "funnel_target", D,
                       counted_symbol="funnel_target_n")
if os.environ.get("SOURCE"):  # the funnel from source: SOURCE=lanes | chain
    sys.path.insert(0, ROOT)
    import bench_secondary as bs
    model = (bk.CTarget.from_source(bs.FUNNEL_LANES_SRC, D, form="lanes", head=1) if os.environ["SOURCE"] == "lanes"
             else bk.CTarget.from_source(bs.FUNNEL_CHAIN_SRC, D, form="chain"))
f64 = dict(dtype=torch.float64, device=dev)
th, rho, g = (torch.randn((D, C), **f64) * 0.1 for _ in range(3))
n_dev = torch.zeros(1, dtype=torch.int32, device=dev)
h = 1e-4


def chain(which):
    for _ in range(REPS):
        if which == "step":  # {gradient, kick, drift} as ONE launch (models that have bk_leapfrog_step)
            model.bk_leapfrog_step(th, rho, None, h, n_dev)
            continue
        if which in ("both", "grad"):
            model.bk_eval(th, g, None, n_dev)
        if which in ("both", "kd"):
            ops.kick_drift(th, th, rho, rho, g, None, h, False, 0.0, True, h, n_dev=n_dev)


res = {}
for which in ("both", "grad", "kd") + (("step",) if hasattr(model, "bk_leapfrog_step") else ()):
    chain(which)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        chain(which)
    for lanes in [int(x) for x in os.environ.get("LANES", "0,64,765,2228,3727,8192,32768").split(",")]:
        n_dev.fill_(lanes)
        gr.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            gr.replay()
        e1.record()
        e1.synchronize()
        res.setdefault(lanes, {})[which] = round(1e3 * e0.elapsed_time(e1) / 5 / REPS, 2)
print("us per step (both; step = the one-launch step) / per launch (grad, kd), by lanes in the set; bound C =", C, "D =", D, "plugin" if os.environ.get("PLUGIN") else os.environ.get("SOURCE", "builtin"))
for lanes, r in res.items():
    print(lanes, r)
