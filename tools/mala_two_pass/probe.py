"""MALA draw at config-3 shape with the step split into two streaming passes (tools/mala_two_pass/probe.hip), the gradient op
and the generator as the library has them: ms per draw against bk.MALA's own path on the same box.  EXPERIMENT.
    python tools/mala_two_pass/probe.py"""
import ctypes, json, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bayes-kit_amd")]
import torch
import bayes_kit_amd as bk
from bayes_kit_amd import _lib

lib = os.path.join("/tmp", "libmala_two_pass_probe.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-ffp-contract=off",
                       os.path.join(HERE, "probe.hip"), "-o", lib])
h = ctypes.CDLL(lib)
P, I, F = ctypes.c_void_p, ctypes.c_int64, ctypes.c_double
h.probe_sums_mask.argtypes = [P, P, P, P, I, P, P, P, F, P, I, I, P]
h.probe_select_propose.argtypes = [P, P, P, P, P, I, P, F, F, I, I, P]
ops = _lib.default_ops()
dev = ops.device
C, D, eps = 65536, 1024, 5e-5
PAD = int(os.environ.get("PAD", 144))
f64 = dict(dtype=torch.float64, device=dev)
lam = torch.logspace(0, 4, D, **f64)
def arr():
    return torch.zeros((D, C + PAD), **f64)[:, :C]
th, g, thp, gp, z = arr(), arr(), arr(), arr(), arr()
th.copy_(torch.randn((D, C), **f64) / torch.sqrt(lam)[:, None])
lp, lp_p, logu = (torch.zeros(C, **f64) for _ in range(3))
mask = torch.zeros(C, dtype=torch.uint8, device=dev)
model = bk.DiagGaussian(lam)
model.bk_eval(th, g, lp)
thp.copy_(th)
ld = th.stride(0)
s2 = (2 * eps) ** 0.5
# the library's generator, as MALA runs it: chain-major normals of the NEXT draw on a side stream
rng = bk.MALA(bk.DiagGaussian(lam), eps, chains=C, seed=7)   # (only to borrow its RNG table and scratch)
zt = torch.empty((C, D), **f64)
side = torch.cuda.Stream()
main = torch.cuda.current_stream()
def stream_of(sv):
    return ctypes.c_void_p(sv.cuda_stream)

def draw(with_gen):
    if with_gen:
        ev = torch.cuda.Event()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            ops.normals_chain_major(rng._rng_kind, rng._rng_state, zt, D, None)
            ops.log_uniform(rng._rng_kind, rng._rng_state, logu)
            ev.record(side)
    model.bk_eval(thp, gp, lp_p)                                   # pass 0: the model's op (16 D)
    h.probe_sums_mask(th.data_ptr(), g.data_ptr(), thp.data_ptr(), gp.data_ptr(), ld, lp.data_ptr(), lp_p.data_ptr(), logu.data_ptr(),
                      eps, mask.data_ptr(), C, D, stream_of(main))  # pass A (32 D)
    if with_gen:
        main.wait_event(ev)
    h.probe_select_propose(th.data_ptr(), g.data_ptr(), thp.data_ptr(), gp.data_ptr(), z.data_ptr(), ld, mask.data_ptr(), eps, s2, C, D,
                           stream_of(main))                         # pass B

def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

res = {"two_pass_no_generator_ms": timed(lambda: draw(False)), "two_pass_with_generator_ms": timed(lambda: draw(True)),
       "accept_rate_last": float(mask.double().mean())}
def only(fn):
    return timed(fn)
res["grad_op_ms"] = only(lambda: model.bk_eval(thp, gp, lp_p))
res["pass_A_ms"] = only(lambda: h.probe_sums_mask(th.data_ptr(), g.data_ptr(), thp.data_ptr(), gp.data_ptr(), ld, lp.data_ptr(), lp_p.data_ptr(),
                                                  logu.data_ptr(), eps, mask.data_ptr(), C, D, stream_of(main)))
mask.copy_((torch.rand(C, device=dev) < 0.57).to(torch.uint8))
res["pass_B_ms_at_accept_0.57"] = only(lambda: h.probe_select_propose(th.data_ptr(), g.data_ptr(), thp.data_ptr(), gp.data_ptr(), z.data_ptr(), ld,
                                                                       mask.data_ptr(), eps, s2, C, D, stream_of(main)))
res["generator_alone_ms"] = only(lambda: ops.normals_chain_major(rng._rng_kind, rng._rng_state, zt, D, None))
s = bk.MALA(bk.DiagGaussian(lam), eps, chains=C, seed=7)
s._theta_dc.mul_((1.0 / torch.sqrt(lam))[:, None])
s.refresh_cache() if hasattr(s, "refresh_cache") else None
res["library_mala_ms"] = timed(s.sample)
res["library_path"] = getattr(s, "path", None)
res["model_88D_at_8TBps_ms"] = 88.0 * D * C / 8e12 * 1e3
print(json.dumps(res))
