"""MALA at config-3 shape with the separable density inlined into the step kernel (model.bk_mala_step): milliseconds per draw
for the generator schedules MALA exposes as class attributes (serialize_step, generate_with, generator_workgroups)."""
import itertools, os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bayes-kit_amd"))
import bayes_kit_amd as bk

C, D = 65536, 1024
lam = torch.logspace(0, 4, D, dtype=torch.float64)

def run(fuse, **attrs):
    s = bk.MALA(bk.DiagGaussian(lam), 5e-5, chains=C, seed=7, path="auto" if fuse else "opaque")
    for k, v in attrs.items():
        setattr(s, k, v)
    s._theta_dc.mul_((1.0 / torch.sqrt(lam)).cuda()[:, None])
    s.refresh_cache()
    for _ in range(4):
        s.sample()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        s.sample()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 30 * 1e3, s._theta_dc.double().sum().item()

print("model-opaque, default schedule: %.3f ms" % run(False)[0])
for ser, gw, wg in itertools.product((True, False), ("grad", "step"), (0, 256, 512)):
    ms, cs = run(True, serialize_step=ser, generate_with=gw, generator_workgroups=wg)
    print("inlined serialize_step=%s generate_with=%s generator_workgroups=%d: %.3f ms  (checksum %.6f)" % (ser, gw, wg, ms, cs))
