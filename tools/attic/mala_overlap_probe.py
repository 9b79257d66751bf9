"""Where the next unit's generator runs inside a two-pass MALA draw (config-3 shape), A/B on one box:
    V0  with the gradient op, the step kernel waits for it            (round 2's schedule)
    V1  with the gradient op, the step kernel does not wait
    V2  with the step kernel, step queued first                        (+ HIGH: the sampler's stream at high priority)
    V3  with the step kernel, generator queued first
Same draws in every variant (checked against V0)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bayes-kit_amd")]
import torch
import bayes_kit_amd as bk

C, D = int(os.environ.get("C", 65536)), int(os.environ.get("D", 1024))
lam = torch.logspace(0, 4, D, dtype=torch.float64)
print("stream priority range (least, greatest):", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else None)


def run(tag, serialize, gen_with, step_first, high, gen_wgs=0):
    bk.MALA.serialize_step, bk.MALA.generate_with, bk.MALA.step_first = serialize, gen_with, step_first
    bk.MALA.generator_workgroups = gen_wgs
    stream = torch.cuda.Stream(priority=-1) if high else torch.cuda.current_stream()
    with torch.cuda.stream(stream):
        s = bk.MALA(bk.DiagGaussian(lam), 5e-5, chains=C, seed=7)
        s._theta_dc.mul_((1.0 / torch.sqrt(lam)).to(s._theta_dc.device)[:, None])
        s.refresh_cache()
        for _ in range(4):
            s.sample()
        torch.cuda.synchronize(); n = 30
        t0 = time.perf_counter()
        for _ in range(n):
            th, lp = s.sample()
        torch.cuda.synchronize(); el = (time.perf_counter() - t0) / n
        sig = float(th[::997, ::13].sum().item()), s.accept_rate()
    del s
    torch.cuda.empty_cache()
    return {"variant": tag, "ms_per_draw": round(1e3 * el, 4), "frac_88D": round(88.0 * D * C / el / 8e12, 4), "sig": sig}


res = []
for rep in range(2):
    for tag, a in [("V0 grad+wait", (True, "grad", True, False)), ("V1 grad no wait", (False, "grad", True, False)),
                   ("V2 step, step first", (False, "step", True, False)), ("V2 HIGH", (False, "step", True, True)),
                   ("V3 step, gen first", (False, "step", False, False)), ("V3 HIGH", (False, "step", False, True)),
                   ("V0 HIGH", (True, "grad", True, True)),
                   # the generator as a background kernel of N workgroups (one wavefront per SIMD at 256), started with
                   # the gradient op, the step kernel not waiting for it
                   ("BG256", (False, "grad", True, False, 256)), ("BG512", (False, "grad", True, False, 512)),
                   ("BG768", (False, "grad", True, False, 768)), ("BG256 HIGH", (False, "grad", True, True, 256)),
                   ("BG512 HIGH", (False, "grad", True, True, 512)),
                   ("BG256 with step", (False, "step", False, False, 256)), ("BG512 wait", (True, "grad", True, False, 512))]:
        r = run(tag, *a)
        res.append(r)
        print(json.dumps(r), flush=True)
assert all(r["sig"] == res[0]["sig"] for r in res), "variants disagree"
