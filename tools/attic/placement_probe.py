"""Placement experiments (DESIGN.md section 3 "Placement"): how the relative placement of equal-size
allocations changes the rate of kernels that stream several arrays at equal offsets.

    python tools/placement_probe.py groups   # K groups of (theta, rho, grad): kick+drift / gradient per group
    python tools/placement_probe.py arrays   # is the rate a property of each array?  single-array passes vs triples
    python tools/placement_probe.py mala     # spread of the step-by-step MALA kernels over role assignments
    python tools/placement_probe.py shifts   # after the best triple: do small base shifts of rho / grad help?
    python tools/placement_probe.py pool     # fastest triple in a pool of 8 vs a pool of N arrays (N=30)
    python tools/placement_probe.py sweep    # what HMCDiag._tune_placement finds for SPARES / TRIALS (env)

Config-3 shape (65,536 x 1024); K, N, SEED, SPARES, TRIALS from the environment.
"""
import itertools
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bayes-kit_amd")]
import torch

import bayes_kit_amd as bk
from bayes_kit_amd import _lib

C, D = 65536, 1024


def timed(fn, n=20):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3  # us


def zeros(ops, n):
    return [torch.zeros((D, C), dtype=torch.float64, device=ops.device) for _ in range(n)]


def kd(ops, th, rho, g, n=20):
    return timed(lambda: ops.kick_drift(th, th, rho, rho, g, None, 0.01, False, 0.0, True, 0.01), n)


def groups(ops):
    lam = torch.logspace(0, 4, D, dtype=torch.float64, device=ops.device)
    for k in range(int(os.environ.get("K", 8))):
        th, rho, g = zeros(ops, 3)
        a = kd(ops, th, rho, g)
        b = timed(lambda: ops.target_grad("diag_gaussian", lam, th, g, None))
        print(f"group {k}: kick+drift {a:.1f} us ({40*D*C/a/1e6:.2f} TB/s)  gradient {b:.1f} us ({16*D*C/b/1e6:.2f} TB/s)  step {a+b:.1f} us")


def arrays(ops):
    N = int(os.environ.get("N", 12))
    arrs = zeros(ops, N)
    single = []
    for i, a in enumerate(arrs):
        us = timed(lambda: ops.target_grad("iso_gaussian", None, a, a, None))
        single.append((us, i))
        print(f"array {i}: in-place pass {us:.1f} us ({16*D*C/us/1e6:.2f} TB/s)  ptr {a.data_ptr():#x}")
    single.sort()
    t3 = lambda idx: kd(ops, *(arrs[i] for i in idx))  # noqa: E731
    best, worst, mid = [i for _, i in single[:3]], [i for _, i in single[-3:]], [i for _, i in single[4:7]]
    print("kick+drift on the 3 best / middle / worst arrays:", f"{t3(best):.1f} / {t3(mid):.1f} / {t3(worst):.1f} us")
    random.seed(1)
    per = dict((i, u) for u, i in single)
    for c in random.sample(list(itertools.combinations(range(N), 3)), 12):
        print("combo", c, f"{t3(c):.1f} us  sum of single-array times {sum(per[i] for i in c):.1f}")


def mala(ops):
    arrs = zeros(ops, 10)
    fwd = torch.empty(C, dtype=torch.float64, device=ops.device); rev = torch.empty_like(fwd)
    mask = (torch.rand(C, device=ops.device) < 0.8).to(torch.uint8)
    rnd, res = random.Random(3), []
    for trial in range(16):
        th, g, thp, gp, z, out = (arrs[i] for i in rnd.sample(range(10), 6))
        a = timed(lambda: ops.mala_propose_from_normals(th, g, z, thp, 0.01, 0.1), 10)
        b = timed(lambda: ops.mala_logq(th, g, thp, gp, 0.01, fwd, rev), 10)
        c = timed(lambda: ops.select_columns(mask, th, thp, g, gp, out), 10)
        res.append(a + b + c)
        print(f"trial {trial}: propose {a:.0f}  logq {b:.0f}  select+copy {c:.0f}  sum {a+b+c:.0f} us")
    print("best", min(res), "worst", max(res))


def shifts(ops):
    SL, N = 1 << 17, 8  # slack in doubles (1 MiB)
    raw = [torch.zeros(D * C + SL, dtype=torch.float64, device=ops.device) for _ in range(N)]
    view = lambda i, sh: raw[i][sh:sh + D * C].view(D, C)  # noqa: E731  (sh: doubles, even)
    rnd, best = random.Random(5), (1e9, None)
    for _ in range(30):
        i, j, k = rnd.sample(range(N), 3)
        us = kd(ops, view(i, 0), view(j, 0), view(k, 0), 10)
        best = min(best, (us, (i, j, k)))
    print("best triple by role assignment:", best)
    i, j, k = best[1]
    res = []
    for trial in range(40):
        s1 = rnd.randrange(0, SL // 2) * 2 if trial else 0
        s2 = rnd.randrange(0, SL // 2) * 2 if trial else 0
        res.append((kd(ops, view(i, 0), view(j, s1), view(k, s2), 10), s1 * 8, s2 * 8))
    res.sort()
    print("shifts (bytes) of rho, grad, fastest 5:", [(round(u, 1), a, b) for u, a, b in res[:5]])
    print("slowest 3:", [(round(u, 1), a, b) for u, a, b in res[-3:]], " unshifted:", [round(u, 1) for u, a, b in res if a == b == 0])


def pool(ops):
    N = int(os.environ.get("N", 30))
    arrs = zeros(ops, N)
    rnd = random.Random(int(os.environ.get("SEED", 1)))
    t3 = lambda t: kd(ops, *(arrs[i] for i in t), n=6)  # noqa: E731
    small = sorted((t3(t), t) for t in [tuple(rnd.sample(range(8), 3)) for _ in range(40)])
    big = sorted((t3(t), t) for t in [tuple(rnd.sample(range(N), 3)) for _ in range(300)])
    print("pool of 8 : best", [(round(u, 1), t) for u, t in small[:2]])
    print(f"pool of {N}: best", [(round(u, 1), t) for u, t in big[:4]], " median", round(big[len(big) // 2][0], 1),
          " worst", round(big[-1][0], 1))


def sweep(ops):
    bk.HMCDiag.TUNE_PLACEMENT_SPARES = int(os.environ.get("SPARES", 3))
    bk.HMCDiag.TUNE_PLACEMENT_TRIALS = int(os.environ.get("TRIALS", 30))
    lam = torch.logspace(0, 4, D, dtype=torch.float64)
    s = bk.HMCDiag(bk.DiagGaussian(lam), 0.006, 64, chains=C, seed=1, fuse_builtin=False)
    print(bk.HMCDiag.TUNE_PLACEMENT_SPARES, bk.HMCDiag.TUNE_PLACEMENT_TRIALS, {k: round(v, 4) for k, v in s.placement.items()})


if __name__ == "__main__":
    modes = dict(groups=groups, arrays=arrays, mala=mala, shifts=shifts, pool=pool, sweep=sweep)
    if len(sys.argv) != 2 or sys.argv[1] not in modes:
        sys.exit(__doc__)
    modes[sys.argv[1]](_lib.default_ops())
