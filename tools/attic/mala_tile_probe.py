"""MALA at config-3 shape as T independent chain tiles stepped round-robin: a tile's six
arrays (theta, grad, theta', grad', z, theta_new) of 4,096 chains are 192 MiB, inside the 256 MiB Infinity Cache, so
the step kernel's reads of theta' / grad' -- written just before by the previous step / the gradient op -- could
come from cache.  Wall time per draw of all 65,536 chains, against the one-tile sampler."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bayes-kit_amd")]
import torch
import bayes_kit_amd as bk

C, D = 65536, 1024
lam = torch.logspace(0, 4, D, dtype=torch.float64)
for T in (1, 2, 4, 8, 16):
    n = C // T
    tiles = []
    for t in range(T):
        s = bk.MALA(bk.DiagGaussian(lam), 5e-5, chains=n, chain_id0=t * n, seed=7)
        s._theta_dc.mul_((1.0 / torch.sqrt(lam)).to(s._theta_dc.device)[:, None])
        s.refresh_cache()
        tiles.append(s)
    def draw():
        for s in tiles:
            s.sample()
    for _ in range(4):
        draw()
    torch.cuda.synchronize(); N = 20
    t0 = time.perf_counter()
    for _ in range(N):
        draw()
    torch.cuda.synchronize(); el = (time.perf_counter() - t0) / N
    print(json.dumps({"tiles": T, "chains_per_tile": n, "MiB_per_array": n * D * 8 / 2 ** 20, "ms_per_draw_all_chains": round(1e3 * el, 4),
                      "frac_88D": round(88.0 * D * C / el / 8e12, 4), "path": tiles[0].path}), flush=True)
    del tiles
    torch.cuda.empty_cache()
