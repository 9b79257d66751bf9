"""Soak test of the wavefront-per-chain generator (k_zig_parallel) against the one-lane-per-chain
kernel over random shapes, stream positions and arguments: outputs, kinetic energies and stream
states must be bit-identical every time.  SECONDS env var = duration (default 60)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bayes-kit_amd")]
import numpy as np
import torch
from bayes_kit_amd import _lib
from bayes_kit_amd._engine import make_streams
ops = _lib.default_ops()
dev = ops.device
rng = np.random.default_rng(int(os.environ.get("SEED", 1)))
budget = float(os.environ.get("SECONDS", 60))
t0, it, normals = time.time(), 0, 0
while time.time() - t0 < budget:
    C = int(rng.choice([1, 2, 3, 63, 64, 65, 127, 257, 1000, 4097, int(rng.integers(1, 20000))]))
    D = int(rng.choice([32, 33, 63, 64, 65, 100, 255, 256, 257, 511, 1024, int(rng.integers(32, 3000))]))
    if C * D > 3e7:
        continue
    seed = int(rng.integers(0, 2**62))
    id0 = int(rng.integers(0, 2**40))
    ka, sa = make_streams(seed, C, id0, False, dev)
    kb, sb = make_streams(seed, C, id0, False, dev)
    u = torch.empty(C, dtype=torch.float64, device=dev)
    work = ops.refresh_work(C, D)
    zt = torch.empty((C, (D + 7) // 8 * 8), dtype=torch.float64, device=dev)
    for rep in range(int(rng.integers(1, 5))):
        for _ in range(int(rng.integers(0, 4))):
            ops.uniform(ka, sa, u); ops.uniform(kb, sb, u)
        use_loc, use_m, use_kin = rng.random() < 0.5, rng.random() < 0.5, rng.random() < 0.7
        loc = torch.randn((D, C), dtype=torch.float64, device=dev) if use_loc else None
        m = (torch.rand(D, dtype=torch.float64, device=dev) + 0.5) if use_m else None
        a = torch.empty((D, C), dtype=torch.float64, device=dev); b = torch.empty_like(a)
        kina = torch.empty(C, dtype=torch.float64, device=dev) if use_kin else None
        kinb = torch.empty(C, dtype=torch.float64, device=dev) if use_kin else None
        mul, sc = float(rng.normal()), float(abs(rng.normal()) + 0.1)
        mode = int(rng.integers(0, 2))
        if mode == 0:
            ops.momentum_refresh(ka, sa, loc, mul, sc, a, m, kina, None, work)
        else:  # chain-major generator + the same arithmetic applied on the host side of the test
            ops.normals_chain_major(ka, sa, zt, D)
            z = zt[:, :D].t()
            a = (loc * mul + sc * z) if use_loc else (0.0 + sc * z)
            kina = None
        ops.momentum_refresh(kb, sb, loc, mul, sc, b, m, kinb, None, None)
        assert torch.equal(a, b), ("values", C, D, rep, mode, seed, id0)
        assert torch.equal(sa, sb), ("state", C, D, rep, mode, seed, id0)
        if use_kin and mode == 0:
            assert torch.equal(kina, kinb), ("kinetic", C, D, rep, seed, id0)
        normals += C * D
    it += 1
print(f"soak ok: {it} random configurations, {normals/1e9:.2f} G normals compared bit for bit in {time.time()-t0:.0f} s")
