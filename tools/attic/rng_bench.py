"""Isolated timing of the momentum-refresh (Philox + ziggurat) kernel."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bayes-kit_amd")]
import torch
from bayes_kit_amd import _lib
from bayes_kit_amd._engine import make_streams
ops = _lib.default_ops()
for C, D in ((65536, 1024), (4096, 128), (32768, 101), (1024, 1024), (256, 64)):
  for wave in (False, True):
    kind, st = make_streams(1, C, 0, False, ops.device)
    out = torch.empty((D, C), dtype=torch.float64, device=ops.device)
    kin = torch.empty(C, dtype=torch.float64, device=ops.device)
    work = ops.refresh_work(C, D) if wave else None
    ops.momentum_refresh(kind, st, None, 0.0, 1.0, out, None, kin, None, work)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ops.momentum_refresh(kind, st, None, 0.0, 1.0, out, None, kin, None, work)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"C={C} D={D} {'wave/chain' if wave else 'lane/chain'}: {ms*1e3:.1f} us  {C*D/ms/1e6:.1f} Gnormals/s")
