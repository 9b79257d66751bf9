"""bk_select_columns at config-3 shape: 8-byte stores for half-accepted lane pairs vs full 16-byte
re-writes (BK_AB_SELECT_FULL=1), with and without the fused output copy, at several accept rates."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bayes-kit_amd")]
import torch
from bayes_kit_amd import _lib
ops = _lib.default_ops(); dev = ops.device
C, D = 65536, 1024
a, b, c, d, out = (torch.randn((D, C), dtype=torch.float64, device=dev) for _ in range(5))
for acc in (0.25, 0.57, 0.8, 0.95):
    mask = (torch.rand(C, device=dev) < acc).to(torch.uint8)
    for two in (False, True):
        for copy in (False, True):
            args = (mask, a, b, c if two else None, d if two else None, out if copy else None)
            ops.select_columns(*args)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.select_columns(*args)
            e1.record(); torch.cuda.synchronize()
            print(f"full={bool(os.environ.get('BK_AB_SELECT_FULL'))} acc={acc} pairs={1+two} copy={copy}: {e0.elapsed_time(e1)*100:.0f} us")
