"""bk_sort_by_key (hand-written LSD radix sort of (double, int64) pairs) against torch.sort (vendor sort), time per call.
usage: sort_bench.py [n ...]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bayes-kit_amd")]
import torch
from bayes_kit_amd import _lib

ops = _lib.default_ops()
for n in [int(a) for a in sys.argv[1:]] or [1 << 16, 1 << 20, 1 << 24, 65_536_000]:
    keys = torch.randn(n, dtype=torch.float64, device=ops.device)
    pay = torch.arange(n, dtype=torch.int64, device=ops.device)

    def timed(fn, reps=5):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    ms = timed(lambda: ops.sort_by_key(keys, pay))
    ms_t = timed(lambda: torch.sort(keys, stable=True))
    print(json.dumps({"n": n, "ms_bk_sort_by_key": round(ms, 3), "ms_torch_sort": round(ms_t, 3),
                      "GBps_on_320B_per_key": round(320.0 * n / ms / 1e6, 1)}))
