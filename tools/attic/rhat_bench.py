"""rhat / split_rhat / rank_normalized_rhat of stored draws [N, C] (one parameter) and the Welford update of [D, C]: time per call."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bayes-kit_amd")]
import torch
import bayes_kit_amd as bk

dev = bk._lib.default_ops().device
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return round(1e3 * (time.perf_counter() - t0) / reps, 3)
for N, C in ((1000, 65536), (1000, 32768), (100, 262144)):
    x = torch.randn((N, C), dtype=torch.float64, device=dev)
    r = {"N": N, "C": C, "ms_rhat": timed(lambda: bk.rhat(x)), "ms_split_rhat": timed(lambda: bk.split_rhat(x))}
    if N * C <= 70_000_000:
        r["ms_rank_normalized_rhat"] = timed(lambda: bk.rank_normalized_rhat(x), 2)
    print(json.dumps(r))
D, C = 1024, 65536
m = bk.RunningMoments(D, C)
th = torch.randn((D, C), dtype=torch.float64, device=dev)
print(json.dumps({"welford_update_1024x65536_ms": timed(lambda: m.update(th)), "rhat_all_dims_ms": timed(lambda: m.rhat())}))
