"""What the pieces of bench.py's config-4 loop cost per draw (un-profiled wall clock, 200 draws each):
the replayed draw alone, + the returned copies, + the Welford update, + the tracked series."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bayes-kit_amd")]
import torch
import bayes_kit_amd as bk

C, D, N = 32768, 101, 200
s = bk.DrGhmcDiag(bk.Funnel(D), 3, [0.2, 0.05, 0.0125], [10, 40, 160], 0.1, chains=C, seed=20242)
mom = bk.RunningMoments(D, C)
rec = bk.DrawRecorder([0, 1, D - 1], 8 * N + 300, C)
for _ in range(50):
    s.sample()


def timed(fn):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / N


def replay_only():
    s._run_draw(s._draw_dev)


def full():
    th, lp = s.sample()
    mom.update(s._theta_dc)
    rec.record(th, lp)


def sample_welford():
    s.sample()
    mom.update(s._theta_dc)


def replay_welford():
    s._run_draw(s._draw_dev)
    mom.update(s._theta_dc)


print({"replay only": timed(replay_only), "sample()": timed(s.sample), "replay + welford": timed(replay_welford),
       "sample() + welford": timed(sample_welford), "sample() + welford + record": timed(full)})
