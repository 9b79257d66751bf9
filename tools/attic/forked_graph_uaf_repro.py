"""PyTorch-only reproducer of the HIP runtime use-after-free behind profiles/r2_heapguard.md: hipGraphs with a
parallel branch (a device-to-device copy and a few kernels forked onto a side stream, joined at the end), two
graphs sharing the side stream, replayed a few times and dropped.  Nothing of this repository is imported.

    gcc -O2 -g -fPIC -shared -o tools/bin/libheapguard.so tools/heapguard.c -ldl -lpthread
    LD_PRELOAD=tools/bin/libheapguard.so python tools/forked_graph_uaf_repro.py [seconds] [linear]

With the page-guard allocator the process stops at the runtime's write into the freed 920-byte block (usually
within 20 s); without it the same runs end in glibc heap aborts or flip the last bit of unrelated host data now
and then.  `linear` captures the same work without the fork: clean."""
import ctypes, faulthandler, os, sys, time
import numpy as np
import torch

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 45.0
fork = "linear" not in sys.argv[2:]
guard = os.environ.get("LD_PRELOAD", "")
rng = np.random.default_rng(int(os.environ.get("SEED", 1)))
t0, n_done = time.time(), 0
while time.time() - t0 < seconds:
    if n_done == 40 and "heapguard" in guard:
        faulthandler.enable()
        ctypes.CDLL(guard.split(":")[0]).heapguard_enable()
    n = int(rng.choice([3, 64, 500]))
    f64 = dict(dtype=torch.float64, device="cuda")
    x, y, z = torch.zeros(n, **f64), torch.zeros((11, n), **f64), torch.zeros((11, n), **f64)
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    graphs = []
    for _ in range(2):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            main = torch.cuda.current_stream()
            if fork:
                side.wait_stream(main)
            with torch.cuda.stream(side if fork else main):
                z.copy_(y)
                y.add_(1.0)
                y.mul_(1.0)
            x.add_(1.0)
            x.mul_(1.0)
            if fork:
                main.wait_stream(side)
        graphs.append(g)
    reps = int(rng.integers(2, 10))
    for i in range(reps):
        graphs[i % 2].replay()
    assert x.cpu().numpy()[0] == reps and float(y[0, 0].item()) == reps
    junk = [np.zeros(int(rng.integers(10, 200))) for _ in range(20)]  # host allocations come and go
    n_done += 1
print(f"ok: {n_done} graph pairs captured, replayed and dropped in {time.time() - t0:.0f} s ({'forked' if fork else 'linear'})")
