"""A/B of two builds of the library on the SMALL generator entry points (one lane per chain): bk_log_uniform and
bk_momentum_refresh at D = 16 (below the wavefront-per-chain path) -- timing and bit equality.
usage: rng_small_ab.py baseline.so candidate.so"""
import ctypes, sys
import torch
P, I, F = ctypes.c_void_p, ctypes.c_int64, ctypes.c_double


def load(path):
    lib = ctypes.CDLL(path)
    lib.bk_rng_init_philox.argtypes = [P, I, ctypes.c_uint64, ctypes.c_uint64, I, P]
    lib.bk_log_uniform.argtypes = [ctypes.c_int, P, I, P, P, I, P]
    lib.bk_momentum_refresh.argtypes = [ctypes.c_int, P, I, P, F, F, P, I, P, P, P, I, I, P, I, P]
    return lib


def run(lib, C, D):
    dev = torch.device("cuda")
    st = torch.zeros((11, C), dtype=torch.int64, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    lib.bk_rng_init_philox(st.data_ptr(), C, 777, 0, C, s)
    u = torch.empty(C, dtype=torch.float64, device=dev)
    out = torch.empty((D, C), dtype=torch.float64, device=dev)
    kin = torch.empty(C, dtype=torch.float64, device=dev)
    res = {}
    for name, fn in (("log_uniform", lambda: lib.bk_log_uniform(0, st.data_ptr(), C, u.data_ptr(), None, C, s)),
                     ("refresh D=%d" % D, lambda: lib.bk_momentum_refresh(0, st.data_ptr(), C, None, 0.0, 1.0, out.data_ptr(), C, None,
                                                                         kin.data_ptr(), None, C, D, None, 0, s))):
        for _ in range(3):
            assert fn() == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / 50 * 1e3
    return res, st.clone(), u.clone(), out.clone(), kin.clone()


a, b = load(sys.argv[1]), load(sys.argv[2])
for C, D in ((65536, 16), (4096, 16), (32768, 8)):
    ra, *xa = run(a, C, D)
    rb, *xb = run(b, C, D)
    same = all(torch.equal(p, q) for p, q in zip(xa, xb))
    print(f"C={C}: " + "  ".join(f"{k}: {ra[k]:.1f} -> {rb[k]:.1f} us" for k in ra) + f"  bit-identical: {same}")
