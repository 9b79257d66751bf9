"""A/B of two builds of csrc/bk_rng.hip: bk_normals_chain_major timing and bit equality of the normals
and of the stream table afterwards.  usage: zig_bench.py baseline.so candidate.so"""
import ctypes, sys
import torch

P, I = ctypes.c_void_p, ctypes.c_int64


def load(path, snap_arg=True):
    """abi: --old-abi = before the `snapshot` argument; round 6 added `max_workgroups` (BK_ZIG_ABI=r5 for an older build)."""
    import os
    lib = ctypes.CDLL(path)
    lib.bk_rng_init_philox.argtypes = [P, I, ctypes.c_uint64, ctypes.c_uint64, I, P]
    r6 = snap_arg and not (os.environ.get("BK_ZIG_ABI") == "r5" or "base" in os.path.basename(path))
    if r6:
        lib.bk_normals_chain_major.argtypes = [ctypes.c_int, P, I, P, I, I, I, P, I, P]
        lib._tail = (None, 0)
    else:
        n = 9 if snap_arg else 8
        lib.bk_normals_chain_major.argtypes = [ctypes.c_int, P, I, P, I, I, I] + [P] * (n - 7)
        lib._tail = (None, ) if snap_arg else ()
    return lib


def run(lib, C, D, calls, time_it):
    dev = torch.device("cuda")
    st = torch.zeros((11, C), dtype=torch.int64, device=dev)
    dp = (D + 7) // 8 * 8
    zt = torch.zeros((C, dp), dtype=torch.float64, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    assert lib.bk_rng_init_philox(st.data_ptr(), C, 12345, 0, C, s) == 0
    for _ in range(calls):
        assert lib.bk_normals_chain_major(0, st.data_ptr(), C, zt.data_ptr(), dp, C, D, *lib._tail, s) == 0
    ms = None
    if time_it:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        keep = st.clone()
        e0.record()
        for _ in range(10):
            lib.bk_normals_chain_major(0, st.data_ptr(), C, zt.data_ptr(), dp, C, D, *lib._tail, s)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        st.copy_(keep)
        lib.bk_normals_chain_major(0, st.data_ptr(), C, zt.data_ptr(), dp, C, D, *lib._tail, s)
    torch.cuda.synchronize()
    return zt[:, :D].clone(), st.clone(), ms


if __name__ == "__main__":
    a, b = load(sys.argv[1], "--old-abi" not in sys.argv), load(sys.argv[2])  # --old-abi: baseline without `snapshot`
    for C, D in ((65536, 1024), (4096, 128), (32768, 101), (1000, 70), (3, 33), (257, 32), (64, 4000)):
        za, sa, ma = run(a, C, D, 3, True)
        zb, sb, mb = run(b, C, D, 3, True)
        same = torch.equal(za, zb) and torch.equal(sa, sb)
        print(f"C={C} D={D}: baseline {ma*1e3:8.1f} us  candidate {mb*1e3:8.1f} us  ({C*D/mb/1e6:6.1f} Gnormals/s)  "
              f"bit-identical: {same}", flush=True)
        if not same:
            bad = (za != zb).nonzero()
            print("   first differences:", bad[:5].tolist(), "states equal:", torch.equal(sa, sb))
