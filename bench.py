#!/usr/bin/env python3
"""Headline benchmark: leapfrog steps/sec, many-chain HMC on BASELINE.json config 3.

    python bench.py --gpus N --steps K --warmup W

Workload: ill-conditioned Gaussian, D = 1024, precision
lam = logspace(0, 4, D); HMC with L = 64 leapfrog steps, eps = 0.006, a streamed
length-D metric of ones; 65,536 chains PER GPU (weak scaling: chains are independent and
shard with no data-path collective; Philox key = (20241, global chain id)).  One bench
"step" = one HMC draw of every chain = 64 leapfrog steps each.  The gradient is supplied
by a separate device op (the C-ABI built-in target), i.e. the model-opaque path whose
algorithmic HBM traffic is 56*D bytes per chain-step (40*D integrator + 16*D gradient).

One JSON line is printed by rank 0.  `roofline` is for the dominant kernel (the fused
kick+drift, 40*D algorithmic bytes per chain per launch), its duration measured live with
HIP events on the launch stream over the timed region.  `cpu_baseline` times the oracle
(NumPy restatement of the reference) on the host cores, rank 0 at N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "bayes-kit_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

D_CFG3 = 1024
L_CFG3 = 64
EPS_CFG3 = 0.006
SEED_CFG3 = 20241
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md)


def make_cfg3_sampler(chains, chain_id0, device, D=D_CFG3, L=L_CFG3, eps=EPS_CFG3, chain_tile=None, fused=False,
                      prefetch_rng=None, tune_placement=None):
    """Config-3 sampler for `chains` chains starting at global chain id `chain_id0`."""
    import torch

    import bayes_kit_amd as bk

    lam = torch.logspace(0, 4, D, dtype=torch.float64)
    model = bk.DiagGaussian(lam)
    s = bk.HMCDiag(model, eps, L, metric_diag=torch.ones(D, dtype=torch.float64), seed=SEED_CFG3,
                   chains=chains, chain_id0=chain_id0, chain_tile=chain_tile, fuse_builtin=fused,
                   prefetch_rng=prefetch_rng, tune_placement=tune_placement)
    # theta0_i ~ N(0,1)/sqrt(lam_i): z comes from each chain's own stream (init=None
    # semantics, hmc.py:24-28), scaled to the target's marginal widths (synthetic start)
    s._theta_dc.mul_((1.0 / torch.sqrt(lam)).to(device)[:, None])
    return s


def _pmc_traffic(C, D):
    """HBM bytes per kick+drift launch from the committed PMC passes (rocprofv3 cannot run
    inside this process); None if the profile is for another shape."""
    try:
        with open(os.path.join(ROOT, "profiles", "r1_cfg3_traffic.json")) as f:
            t = json.load(f)
        if t["chains"] == C and t["dims"] == D:
            return t["traffic_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        pass
    return None


def cpu_baseline(seconds_hint=12.0):
    """Oracle (NumPy restatement of the reference samplers) on the host cores: one sampler
    object per chain, chains spread over P processes, config-3 shape, bounded sample."""
    import subprocess

    P = max(1, min(os.cpu_count() or 1, 32))
    chains_per_proc, draws = 8, 400  # 8*400*64 = 205k leapfrog steps per process (~1.5 s each)
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")

    def fan_out(argv_of):
        t0 = time.perf_counter()
        procs = [subprocess.Popen([sys.executable, "-m", "oracle.cpu_baseline"] + [str(a) for a in argv_of(p)],
                                  cwd=ROOT, env=env, stdout=subprocess.PIPE, text=True) for p in range(P)]
        res = []
        for pr in procs:
            out, _ = pr.communicate(timeout=600)
            if pr.returncode != 0:
                raise RuntimeError("cpu baseline worker failed")
            r = json.loads(out.strip().splitlines()[-1])
            res.append((r["steps"], r["seconds"]))
        wall = time.perf_counter() - t0
        busy = max(r[1] for r in res)  # slowest worker's compute time (excludes process start-up)
        return sum(r[0] for r in res) / busy, wall

    rate, wall = fan_out(lambda p: [p * chains_per_proc, chains_per_proc, draws, D_CFG3, L_CFG3, EPS_CFG3, SEED_CFG3])
    out = {
        "value": rate,
        "unit": "leapfrog steps/sec",
        "cores": P,
        "kind": "port",
        "sample": f"{P} procs x {chains_per_proc} chains x {draws} draws x L={L_CFG3} at D={D_CFG3} "
                  f"(oracle/samplers.py HMCDiag, one object per chain); wall incl. spawn {wall:.1f}s",
    }
    try:
        # context only: the same arithmetic hand-vectorised over [D, C] arrays (not how the reference runs)
        bc, bd = 512, 6
        brate, bwall = fan_out(lambda p: ["batched", bc, bd, D_CFG3, L_CFG3, EPS_CFG3, SEED_CFG3 + 1 + p])
        out["batched_numpy"] = {"value": brate, "unit": "leapfrog steps/sec", "cores": P,
                                "sample": f"{P} procs x {bc} chains x {bd} draws, [D, C] arrays, in-place ufuncs; "
                                          f"wall incl. spawn {bwall:.1f}s"}
    except Exception as e:  # the context figure must not cost the baseline
        out["batched_numpy"] = {"error": repr(e)}
    return out


def _device_index(local_rank):
    """LOCAL_RANK, unless BK_BENCH_SHARE_GPU=1: then ranks share the visible GPUs round-robin and
    rendezvous over gloo.  That mode exists only to exercise the N > 1 code path on a one-GPU box
    (RCCL refuses two ranks on one device); its numbers are not measurements."""
    if os.environ.get("BK_BENCH_SHARE_GPU"):
        import torch

        return local_rank % max(1, torch.cuda.device_count())
    return local_rank


def run_other_config(args, rank, local_rank, world):
    """Secondary workloads (parity-test configs of BASELINE.json), not the headline line:
    --config 2: iso-Gaussian D=128, HMC L=32, 4096 chains (cache-resident, launch-bound);
    --config 4: Neal's funnel D=101, DRGHMC K=3, 32,768 chains per GPU + R-hat / ESS."""
    import torch
    import torch.distributed as dist

    import bayes_kit_amd as bk

    local_rank = _device_index(local_rank)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        if os.environ.get("BK_BENCH_SHARE_GPU"):
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            bk.dist.init_from_env()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.config == 2:
        C, D, L = args.chains or 4096, 128, 32
        s = bk.HMCDiag(bk.IsoGaussian(D), 0.05, L, metric_diag=torch.ones(D, dtype=torch.float64), seed=20240,
                       chains=C, chain_id0=rank * C, graph=False if args.no_graph else None)
        for _ in range(args.warmup):
            s.sample()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            s.sample()
        barrier()
        el = time.perf_counter() - t0
        out = {"metric": "leapfrog steps/sec, iso-Gaussian D=128 x 4096 chains per GPU, HMC L=32", "value": C * world * L * args.steps / el,
               "unit": "leapfrog steps/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": 1e3 * el / args.steps, "us_per_leapfrog_step": 1e6 * el / args.steps / L,
               "accept_rate": s.accept_rate(), "dtype": "f64", "data": "synthetic",
               "config": {"workload": "BASELINE.json configs[1]", "chains_per_gpu": C, "dims": D, "leapfrog_steps": L},
               "note": "4 MiB arrays: cache-resident and launch-bound; HBM fraction not meaningful"}
    else:
        C, D = args.chains or 32768, 101
        s = bk.DrGhmcDiag(bk.Funnel(D), 3, [0.2, 0.05, 0.0125], [10, 40, 160], 0.1, chains=C, chain_id0=rank * C,
                          seed=20242)
        mom = bk.RunningMoments(D, C)
        N = args.steps
        rec = bk.DrawRecorder([0, 1, D - 1], N, C)
        for _ in range(args.warmup):
            s.sample()
        lane_steps = 0
        barrier()
        t0 = time.perf_counter()
        for n in range(N):
            th, lp = s.sample()
            lane_steps += s.last_lane_steps
            mom.update(s._theta_dc)
            rec.record(th, lp)
        barrier()
        el = time.perf_counter() - t0
        rh = mom.rhat() if N >= 2 else torch.full((D,), float("nan"))
        if N >= 4:  # the reference's estimator needs 4 draws (ess.py:67-68)
            ess = rec.ess()
            ess = torch.where(ess > 0, ess, torch.full_like(ess, float(N))).clamp(max=float(N)).min(dim=0).values
            ess_total = bk.dist.sum_over_ranks(float(ess.sum().item()), device)
        else:
            ess_total = None
        lane_total = bk.dist.sum_over_ranks(float(lane_steps), device)
        out = {"metric": "DRGHMC funnel D=101 K=3: gradient evaluations/sec (chain-steps actually run)",
               "value": lane_total / el, "unit": "gradient evaluations/sec", "n_gpus": world, "steps": N,
               "warmup": args.warmup, "ms_per_step": 1e3 * el / N, "draws_per_sec": C * world * N / el,
               "mean_grad_evals_per_draw": lane_total / (C * world * N), "rhat_max": float(rh.max()) if N >= 2 else None,
               "rhat_v": float(rh[0]) if N >= 2 else None,
               "ess_per_sec": None if ess_total is None else ess_total / el, "dtype": "f64", "data": "synthetic",
               "config": {"workload": "BASELINE.json configs[3]", "chains_per_gpu": C, "dims": D, "max_proposals": 3}}
    out.update({"higher_is_better": True, "scaling": "weak", "vs_baseline": None})
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--chains", type=int, default=None, help="chains per GPU (default: the config's)")
    ap.add_argument("--config", type=int, default=3, choices=[2, 3, 4],
                    help="3 = headline (default); 2 and 4 = secondary workloads")
    ap.add_argument("--chain-tile", type=int, default=None,
                    help="chains per Infinity-Cache tile (default: no tiling)")
    ap.add_argument("--no-graph", action="store_true", help="config 2: eager launches instead of hipGraph replay")
    ap.add_argument("--no-rng-prefetch", action="store_true",
                    help="generate each draw's randomness in line instead of on the side stream (experiments)")
    ap.add_argument("--no-placement-tuning", action="store_true",
                    help="keep the scratch arrays in the roles they were allocated for (experiments)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--no-fused-extra", action="store_true")
    ap.add_argument("--ess-draws", type=int, default=50, help="extra draws for the ESS/sec figure (0 = skip)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("launch multi-GPU runs with torch.distributed.run (one process per GPU)")
        args.gpus = world

    if args.config != 3:
        return run_other_config(args, rank, local_rank, world)
    args.chains = args.chains or 65536

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline()  # before the GPU is initialised (worker processes never touch it)

    import torch
    import torch.distributed as dist

    local_rank = _device_index(local_rank)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if os.environ.get("BK_BENCH_SHARE_GPU"):
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    C = args.chains
    D, L = D_CFG3, L_CFG3
    s = make_cfg3_sampler(C, rank * C, device, chain_tile=args.chain_tile,
                          prefetch_rng=False if args.no_rng_prefetch else None,
                          tune_placement=False if args.no_placement_tuning else None)
    Ct = s._chain_tile
    ops = s._ops

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        s.sample()
    if not args.no_kernel_events:
        ops.timed = {"bk_leapfrog_kick_drift": [], "bk_target_diag_gaussian_grad": []}
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        s.sample()
    barrier()
    elapsed = time.perf_counter() - t0
    timed, ops.timed = ops.timed, None

    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    accept = s.accept_rate()

    total_steps = float(C) * world * L * args.steps
    value = total_steps / elapsed
    try:  # BASELINE.json's metric string, verbatim
        with open(os.path.join(ROOT, "BASELINE.json")) as f:
            metric_name = json.load(f)["metric"]
    except (OSError, KeyError, ValueError):
        metric_name = "leapfrog steps/sec (whole node), D=1024 x 65,536 chains; ESS/sec"
    out = {
        "metric": metric_name,
        "value": value,
        "unit": "leapfrog steps/sec",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": "BASELINE.json configs[2]: ill-conditioned Gaussian D=1024 (lam=logspace(0,4)), HMC L=64 "
                        "eps=0.006, diag metric of ones, model-opaque gradient op",
            "chains_per_gpu": C, "dims": D, "leapfrog_steps": L, "parallelism": f"chains sharded x{world}",
            "chain_tile": Ct,
        },
        "accept_rate": accept,
        "placement": s.placement,  # which allocation plays which role was chosen by timing (HMCDiag._tune_placement)
        # whole-path figure: 56*D algorithmic bytes per chain-step over the wall clock
        "path_hbm_frac": value / world * 56.0 * D / 1e9 / HBM_PEAK_GBPS,
    }
    if timed:
        kd = [a.elapsed_time(b) for a, b in timed["bk_leapfrog_kick_drift"]]
        gr = [a.elapsed_time(b) for a, b in timed["bk_target_diag_gaussian_grad"]]
        kd_ms = sum(kd) / len(kd)
        bytes_per_launch = 40.0 * D * Ct  # one launch advances one tile of Ct chains by one step
        achieved = bytes_per_launch / (kd_ms * 1e-3) / 1e9
        out["roofline"] = {
            "kernel": "k_kick_drift_v2 (bk_leapfrog_kick_drift)",
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS,
            "traffic": _pmc_traffic(C, D),
            "avg_launch_ms": kd_ms,
            "launches": len(kd),
            "algorithmic_bytes_per_launch": bytes_per_launch,
        }
        if gr:
            g_ms = sum(gr) / len(gr)
            out["roofline"]["gradient_kernel"] = {
                "kernel": "k_gauss_grad_v2 (bk_target_diag_gaussian_grad)",
                "avg_launch_ms": g_ms,
                "achieved": 16.0 * D * Ct / (g_ms * 1e-3) / 1e9,
                "algorithmic_bytes_per_launch": 16.0 * D * Ct,
            }
        try:
            # context for `frac`: what a plain device-to-device copy of the same arrays sustains on this
            # box right now (the guide's measured copy ceiling is 6.29 TB/s = 79 % of the 8 TB/s spec)
            src, dst = s._theta_p, s._grad_p
            dst.copy_(src)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                dst.copy_(src)
            e1.record()
            torch.cuda.synchronize()
            out["roofline"]["device_copy_GBps"] = 10 * 2.0 * src.numel() * 8 / (e0.elapsed_time(e1) * 1e-3) / 1e9
        except Exception as e:  # context only
            out["roofline"]["device_copy_GBps"] = None
    if args.ess_draws > 0:
        try:
            # Second half of BASELINE.json's metric: ESS/sec.  Extra draws (outside the timed region
            # above), three tracked coordinates + the returned log density, ESS by the reference's
            # estimator (ess.py:52-69 -> bk_ess), summed over chains of the per-chain minimum.
            import bayes_kit_amd as bk

            N = args.ess_draws
            series = torch.empty((4, N, C), dtype=torch.float64, device=device)
            barrier()
            t0 = time.perf_counter()
            for n in range(N):
                th, lp = s.sample()
                series[0, n], series[1, n], series[2, n], series[3, n] = th[:, 0], th[:, D // 2], th[:, D - 1], lp
            barrier()
            eel = time.perf_counter() - t0
            ess = torch.stack([bk.ess(series[i]) for i in range(4)])
            # the reference's estimator returns N/IAT with IAT <= 0 possible for antithetic chains
            # (iat.py:151-152); for a throughput figure each series counts as at most N draws
            ess = torch.where(ess > 0, ess, torch.full_like(ess, float(N))).clamp(max=float(N))
            ess_min = ess.min(dim=0).values
            tot = bk.dist.sum_over_ranks(float(ess_min.sum().item()), device)
            out["ess"] = {"draws": N, "tracked": ["theta[0]", "theta[D/2]", "theta[D-1]", "logp"],
                          "ess_per_sec": tot / eel, "mean_min_ess_per_chain": tot / (C * world),
                          "mean_ess_per_tracked": [float(e.mean().item()) for e in ess],
                          "note": "whole job; per-chain minimum over the tracked series (each clipped to "
                                  "(0, N]: IAT <= 0 counts as N), summed over chains"}
            del series
        except Exception as e:
            out["ess"] = {"error": repr(e)}
    if not args.no_fused_extra:
        try:
            # Separately reported (never priced on the 56*D model): the same workload through the
            # built-in target's register-resident trajectory kernel (bk_hmc_trajectory_gaussian).
            # Same results bit for bit; bound by the fp64 vector rate and by the per-draw RNG.
            del s
            torch.cuda.empty_cache()
            f = make_cfg3_sampler(C, rank * C, device, fused=True)
            ops.timed = {"bk_hmc_trajectory_gaussian": []}
            for _ in range(2):
                f.sample()
            ops.timed = {"bk_hmc_trajectory_gaussian": []}
            barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                f.sample()
            barrier()
            fel = time.perf_counter() - t0
            tj = [a.elapsed_time(b) for a, b in ops.timed["bk_hmc_trajectory_gaussian"]]
            ops.timed = None
            tj_ms = sum(tj) / len(tj)
            flop = 6.0 * D * C * L  # 4 mul + 2 add per element-step, individually rounded (no FMA)
            out["fused_builtin"] = {
                "what": "built-in DiagGaussian, whole trajectory in registers; NOT the model-opaque path, reported "
                        "separately from `value`",
                "value": float(C) * L * args.steps / fel, "unit": "leapfrog steps/sec (this rank)",
                "ms_per_step": 1e3 * fel / args.steps,
                "trajectory_kernel_ms": tj_ms, "trajectory_kernel_tflops_fp64": flop / (tj_ms * 1e-3) / 1e12,
                "fp64_vector_peak_tflops_spec": 78.6,
                "note": "peak counts an FMA as 2 flop; this kernel may not contract (bit-parity), ceiling 39.3",
            }
        except Exception as e:  # the extra must never cost the headline line
            out["fused_builtin"] = {"error": repr(e)}
    if cpu is not None:
        out["cpu_baseline"] = cpu
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
