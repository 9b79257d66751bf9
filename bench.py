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

Launching.  With WORLD_SIZE in the environment (torch.distributed.run, one process per GPU)
this process is one rank.  Without it, `--gpus N` with N > 1 makes this process a LAUNCHER:
it starts N rank processes itself (before anything touches the GPU), relays rank 0's JSON
line and exits non-zero if any rank fails.  Ranks rendezvous over RCCL (`nccl` backend); on a
box with fewer GPUs than ranks they share the visible GPUs and rendezvous over gloo instead
(`"shared_gpu": true`: exercises the N > 1 code path, its numbers are not measurements).

One JSON line is printed by rank 0.  `value` is the weak-scaling figure (65,536 chains per
GPU); `value_strong` is the same metric with 65,536 chains per NODE (65,536 / N per GPU).
`roofline` is for the dominant kernel (the fused kick+drift, 40*D algorithmic bytes per chain
per launch), its duration measured live with HIP events on the launch stream over the timed
region -- around every 8th launch, spread evenly (`roofline.event_stride`): an event record is a
packet of its own on the queue and bracketing all 128 launches of a draw cost the draw 4 %.
`cpu_baseline` times the oracle (NumPy restatement of the reference) on the host cores that can
run at once (affinity capped by the cgroup quota), rank 0 at N=1 only.  `secondary` (N=1) carries the
other BASELINE.json configs, each with its own bound and bytes / flop model; at N > 1 config 4 runs at
every world size with R-hat / ESS over the process group, and every rank pins itself to its GPU's NUMA
share of the CPUs before torch starts.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "bayes-kit_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

from bench_config import (C_CFG3, D_CFG3, EPS_CFG3, FP64_MFMA_PEAK_TFLOPS, FP64_VECTOR_PEAK_TFLOPS,  # noqa: E402,F401
                          HBM_PEAK_GBPS, L_CFG3, SEED_CFG3)
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "cfg3_traffic.json")
KD_SOURCE = os.path.join(ROOT, "bayes-kit_amd", "csrc", "bk_integrator.hip")


# ---------------------------------------------------------------------------------------------
# launcher: N rank processes on one node
# ---------------------------------------------------------------------------------------------
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n, child_argv, extra_env=None, timeout=3000.0, out=None):
    """Start `child_argv` n times, one process per rank, with RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT set (what torch.distributed.run would set).  Rank 0's stdout is
    relayed to `out` (default: this process's stdout); other ranks' stdout goes to stderr.
    Returns 0 when every rank exited 0; otherwise the remaining ranks are terminated and the
    first failing exit code is returned.  The launcher itself never touches the GPU."""
    out = out if out is not None else sys.stdout
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "MASTER_ADDR": "127.0.0.1",
                    "MASTER_PORT": str(port),
                    "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
        env.update(extra_env or {})
        procs.append(subprocess.Popen(list(child_argv), env=env, stdout=subprocess.PIPE, stderr=None, text=True))
    deadline = time.monotonic() + timeout
    rc = 0
    pending = set(range(n))
    outputs = {}
    while pending:
        for r in sorted(pending):
            p = procs[r]
            try:
                o, _ = p.communicate(timeout=0.2)
            except subprocess.TimeoutExpired:
                continue
            pending.discard(r)
            outputs[r] = o
            if p.returncode != 0 and rc == 0:
                rc = p.returncode or 1
        if (rc != 0 or time.monotonic() > deadline) and pending:
            rc = rc or 124
            for r in pending:
                procs[r].terminate()
            for r in sorted(pending):
                try:
                    outputs[r], _ = procs[r].communicate(timeout=10)
                except subprocess.TimeoutExpired:
                    procs[r].kill()
                    outputs[r], _ = procs[r].communicate()
            pending.clear()
    for r in range(1, n):
        if outputs.get(r):
            sys.stderr.write(outputs[r])
    # rank 0's JSON line(s) are the launcher's stdout; anything else a library printed there (gloo
    # announces its connections on stdout) goes to stderr, so that stdout stays one JSON line
    for line in (outputs.get(0) or "").splitlines():
        (out if line.lstrip().startswith("{") else sys.stderr).write(line + "\n")
    out.flush()
    return rc


def _visible_gpus():
    """Number of GPUs this process's children will see, found WITHOUT loading the HIP runtime here: the
    launcher must not touch the GPU before its ranks do.  Order: an explicit visibility list in the
    environment; the kernel driver's topology (KFD nodes that have SIMDs are GPUs); render nodes; and, if
    none of that is readable, a child process that asks PyTorch (the child, not the launcher, loads HIP)."""
    for var in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([x for x in v.split(",") if x.strip()])
    try:
        import glob

        n, seen = 0, 0
        for path in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
            with open(path) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
            seen += 1
            if int(props.get("simd_count", "0")) > 0:
                n += 1
        if seen:
            return n
    except (OSError, ValueError):
        pass
    try:
        import glob

        n = len(glob.glob("/dev/dri/renderD*"))
        if n:
            return n
    except OSError:
        pass
    try:
        out = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"],
                             capture_output=True, text=True, timeout=300)
        return int(out.stdout.strip().splitlines()[-1])
    except Exception:
        return 0


# ---------------------------------------------------------------------------------------------
# CPU placement of a rank
# ---------------------------------------------------------------------------------------------
def _parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        cpus.update(range(int(a), int(b or a) + 1))
    return cpus


def _gpu_numa_node(index):
    """NUMA node of the index-th render node (no HIP: sysfs), or None."""
    try:
        import glob

        # (AMD render nodes only: vendor 0x1002 -- a box may have other GPUs' render nodes as well)
        cards = [p for p in sorted(glob.glob("/sys/class/drm/renderD*/device/numa_node"),
                                   key=lambda p: int(p.split("renderD")[1].split("/")[0]))
                 if open(os.path.join(os.path.dirname(p), "vendor")).read().strip().lower() == "0x1002"]
        node = int(open(cards[index]).read())
        return node if node >= 0 else None
    except (OSError, ValueError, IndexError):
        return None


def rank_cpu_set(local_rank, world, affinity=None, numa_of_gpu=_gpu_numa_node, node_cpus=None):
    """The CPUs rank `local_rank` of `world` ranks on this node should run on: those of its GPU's NUMA node that this
    process may use, shared evenly between the ranks whose GPUs sit on the same node (launch-bound workloads -- config 2
    at 8 us per step, config 4's twelve launches per draw -- are the ones a host thread migrating across sockets hurts);
    when the topology is not readable, an even contiguous slice of the allowed CPUs.  Pure: inputs can be injected."""
    if affinity is None:
        try:
            affinity = os.sched_getaffinity(0)
        except (AttributeError, OSError):
            affinity = set(range(os.cpu_count() or 1))
    allowed = sorted(affinity)
    if world <= 1 or len(allowed) <= 1:
        return set(allowed)

    def cpus_of(node):
        if node_cpus is not None:
            return set(node_cpus.get(node, ()))
        try:
            return _parse_cpulist(open(f"/sys/devices/system/node/node{node}/cpulist").read())
        except OSError:
            return set()

    nodes = [numa_of_gpu(r) for r in range(world)]
    mine = nodes[local_rank]
    if mine is not None:
        pool = sorted(cpus_of(mine) & set(allowed))
        peers = [r for r in range(world) if nodes[r] == mine]
        if len(pool) >= len(peers):
            k, per = peers.index(local_rank), len(pool) // len(peers)
            return set(pool[k * per:(k + 1) * per])
    per = max(1, len(allowed) // world)
    k = local_rank % max(1, len(allowed) // per)
    return set(allowed[k * per:(k + 1) * per])


def pin_rank(local_rank, world):
    """Apply rank_cpu_set() to this process (BEFORE torch / HIP start their threads: they inherit it).  Returns a
    record for the JSON line; BK_BENCH_NO_PIN=1 leaves the affinity alone."""
    if world <= 1 or os.environ.get("BK_BENCH_NO_PIN"):
        return None
    masks = [v for v in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES") if os.environ.get(v)]
    if masks:
        # device index -> render node is the identity only without a visibility mask (and a reordered or partial list
        # cannot be resolved without the HIP runtime, which must not start before the affinity is set): leave the
        # affinity alone
        return {"skipped": "visibility mask set (" + ", ".join(masks) + "): GPU -> NUMA node mapping "
                                                                        "not resolvable without HIP"}
    try:
        cpus = rank_cpu_set(local_rank, world)
        os.sched_setaffinity(0, cpus)
        return {"cpus": len(cpus), "first_cpu": min(cpus), "numa_node_of_gpu": _gpu_numa_node(local_rank)}
    except (AttributeError, OSError, ValueError) as e:
        return {"error": repr(e)}


# ---------------------------------------------------------------------------------------------
# workload builders
# ---------------------------------------------------------------------------------------------
def make_cfg3_sampler(chains, chain_id0, device, D=D_CFG3, L=L_CFG3, eps=EPS_CFG3, chain_tile=None, fused=False,
                      prefetch_rng=None, tune_placement=None):
    """Config-3 sampler for `chains` chains starting at global chain id `chain_id0`."""
    import torch

    import bayes_kit_amd as bk

    lam = torch.logspace(0, 4, D, dtype=torch.float64)
    model = bk.DiagGaussian(lam)
    s = bk.HMCDiag(model, eps, L, metric_diag=torch.ones(D, dtype=torch.float64), seed=SEED_CFG3,
                   chains=chains, chain_id0=chain_id0, chain_tile=chain_tile,
                   # (the headline's path: the gradient a SEPARATE op per leapfrog step, whatever the model offers)
                   path="auto" if fused else "opaque",
                   prefetch_rng=prefetch_rng, tune_placement=tune_placement)
    # theta0_i ~ N(0,1)/sqrt(lam_i): z comes from each chain's own stream (init=None
    # semantics, hmc.py:24-28), scaled to the target's marginal widths (synthetic start)
    s._theta_dc.mul_((1.0 / torch.sqrt(lam)).to(device)[:, None])
    return s


def source_hash(path=None):
    """Hash of the kick+drift kernel's source text (the region of bk_integrator.hip between its
    section marker and the next kernel), so that edits elsewhere in the file do not stale the stamp."""
    with open(path or KD_SOURCE, "r") as f:
        text = f.read()
    a = text.find("// ---- fused kick + drift")
    b = text.find("// any layout of the gradient")
    region = text[a:b] if 0 <= a < b else text
    return hashlib.sha256(region.encode()).hexdigest()[:16]


LIB_FILE = os.path.join(ROOT, "bayes-kit_amd", "bayes_kit_amd", "lib", "libbkhip.so")


def lib_hash(path=None):
    """sha256 (16 hex digits) of the shared library the process loads: what the PMC passes were taken with."""
    try:
        h = hashlib.sha256()
        with open(path or LIB_FILE, "rb") as f:
            for chunk in iter(lambda: f.read(1 << 20), b""):
                h.update(chunk)
        return h.hexdigest()[:16]
    except OSError:
        return None


def _pmc_traffic(C, D):
    """(HBM bytes per kick+drift launch, how the stamp matched) from the committed PMC passes (rocprofv3 cannot run
    inside this process).  The file is stamped with the sha256 of the libbkhip.so the passes ran (`lib_sha256_16`) and
    with the hash of the kernel's source text: the traffic is reported when the LIBRARY is the same binary
    ("library"), or -- a rebuilt library whose kick+drift source is unchanged -- when the source text is
    ("kernel source"); otherwise None."""
    try:
        with open(TRAFFIC_FILE) as f:
            t = json.load(f)
        if t["chains"] == C and t["dims"] == D:
            if t.get("lib_sha256_16") and t.get("lib_sha256_16") == lib_hash():
                return t["traffic_bytes_per_launch"], "library"
            if t.get("source_sha256_16") == source_hash():
                return t["traffic_bytes_per_launch"], "kernel source"
    except (OSError, KeyError, ValueError):
        pass
    return None, None


# ---------------------------------------------------------------------------------------------
# which device, at which clocks (sysfs only: no HIP call, safe in the launcher too)
# ---------------------------------------------------------------------------------------------
def _amd_cards():
    import glob

    cards = []
    for p in sorted(glob.glob("/sys/class/drm/renderD*/device"), key=lambda p: int(p.split("renderD")[1].split("/")[0])):
        try:
            if open(os.path.join(p, "vendor")).read().strip().lower() == "0x1002":
                cards.append(p)
        except OSError:
            pass
    return cards


def _current_mhz(path):
    """The starred level of a pp_dpm_* file, in MHz."""
    try:
        for line in open(path):
            if "*" in line:
                return float(line.split(":")[1].strip().split("M")[0].strip())
    except (OSError, ValueError, IndexError):
        pass
    return None


def device_identity(index):
    """PCI / unique id of the index-th AMD render node: lets two bench lines be told apart by box."""
    try:
        dev = _amd_cards()[index]
    except IndexError:
        return None
    out = {}
    for key, name in (("unique_id", "unique_id"), ("pci_device", "device"), ("pci_revision", "revision"),
                      ("vbios", "vbios_version")):
        try:
            out[key] = open(os.path.join(dev, name)).read().strip()
        except OSError:
            out[key] = None
    try:
        out["pci_slot"] = os.path.basename(os.path.realpath(dev))
    except OSError:
        out["pci_slot"] = None
    return out


def _device_record(index, clocks):
    """The `device` object of the line; whatever cannot be read is None (it must never cost the headline)."""
    rec = {"name": None, "identity": None, "clocks": None, "lib_sha256_16": None}
    try:
        import torch

        rec["name"] = torch.cuda.get_device_name(index)
    except Exception:
        pass
    for key, fn in (("identity", lambda: device_identity(index)), ("clocks", clocks.summary), ("lib_sha256_16", lib_hash)):
        try:
            rec[key] = fn()
        except Exception:
            pass
    return rec


class ClockSampler:
    """Shader / memory clock of one card sampled from sysfs by a sleeping thread while a timed region runs (a read every
    `period` seconds: a few dozen samples per region, no HIP, nothing on the GPU's queues)."""

    def __init__(self, index, period=0.02):
        import threading

        try:
            cards = _amd_cards()
        except Exception:
            cards = []
        self._dev = cards[index] if index < len(cards) else None
        self._period, self._stop, self.sclk, self.mclk, self.power = period, threading.Event(), [], [], []
        # hwmon reports the clocks the part actually runs at (freq1_input / freq2_input, Hz) and the socket power; the
        # pp_dpm_* files only say which DPM level is selected
        self._hw = None
        if self._dev:
            import glob

            hw = sorted(glob.glob(os.path.join(self._dev, "hwmon", "hwmon*")))
            self._hw = hw[0] if hw else None
        self._thread = threading.Thread(target=self._run, daemon=True) if self._dev else None

    def _run(self):
        while not self._stop.is_set():
            a = self._hwmon("freq1_input", 1e-6) or _current_mhz(os.path.join(self._dev, "pp_dpm_sclk"))
            b = self._hwmon("freq2_input", 1e-6) or _current_mhz(os.path.join(self._dev, "pp_dpm_mclk"))
            w = self._hwmon("power1_average", 1e-6) or self._hwmon("power1_input", 1e-6)
            if w is not None:
                self.power.append(w)
            if a is not None:
                self.sclk.append(a)
            if b is not None:
                self.mclk.append(b)
            self._stop.wait(self._period)

    def _hwmon(self, name, scale):
        if not self._hw:
            return None
        try:
            return float(open(os.path.join(self._hw, name)).read()) * scale
        except (OSError, ValueError):
            return None

    def __enter__(self):
        if self._thread:
            self._thread.start()
        return self

    def __exit__(self, *exc):
        if self._thread:
            self._stop.set()
            self._thread.join(timeout=1.0)

    def summary(self):
        def stat(v):
            if not v:
                return None
            w = sorted(v)
            return {"min": w[0], "median": w[len(w) // 2], "max": w[-1], "samples": len(w)}
        return {"sclk_mhz": stat(self.sclk), "mclk_mhz": stat(self.mclk), "power_w": stat(self.power),
                "power_cap_w": self._hwmon("power1_cap", 1e-6),
                "source": ("sysfs hwmon freq1_input / freq2_input / power1_average" if self._hw else
                           "sysfs pp_dpm_sclk / pp_dpm_mclk (selected DPM level)") + " during the timed region"}


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return None


def _cgroup_cpu_limit():
    """CPUs this container may use at once per its cgroup quota (None: no quota readable)."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        return None if quota == "max" else float(quota) / float(period)
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            q = int(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            per = int(f.read())
        return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


def cpu_cores_that_can_run():
    """Threads this process can actually run at once: its CPU affinity, capped by the cgroup CPU quota."""
    import math

    total = os.cpu_count() or 1
    try:
        usable = len(os.sched_getaffinity(0))  # (a container may be pinned to fewer)
    except (AttributeError, OSError):
        usable = total
    quota = _cgroup_cpu_limit()
    can_run = max(1, usable if quota is None else min(usable, int(math.ceil(quota))))
    return total, usable, quota, can_run


def cpu_baseline():
    """Oracle (NumPy restatement of the reference samplers) on the host cores: one sampler object per
    chain -- the reference's execution model -- chains spread over P worker processes, config-3 shape, a
    bounded sample (~1.5 s per process).  P = the threads that can run at once: min(CPU affinity,
    ceil(cgroup CPU quota)); `cores` is that P.  One point (round 3 also spawned one process per hardware
    thread of the host: 16x oversubscribed under the container's quota, noise)."""
    total, usable, quota, P = cpu_cores_that_can_run()
    P = min(P, 64)
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")

    def fan_out(P, argv_of):
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
        return sum(r[0] for r in res) / busy, wall, sum(r[1] for r in res)

    chains_per_proc, draws = 8, 400  # 8 x 400 x 64 = 205k leapfrog steps per process
    rate, wall, cpu_s = fan_out(P, lambda p: [p * chains_per_proc, chains_per_proc, draws, D_CFG3, L_CFG3, EPS_CFG3, SEED_CFG3])
    out = {
        "value": rate,
        "unit": "leapfrog steps/sec",
        "cores": P,
        "host_cores_total": total,
        "host_cores_usable": usable,
        "cgroup_cpu_limit": quota,
        "cpu_model": _cpu_model(),
        "kind": "port",
        "sample": f"{P} procs x {chains_per_proc} chains x {draws} draws x L={L_CFG3} at D={D_CFG3} "
                  f"(oracle/samplers.py HMCDiag, one object per chain); {cpu_s:.0f} CPU-seconds of work in all, wall incl. "
                  f"spawn {wall:.1f}s",
        "cpu_seconds": cpu_s,
        "leg_seconds": wall,
    }
    if os.environ.get("BK_BENCH_BATCHED_NUMPY"):
        # context only (opt-in): the same arithmetic hand-vectorised over [D, C] arrays (not how the reference runs)
        try:
            bc, bd = 512, 6
            brate, bwall, _ = fan_out(P, lambda p: ["batched", bc, bd, D_CFG3, L_CFG3, EPS_CFG3, SEED_CFG3 + 1 + p])
            out["batched_numpy"] = {"value": brate, "unit": "leapfrog steps/sec", "cores": P,
                                    "sample": f"{P} procs x {bc} chains x {bd} draws, [D, C] arrays, in-place ufuncs; "
                                              f"wall incl. spawn {bwall:.1f}s"}
        except Exception as e:  # the context figure must not cost the baseline
            out["batched_numpy"] = {"error": repr(e)}
    return out


def strong_shard_entries(ctx, args, full_rate):
    """The metric's OWN per-rank shard, on one GPU: BASELINE's metric is "65,536 chains whole node", so at N = 8 / 4 / 2
    a GPU holds 8,192 / 16,384 / 32,768 chains x 1,024 -- 64 / 128 / 256 MiB per array, at or inside the 256-MiB Infinity
    Cache, ~50-200 us per launch.  Same sampler, same path (gradient a separate op), chains = the FIRST shard's global
    ids; per-launch times from HIP events.  `rate_vs_full_per_chain` = (steps/s per chain here) / (steps/s per chain at
    65,536): what strong scaling keeps per GPU; `predicted_node_steps_per_s` = N x this GPU's rate -- a PREDICTION from
    one GPU (no exchange on the path: chains are independent), not a measurement of N GPUs."""
    import torch

    out = {}
    ops = None
    for n_gpus in (8, 4, 2):
        n = C_CFG3 // n_gpus
        try:
            ss = make_cfg3_sampler(n, 0, ctx.device)
            ops = ss._ops
            for _ in range(max(3, args.warmup)):
                ss.sample()
            steps = max(args.steps, 20)
            el = ctx.timed_loop(ss.sample, steps)
            ops.timed = {"bk_leapfrog_kick_drift": [], "bk_target_diag_gaussian_grad": []}
            ops.timed_stride = 8
            for _ in range(4):
                ss.sample()
            torch.cuda.synchronize()
            timed, ops.timed, ops.timed_stride = ops.timed, None, 1
            kd = [a.elapsed_time(b) for a, b in timed["bk_leapfrog_kick_drift"]]
            gr = [a.elapsed_time(b) for a, b in timed["bk_target_diag_gaussian_grad"]]
            rate = float(n) * L_CFG3 * steps / el
            per_array_mib = n * D_CFG3 * 8 / 2**20
            e = {"chains": n, "share_of": f"N = {n_gpus}", "ms_per_draw": 1e3 * el / steps, "value": rate,
                 "unit": "leapfrog steps/sec (this GPU)", "us_per_leapfrog_step": 1e6 * el / steps / L_CFG3,
                 "rate_vs_full_per_chain": (rate / n) / (full_rate / C_CFG3),
                 "predicted_node_steps_per_s": rate * n_gpus, "predicted_speedup_vs_1gpu": rate * n_gpus / full_rate,
                 "hipgraph": bool(ss._use_graph), "array_mib": per_array_mib,
                 "regime": ("Infinity-Cache resident (3 arrays <= 256 MiB)" if 3 * per_array_mib <= 256 else
                            "partly cache resident" if per_array_mib <= 256 else "HBM"),
                 "path_bytes_model_GBps": rate * 56.0 * D_CFG3 / 1e9}
            if kd:
                e["kick_drift_us"] = 1e3 * sum(kd) / len(kd)
                e["kick_drift_GBps"] = 40.0 * D_CFG3 * ss._chain_tile / (sum(kd) / len(kd) * 1e-3) / 1e9
            if gr:
                e["gradient_us"] = 1e3 * sum(gr) / len(gr)
                e["gradient_GBps"] = 16.0 * D_CFG3 * ss._chain_tile / (sum(gr) / len(gr) * 1e-3) / 1e9
            out[str(n)] = e
            del ss
            torch.cuda.empty_cache()
        except Exception as ex:  # an extra must never cost the headline line
            if ops is not None:
                ops.timed, ops.timed_stride = None, 1
            out[str(n)] = {"error": repr(ex)}
    return out


def _device_index(local_rank):
    """LOCAL_RANK, unless BK_BENCH_SHARE_GPU=1: then ranks share the visible GPUs round-robin and
    rendezvous over gloo.  That mode exists only to exercise the N > 1 code path on a one-GPU box
    (RCCL refuses two ranks on one device); its numbers are not measurements."""
    if os.environ.get("BK_BENCH_SHARE_GPU"):
        import torch

        return local_rank % max(1, torch.cuda.device_count())
    return local_rank


class RankContext:
    """Device, process group and barrier of one rank."""

    def __init__(self, rank, local_rank, world):
        import torch
        import torch.distributed as dist

        self.rank, self.world = rank, world
        self.shared_gpu = bool(os.environ.get("BK_BENCH_SHARE_GPU")) and world > 1
        self.local = _device_index(local_rank)
        torch.cuda.set_device(self.local)
        self.device = torch.device("cuda", self.local)
        self.backend = None
        # BK_BENCH_FORCE_GROUP=1: a process group even for ONE rank (RCCL allows a single-rank group on one GPU) -- with
        # BK_DIST_FORCE_COLLECTIVES=1 the summaries' collectives then run on the backend the 8-GPU run will use
        self.grouped = world > 1 or (bool(os.environ.get("BK_BENCH_FORCE_GROUP")) and "MASTER_PORT" in os.environ)
        if self.grouped:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if self.shared_gpu:
                dist.init_process_group("gloo", rank=rank, world_size=world)
            else:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=self.device)
            self.backend = dist.get_backend()

    def barrier(self):
        import torch
        import torch.distributed as dist

        if self.grouped:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(self, seconds):
        import torch
        import torch.distributed as dist

        if self.world == 1:
            return seconds
        t = torch.tensor([seconds], dtype=torch.float64, device="cpu" if self.shared_gpu else self.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def collective_ranks(self):
        """The number of ranks, as counted by a collective that actually completed on the process group
        (an all_reduce of ones on the device; RCCL when the backend is nccl); 0 without a group."""
        import torch
        import torch.distributed as dist

        if not self.grouped:
            return 0
        one = torch.ones(1, dtype=torch.float64, device=self.device)
        dist.all_reduce(one)
        torch.cuda.synchronize()
        return int(round(float(one.item())))

    def timed_loop(self, fn, steps):
        """EXACTLY `steps` calls of fn bracketed by barrier + synchronize; max over ranks."""
        # No cyclic garbage collection inside the timed region: with PyTorch loaded a generation-2 collection is a 38-40
        # ms pause of the launching thread (measured: one draw's enqueue 1.5 -> 40 ms).  Mid-run the host is a draw or
        # two ahead of the GPU and the pause is absorbed; right after the barrier it is not, and WHERE the collection
        # lands depends on the allocation count since start-up (it moved into the first timed draw when the package
        # grew: 38.2 -> 42 ms per draw over 10 draws, kernel times unchanged).  Collected before, switched off during,
        # restored after.
        import gc

        gc.collect()
        was_enabled = gc.isenabled()
        gc.disable()
        try:
            self.barrier()
            t0 = time.perf_counter()
            for _ in range(steps):
                fn()
            self.barrier()
            elapsed = time.perf_counter() - t0
        finally:
            if was_enabled:
                gc.enable()
        return self.max_over_ranks(elapsed)

    def close(self):
        import torch.distributed as dist

        if self.grouped:
            dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------
# one rank
# ---------------------------------------------------------------------------------------------
def _claim_stdout():
    """stdout must carry ONE JSON line.  Libraries write there too -- RCCL prints a version banner through C stdio,
    which is flushed at exit when stdout is a pipe, i.e. AFTER the JSON line -- so from here on file descriptor 1 is
    stderr, and the line goes to the saved descriptor at the very end (_emit)."""
    sys.stdout.flush()
    real = os.dup(1)
    os.dup2(2, 1)
    return real


def _emit(real_stdout, obj):
    import ctypes

    sys.stdout.flush()
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    os.write(real_stdout, (json.dumps(obj) + "\n").encode())


def run_rank(args):
    real_stdout = _claim_stdout()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.only is None:
        cpu = cpu_baseline()  # before the GPU is initialised (worker processes never touch it)
    pinned = pin_rank(local_rank, world)  # before torch / HIP create their threads

    import torch

    import bayes_kit_amd as bk

    ctx = RankContext(rank, local_rank, world)
    device = ctx.device

    if args.only is not None:  # one secondary workload on its own (profiling runs)
        kw = {}
        if args.only == "cfg4":
            kw = dict(full_rhat=True, **({"chains": args.chains} if args.chains else {}),
                      **({"draws": args.steps} if args.steps_given else {}))
        from bench_secondary import run_secondary

        out = run_secondary(ctx, [args.only], **kw)[args.only]
        out.update({"n_gpus": world, "secondary_only": args.only})
        if rank == 0:
            _emit(real_stdout, out)
        ctx.close()
        return

    C = args.chains or C_CFG3
    D, L = D_CFG3, L_CFG3
    s = make_cfg3_sampler(C, rank * C, device, chain_tile=args.chain_tile,
                          prefetch_rng=False if args.no_rng_prefetch else None,
                          tune_placement=False if args.no_placement_tuning else None)
    Ct = s._chain_tile
    ops = s._ops

    for _ in range(args.warmup):
        s.sample()
    if not args.no_kernel_events:
        ops.timed = {"bk_leapfrog_kick_drift": [], "bk_target_diag_gaussian_grad": []}
        # every 8th launch of each is bracketed by HIP events (a sample spread evenly over the timed region): an event
        # record is a packet of its own on the queue, and 256 of them per draw cost the draw itself 4 %
        ops.timed_stride = int(os.environ.get("BK_BENCH_EVENT_STRIDE", "8"))
    with ClockSampler(ctx.local) as clocks:
        elapsed = ctx.timed_loop(s.sample, args.steps)
    timed, ops.timed = ops.timed, None
    event_stride, ops.timed_stride = ops.timed_stride, 1
    accept = s.accept_rate()

    # an N > 1 run is a measurement only if every rank is really in the RCCL group: an all_reduce of ones that completed
    # on it must count `world` ranks, or the run exits non-zero (ranks sharing a GPU over gloo are a code-path exercise
    # and say so)
    rccl_ranks = ctx.collective_ranks() if ctx.backend == "nccl" else 0
    if world > 1 and not ctx.shared_gpu and rccl_ranks != world:
        sys.stderr.write(f"bench.py: {world} ranks were launched but the RCCL group counted {rccl_ranks}\n")
        ctx.close()
        sys.exit(3)
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
        "comm_backend": ctx.backend,
        # ranks counted by an all_reduce that completed on the group, reported as RCCL's only when it ran on nccl
        "rccl_ranks": rccl_ranks,
        "shared_gpu": ctx.shared_gpu,
        "value_weak": value,
        "accept_rate": accept,
        "placement": s.placement,  # which allocation plays which role was chosen by timing (HMCDiag._tune_placement)
        # whole-path figure: 56*D algorithmic bytes per chain-step over the wall clock
        "path_hbm_frac": value / world * 56.0 * D / 1e9 / HBM_PEAK_GBPS,
        # which box and at which clocks (box-to-box spread of this pool is a few per cent: VERDICT r5 item 7)
        "device": _device_record(ctx.local, clocks),
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
            "traffic": _pmc_traffic(C, D)[0],
            "traffic_stamp": _pmc_traffic(C, D)[1],
            "avg_launch_ms": kd_ms,
            "launches": len(kd),
            "launches_in_timed_region": len(kd) * event_stride,
            "event_stride": event_stride,
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
            # (flat copies: a parser that keeps only the scalar keys of `roofline` still sees the second kernel)
            out["roofline"]["gradient_avg_launch_ms"] = g_ms
            out["roofline"]["gradient_achieved"] = 16.0 * D * Ct / (g_ms * 1e-3) / 1e9
            out["roofline"]["gradient_frac"] = 16.0 * D * Ct / (g_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS
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
        except Exception:  # context only
            out["roofline"]["device_copy_GBps"] = None
    if args.ess_draws > 0:
        try:
            # Second half of BASELINE.json's metric: ESS/sec.  Extra draws (outside the timed region
            # above), three tracked coordinates + the returned log density, ESS by the reference's
            # estimator (ess.py:52-69 -> bk_ess), summed over chains of the per-chain minimum.
            N = args.ess_draws
            series = torch.empty((4, N, C), dtype=torch.float64, device=device)
            it = {"n": 0}

            def one():
                th, lp = s.sample()
                n = it["n"]
                series[0, n], series[1, n], series[2, n], series[3, n] = th[:, 0], th[:, D // 2], th[:, D - 1], lp
                it["n"] = n + 1

            eel = ctx.timed_loop(one, N)
            ess = torch.stack([bk.ess(series[i]) for i in range(4)])
            # the reference's estimator returns N/IAT with IAT <= 0 possible for antithetic chains
            # (iat.py:151-152); for a throughput figure each series counts as at most N draws
            ess = torch.where(ess > 0, ess, torch.full_like(ess, float(N))).clamp(max=float(N))
            ess_min = ess.min(dim=0).values
            tot = bk.dist.sum_over_ranks(float(ess_min.sum().item()), device)
            tot2 = bk.dist.sum_over_ranks(float((ess_min * ess_min).sum().item()), device)
            n_ch = C * world
            # between-chain variance of the per-chain ESS
            var_ch = max(0.0, (tot2 - tot * tot / n_ch) / max(1, n_ch - 1))
            out["ess"] = {"draws": N, "tracked": ["theta[0]", "theta[D/2]", "theta[D-1]", "logp"],
                          "ess_per_sec": tot / eel, "mean_min_ess_per_chain": tot / n_ch,
                          # Monte-Carlo standard error of the figure: the sum over n_ch independent chains of a
                          # per-chain estimate with between-chain variance var_ch, over a wall time measured once
                          "ess_per_sec_mcse": (var_ch * n_ch) ** 0.5 / eel,
                          "sd_min_ess_per_chain": var_ch ** 0.5,
                          "mean_ess_per_tracked": [float(e.mean().item()) for e in ess],
                          "note": "whole job; per-chain minimum over the tracked series (each clipped to "
                                  "(0, N]: IAT <= 0 counts as N), summed over chains"}
            del series
        except Exception as e:
            out["ess"] = {"error": repr(e)}
    del s
    torch.cuda.empty_cache()

    # Strong scaling: the same metric with 65,536 chains per NODE (65,536 / N per GPU).  At N = 1 it
    # is the headline run itself.  At N = 8 each GPU holds 8,192 chains = 64 MiB arrays, which live
    # in the Infinity Cache: a different regime from the HBM-bound weak figure, reported beside it.
    if world == 1 or args.no_strong:
        out["value_strong"] = value if world == 1 else None
    else:
        try:
            first, n = bk.dist.shard(C_CFG3, rank, world)
            n -= n % 2
            ss = make_cfg3_sampler(n, first, device)
            for _ in range(args.warmup):
                ss.sample()
            sel = ctx.timed_loop(ss.sample, args.steps)
            out["value_strong"] = float(C_CFG3) * L * args.steps / sel
            out["strong"] = {"chains_per_gpu": n, "chains_per_node": C_CFG3, "ms_per_step": 1e3 * sel / args.steps,
                             "hipgraph": bool(ss._use_graph)}
            del ss
            torch.cuda.empty_cache()
        except Exception as e:
            out["value_strong"] = None
            out["strong"] = {"error": repr(e)}

    if world == 1 and not args.no_strong_shard:
        out["cfg3_strong_shard"] = strong_shard_entries(ctx, args, value)
    if not args.no_fused_extra:
        try:
            # Separately reported (never priced on the 56*D model): the same workload through the
            # built-in target's register-resident whole-draw kernel (bk_hmc_draw_gaussian: trajectory
            # + both kinetic energies + end-point log density in one pass over the state, momentum read
            # chain-major from the generator).  Same results bit for bit; bound by the fp64 vector rate.
            f = make_cfg3_sampler(C, rank * C, device, fused=True)
            for _ in range(3):
                f.sample()
            n_f = max(args.steps, 20)  # (a draw is ~1.3 ms: enough of them that the pipeline's fill and drain vanish)
            fel = ctx.timed_loop(f.sample, n_f) * args.steps / n_f
            # the draw kernel's own time, in a second pass: the event pairs around it are extra markers on
            # the main stream and cost the draw a few per cent, so they stay out of the loop timed above
            ops.timed = {"bk_hmc_draw_gaussian": []}
            for _ in range(args.steps):
                f.sample()
            torch.cuda.synchronize()
            tj = [a.elapsed_time(b) for a, b in ops.timed["bk_hmc_draw_gaussian"]]
            ops.timed = None
            tj_ms = sum(tj) / len(tj)
            # per element-step, individually rounded (no FMA): 3 mul + 2 add, + 1 mul by the metric unless it
            # is all ones (BASELINE's config: x * 1.0 is x bit for bit, so the kernel is launched without it)
            flop = (5.0 if f._metric_identity else 6.0) * D * C * L
            out["fused_builtin"] = {
                "what": "built-in DiagGaussian, whole draw (trajectory + energies) in registers; NOT the model-opaque "
                        "path, reported separately from `value`",
                "value": float(C) * world * L * args.steps / fel, "unit": "leapfrog steps/sec (whole job)",
                "ms_per_step": 1e3 * fel / args.steps, "bound": "fp64 VALU (no FMA: bit parity)",
                "trajectory_kernel_ms": tj_ms, "trajectory_kernel_tflops_fp64": flop / (tj_ms * 1e-3) / 1e12,
                "trajectory_kernel_frac_of_no_fma_ceiling":
                    flop / (tj_ms * 1e-3) / 1e12 / (FP64_VECTOR_PEAK_TFLOPS / 2),
                "draw_tflops_fp64": flop / (fel / args.steps) / 1e12,
                "fp64_vector_peak_tflops_spec": FP64_VECTOR_PEAK_TFLOPS,
                "flop_per_element_step": 5 if f._metric_identity else 6,
                "note": "peak counts an FMA as 2 flop; this kernel may not contract (bit-parity), ceiling 39.3; "
                        "trajectory_kernel_ms is measured while the next draw's generator shares the ALUs",
            }
            # the same draws with the metric multiplied in (6 flop per element-step): one entry an ulp off 1.0
            # keeps the chain what it was and makes the metric a non-identity
            m = torch.ones(D, dtype=torch.float64)
            m[0] = 1.0 + 2.0 ** -52
            f._metric = m
            for _ in range(2):
                f.sample()
            fel6 = ctx.timed_loop(f.sample, n_f)
            out["fused_builtin"]["ms_per_step_metric_multiplied_in"] = 1e3 * fel6 / n_f
            del f
            torch.cuda.empty_cache()
        except Exception as e:  # the extra must never cost the headline line
            ops.timed = None
            out["fused_builtin"] = {"error": repr(e)}
    if not args.no_secondary:
        # N = 1: every other BASELINE.json config.  N > 1: config 4, the one north_star shards with a
        # cross-rank reduction (32,768 chains per rank, R-hat / ESS over the process group).
        from bench_secondary import run_secondary

        out["secondary"] = run_secondary(ctx,
            ["cfg2", "cfg4", "mala", "torch_model", "cfg5"] if world == 1 else ["cfg4"], extras=args.full_secondary)
    if cpu is not None:
        out["cpu_baseline"] = cpu
    if pinned is not None:
        out["rank0_cpu_affinity"] = pinned
    if rank == 0:
        _emit(real_stdout, out)
    ctx.close()


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--chains", type=int, default=None, help="chains per GPU (default: 65,536)")
    ap.add_argument("--only", choices=["cfg2", "cfg4", "mala", "torch_model", "cfg5"], default=None,
                    help="run ONE secondary workload alone and print its record (profiling runs)")
    ap.add_argument("--chain-tile", type=int, default=None,
                    help="chains per Infinity-Cache tile (default: no tiling)")
    ap.add_argument("--no-rng-prefetch", action="store_true",
                    help="generate each draw's randomness in line instead of on the side stream (experiments)")
    ap.add_argument("--no-placement-tuning", action="store_true",
                    help="keep the scratch arrays in the roles they were allocated for (experiments)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--no-fused-extra", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the other configs' records (N=1 only)")
    ap.add_argument("--full-secondary", action="store_true",
                    help="also the provider variants (plugin / from-source / traced models) of configs 4 and 3: what "
                         "--only cfg4 / --only torch_model run; the default line keeps the entries SURVEY 8's rows need")
    ap.add_argument("--no-strong", action="store_true", help="skip the strong-scaling leg (N>1 only)")
    ap.add_argument("--no-strong-shard", action="store_true",
                    help="skip config 3 at the per-rank shard sizes of the metric (8,192 / 16,384 / 32,768 chains; N=1 only)")
    ap.add_argument("--ess-draws", type=int, default=200, help="extra draws for the ESS/sec figure (0 = skip)")
    args = ap.parse_args(argv)
    args.steps_given = any(a == "--steps" or a.startswith("--steps=") for a in (sys.argv[1:] if argv is None else argv))
    return args


def main(argv=None):
    args = parse_args(argv)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # launcher: start the ranks before anything in this process touches the GPU
        extra = {}
        have = _visible_gpus()
        if have < args.gpus and not os.environ.get("BK_BENCH_SHARE_GPU"):
            if have == 0:
                sys.exit("bench.py needs a GPU: none visible")
            extra["BK_BENCH_SHARE_GPU"] = "1"
            sys.stderr.write(f"bench.py: {have} GPU(s) visible for {args.gpus} ranks: ranks share them and rendezvous "
                             "over gloo (code-path exercise, not a measurement)\n")
        child = [sys.executable, os.path.abspath(__file__)] + (sys.argv[1:] if argv is None else list(argv))
        sys.exit(launch_ranks(args.gpus, child, extra))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and int(os.environ.get("RANK", "0")) == 0:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: using {world}\n")
    run_rank(args)


if __name__ == "__main__":
    main()
