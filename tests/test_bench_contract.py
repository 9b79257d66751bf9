"""bench.py contract pieces that do not need a GPU: CLI, the CPU-baseline leg, traffic file."""
import json
import os
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_cli_help_and_defaults():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True,
                         timeout=120)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup", "--only"):
        assert flag in out.stdout


def test_cpu_baseline_worker_is_the_oracle_and_reports_steps():
    # one tiny worker run: the thing bench.py times as `cpu_baseline` (kind "port")
    out = subprocess.run([sys.executable, "-m", "oracle.cpu_baseline", "0", "2", "3", "32", "4", "0.05", "7"], cwd=ROOT,
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    r = json.loads(out.stdout.strip().splitlines()[-1])
    assert r["steps"] == 2 * 3 * 4 and r["seconds"] > 0


def test_committed_traffic_matches_the_algorithmic_bytes(tmp_path, monkeypatch):
    sys.path.insert(0, ROOT)
    import bench

    t, stamp = bench._pmc_traffic(65536, 1024)
    assert t is not None  # the committed PMC passes are for the library (or at least the kernel source) as it stands
    assert stamp in ("library", "kernel source")
    algorithmic = 40.0 * 1024 * 65536
    assert abs(t / algorithmic - 1.0) < 0.02  # PMC traffic == algorithmic bytes (no re-reads)
    assert bench._pmc_traffic(4096, 128) == (None, None)
    # the figure is tied to the kernel's source text: any edit of the kernel nulls it
    src = open(bench.KD_SOURCE).read()
    edited = tmp_path / "bk_integrator.hip"
    edited.write_text(src.replace("double t = has_m ? m * g : g;", "double t = has_m ? g * m : g;"))
    assert bench.source_hash(str(edited)) != bench.source_hash()
    monkeypatch.setattr(bench, "KD_SOURCE", str(edited))
    monkeypatch.setattr(bench, "LIB_FILE", str(tmp_path / "another_build.so"))   # ... and a different binary
    assert bench._pmc_traffic(65536, 1024) == (None, None)


def test_device_identity_and_clock_sampler_read_sysfs_only(tmp_path, monkeypatch):
    """bench.py's box identity and clocks come from sysfs (no HIP): parsed from a fake card directory here."""
    sys.path.insert(0, ROOT)
    import bench

    dev = tmp_path / "renderD128" / "device"
    dev.mkdir(parents=True)
    (dev / "vendor").write_text("0x1002\n")
    (dev / "unique_id").write_text("abc123\n")
    (dev / "device").write_text("0x75a3\n")
    (dev / "pp_dpm_sclk").write_text("0: 132Mhz\n1: 2104Mhz *\n2: 2400Mhz\n")
    (dev / "pp_dpm_mclk").write_text("0: 900Mhz\n1: 2000Mhz *\n")
    monkeypatch.setattr(bench, "_amd_cards", lambda: [str(dev)])
    ident = bench.device_identity(0)
    assert ident["unique_id"] == "abc123" and ident["pci_device"] == "0x75a3" and ident["vbios"] is None
    assert bench.device_identity(1) is None
    import time

    with bench.ClockSampler(0, period=0.005) as c:
        time.sleep(0.05)
    sm = c.summary()
    assert sm["sclk_mhz"]["median"] == 2104.0 and sm["mclk_mhz"]["max"] == 2000.0 and sm["sclk_mhz"]["samples"] >= 2
    assert sm["power_w"] is None and "pp_dpm" in sm["source"]
    hw = dev / "hwmon" / "hwmon3"           # with hwmon: the clocks the part runs at, and the power
    hw.mkdir(parents=True)
    (hw / "freq1_input").write_text("1987000000\n")
    (hw / "freq2_input").write_text("2000000000\n")
    (hw / "power1_average").write_text("742000000\n")
    with bench.ClockSampler(0, period=0.005) as c:
        time.sleep(0.03)
    sm = c.summary()
    assert sm["sclk_mhz"]["median"] == 1987.0 and sm["power_w"]["max"] == 742.0 and "hwmon" in sm["source"]
    with bench.ClockSampler(3) as c:   # no such card: an empty summary, no thread
        pass
    assert c.summary()["sclk_mhz"] is None


def test_launcher_starts_one_process_per_rank_and_relays_rank0(capsys):
    # bench.py --gpus N without WORLD_SIZE: the launcher path (no GPU involved: a rank stub over gloo)
    sys.path.insert(0, ROOT)
    import io

    import bench

    stub = [sys.executable, os.path.join(ROOT, "tests", "bench_rank_stub.py")]
    buf = io.StringIO()
    rc = bench.launch_ranks(2, stub + ["ok"], {"OMP_NUM_THREADS": "1"}, timeout=240, out=buf)
    assert rc == 0
    r = json.loads(buf.getvalue().strip().splitlines()[-1])
    assert r == {"n_gpus": 2, "max_over_ranks": 2.0, "backend": "gloo"}
    assert "rank 1 done" in capsys.readouterr().err  # other ranks' output goes to stderr
    # a failing rank fails the launch and the other rank is not left hanging in the rendezvous
    buf = io.StringIO()
    rc = bench.launch_ranks(2, stub + ["fail1"], {"OMP_NUM_THREADS": "1"}, timeout=240, out=buf)
    assert rc == 3


def test_bench_refuses_multi_gpu_launch_without_any_gpu():
    # on a box with no GPU the launcher must say so instead of spawning ranks that cannot run
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], env=env,
                         capture_output=True, text=True, timeout=240)
    import torch

    if torch.cuda.device_count() == 0:
        assert out.returncode != 0 and "needs a GPU" in out.stderr


def test_launcher_counts_gpus_without_loading_hip(tmp_path, monkeypatch):
    """The launcher must not touch the GPU before its ranks do: the GPU count comes from the environment or
    from the kernel driver's topology files, never from torch / HIP in the launcher process."""
    code = (
        "import sys, os; sys.path.insert(0, %r); import bench\n"
        "os.environ['HIP_VISIBLE_DEVICES'] = '0,1,2'\n"
        "assert bench._visible_gpus() == 3\n"
        "os.environ['HIP_VISIBLE_DEVICES'] = ''\n"
        "assert bench._visible_gpus() == 0\n"
        "assert 'torch' not in sys.modules\n" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    # the topology reader: nodes with SIMDs are GPUs, CPU nodes have simd_count 0
    sys.path.insert(0, ROOT)
    import glob as globmod

    import bench

    for i, simd in enumerate([0, 256, 256, 0]):
        d = tmp_path / str(i)
        d.mkdir()
        (d / "properties").write_text(f"cpu_cores_count {64 if simd == 0 else 0}\nsimd_count {simd}\nmem_banks_count 1\n")
    real = globmod.glob
    monkeypatch.setattr(globmod, "glob", lambda pat: real(str(tmp_path / "*" / "properties")) if "kfd" in pat else real(pat))
    for var in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    assert bench._visible_gpus() == 2


def test_eight_ranks_on_the_cpu_stand_in_reduce_rhat_over_all_chains_and_get_their_own_cpus():
    """8-GPU readiness without an 8-GPU box: the launcher starts 8 ranks; each pins itself to its
    share of the allowed CPUs before anything else, runs config 4's algorithm on C chains with global chain ids
    rank*C.., and the R-hat over all 8 C chains through the process group equals one process holding every chain."""
    sys.path.insert(0, ROOT)
    import io

    import bench

    stub = [sys.executable, os.path.join(ROOT, "tests", "bench_rank_stub.py")]
    buf = io.StringIO()
    rc = bench.launch_ranks(8, stub + ["cfg4"], {"OMP_NUM_THREADS": "1"}, timeout=600, out=buf)
    assert rc == 0
    r = json.loads(buf.getvalue().strip().splitlines()[-1])
    assert r["n_gpus"] == 8 and r["rhat_over_chains"] == 8 * 6 and r["chain_id0"] == [6 * k for k in range(8)]
    assert r["rhat_equals_one_process"]
    if len(os.sched_getaffinity(0)) >= 8:
        assert r["affinity_disjoint"] and min(r["affinity_sizes"]) >= 1


def test_rank_cpu_sets_follow_the_numa_node_of_the_gpu():
    sys.path.insert(0, ROOT)
    import bench

    # two sockets of 8 CPUs, GPUs 0-3 on node 0 and 4-7 on node 1, the container allowed 12 of the 16 CPUs
    node_cpus = {0: range(0, 8), 1: range(8, 16)}
    allowed = set(range(2, 14))
    sets = [bench.rank_cpu_set(r, 8, affinity=allowed, numa_of_gpu=lambda i: i // 4, node_cpus=node_cpus) for r in range(8)]
    assert all(s and s <= allowed for s in sets)
    assert all(s <= set(node_cpus[r // 4]) for r, s in enumerate(sets))          # on the GPU's own node
    assert all(not (sets[i] & sets[j]) for i in range(8) for j in range(i))      # nobody shares a CPU
    # no topology: even slices of what is allowed; one rank: everything
    flat = [bench.rank_cpu_set(r, 4, affinity=set(range(8)), numa_of_gpu=lambda i: None) for r in range(4)]
    assert flat == [{0, 1}, {2, 3}, {4, 5}, {6, 7}]
    assert bench.rank_cpu_set(0, 1, affinity={3, 4}) == {3, 4}
    # more ranks than CPUs: ranks share, nobody gets an empty set
    assert all(bench.rank_cpu_set(r, 8, affinity={0, 1}, numa_of_gpu=lambda i: None) for r in range(8))
