#!/usr/bin/env python3
"""In-tree build of libbkhip.so (HIP kernels + C ABI) for gfx950.

    python bayes-kit_amd/build.py [--force]

hipcc cross-compiles without a GPU.  The .so lands in bayes_kit_amd/lib/ (git-ignored, but
it travels to the GPU box with the working tree).  -ffp-contract=off is REQUIRED: results
are specified as individually rounded IEEE operations (see include/bkhip.h).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT_DIR = os.path.join(HERE, "bayes_kit_amd", "lib")
OBJ_DIR = os.path.join(HERE, "build")
LIB = os.path.join(OUT_DIR, "libbkhip.so")
SOURCES = ["bk_rng.hip", "bk_integrator.hip", "bk_targets.hip", "bk_targets_gauss_lanes.hip", "bk_diag.hip", "bk_dr.hip", "bk_dense.hip", "bk_smc.hip", "bk_mala.hip", "bk_sort.hip", "bk_fft.hip"]
HEADERS = ["bk_common.hpp", "bk_rng.hpp", "bk_lanes.hpp", "bk_elementwise.hpp", "bk_mala_step.hpp", "ziggurat_tables.inc", os.path.join("..", "..", "include", "bkhip.h"),
           os.path.join("..", "..", "include", "bkhip_math.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-Wall", "-Wno-unused-function"]
# Per-file flags.  bk_rng.hip: machine-level loop-invariant code motion lifts every constant of exp() / log1p() (the
# rare ziggurat branches) out of the generator's pass loop into registers -- 149 instead of 93 VGPRs for k_zig_parallel<16>,
# 3 instead of 5 wavefronts per SIMD, 332 instead of 286 us at 65,536 x 1024 (profiles/r6_generator.md).
FILE_FLAGS = {"bk_rng.hip": ["-mllvm", "-disable-machine-licm"]}


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    os.makedirs(OUT_DIR, exist_ok=True)
    os.makedirs(OBJ_DIR, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)]
    hipcc = _hipcc()
    jobs = []
    objs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ_DIR, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            jobs.append([hipcc] + FLAGS + FILE_FLAGS.get(s, []) + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


PLUGIN_DIR = os.path.join(HERE, "..", "examples", "plugin_target")
PLUGINS = ["ar1_target", "funnel_target"]  # <name>.hip -> lib<name>.so
PLUGIN_LIB = os.path.join(PLUGIN_DIR, "libar1_target.so")


def build_example_plugin(force=False, verbose=True):
    """The example USER targets (plugin ABI bk_target_fn / bk_target_fn_n): shared libraries of their own, not
    part of libbkhip.so; built here so that the tests can load them on the GPU box."""
    libs = []
    for name in PLUGINS:
        src = os.path.abspath(os.path.join(PLUGIN_DIR, name + ".hip"))
        lib = os.path.abspath(os.path.join(PLUGIN_DIR, "lib" + name + ".so"))
        public = [os.path.abspath(os.path.join(HERE, "..", "include", h)) for h in ("bkhip.h", "bkhip_math.h")]
        if force or _stale(lib, [src] + public):  # (the plugins include the public headers: bk_exp lives there)
            cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                   src, "-o", lib]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        libs.append(lib)
    return libs[0]


C_HOST_SRC = os.path.join(HERE, "..", "examples", "c_host", "hmc_main.c")
C_HOST_BIN = os.path.join(HERE, "..", "examples", "c_host", "hmc_main")


def build_c_host_example(force=False, verbose=True):
    """examples/c_host/hmc_main.c: a many-chain HMC loop in plain C (gcc) on the C ABI alone."""
    src, exe = os.path.abspath(C_HOST_SRC), os.path.abspath(C_HOST_BIN)
    if force or _stale(exe, [src, LIB, os.path.join(HERE, "..", "include", "bkhip.h")]):
        rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
        cmd = ["gcc", "-std=gnu11", "-O2", "-D__HIP_PLATFORM_AMD__", src, "-I" + os.path.join(rocm, "include"),
               "-I" + os.path.abspath(os.path.join(HERE, "..", "include")), "-L" + OUT_DIR, "-lbkhip",
               "-L" + os.path.join(rocm, "lib"), "-lamdhip64",
               "-Wl,-rpath,$ORIGIN/../../bayes-kit_amd/bayes_kit_amd/lib", "-Wl,-rpath," + os.path.join(rocm, "lib"),
               "-o", exe]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return exe


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
    print(build_example_plugin(force="--force" in sys.argv))
    print(build_c_host_example(force="--force" in sys.argv))
