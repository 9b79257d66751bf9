"""BASELINE.json config 4 as a user would run it: delayed-rejection HMC on Neal's funnel, many chains
per GPU, streaming R-hat over ALL chains of ALL ranks and ESS from a few recorded coordinates.

    python examples/funnel_drghmc_rhat.py                                  # one GPU
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 \\
        examples/funnel_drghmc_rhat.py                                     # one process per GPU (RCCL)

Chains are sharded by GLOBAL chain id (Philox key = (seed, chain id)), so chain c produces the same
draws whatever the number of GPUs; the only collective is the few-KB all_gather inside rhat().
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "bayes-kit_amd")]

import torch

import bayes_kit_amd as bk

rank, local_rank, world = bk.dist.init_from_env()
torch.cuda.set_device(local_rank)

D, total_chains, draws = 101, 32768 * world, 200
first, n = bk.dist.shard(total_chains)                     # this rank's contiguous block of chain ids
sampler = bk.DrGhmcDiag(bk.Funnel(D), max_proposals=3, leapfrog_step_sizes=[0.2, 0.05, 0.0125],
                        leapfrog_step_counts=[10, 40, 160], damping=0.1, chains=n, chain_id0=first, seed=20242)
moments = bk.RunningMoments(D, n)                          # Welford per chain and dimension, on the device
recorder = bk.DrawRecorder([0, 1, D - 1], draws, n)        # full series of three coordinates (+ logp) for ESS

for _ in range(draws):
    theta, logp = sampler.sample()                         # (n, D) and (n,) device tensors
    moments.update(theta)
    recorder.record(theta, logp)

rhat = moments.rhat()                                      # over all chains of all ranks
ess = recorder.ess().clamp(min=0.0)                        # (4, n): per tracked series and chain
total_ess = bk.dist.sum_over_ranks(float(ess.min(dim=0).values.sum()), torch.device("cuda", local_rank))
if rank == 0:
    print(f"{total_chains} chains x {draws} draws on {world} GPU(s)")
    print(f"R-hat: v = theta[0] {float(rhat[0]):.3f}, max over dims {float(rhat.max()):.3f}")
    print(f"sum over chains of min-ESS: {total_ess:.3g}")
