"""One rank of tools/first_contact_8gpu.sh: the exchange step of the LDS E-step on a real multi-GPU node, checked.

Launched by torch.distributed.run with one rank per GPU.  Each rank runs the E-step on its own shard of 512 sequences
(T = 200, n = 10: BASELINE configs[2] at N = 8), reduces its statistics on the device, all-reduces the packed buffer
through the exchange the environment selects (RCCL, or the IPC mailbox kernel with SVAE_BENCH_ALLREDUCE=mailbox) and
asserts:  world size and backend;  every rank holds the SAME BITS after the all-reduce;  they equal the float64 sum
of the per-rank buffers (gathered before the reduction) to rounding.  Rank 0 writes one JSON."""
import argparse
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--rounds", type=int, default=20)
    a = ap.parse_args()
    rank, local, world = int(os.environ["RANK"]), int(os.environ["LOCAL_RANK"]), int(os.environ["WORLD_SIZE"])
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", device_id=dev)
    assert dist.get_world_size() == world and dist.get_backend() == "nccl"
    assert torch.cuda.device_count() >= world, "one GPU per rank"
    from svae_amd.lds.lds_inference import LDSEStepPlan
    from svae_amd.lds.synthetic_data import rand_lds_natparam, rand_node_potentials
    from svae_amd.parallel import allreduce_global_stats, use_mailbox_allreduce
    mailbox = os.environ.get("SVAE_BENCH_ALLREDUCE", "") == "mailbox"
    ar = use_mailbox_allreduce(4 * 10 * 10 + 10 + 2) if mailbox else None
    B, T, n = 512, 200, 10
    init, pair = rand_lds_natparam(n, np.random.default_rng(0))                      # replicated
    nJ, nh = rand_node_potentials((B, T, n), np.random.default_rng(1000 + rank))     # this rank's shard
    t = lambda x: torch.as_tensor(np.asarray(x, float), dtype=torch.float64, device=dev).contiguous()
    args = [t(init[0]), t(init[1]), t(init[2]).reshape(1), t(pair[0]), t(pair[1]), t(pair[2]), t(pair[3]).reshape(1),
            t(nJ), t(nh), None]
    plan = LDSEStepPlan(B, T, n, dev)
    worst, identical = 0.0, True
    for it in range(a.rounds):
        plan.launch(*args)
        mine = plan.reduce().clone()
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)                                  # the per-rank buffers, for the reference sum
        want = torch.stack(parts).sum(0)
        got = allreduce_global_stats(mine.clone())
        torch.cuda.synchronize()
        every = [torch.empty_like(got) for _ in range(world)]
        dist.all_gather(every, got)
        identical = identical and all(torch.equal(e, every[0]) for e in every)
        worst = max(worst, float(((got - want).abs() / want.abs().clamp_min(1e-300)).max()))
    if ar is not None:
        ar.check()
    plan.check_info()
    ok = identical and worst < 1e-13
    res = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device=dev)
    dist.all_reduce(res, op=dist.ReduceOp.MIN)
    if rank == 0:
        out = {"world_size": world, "backend": dist.get_backend(), "exchange": "ipc mailbox" if mailbox else "RCCL",
               "rounds": a.rounds, "bit_identical_on_every_rank": bool(identical), "max_rel_vs_fp64_sum": worst,
               "ok_on_every_rank": bool(res.item() == 1.0),
               "devices": [torch.cuda.get_device_name(i) for i in range(world)]}
        with open(a.out, "w") as f:
            json.dump(out, f, indent=1)
        print(json.dumps(out))
    dist.barrier()
    dist.destroy_process_group()
    if not ok:
        raise SystemExit("first contact: exchange step check FAILED on rank %d" % rank)


if __name__ == "__main__":
    main()
