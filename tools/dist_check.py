"""Cell-sharded fit under torch.distributed.run: every rank builds its shard's engine, the ranks run the whole loop (ca_run)
on a shared eps stream, rank 0 writes the ELBO trace, the transport the engine reports, the replicated parameters of every
rank and the engine's all-reduce payload size to --out (JSON).  Used by tests/test_gpu_multi.py (2 ranks on 2 GPUs, or 2 ranks
on ONE GPU with --same-device for the peer-to-peer transport) -- started as a fresh process before anything touches the GPU.

  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P tools/dist_check.py \
         --transport p2p|rccl|host --cells 6000 --genes 700 --clones 5 --iters 6 --out /tmp/x.json [--same-device]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--transport", default="p2p")
    ap.add_argument("--cells", type=int, default=6000)
    ap.add_argument("--genes", type=int, default=700)
    ap.add_argument("--clones", type=int, default=5)
    ap.add_argument("--iters", type=int, default=6)
    ap.add_argument("--mc-samples", type=int, default=1)
    ap.add_argument("--seed", type=int, default=5)
    ap.add_argument("--same-device", action="store_true")
    ap.add_argument("--out", required=True)
    ap.add_argument("--extra-call-rank", type=int, default=-1,
                    help="fault injection: this rank makes ONE collective call (ca_elbo) more than its peers after the fit; the call must "
                         "end in CA_ERR_COMM within --comm-timeout-ms and the process must exit non-zero")
    ap.add_argument("--comm-timeout-ms", type=int, default=0)
    ap.add_argument("--variant-off", default="", help="comma-separated engine variants to switch off (engine.VARIANTS)")
    args = ap.parse_args()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = 0 if args.same_device else int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    import torch.distributed as dist
    from clonealign_amd import sharding
    from clonealign_amd.engine import HipEngine, comm_unique_id
    from tests._cases import eps_for, make_case
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    S = args.mc_samples
    case = make_case(seed=args.seed, N=args.cells, G=args.genes, C=args.clones, K=1, S=S)
    lo, hi = sharding.cell_range(args.cells, rank, world)
    kw = {}
    if world > 1:
        if args.transport == "p2p":
            def exchange(hd):
                box = [None] * world
                dist.all_gather_object(box, hd)
                return box
            kw["p2p_exchange"] = exchange
        elif args.transport == "rccl":
            box = [comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            kw["comm_id"] = box[0]
        else:
            import torch

            def gloo_sum(buf):
                t = torch.from_numpy(buf.copy())
                dist.all_reduce(t)
                buf[:] = t.numpy()
            kw["host_allreduce"] = gloo_sum
    eng = HipEngine(case["Y"][lo:hi], case["L"], case["psi0"][lo:hi], case["loc0"], 1, S, device=local, rank=rank, world=world,
                    comm_timeout_ms=args.comm_timeout_ms, variant_off=tuple(v for v in args.variant_off.split(",") if v), **kw)
    selftest_bad = eng.comm_selftest(27) if world > 1 else 0     # known-answer all-reduces on the transport in use, before the fit
    eps = np.stack([eps_for(S, args.genes, 300 + i) for i in range(2 + 2 * args.iters + 4)])
    trace = eng.run(eps, args.iters, 1e-12)
    finals = eng.final_elbo(eps[2 + 2 * args.iters:], 4)
    info = eng.info()
    if rank == args.extra_call_rank and world > 1:
        import time
        from clonealign_amd.engine import EngineError
        t0 = time.perf_counter()
        try:
            eng.elbo(eps[0])
        except EngineError as ex:
            print(f"[rank {rank}] dist_check: extra collective call failed after {time.perf_counter() - t0:.2f} s with code {ex.code}: {ex}",
                  file=sys.stderr, flush=True)
            raise SystemExit(7)
        print(f"[rank {rank}] dist_check: the extra collective call RETURNED (no peer took part in it)", file=sys.stderr, flush=True)
        raise SystemExit(0)
    rep = {n: eng.get(n).tolist() for n in ("W", "loc", "ls", "alpha_unconstr", "v")}
    mine = dict(rank=rank, trace=trace.tolist(), finals=finals.tolist(), transport=info["transport_name"], red_n=int(info["red_n"]),
                fwd_block_cells=int(info["fwd_block_cells"]), fwd_balanced=int(info.get("fwd_balanced", 0)), selftest_bad=int(selftest_bad),
                rep=rep, psi_head=eng.get("psi")[:5, 0].tolist(), lo=lo, hi=hi)
    eng.close()
    if world > 1:
        box = [None] * world
        dist.all_gather_object(box, mine)
        dist.barrier()
        dist.destroy_process_group()
    else:
        box = [mine]
    if rank == 0:
        json.dump(dict(world=world, ranks=box, plan=sharding.reduce_plan(args.genes, args.clones, 1, 0, S)), open(args.out, "w"))


if __name__ == "__main__":
    main()
