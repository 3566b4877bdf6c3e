"""world_size-2 gloo tests (CPU): the cell-sharded evaluation -- what is local, what is replicated, what is
all-reduced and where -- reproduces the single-process fit.  The oracle stands in for the engine (the HIP
engine implements the same plan and is checked against a single-GPU run in test_gpu_sharding.py)."""
import os
import socket

import numpy as np
import pytest

from clonealign_amd.sharding import cell_range, reduce_plan
from oracle.fused_numpy import FusedModel
from tests._cases import eps_for, make_case

CASES = {"k1": dict(N=61, G=23, C=3, K=1), "k2p1s2": dict(N=50, G=19, C=4, K=2, P=1, S=2, extra=True),
         "k0": dict(N=40, G=17, C=3, K=0)}
N_ITER = 6


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(model, S, G):
    model.gamma_init(eps_for(S, G, 0))
    trace = [model.elbo(eps_for(S, G, 1))]
    for i in range(1, N_ITER + 1):
        model.step(eps_for(S, G, 2 * i))
        trace.append(model.elbo(eps_for(S, G, 2 * i + 1)))
    return np.array(trace)


def _worker(rank, world, port, name, outdir):
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    case = make_case(seed=5, **CASES[name])
    N = case["Y"].shape[0]
    lo, hi = cell_range(N, rank, world)
    shard = dict(case)
    for k in ("Y", "psi0", "X", "extra_loglik"):
        if shard.get(k) is not None:
            shard[k] = shard[k][lo:hi]
    m = FusedModel(**shard)
    sizes = []

    def allreduce(v):
        t = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float64).copy())
        dist.all_reduce(t)
        sizes.append(t.numel())
        return t.numpy()

    m.set_allreduce(allreduce)
    trace = _run(m, m.S, m.G)
    st = m.get_state()
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), trace=trace, lo=lo, hi=hi, sizes=np.array(sizes), **st)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("name", list(CASES))
def test_two_rank_sharded_fit_equals_single_process(name, tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, name, str(tmp_path)), nprocs=2, join=True)
    case = make_case(seed=5, **CASES[name])
    ref = FusedModel(**case)
    tr = _run(ref, ref.S, ref.G)
    sr = ref.get_state()
    r = [np.load(tmp_path / f"rank{i}.npz") for i in range(2)]
    for i in range(2):
        np.testing.assert_allclose(r[i]["trace"], tr, rtol=1e-10)          # every rank sees the global ELBO
        for n in ("W", "v", "beta", "alpha_unconstr", "loc", "ls"):          # replicated variables
            np.testing.assert_allclose(r[i][n], sr[n], rtol=1e-7, atol=1e-9)
            np.testing.assert_array_equal(r[i][n], r[0][n])                  # bit-identical across ranks
        lo, hi = int(r[i]["lo"]), int(r[i]["hi"])
        for n in ("psi", "gamma_logits"):                                    # cell-local variables
            np.testing.assert_allclose(r[i][n], sr[n][lo:hi], rtol=1e-7, atol=1e-8)
    # the train-pass payload is what sharding.reduce_plan says the engine reduces
    plan = reduce_plan(ref.G, ref.C, ref.K, ref.P, ref.S)
    assert int(r[0]["sizes"].max()) == plan["total"] - 3


def test_cell_range_partitions_exactly():
    for N in (1, 7, 100, 100_000):
        for W in (1, 2, 3, 8):
            if W > N:
                continue
            rs = [cell_range(N, r, W) for r in range(W)]
            assert rs[0][0] == 0 and rs[-1][1] == N
            assert all(rs[i][1] == rs[i + 1][0] for i in range(W - 1))
            sz = [b - a for a, b in rs]
            assert max(sz) - min(sz) <= 1
    with pytest.raises(ValueError):
        cell_range(10, 2, 2)


def _bench_mod():
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    return b


def test_agreed_calls_is_a_pure_function_of_agreed_numbers():
    b = _bench_mod()
    assert b.agreed_calls(60.0, 1.4) == 43 and b.agreed_calls(60.0, 100.0) == 1 and b.agreed_calls(60.0, 0.0) == 1
    assert b.agreed_calls(1e9, 1.0) == 4096 and b.agreed_calls(60.0, float("nan")) == 1


def _preheat_worker(rank, world, port, outdir):
    """bench.py's pre-heat with a collective call whose duration differs per rank and per call: each rank's OWN clock would stop
    the loop at a different count (the r03 failure); the agreed count must be equal, and no collective is left unmatched."""
    import time
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    b = _bench_mod()
    made = []
    rng = np.random.default_rng(rank)

    def call():
        time.sleep(float(rng.uniform(0.0, 0.004)) * (1 + rank))     # host-side skew in front of the collective
        t = torch.ones(4)
        dist.all_reduce(t)                                           # the collective: returns only when every rank is in it
        made.append(float(t[0]))
    counts = [b.run_agreed_calls(call, 40.0, dist) for _ in range(5)]
    dist.barrier()
    np.savez(os.path.join(outdir, f"pre{rank}.npz"), counts=np.array(counts), made=len(made))
    dist.destroy_process_group()


def test_bench_preheat_call_count_is_agreed_over_two_ranks(tmp_path):
    import torch.multiprocessing as mp
    mp.spawn(_preheat_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    a, b = (np.load(tmp_path / f"pre{i}.npz") for i in range(2))
    assert np.array_equal(a["counts"], b["counts"]) and int(a["made"]) == int(b["made"]) == int(a["counts"].sum())
    assert (a["counts"] >= 2).all()


def _bare_bench(args, **env_extra):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "CLONEALIGN_BENCH_DEVICE")}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(root, "bench.py"), *args], env=env, capture_output=True, text=True, timeout=600, cwd=root)


def test_bare_bench_with_more_ranks_than_devices_is_refused_not_run_on_one():
    """`python bench.py --gpus 8` with no launcher around it ran ONE rank and printed n_gpus 1.  It must start its
    ranks itself or fail: here (no GPU) it refuses before any rank starts, and a WORLD_SIZE that contradicts --gpus is refused too."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("a multi-GPU box: the GPU suite runs the bare command for real")
    r = _bare_bench(["--gpus", "2", "--steps", "2"])
    assert r.returncode == 2 and "GPU(s) visible" in r.stderr and "{" not in r.stdout, (r.returncode, r.stderr[-500:])
    r = _bare_bench(["--gpus", "4", "--steps", "2"], WORLD_SIZE="1", RANK="0")
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr and "{" not in r.stdout, (r.returncode, r.stderr[-500:])
    r = _bare_bench(["--gpus", "1", "--steps", "2"], WORLD_SIZE="2", RANK="0")
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr, (r.returncode, r.stderr[-500:])


def test_bare_bench_launcher_starts_fresh_ranks_and_exits_with_the_worst_code():
    """The launcher half of bench.py on a box without GPUs: with the plumbing override it starts its two ranks (fresh processes, RANK /
    WORLD_SIZE / MASTER_* set), each of which refuses to run without an MI355X -- the launcher reports both and exits non-zero."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a box WITHOUT a GPU (the ranks must fail)")
    r = _bare_bench(["--gpus", "2", "--steps", "2"], CLONEALIGN_BENCH_DEVICE="0")
    assert r.returncode != 0 and "{" not in r.stdout
    assert r.stderr.count("needs an MI355X") == 2 and "launcher: rank 0 exited" in r.stderr and "launcher: rank 1 exited" in r.stderr, r.stderr[-800:]


def test_reduce_plan_of_a_series_capable_engine_is_a_prefix_layout():
    """Round 6: an engine whose shape takes the series form lays the reduction buffer out so that what a series pass reduces (cell sums, Y^T psi, the backward
    moments, a max |psi| slot per rank) is a contiguous PREFIX and the per-gene sums come last; the classic layout is unchanged."""
    G, C, W = 5000, 8, 8
    classic = reduce_plan(G, C, 1, 0, 1)
    assert classic["total"] == 3 + C + G * 2 + G == 15011 and classic["gene"][0] == 3 + C
    ser = reduce_plan(G, C, 1, 0, 1, series=True, world=W)
    Gp = 5120
    assert ser["ytpsi"] == (3 + C, G) and ser["moments"] == (3 + C + Gp, 32 * 22 * 8) and ser["max_psi"] == (3 + C + Gp + 5632, W)
    assert ser["series_total"] == ser["gene"][0] == 3 + C + Gp + 5632 + W == 10771
    assert ser["total"] == ser["series_total"] + 2 * G
    # every region inside the buffer, in order, without overlap
    regions = sorted(v for k, v in ser.items() if isinstance(v, tuple))
    for (o0, n0), (o1, _n1) in zip(regions, regions[1:]):
        assert o0 + n0 <= o1
    assert regions[-1][0] + regions[-1][1] == ser["total"]
