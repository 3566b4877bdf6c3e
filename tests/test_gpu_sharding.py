"""The engine's cell-sharded path on ONE GPU: two handles (rank 0/1 of world 2) on two host threads, joined
by the host all-reduce hook of the C ABI (ca_set_host_allreduce).  Everything but the RCCL transport itself
is the code the 8-GPU bench runs; the result must match a single-handle fit of all cells."""
import threading

import numpy as np
import pytest

from clonealign_amd.sharding import cell_range
from tests._cases import eps_for, make_case

pytestmark = pytest.mark.gpu


class _HostAllreduce:
    def __init__(self, world):
        self.world = world
        self.bar = threading.Barrier(world)
        self.slots = [None] * world
        self.sizes = []

    def make(self, rank):
        def fn(buf):
            self.slots[rank] = buf.copy()
            self.bar.wait()
            tot = self.slots[0].copy()
            for r in range(1, self.world):
                tot += self.slots[r]
            self.bar.wait()
            buf[:] = tot
            if rank == 0:
                self.sizes.append(len(buf))
        return fn


@pytest.mark.parametrize("kw", [dict(N=301, G=140, C=3, K=1), dict(N=260, G=90, C=4, K=2, P=1, S=2, extra=True)])
def test_two_shards_on_one_gpu_match_single_handle(kw):
    from clonealign_amd.engine import HipEngine
    case = make_case(seed=9, **kw)
    N, G, S = case["Y"].shape[0], case["Y"].shape[1], case["S"]
    n_iter = 8

    def drive(eng):
        eng.gamma_init(eps_for(S, G, 0))
        tr = [eng.elbo(eps_for(S, G, 1))]
        for i in range(1, n_iter + 1):
            eng.step(eps_for(S, G, 2 * i))
            tr.append(eng.elbo(eps_for(S, G, 2 * i + 1)))
        return np.array(tr), eng.get_state()

    ref = HipEngine(**case)
    tr_ref, st_ref = drive(ref)
    ref.close()

    ar = _HostAllreduce(2)
    out = [None, None]

    def worker(rank):
        lo, hi = cell_range(N, rank, 2)
        shard = dict(case)
        for k in ("Y", "psi0", "X", "extra_loglik"):
            if shard.get(k) is not None:
                shard[k] = shard[k][lo:hi]
        eng = HipEngine(**shard, rank=rank, world=2, host_allreduce=ar.make(rank))
        out[rank] = drive(eng) + ((lo, hi),)
        eng.close()

    ts = [threading.Thread(target=worker, args=(r,)) for r in range(2)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    for r in range(2):
        tr, st, (lo, hi) = out[r]
        assert np.abs(tr - tr_ref).max() <= 2e-6 * np.abs(tr_ref).max()
        for n in ("W", "v", "beta", "alpha_unconstr", "loc", "ls"):
            assert np.abs(st[n] - st_ref[n]).max(initial=0) <= 2e-5 * max(np.abs(st_ref[n]).max(initial=0), 1e-30), n
            assert np.array_equal(st[n], out[0][1][n])          # replicas stay bit-identical
        for n in ("psi", "gamma_logits"):
            assert np.abs(st[n] - st_ref[n][lo:hi]).max(initial=0) <= 2e-5 * max(np.abs(st_ref[n]).max(initial=0), 1e-30), n


@pytest.mark.parametrize("kw", [dict(N=301, G=140, C=3, K=1), dict(N=1300, G=700, C=8, K=1), dict(N=260, G=90, C=4, K=2, P=1),
                                dict(N=777, G=333, C=5, K=1, P=1), dict(N=515, G=97, C=2, K=2)])
def test_two_shards_whole_loop_matches_single_handle(kw):
    """ca_run / ca_iterate (the fused two-eps sweep, what bench.py drives on N GPUs) on two shards through the host hook
    against the single-handle loop: same ELBO trace, same parameters, replicas bit-identical, and the number of
    collectives per iteration the design promises."""
    from clonealign_amd.engine import HipEngine
    from clonealign_amd.rng import EpsStream
    case = make_case(seed=11, **kw)
    N, G = case["Y"].shape[0], case["Y"].shape[1]
    n_iter = 6

    counts = {}

    def drive(eng, ar=None, rank=None):
        c0 = len(ar.sizes) if ar else 0
        tr = eng.run(EpsStream(77, 1, G), n_iter, 1e-12)
        c1 = len(ar.sizes) if ar else 0
        last = eng.iterate(3, np.stack([eps_for(1, G, 500 + i) for i in range(6)]))
        c2 = len(ar.sizes) if ar else 0
        fin = eng.final_elbo(EpsStream(78, 1, G), 3)
        if ar and rank == 0:
            counts["run"], counts["iterate"], counts["mfma"] = c1 - c0, c2 - c1, eng.info()["bwd_mfma"]
        return np.asarray(tr), last, np.asarray(fin), eng.get_state()

    ref = HipEngine(**case)
    tr_ref, last_ref, fin_ref, st_ref = drive(ref)
    ref.close()

    ar = _HostAllreduce(2)
    out = [None, None]
    err = []

    def worker(rank):
        try:
            lo, hi = cell_range(N, rank, 2)
            shard = dict(case)
            for k in ("Y", "psi0", "X", "extra_loglik"):
                if shard.get(k) is not None:
                    shard[k] = shard[k][lo:hi]
            eng = HipEngine(**shard, rank=rank, world=2, host_allreduce=ar.make(rank))
            out[rank] = drive(eng, ar, rank) + ((lo, hi),)
            eng.close()
        except Exception as e:  # noqa: BLE001
            err.append(e)
            ar.bar.abort()

    ts = [threading.Thread(target=worker, args=(r,)) for r in range(2)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not err, err
    # ONE collective per iteration in the loop: the monitor pass's cell sums travel with the next train pass's gene sums
    # (ca_run: n_iter merged + the last, unfused monitor pass; ca_iterate(3): first train pass, 2 merged, last monitor pass)
    # The general backward sweep (here K + P = 3) keeps the plain sequence: one collective per pass.
    if counts.pop("mfma"):
        assert counts == {"run": n_iter + 1, "iterate": 4}, counts
    else:
        assert counts == {"run": 2 * n_iter + 1, "iterate": 6}, counts
    for r in range(2):
        tr, last, fin, st, (lo, hi) = out[r]
        assert len(tr) == n_iter + 1
        assert np.abs(tr - tr_ref).max() <= 2e-6 * np.abs(tr_ref).max()
        assert abs(last - last_ref) <= 2e-6 * abs(last_ref)
        assert np.abs(fin - fin_ref).max() <= 2e-6 * np.abs(fin_ref).max()
        assert np.array_equal(tr, out[0][0]) and last == out[0][1]      # every rank sees the same ELBOs
        for n in ("W", "v", "beta", "alpha_unconstr", "loc", "ls"):
            assert np.abs(st[n] - st_ref[n]).max(initial=0) <= 5e-5 * max(np.abs(st_ref[n]).max(initial=0), 1e-30), n
            assert np.array_equal(st[n], out[0][3][n])
        for n in ("psi", "gamma_logits"):
            assert np.abs(st[n] - st_ref[n][lo:hi]).max(initial=0) <= 5e-5 * max(np.abs(st_ref[n]).max(initial=0), 1e-30), n


@pytest.mark.parametrize("kw", [dict(N=640, G=300, C=6, K=1, S=2), dict(N=9000, G=500, C=4, K=1, S=2, extra=True)])
def test_two_shards_whole_loop_with_two_mc_samples_matches_single_handle(kw):
    """mc_samples = 2 sharded: since round 4 the monitor pass's two samples share a forward sweep with the next train pass's two, so a pending
    monitor tail now meets a (two-sample) backward sweep's all-reduce on this path too.  ca_run, ca_iterate and the final ELBOs on two shards
    through the host hook against the single handle; replicas bit-identical."""
    from clonealign_amd.engine import HipEngine
    from clonealign_amd.rng import EpsStream
    case = make_case(seed=13, **kw)
    N, G = case["Y"].shape[0], case["Y"].shape[1]
    n_iter = 5

    def drive(eng):
        assert eng.info()["fused_sweep"] == 1
        tr = eng.run(EpsStream(81, 2, G), n_iter, 1e-12)
        last = eng.iterate(3, np.stack([eps_for(2, G, 600 + i) for i in range(6)]))
        fin = eng.final_elbo(EpsStream(82, 2, G), 3)
        return np.asarray(tr), last, np.asarray(fin), eng.get_state()

    ref = HipEngine(**case)
    tr_ref, last_ref, fin_ref, st_ref = drive(ref)
    ref.close()
    ar = _HostAllreduce(2)
    out, err = [None, None], []

    def worker(rank):
        try:
            lo, hi = cell_range(N, rank, 2)
            shard = dict(case)
            for k in ("Y", "psi0", "X", "extra_loglik"):
                if shard.get(k) is not None:
                    shard[k] = shard[k][lo:hi]
            eng = HipEngine(**shard, rank=rank, world=2, host_allreduce=ar.make(rank))
            out[rank] = drive(eng) + ((lo, hi),)
            eng.close()
        except Exception as e:  # noqa: BLE001
            err.append(e)
            ar.bar.abort()

    ts = [threading.Thread(target=worker, args=(r,)) for r in range(2)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not err, err
    for r in range(2):
        tr, last, fin, st, (lo, hi) = out[r]
        assert len(tr) == n_iter + 1
        assert np.abs(tr - tr_ref).max() <= 2e-6 * np.abs(tr_ref).max()
        assert abs(last - last_ref) <= 2e-6 * abs(last_ref)
        assert np.abs(fin - fin_ref).max() <= 2e-6 * np.abs(fin_ref).max()
        assert np.array_equal(tr, out[0][0]) and last == out[0][1]
        for n in ("W", "v", "beta", "alpha_unconstr", "loc", "ls"):
            assert np.abs(st[n] - st_ref[n]).max(initial=0) <= 5e-5 * max(np.abs(st_ref[n]).max(initial=0), 1e-30), n
            assert np.array_equal(st[n], out[0][3][n])
        for n in ("psi", "gamma_logits"):
            assert np.abs(st[n] - st_ref[n][lo:hi]).max(initial=0) <= 5e-5 * max(np.abs(st_ref[n]).max(initial=0), 1e-30), n


def test_rccl_communicator_of_one_rank_runs():
    """ncclCommInitRank with one rank on the visible GPU: the RCCL code path (dlopen, init, all-reduce, destroy)."""
    from clonealign_amd.engine import HipEngine, comm_unique_id
    case = make_case(seed=2, N=200, G=64, C=3, K=1)
    a = HipEngine(**case)
    ea = a.elbo(eps_for(1, 64, 3))
    a.close()
    b = HipEngine(**case, rank=0, world=1)
    b._ck(b.lib.ca_comm_init(b.h, comm_unique_id()))   # 1-rank communicator: every pass now goes through ncclAllReduce
    eb = b.elbo(eps_for(1, 64, 3))
    b.close()
    assert ea == eb


class _ThreadExchange:
    """An in-process all-gather of byte strings for W host threads (the p2p_exchange hook of HipEngine): what an R session that
    drives several devices from worker threads would use instead of torch.distributed / MPI."""
    def __init__(self, world):
        self.world, self.bar, self.box = world, threading.Barrier(world), [None] * world

    def make(self, rank):
        def fn(payload):
            self.box[rank] = payload
            self.bar.wait()
            out = list(self.box)
            self.bar.wait()
            return out
        return fn


def _n_gpus():
    import torch
    return torch.cuda.device_count()


@pytest.mark.skipif(_n_gpus() < 2, reason="needs two GPUs: one process may hold ONE rank per device (two ranks of one process on one "
                                           "device are refused, see test_two_ranks_of_one_process_on_one_device_are_refused)")
@pytest.mark.parametrize("world", [2, 3])
def test_handles_of_one_process_join_through_the_device_transport(world):
    """SURVEY section 8b "one process / 8 devices": the R session that calls clonealign() is ONE process.  W handles on W host
    threads of this process, one per device, are joined by the one-shot peer-to-peer all-reduce itself -- the handles of
    the own process are mapped by address, not through IPC (which cannot open a handle in the process that made it) -- and
    the fit equals the single-handle fit; replicas bit-identical."""
    from clonealign_amd.engine import HipEngine
    from clonealign_amd.rng import EpsStream
    if _n_gpus() < world:
        pytest.skip(f"needs {world} GPUs")
    case = make_case(seed=21, N=1500, G=400, C=5, K=1)
    N, G = case["Y"].shape[0], case["Y"].shape[1]
    ref = HipEngine(**case)
    tr_ref = np.asarray(ref.run(EpsStream(5, 1, G), 6, 1e-12))
    st_ref = ref.get_state()
    ref.close()
    ex = _ThreadExchange(world)
    out, err = [None] * world, []

    def worker(rank):
        try:
            lo, hi = cell_range(N, rank, world)
            shard = dict(case)
            for k in ("Y", "psi0"):
                shard[k] = shard[k][lo:hi]
            eng = HipEngine(**shard, rank=rank, world=world, device=rank, p2p_exchange=ex.make(rank), comm_timeout_ms=20000)
            assert eng.info()["transport_name"] == "p2p"
            tr = np.asarray(eng.run(EpsStream(5, 1, G), 6, 1e-12))
            us = eng.comm_benchmark("p2p", 50)
            out[rank] = (tr, eng.get_state(), us)
            eng.close()
        except BaseException as e:  # noqa: BLE001
            err.append((rank, repr(e)))
            ex.bar.abort()
    ts = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not err, err
    for r in range(world):
        tr, st, us = out[r]
        assert np.abs(tr - tr_ref).max() <= 1e-6 * np.abs(tr_ref).max()
        assert np.array_equal(tr, out[0][0]) and 0 < us < 1e5
        for n in ("W", "v", "alpha_unconstr", "loc", "ls"):
            assert np.array_equal(st[n], out[0][1][n]), n
            assert np.abs(st[n] - st_ref[n]).max(initial=0) <= 2e-5 * max(np.abs(st_ref[n]).max(initial=0), 1e-30), n


def test_two_ranks_of_one_process_on_one_device_are_refused():
    """Two handles of ONE process on ONE device cannot be joined by the device transport: a device-wide synchronising runtime call made
    for one of them (hipFree of a grown buffer, hipMalloc) waits for every kernel on the device, including the other's all-reduce
    kernel, which waits for this rank -- measured as a time-out of the pair's second all-reduce.  ca_p2p_connect says so at once, on
    both ranks (two-phase setup: nobody is left waiting), and both engines close cleanly."""
    from clonealign_amd.engine import EngineError, HipEngine
    case = make_case(seed=23, N=300, G=90, C=3, K=1)
    ex = _ThreadExchange(2)
    err = [None, None]

    def worker(rank):
        lo, hi = cell_range(300, rank, 2)
        shard = {k: (v[lo:hi] if k in ("Y", "psi0") else v) for k, v in case.items()}
        try:
            HipEngine(**shard, rank=rank, world=2, p2p_exchange=ex.make(rank)).close()
        except EngineError as e:
            err[rank] = e
    ts = [threading.Thread(target=worker, args=(r,)) for r in range(2)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert all(e is not None and e.code == 5 and "one process on one device" in str(e) for e in err), err


def test_peer_to_peer_gives_up_on_a_missing_peer_instead_of_hanging():
    """VERDICT r2 / ADVICE: a dead or out-of-step peer must end in CA_ERR_COMM, not in a device spin.  Rank 0 of a world of two
    commits the transport; rank 1 exists (its slab is mapped) but never reduces.  The first all-reduce of rank 0 (the setup sums
    inside ca_p2p_commit) runs into the device-side time limit, the engine reports the error, stays destroyable, and the GPU
    goes on working for the next engine."""
    import time
    import ctypes as C
    from clonealign_amd import engine as E
    from clonealign_amd.engine import EngineError, HipEngine
    case = make_case(seed=22, N=400, G=120, C=3, K=1)
    half = {k: (v[:200] if k in ("Y", "psi0") else v) for k, v in case.items()}
    other = {k: (v[200:] if k in ("Y", "psi0") else v) for k, v in case.items()}
    silent = HipEngine(**other, rank=1, world=2, defer_transport=True)          # exports a slab, never takes part
    buf = C.create_string_buffer(E.P2P_HANDLE_BYTES)
    assert silent.lib.ca_p2p_export(silent.h, buf) == 0
    calls = []

    def exchange(payload):                    # rank 0's view of the all-gather: its own payload + rank 1's
        calls.append(payload)
        if len(calls) == 1:
            return [payload, bytes(buf.raw)]
        return [payload, bytes([1]) + bytes(E.P2P_HANDLE_BYTES - 1)]
    t0 = time.perf_counter()
    with pytest.raises(EngineError) as ei:
        HipEngine(**half, rank=0, world=2, p2p_exchange=exchange, comm_timeout_ms=300, variant_on=("p2p_same_device",))
    dt = time.perf_counter() - t0
    assert ei.value.code == 5 and "did not arrive" in str(ei.value), str(ei.value)
    assert dt < 20.0, dt
    silent.close()
    # a one-sided mapping failure is agreed on BEFORE anybody waits on the device: the second exchange carries a 0, every rank raises
    def exchange_bad(payload):                # the other rank's export failed: an all-zero handle, then an all-zero status
        return [payload, bytes(E.P2P_HANDLE_BYTES)]
    with pytest.raises(EngineError) as ei:
        HipEngine(**half, rank=0, world=2, p2p_exchange=exchange_bad)
    assert ei.value.code == 5
    # the device is fine
    eng = HipEngine(**case)
    eng.gamma_init(eps_for(1, 120, 0))
    assert np.isfinite(eng.elbo(eps_for(1, 120, 1)))
    eng.close()
