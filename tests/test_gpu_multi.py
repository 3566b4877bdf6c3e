"""The cell-sharded fit across processes (BASELINE.json configs[3]) with the transports the engine offers:
the one-shot peer-to-peer all-reduce (two ranks on ONE GPU suffice: the slabs are IPC-mapped between processes), the host
callback over gloo, and -- where the box has two GPUs -- RCCL and peer-to-peer across devices.  Every run is a fresh pair of
processes started before anything touches the GPU."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPE = ["--cells", "6000", "--genes", "700", "--clones", "5", "--iters", "6"]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(world, transport, out, same_device):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    script = [os.path.join(ROOT, "tools", "dist_check.py"), "--transport", transport, "--out", str(out), *SHAPE]
    if same_device:
        script.append("--same-device")
    if world == 1:
        cmd = [sys.executable, *script]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), *script]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return json.load(open(out))


def _gpus():
    import torch
    return torch.cuda.device_count()       # (counting devices does not initialise the GPU)


@pytest.fixture(scope="module")
def single(tmp_path_factory):
    return _run(1, "none", tmp_path_factory.mktemp("dist") / "one.json", True)["ranks"][0]


def _check(res, single, transport):
    ranks = res["ranks"]
    assert [r["transport"] for r in ranks] == [transport] * len(ranks)
    # the engine's all-reduce payload is the plan shared with the host side (clonealign_amd/sharding.py)
    assert all(r["red_n"] == res["plan"]["total"] for r in ranks)
    t0 = np.array(ranks[0]["trace"])
    for r in ranks[1:]:                                    # replicas bit-identical: trace, final ELBOs, replicated variables
        assert np.array_equal(np.array(r["trace"]), t0) and r["finals"] == ranks[0]["finals"]
        for n, v in r["rep"].items():
            assert v == ranks[0]["rep"][n], n
    ts = np.array(single["trace"])                         # same fit as one handle holding all cells (summation order differs)
    assert t0.shape == ts.shape and np.abs(t0 - ts).max() <= 1e-7 * np.abs(ts).max()
    # (final ELBOs: one handle takes two draws per matrix-core sweep, shards take plain fp32 passes -- fp32 rounding apart)
    np.testing.assert_allclose(np.array(ranks[0]["finals"]), np.array(single["finals"]), rtol=2e-6)
    for n, v in ranks[0]["rep"].items():
        a, b = np.array(v), np.array(single["rep"][n])
        assert np.abs(a - b).max() <= 1e-5 * max(np.abs(b).max(), 1e-30), n
    assert ranks[0]["lo"] == 0 and ranks[-1]["hi"] == 6000 and all(a["hi"] == b["lo"] for a, b in zip(ranks, ranks[1:]))


def test_two_ranks_on_one_gpu_peer_to_peer_allreduce(tmp_path, single):
    """One-shot P2P all-reduce between two PROCESSES sharing device 0: IPC-mapped inbox slabs, sequence flags, rank-order sum."""
    _check(_run(2, "p2p", tmp_path / "p2p.json", True), single, "p2p")


def test_three_ranks_on_one_gpu_peer_to_peer_allreduce(tmp_path, single):
    _check(_run(3, "p2p", tmp_path / "p2p3.json", True), single, "p2p")


def test_two_ranks_on_one_gpu_host_callback_over_gloo(tmp_path, single):
    _check(_run(2, "host", tmp_path / "host.json", True), single, "host")


@pytest.mark.skipif(_gpus() < 2, reason="needs two GPUs (RCCL refuses two ranks on one device)")
def test_two_gpus_rccl_allreduce(tmp_path, single):
    _check(_run(2, "rccl", tmp_path / "rccl.json", False), single, "rccl")


@pytest.mark.skipif(_gpus() < 2, reason="needs two GPUs")
def test_two_gpus_peer_to_peer_allreduce(tmp_path, single):
    _check(_run(2, "p2p", tmp_path / "p2p2.json", False), single, "p2p")


def _bench(world, extra, same_device=True):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    if same_device:
        env["CLONEALIGN_BENCH_DEVICE"] = "0"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "6", "--warmup", "2",
           "--repeats", "2", "--cells", "20000", "--genes", "1000", "--clones", "4", "--no-cpu-baseline", "--busy-seconds", "0", *extra]
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=420, cwd=ROOT)   # (a run takes 15-60 s)


def test_bench_two_ranks_use_a_device_transport_and_fail_loudly_without_one():
    """bench.py --gpus 2 (both ranks on device 0 here): the all-reduce runs on the device (peer-to-peer), the JSON line says so;
    asking for RCCL -- which refuses two ranks on one device -- must exit non-zero instead of quietly measuring a host path,
    and only --allow-host-fallback lets such a run through (marked as what it is)."""
    r = _bench(2, [])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["collective"] == "p2p" and line["config"]["collectives_tried"] == ["p2p"]
    assert line["scaling"] == "strong" and line["value"] > 0 and line["repeats"]["n"] == 2
    assert line["config"]["allreduce_doubles_per_train_pass"] == 3 + 4 + 1000 * 2 + 1000
    if _gpus() < 2:
        r = _bench(2, ["--collective", "rccl"])
        assert r.returncode != 0 and "refusing to report" in r.stderr
        r = _bench(2, ["--collective", "rccl", "--allow-host-fallback"])
        assert r.returncode == 0, r.stderr[-2000:]
        line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        assert line["config"]["collective"] == "gloo-host-fallback" and line["config"]["collectives_tried"] == ["rccl", "host"]


@pytest.mark.skipif(_gpus() < 8, reason="needs an 8-GPU node (BASELINE.json configs[3]: 100k x 5k x 8 cell-sharded over 8 MI355X)")
def test_bench_on_eight_gpus_uses_a_device_collective_and_replicas_agree(tmp_path):
    """The driver's scaling run, as a test for the day a node is available: bench.py --gpus 8 --steps 20 at the full BASELINE
    size must come up on a DEVICE transport (peer-to-peer over xGMI, else RCCL), report the collective's own cost, and the
    8-rank fit of tools/dist_check.py must leave bit-identical replicas that match the one-handle fit."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "20", "--warmup", "5"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1800, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 8 and line["config"]["collective"] in {"p2p", "rccl"}, line["config"]
    assert line["config"]["allreduce_doubles_per_train_pass"] == 3 + 8 + 5000 * 2 + 5000
    assert line["allreduce_us"].get(line["config"]["collective"], 0) > 0, line["allreduce_us"]
    single = _run(1, "none", tmp_path / "one.json", True)["ranks"][0]
    for tr in ("p2p", "rccl"):
        _check(_run(8, tr, tmp_path / f"{tr}8.json", False), single, tr)
