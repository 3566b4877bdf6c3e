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

from tests._cases import child_report

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPE = ["--cells", "6000", "--genes", "700", "--clones", "5", "--iters", "6"]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _launch(world, transport, out, same_device, shape=None, extra=()):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    script = [os.path.join(ROOT, "tools", "dist_check.py"), "--transport", transport, "--out", str(out), *(shape or SHAPE), *extra]
    if same_device:
        script.append("--same-device")
    if world == 1:
        cmd = [sys.executable, *script]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), *script]
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)


def _run(world, transport, out, same_device, shape=None):
    r = _launch(world, transport, out, same_device, shape)
    assert r.returncode == 0, child_report(r)
    return json.load(open(out))


def _gpus():
    import torch
    return torch.cuda.device_count()       # (counting devices does not initialise the GPU)


@pytest.fixture(scope="module")
def single(tmp_path_factory):
    return _run(1, "none", tmp_path_factory.mktemp("dist") / "one.json", True)["ranks"][0]


def _check(res, single, transport, cells=6000):
    ranks = res["ranks"]
    assert [r["transport"] for r in ranks] == [transport] * len(ranks)
    assert [r["selftest_bad"] for r in ranks] == [0] * len(ranks)      # known-answer all-reduces (ca_comm_selftest) on the transport in use
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
    assert ranks[0]["lo"] == 0 and ranks[-1]["hi"] == cells and all(a["hi"] == b["lo"] for a, b in zip(ranks, ranks[1:]))


def test_two_ranks_on_one_gpu_peer_to_peer_allreduce(tmp_path, single):
    """One-shot P2P all-reduce between two PROCESSES sharing device 0: IPC-mapped inbox slabs, sequence flags, rank-order sum."""
    _check(_run(2, "p2p", tmp_path / "p2p.json", True), single, "p2p")


def test_three_ranks_on_one_gpu_peer_to_peer_allreduce(tmp_path, single):
    _check(_run(3, "p2p", tmp_path / "p2p3.json", True), single, "p2p")


def test_two_ranks_on_one_gpu_host_callback_over_gloo(tmp_path, single):
    _check(_run(2, "host", tmp_path / "host.json", True), single, "host")


# Per-rank shard sizes on BOTH sides of every threshold of the engine's kernel selection (clonealign_hip.hip create_impl: forward
# blocks of 16 cells below 8192 cells per rank, 32 below 32768, 96 above; the small-problem folds are off when sharded), each
# against the one-handle fit of the same cells -- which itself takes a different decomposition.
THRESHOLD_SHAPES = {
    "2x3k_blocks16": (2, ["--cells", "6000", "--genes", "700", "--clones", "5", "--iters", "6"], 16),
    "2x10k_blocks32": (2, ["--cells", "20000", "--genes", "400", "--clones", "4", "--iters", "4"], 32),
    "3x33k_blocks96": (3, ["--cells", "100000", "--genes", "300", "--clones", "8", "--iters", "3"], 96),
    "2x50k_blocks96": (2, ["--cells", "100000", "--genes", "260", "--clones", "3", "--iters", "3"], 96),
}


@pytest.mark.parametrize("name", list(THRESHOLD_SHAPES))
def test_ranks_on_one_gpu_at_shard_sizes_across_the_kernel_selection_thresholds(tmp_path, name):
    world, shape, block = THRESHOLD_SHAPES[name]
    cells = int(shape[1])
    one = _run(1, "none", tmp_path / "one.json", True, shape)["ranks"][0]
    res = _run(world, "p2p", tmp_path / "p2p.json", True, shape)
    assert [r["fwd_block_cells"] for r in res["ranks"]] == [block] * world, [r["fwd_block_cells"] for r in res["ranks"]]
    _check(res, one, "p2p", cells)


@pytest.mark.parametrize("world,cells", [(2, 12000), (3, 26000)])
def test_ranks_whose_shards_take_the_balanced_forward_sweep(tmp_path, world, cells):
    """Round 5: shards of 4096 ... ~28k cells with 3072+ genes run the balanced eight-wave forward sweep (k_fwd_bal_ys: left-over tiles exchanged
    between blocks through tagged words) inside a sharded fit -- its block partials feed the same all-reduced sums.  Two ranks of 6000 cells
    (one tile per CU + 119 left over) and three of ~8667 (two per CU + 30): replicas bit-identical, the fit of the one-handle engine (which takes
    another decomposition: three / six tiles per CU)."""
    shape = ["--cells", str(cells), "--genes", "3100", "--clones", "5", "--iters", "3"]
    one = _run(1, "none", tmp_path / "one.json", True, shape)["ranks"][0]
    res = _run(world, "p2p", tmp_path / "p2p.json", True, shape)
    assert all(r["fwd_balanced"] >= 1 for r in res["ranks"]) and one["fwd_balanced"] >= 1, [r["fwd_balanced"] for r in res["ranks"]]
    _check(res, one, "p2p", cells)


def test_two_ranks_on_one_gpu_with_two_mc_samples(tmp_path):
    """mc_samples = 2 across two processes over the peer-to-peer transport: the four-draw forward sweep, the two-sample backward sweep whose
    column sums (three columns per gene: one per sample, one for W) are folded inside the all-reduce's launch, the pending monitor tail
    travelling with the train pass's sums -- against the one-handle fit."""
    shape = ["--cells", "9000", "--genes", "500", "--clones", "6", "--iters", "5", "--mc-samples", "2"]
    one = _run(1, "none", tmp_path / "one.json", True, shape)["ranks"][0]
    _check(_run(2, "p2p", tmp_path / "p2p.json", True, shape), one, "p2p", 9000)


@pytest.mark.parametrize("world", [2, 3])
def test_work_riding_in_the_all_reduce_launch_equals_the_separate_launches(tmp_path, world):
    """Round 4, sharded over the peer-to-peer transport: the backward sweep's column sums, the int8 stream's finishing sums and a pending
    monitor pass's psi.(YW) sum ride in the sweep's and the all-reduce's launches (four launches per iteration instead of six).  Against the
    separate launches (variant p2p_ride off): same fit to fp64 summation order (1e-10 on the ELBO trace), replicas bit-identical either way,
    both equal to the one-handle fit."""
    shape = ["--cells", "9000", "--genes", "800", "--clones", "6", "--iters", "8"]
    one = _run(1, "none", tmp_path / "one.json", True, shape)["ranks"][0]
    on = _run(world, "p2p", tmp_path / "on.json", True, shape)
    r = _launch(world, "p2p", tmp_path / "off.json", True, shape, extra=["--variant-off", "p2p_ride"])
    assert r.returncode == 0, child_report(r)
    off = json.load(open(tmp_path / "off.json"))
    _check(on, one, "p2p", 9000)
    _check(off, one, "p2p", 9000)
    a, b = np.array(on["ranks"][0]["trace"]), np.array(off["ranks"][0]["trace"])
    assert a.shape == b.shape and np.abs(a - b).max() <= 1e-10 * np.abs(b).max(), (a, b)


def test_eight_ranks_on_one_gpu_with_the_cfg4_gene_count(tmp_path):
    """W = 8 flag lanes / inbox slabs and the all-reduce payload of BASELINE.json configs[3] (G = 5000, C = 8: 15 011 doubles per
    train pass) -- eight PROCESSES sharing device 0, 1000 cells each, against the one-handle fit."""
    shape = ["--cells", "8000", "--genes", "5000", "--clones", "8", "--iters", "3"]
    one = _run(1, "none", tmp_path / "one.json", True, shape)["ranks"][0]
    res = _run(8, "p2p", tmp_path / "p2p8.json", True, shape)
    assert len(res["ranks"]) == 8 and res["ranks"][0]["red_n"] == 3 + 8 + 5000 * 2 + 5000
    _check(res, one, "p2p", 8000)


@pytest.mark.parametrize("world,who", [(2, 0), (3, 2)])
def test_a_rank_that_makes_one_collective_call_too_many_fails_within_the_time_limit(tmp_path, world, who):
    """The failure mode of r03's red record, as a test: one rank enters a collective call its peers never make.  The device-side
    wait gives up after comm_timeout_ms (1.5 s here), the call returns CA_ERR_COMM, that process exits non-zero, the launcher
    takes the others down -- nothing hangs, nothing reports a number."""
    import time
    t0 = time.perf_counter()
    r = _launch(world, "p2p", tmp_path / "x.json", True, extra=["--extra-call-rank", str(who), "--comm-timeout-ms", "1500"])
    dt = time.perf_counter() - t0
    assert r.returncode != 0, child_report(r)
    said = [l for l in r.stderr.splitlines() if l.startswith(f"[rank {who}] dist_check: extra collective call failed")]
    assert said and "code 5" in said[0] and "did not arrive" in said[0], child_report(r)
    assert float(said[0].split("after ")[1].split(" s")[0]) < 6.0, said[0]
    assert not os.path.exists(tmp_path / "x.json") and dt < 240, dt


@pytest.mark.skipif(_gpus() < 2, reason="needs two GPUs (RCCL refuses two ranks on one device)")
def test_two_gpus_rccl_allreduce(tmp_path, single):
    _check(_run(2, "rccl", tmp_path / "rccl.json", False), single, "rccl")


@pytest.mark.skipif(_gpus() < 2, reason="needs two GPUs")
def test_two_gpus_peer_to_peer_allreduce(tmp_path, single):
    _check(_run(2, "p2p", tmp_path / "p2p2.json", False), single, "p2p")


def _bench(world, extra, same_device=True, shape=("--cells", "20000", "--genes", "1000", "--clones", "4"), api_leg=False):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    if same_device:
        env["CLONEALIGN_BENCH_DEVICE"] = "0"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "6", "--warmup", "2",
           "--repeats", "2", *shape, "--no-cpu-baseline", "--busy-seconds", "0", *([] if api_leg else ["--no-through-api-leg"]), *extra]
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=420, cwd=ROOT)   # (a run takes 15-60 s)


def _bench_bare(world, extra=(), same_device=True, shape=("--cells", "20000", "--genes", "1000", "--clones", "4"), timeout=600):
    """`python bench.py --gpus N ...` with NO launcher around it and no WORLD_SIZE in the environment -- the shape of the driver's 1-GPU
    command: bench.py must start its N ranks itself."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "CLONEALIGN_BENCH_DEVICE")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    if same_device:
        env["CLONEALIGN_BENCH_DEVICE"] = "0"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "6", "--warmup", "2", "--repeats", "2", *shape,
           "--no-cpu-baseline", "--busy-seconds", "0", "--steady-steps", "0", *([] if ("--through-api" in extra or "--api-leg" in extra) else ["--no-through-api-leg"]),
           *[e for e in extra if e != "--api-leg"]]
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)


def test_bench_through_the_api_runs_one_sharded_fit_inside_one_process():
    """`bench.py --through-api --gpus N` is ONE process driving N devices through the device group (what
    inference_tflow(devices=) / C_clonealign_fit(devices) run); and the driver's multi-GPU command -- one process per GPU -- carries the
    same measurement as an extra leg made by rank 0 in a fresh child process, so that the driver's first multi-GPU contact exercises the
    drop-in's own path.  Here both ranks sit on device 0: the group settles on the host reduction and says why."""
    r = _bench_bare(2, ["--through-api"])
    assert r.returncode == 0, child_report(r)
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    t = line["through_api"]
    assert line["n_gpus"] == 2 and line["config"]["collective"] == "host" and t["transport"] == "host" and "more than one rank" in t["note"]
    assert t["cells_per_rank"] == [10000, 10000] and t["finite"] and line["value"] == t["value"] > 0 and t["fit_wallclock"]["iterations"] >= 10
    r = _bench_bare(2, ["--api-leg", "--preheat-ms", "20"])
    assert r.returncode == 0, child_report(r)
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["config"]["collective"] == "p2p" and "error" not in line["through_api"], line.get("through_api")
    assert line["through_api"]["transport"] == "host" and line["through_api"]["value"] > 0 and line["through_api"]["devices"] == [0, 0]


@pytest.mark.parametrize("world", [2, 8])
def test_bare_bench_command_starts_its_own_ranks(world):
    """`python bench.py --gpus N` without torch.distributed.run around it used to run ONE rank and print n_gpus 1.  Now the
    process becomes the launcher (before anything touches a GPU): N fresh ranks, rank 0's line relayed, the worst rank's exit code."""
    r = _bench_bare(world, ["--preheat-ms", "30"], shape=("--cells", "24000", "--genes", "600", "--clones", "8") if world == 8 else ("--cells", "20000", "--genes", "1000", "--clones", "4"))
    assert r.returncode == 0, child_report(r)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                      # exactly ONE JSON line: rank 0's
    line = json.loads(lines[0])
    assert line["n_gpus"] == world and line["config"]["collective"] == "p2p" and line["scaling"] == "strong"
    assert line["config"]["allreduce_selftest"] == {"p2p": True} and line["value"] > 0
    assert line["config"]["parallelism"] == f"cells/{world}" and line["replicas_bit_identical_after_timed_regions"] is True


def test_bare_bench_command_refuses_more_ranks_than_devices_and_relays_a_failing_rank():
    """Never a silent run at another width: more ranks than visible devices (and no plumbing override) exits non-zero before any rank starts;
    a wrapper that sets WORLD_SIZE to something else than --gpus is refused too; and when a rank fails the launcher exits non-zero."""
    if _gpus() < 8:
        r = _bench_bare(8, same_device=False)
        assert r.returncode != 0 and "GPU(s) visible" in r.stderr and not [l for l in r.stdout.splitlines() if l.startswith("{")], child_report(r)
    env = dict(os.environ, WORLD_SIZE="1", RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "2"], env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr, child_report(r)
    r = _bench_bare(2, ["--collective", "p2p", "--selftest-fail", "p2p"])
    assert r.returncode != 0 and "refusing to report" in r.stderr and "launcher: rank" in r.stderr, child_report(r)
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_bench_two_ranks_use_a_device_transport_and_fail_loudly_without_one():
    """bench.py --gpus 2 (both ranks on device 0 here): the all-reduce runs on the device (peer-to-peer), the JSON line says so;
    asking for RCCL -- which refuses two ranks on one device -- must exit non-zero instead of quietly measuring a host path,
    and only --allow-host-fallback lets such a run through (marked as what it is)."""
    r = _bench(2, [])
    assert r.returncode == 0, child_report(r)
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["collective"] == "p2p" and line["config"]["collectives_tried"] == ["p2p"]
    assert line["config"]["allreduce_selftest"] == {"p2p": True}
    assert line["scaling"] == "strong" and line["value"] > 0 and line["repeats"]["n"] == 2
    assert line["config"]["allreduce_doubles_per_train_pass"] == 3 + 4 + 1000 * 2 + 1000
    # a transport that comes up but does not ADD is not used: one rank's failed known-answer test moves every rank on (here: to nothing)
    r = _bench(2, ["--collective", "p2p", "--selftest-fail", "p2p"])
    assert r.returncode != 0 and "known-answer all-reduce wrong" in r.stderr and "refusing to report" in r.stderr, child_report(r)
    if _gpus() < 2:
        r = _bench(2, ["--collective", "rccl"])
        assert r.returncode != 0 and "refusing to report" in r.stderr
        r = _bench(2, ["--collective", "rccl", "--allow-host-fallback"])
        assert r.returncode == 0, child_report(r)
        line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        assert line["config"]["collective"] == "gloo-host-fallback" and line["config"]["collectives_tried"] == ["rccl", "host"]


def test_bench_preheat_makes_the_same_number_of_collective_calls_on_every_rank():
    """the clock pre-heat is a loop of collective calls; its length must be agreed on, not decided by each
    rank's own clock.  A long pre-heat of short calls (about 150 calls here) gives a per-rank clock every chance to disagree;
    three ranks, 96-cell blocks (34k cells per rank), then two ranks again."""
    r = _bench(3, ["--preheat-ms", "400"], shape=("--cells", "102000", "--genes", "300", "--clones", "8"))
    assert r.returncode == 0, child_report(r)
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 3 and line["config"]["collective"] == "p2p" and line["preheat"]["calls"] >= 2
    assert line["preheat"]["iterations"] == line["preheat"]["calls"] * 6
    r = _bench(2, ["--preheat-ms", "300"])
    assert r.returncode == 0, child_report(r)
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["preheat"]["calls"] >= 20, line["preheat"]


def test_bench_eight_ranks_on_one_device_run_the_driver_command_shape():
    """The driver's scaling command at W = 8 -- `bench.py --gpus 8 --steps K --warmup W` under torch.distributed.run -- with every rank on
    device 0 and a small matrix: transport bring-up over eight processes, agreed pre-heat, timed regions, the all-reduce benchmark, the fit
    with its stop rule on eight replicas.  (The numbers mean nothing: eight processes time-slice one GPU.)"""
    r = _bench(8, ["--preheat-ms", "30"], shape=("--cells", "24000", "--genes", "600", "--clones", "8"))
    assert r.returncode == 0, child_report(r)
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 8 and line["config"]["collective"] == "p2p" and line["scaling"] == "strong"
    assert line["config"]["allreduce_doubles_per_train_pass"] == 3 + 8 + 600 * 2 + 600 and line["allreduce_us"]["p2p"] > 0
    assert line["fit_wallclock"]["iterations"] >= 10 and np.isfinite(line["final_elbo"])


@pytest.mark.skipif(_gpus() < 8, reason="needs an 8-GPU node (BASELINE.json configs[3]: 100k x 5k x 8 cell-sharded over 8 MI355X)")
def test_bench_on_eight_gpus_uses_a_device_collective_and_replicas_agree(tmp_path):
    """The driver's scaling run, as a test for the day a node is available: bench.py --gpus 8 --steps 20 at the full BASELINE
    size must come up on a DEVICE transport (peer-to-peer over xGMI, else RCCL), report the collective's own cost, and the
    8-rank fit of tools/dist_check.py must leave bit-identical replicas that match the one-handle fit."""
    # the BARE command (no launcher, no WORLD_SIZE): bench.py starts its eight ranks itself, one per device
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "CLONEALIGN_BENCH_DEVICE")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "20", "--warmup", "5"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1800, cwd=ROOT)
    assert r.returncode == 0, child_report(r)
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 8 and line["config"]["collective"] in {"p2p", "rccl"}, line["config"]
    assert line["config"]["allreduce_doubles_per_train_pass"] == 3 + 8 + 5000 * 2 + 5000
    assert line["allreduce_us"].get(line["config"]["collective"], 0) > 0, line["allreduce_us"]
    single = _run(1, "none", tmp_path / "one.json", True)["ranks"][0]
    for tr in ("p2p", "rccl"):
        _check(_run(8, tr, tmp_path / f"{tr}8.json", False), single, tr)


def test_bench_two_ranks_take_the_series_form_sharded():
    """Round 6, late: one process per rank (IPC-mapped peer-to-peer inboxes), a shape whose ranks take the series form of the contraction (35 000 cells x 5000
    genes per rank; sharded, the pick asks for G >= 2000 + 7.5e7 / N of the rank): every pass on the series form on both ranks, replicas bit-identical after the timed regions, and the payload of the collective is the
    series prefix of sharding.reduce_plan (cell sums, Y^T psi, backward moments, a max |psi| slot per rank) -- not the per-gene sums."""
    from clonealign_amd import sharding
    r = _bench(2, [], shape=("--cells", "70000", "--genes", "5000", "--clones", "8"))
    if r.returncode != 0 and ("time limit" in r.stderr or "CA_ERR_COMM" in r.stderr or "error 5" in r.stderr):
        # two PROCESSES on one device: a rank's all-reduce kernel spins for its peer's data while the peer's kernels wait for the same GPU -- the one-device rig's
        # own hazard (bounded by comm_timeout_ms, reported as CA_ERR_COMM; seen once in a dozen runs of this test).  Distinct devices have no such wait.  Once more.
        r = _bench(2, [], shape=("--cells", "70000", "--genes", "5000", "--clones", "8"))
    assert r.returncode == 0, child_report(r)
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["collective"] == "p2p" and line["config"]["series_form"] is True, line["config"]
    sf = line["roofline"]["series_form"]
    assert sf["passes"] > 0 and sf["handed_to_the_sweeps"] == 0, sf
    assert line["replicas_bit_identical_after_timed_regions"] is True
    plan = sharding.reduce_plan(5000, 8, 1, 0, 1, series=True, world=2)
    assert line["config"]["allreduce_doubles_per_train_pass"] == plan["series_total"] == 3 + 8 + 5120 + 32 * 22 * 8 + 2
    assert line["roofline"]["kernel"] == "ypass" and line["roofline"]["bound"] == "hbm" and 0 < line["roofline"]["frac"] < 1


def test_bench_four_ranks_on_one_device_keep_their_collectives_in_step_when_a_gated_update_gives_up():
    """Found late in round 6: four rank PROCESSES sharing one device are slow enough for a rank's gated update to give up now and then (its host's decision came
    late: the launch stores nothing and is queued again -- a rank's own affair).  The sharded series form used to invalidate its max |psi| slots on that path, the
    next pass refreshed them with a small collective of ITS OWN, and that rank's four doubles paired with its peers' next all-reduce: their cell sums landed in its
    slots ("cannot cover the exponent range"), the peers timed out -- every other run of this command.  The slots' step count is part of the gate's snapshot now.
    cfg-3 over four ranks (25 000 cells each: the series form), the driver's command shape, twice."""
    shape = ("--cells", "100000", "--genes", "5000", "--clones", "8")
    for _ in range(2):
        r = _bench(4, [], shape=shape)
        if r.returncode != 0 and "time limit" in r.stderr and "cannot cover" not in r.stderr:
            r = _bench(4, [], shape=shape)   # (the one-device rig's own hazard: a rank's all-reduce spinning for a peer whose kernels wait for the same GPU)
        assert r.returncode == 0, child_report(r)
        line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        assert line["n_gpus"] == 4 and line["config"]["series_form"] is True and line["replicas_bit_identical_after_timed_regions"] is True, line["config"]
        assert line["fit_wallclock"]["iterations"] >= 1 and np.isfinite(line["fit_wallclock"]["final_elbo_mean"])
