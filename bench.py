#!/usr/bin/env python3
"""ELBO iterations/sec of the clonealign VI hot path on MI355X (BASELINE.json metric).

One "step" = one iteration of the reference's loop (R/inference-tflow.R:401,403): a train
pass (forward + backward + Adam, fresh eps) followed by a monitor pass (forward, fresh eps).
Workload at N=1: BASELINE.json configs[2] -- synthetic 100k cells x 5k genes x 8 clones
(the configuration the metric is quoted on).  With N>1 GPUs the SAME 100k cells are sharded
across the ranks (configs[3]: strong scaling) with ONE all-reduce of the per-gene gradient sums
per train pass on the device (one-shot peer-to-peer over xGMI, else RCCL).  A run whose data path
would fall back to the host exits non-zero unless --allow-host-fallback is given.

  python bench.py --gpus 1 --steps 50 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import glob
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_HBM_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
PEAK_F32_TFLOPS = 157.3      # MI355X_MICROARCH.md: fp32 vector == fp32 MFMA peak


EVENT_STRIDE = 8


def algorithmic_work(kernel, N, G, C, K, fused=False, steps=1, ride=False):
    """Per-launch algorithmic work of each kernel class.  ``ride``: the Y stream's blocks run inside
    the forward sweep's launch (k_fwd_cell_mix_y), so that launch also does the two count-matrix products, 2 flop each per
    (count, latent dimension); its HBM side is reported by roofline_ystream."""
    if kernel == "ypass":     # one pass over Y: Y.W and Y^T.psi; canonical 4 B/elem + outputs
        return "hbm", N * G * 4.0 + (N + G) * K * 4.0 * 2
    if kernel == "fwd":       # eta 2K, Z 2C, +1 (exp not counted)
        plain = N * G * (2.0 * C + 2.0 * K + 1.0)
        if not fused:
            return "mfma", plain
        # fused two-eps sweep: 2C columns share one eta/exp; per timed call: steps-1 fused launches + 2 plain ones
        # (since r02 the first and the last sweep of a ca_iterate call are fused launches carrying ONE draw twice: still `plain` useful work)
        two = N * G * (4.0 * C + 2.0 * K + 1.0)
        return "mfma", ((steps - 1) * two + 2 * plain) / (steps + 1) + (N * G * 4.0 * K if ride else 0.0)
    if kernel == "bwd":       # eta 2K, t 2C, dM/dmu 2C-equivalent, deta 1, dpsi 2K, dW 2K
        return "mfma", N * G * (4.0 * C + 6.0 * K + 1.0)
    return "hbm", 0.0


def draws_for_calls(rng, steps, S, G):
    """2 steps + 1 draws for back-to-back ca_iterate(steps) calls on the SAME array: the extra draw is a copy of the first, so that every call's
    last sweep makes the forward half of the next call's first train pass (ca_iterate, ABI 6) and a call of K iterations runs K sweeps, as in the
    steady state of the reference's loop -- not K + 1 with a duplicate half at the end (r5: the 20-step command paid 21/20)."""
    eps = rng.normal(size=(2 * steps + 1, S, G)).astype(np.float32)
    eps[-1] = eps[0]
    return eps


def agreed_calls(budget_ms, call_ms, cap=4096):
    """How many equal calls fill ``budget_ms`` when one takes ``call_ms`` -- a pure function of two numbers every rank holds
    identically (call_ms is the all-reduced MAX), so a loop of collective calls sized by it has the same length on every rank."""
    if not (call_ms > 0.0):
        return 1
    return int(min(max(1, -(-budget_ms // call_ms)), cap))


def run_agreed_calls(call, budget_ms, dist=None):
    """Fill about ``budget_ms`` with equal calls of ``call`` -- which may be COLLECTIVE (every rank must make the same number of
    them): one call is timed, the slowest rank's time is agreed on (all-reduce MAX over ``dist``, the control-plane group), the
    count follows from that one number, the remaining calls run.  Returns the number of calls made (the same on every rank)."""
    t0 = time.perf_counter()
    call()
    call_ms = (time.perf_counter() - t0) * 1e3
    if dist is not None:
        import torch
        t = torch.tensor([call_ms], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        call_ms = float(t[0])
    n = agreed_calls(budget_ms, call_ms)
    for _ in range(n - 1):
        call()
    return n


def pmc_traffic(build_id, kernel_class):
    """HBM bytes per launch of a kernel class from the PMC passes committed under profiles/ -- only from a file that was
    collected on THIS build of the library (matching ca_build_id) and only under the class's own key; else None."""
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_hbm.json")), reverse=True):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("build_id") == build_id and kernel_class in d.get("classes", {}):
            c = d["classes"][kernel_class]
            return c.get("hbm_bytes"), {"file": os.path.basename(f), "kernel": c.get("kernel")}
    return None, None


def cpu_baseline(Ycells, L, psi0, loc0, K, N_full, budget_s=22.0):
    """The CPU ports of the oracle (kind 'port') timed on the host cores.  Headline: the float32 SIMD port
    (oracle/c/clonealign_simd.c: unit-stride gene loops, libmvec exp/log, AVX-512 where the host has it) at up to FOUR sample
    sizes of the same workload, the largest being the FULL matrix when the host has the memory for it -- then the quoted rate is
    not extrapolated; the smaller samples show how time per iteration grows with the cell count.  Beside it the scalar
    float64 port (oracle/c/clonealign_oracle.c, the tests' checker) on the smallest sample."""
    from oracle import c_port
    sizes = [n for n in (4096, 16384, 65536) if n < Ycells.shape[0]] + [Ycells.shape[0]]
    rows, cores = [], 1
    for n in sizes:
        r = c_port.time_baseline(Ycells[:n], L, psi0, loc0, K, N_full, budget_s * n / sum(sizes), simd=True)
        rows.append({"cells": n, "scaled_it_per_s": r["value"], "s_per_iter_sample": n / N_full / r["value"],
                     "us_per_cell": 1e6 / (N_full * r["value"])})   # flat over the samples = time linear in the cell count
        cores = r["cores"]
    scalar = c_port.time_baseline(Ycells[:sizes[0]], L, psi0, loc0, K, N_full, 4.0)
    big = rows[-1]
    full = big["cells"] == N_full
    return {"value": big["scaled_it_per_s"], "unit": "iterations/s", "cores": cores, "kind": "port",
            "sample": f"C + OpenMP float32 SIMD port (oracle/c/clonealign_simd.c) on "
                      + (f"ALL {N_full} cells (not extrapolated)" if full else f"the first {big['cells']} of {N_full} cells, rate scaled by {big['cells']}/{N_full}")
                      + f" x {L.shape[0]} genes; time per iteration by sample (fixed cost of the thread team shows at the small ones): "
                      + ", ".join(f"{r['cells']} cells {r['s_per_iter_sample'] * 1e3:.1f} ms/iter" for r in rows),
            "samples": rows, "extrapolated": not full,
            "scalar_f64": {"value": scalar["value"], "unit": "iterations/s", "cores": scalar["cores"], "sample": scalar["sample"]}}


def parity_check(eng, Ycells, L, psi0, loc0, K, iters=2):
    """Outside the timed region: the engine, restarted from its initial values, against the float64 C oracle on the SAME matrix
    through the whole-loop call (gamma init, initial ELBO, `iters` iterations of ca_run vs the oracle driven call by call,
    R/inference-tflow.R:368-417).  Reports the largest relative ELBO difference and the largest relative parameter difference."""
    from clonealign_amd.inference import run_vi_loop
    from clonealign_amd.rng import EpsStream
    from oracle.c_port import CPortModel
    G = Ycells.shape[1]
    t0 = time.perf_counter()
    ora = CPortModel(Ycells, L, psi0, loc0, K, dtype="float32")
    try:
        eng.reinit(psi0, loc0)
        tr = np.asarray(eng.run(EpsStream(77, 1, G), iters, 1e-12))
        to = np.asarray(run_vi_loop(ora, EpsStream(77, 1, G), iters, 1e-12))
        se, so = eng.get_state(), ora.get_state()
        perr = max(float(np.abs(se[n] - so[n]).max(initial=0) / max(np.abs(so[n]).max(initial=0), 1e-30)) for n in so)
        # clone_assignment (R/inference-tflow.R:22-29) on both posteriors: argmax if max >= 0.95 else "unassigned"
        pe, po = eng.get("clone_probs"), ora.get_params()["clone_probs"]
        le = np.where(pe.max(1) >= 0.95, pe.argmax(1), -1)
        lo_ = np.where(po.max(1) >= 0.95, po.argmax(1), -1)
        flips = int((le != lo_).sum())
    finally:
        ora.close()
    return {"iters": iters, "cells": int(Ycells.shape[0]), "max_rel_elbo": float(np.abs(tr - to).max() / np.abs(to).max()),
            "max_rel_param": perr, "label_flips": flips, "labels_unassigned_oracle": int((lo_ < 0).sum()), "elbo_engine": tr.tolist(), "elbo_oracle": to.tolist(), "seconds": time.perf_counter() - t0,
            "what": "ca_run on the bench engine (restarted from its initial values) vs oracle/c/clonealign_oracle.c (float64 arithmetic, "
                    "float32 variables) on the same cells, same eps stream; not part of any timed region"}


def cpu_ref_dataflow(budget_s=10.0):
    """BASELINE.md section 3 'CPU-ref-dataflow': the literal restatement that follows the reference's dataflow op for op
    (materialises [S,G,C,N], recomputes the lgamma terms every pass, autodiff) in float32 on the host cores, at cfg-2."""
    import torch
    from clonealign_amd.hostprep import mu_guess, safe_inverse_softplus
    from oracle.literal_torch import LiteralModel
    N, G, C = 10_000, 2_000, 4
    rng = np.random.default_rng(20242)
    L = rng.integers(1, 5, size=(G, C)).astype(np.float64)
    Y = rng.poisson(rng.lognormal(-1.0, 1.0, G)[None, :] * L[:, rng.integers(0, C, N)].T * 0.5).astype(np.float64)
    Y[:, 0] += 1
    loc0 = safe_inverse_softplus(np.maximum(mu_guess(Y, True), 1e-6))
    m = LiteralModel(Y, L, np.random.default_rng(1).normal(size=(N, 1)), loc0, 1, 1, dtype="float32")
    e = lambda i: np.random.default_rng(i).normal(size=(1, G)).astype(np.float32)  # noqa: E731
    m.gamma_init(e(0)); m.step(e(1)); m.elbo(e(2))
    t0, it = time.perf_counter(), 0
    while it < 20:
        m.step(e(3 + 2 * it)); m.elbo(e(4 + 2 * it)); it += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return {"value": it / dt, "unit": "iterations/s", "cores": torch.get_num_threads(), "kind": "port",
            "workload": f"synthetic {N} cells x {G} genes x {C} clones (BASELINE.json configs[1])",
            "what": "oracle/literal_torch.py in float32: materialising dataflow of R/inference-tflow.R:288-296, two full passes "
                    f"per iteration, {it} iterations in {dt:.1f} s"}


def side_config(name, N, G, C, K=1, S=1, P=0, steps=200, regions=3, seed=20251, device=0, what="", kernel_classes=False, engine_kw=None):
    """One more shape on the driver's clock, OUTSIDE the headline regions: a fresh synthetic problem of that shape (generated on the
    GPU), a fresh engine, gamma init, 20 warm-up iterations, then `regions` regions of one ca_iterate(steps) call between
    synchronisations; the median region is reported against the shape's own roof (the larger of the canonical
    HBM time and the fp32 time of the iteration's algorithmic work)."""
    import torch
    import synth_data as synth
    from clonealign_amd.engine import HipEngine
    from clonealign_amd.hostprep import safe_inverse_softplus
    dev = f"cuda:{device}"
    Yd, aux = synth.make_problem_torch(N, G, C, seed=seed, device=dev)
    rm = Yd.sum(1, keepdim=True).to(torch.float64) / G
    col = torch.zeros(G, dtype=torch.float64, device=dev)
    for b0 in range(0, N, 8192):
        col += (Yd[b0:b0 + 8192].to(torch.float64) / rm[b0:b0 + 8192]).sum(0)
    loc0 = safe_inverse_softplus(np.maximum(col.cpu().numpy() / N, 1e-6))
    rng = np.random.default_rng(seed + 1)
    psi0 = rng.normal(size=(N, K))
    X = rng.normal(size=(N, P)) if P > 0 else None
    torch.cuda.synchronize()
    eng = HipEngine(None, aux["L"], psi0, loc0, K, S, X=X, y_device_ptr=Yd.data_ptr(), y_device_dtype=np.int32, shape=(N, G), device=device, profile=0, **(engine_kw or {}))
    classes = None
    try:
        info = eng.info()
        eng.gamma_init(rng.normal(size=(S, G)).astype(np.float32))
        eps = draws_for_calls(rng, steps, S, G)
        eng.iterate(min(steps, 20), eps[:2 * min(steps, 20)])
        eng.iterate(steps, eps)                       # (untimed: the first timed call then starts from a carried half like every later one)
        eng.synchronize()
        ts = []
        for _ in range(regions):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            last = eng.iterate(steps, eps)
            eng.synchronize()
            ts.append(time.perf_counter() - t0)
        dt = float(np.median(ts)) / steps
        info_end = eng.info()
        if kernel_classes:   # (tools/side_time.py: after the timed regions, events around every launch)
            eng.set_profile(0x1F)
            eng.iterate(min(steps, 20), eps[:2 * min(steps, 20) + 1])
            classes = {k: round(v[0] / min(steps, 20) * 1e3, 1) for k, v in eng.kernel_times(reset=True).items()}
    finally:
        eng.close()
        del Yd
        torch.cuda.empty_cache()
    D = K + P if K > 0 else 0
    flops = S * N * G * (8.0 * C + 12.0 * D + 3.0)
    bytes_c = N * G * 4.0 + N * (8.0 * C + 6.0 * K + 2.0) * 4.0
    series = bool(info.get("fwd_series")) and info_end.get("series_passes", 0) > 0
    if series:
        # the contraction ran in its series form: the N G C flops of SURVEY.md section 8d are not executed, and the O(N G) work left is one pass over the matrix
        # at its STORED width -- that is the roof of such an iteration (the canonical float32 bytes would put the measured time above "the roof")
        flops = 0.0
        bytes_c = N * G * float(info["y_bytes_per_elem"]) + N * (8.0 * C + 6.0 * K + 2.0) * 4.0
    t_roof = max(flops / (PEAK_F32_TFLOPS * 1e12), bytes_c / (PEAK_HBM_GBS * 1e9))
    frac = t_roof / dt
    over = {}
    if frac > 1.0:
        # the fp32 roof prices the reference's formulation on the VECTOR unit; a shape whose sweeps run on the matrix cores (two or three bf16 parts per operand)
        # is not bound by it, and a "fraction of the roof" above one says nothing: it is not quoted, the ratio is kept under its own name
        over = {"times_the_fp32_vector_roof": frac, "roof_note": "faster than the fp32 vector roof of the reference's formulation: the contraction runs on the matrix cores in bf16 parts"}
        frac = None
    return {**({"kernel_class_us_per_iter": classes} if classes else {}), **over,
            "workload": f"synthetic {N} cells x {G} genes x {C} clones, K={K}, P={P}, S={S}" + (f" ({what})" if what else ""),
            "it_per_s": 1.0 / dt, "us_per_iter": dt * 1e6, "steps": steps, "regions": regions,
            "roof_us": t_roof * 1e6, "frac_of_roof": frac, "roof_is": "hbm (stored bytes; series form)" if series else "fp32" if flops / (PEAK_F32_TFLOPS * 1e12) >= bytes_c / (PEAK_HBM_GBS * 1e9) else "hbm",
            "fwd_mfma": bool(info["fwd_mfma"]), "bwd_mfma": bool(info["bwd_mfma"]), "fused_sweep": bool(info["fused_sweep"]), "series_form": series,
            "fwd_block_cells": int(info["fwd_block_cells"]), "update_merge": bool(info["update_merge"]), "final_elbo_finite": bool(np.isfinite(last))}


def through_api(N, G, C, K, devices, steps, warmup, regions=3, seed=20243, transport="auto", budget_ms=60.0):
    """ONE fit of the workload, cell-sharded over `devices` INSIDE THIS PROCESS -- the path the drop-in takes for
    `inference_tflow(..., devices=)` / `C_clonealign_fit(..., devices)` (ca_group_*: one engine handle + one host thread per device, joined by
    the first transport that passes its known-answer test: peer-to-peer by address -> RCCL -> host reduction).  The count matrix is handed
    over from HOST memory as one N x G matrix, like R's; loc0 = NULL (the data-driven initial values are made on the devices, over all cells).
    Times `regions` regions of `steps` ca_group_iterate iterations (median), after `warmup` iterations and ~budget_ms of pre-heat."""
    import synth_data as synth
    from clonealign_amd.engine import HipGroupEngine
    prob = synth.make_problem(N, G, C, seed=seed)
    Y = prob["Y"]
    if Y.max() <= 255:
        Y = Y.astype(np.uint8)
    rng = np.random.default_rng(seed + 1)
    psi0 = rng.normal(size=(N, K))
    t0 = time.perf_counter()
    grp = HipGroupEngine(Y, prob["L"], psi0, None, K, 1, devices=list(devices), transport=transport)
    create_s = time.perf_counter() - t0
    try:
        gi = grp.group_info()
        eps0 = rng.normal(size=(1, G)).astype(np.float32)
        grp.gamma_init(eps0)
        eps_w = rng.normal(size=(2 * max(warmup, 1), 1, G)).astype(np.float32)
        grp.iterate(max(warmup, 1), eps_w)
        eps_t = draws_for_calls(rng, steps, 1, G)
        grp.iterate(steps, eps_t)
        tb = time.perf_counter()
        while (time.perf_counter() - tb) * 1e3 < budget_ms:
            grp.iterate(steps, eps_t)
        times, last = [], float("nan")
        for _ in range(max(regions, 1)):
            t1 = time.perf_counter()
            last = grp.iterate(steps, eps_t)      # (returns after every rank has synchronised its stream and read its ELBO back)
            times.append(time.perf_counter() - t1)
        dt = float(np.median(times))
        # the default fit through the same object, from its initial values: ca_group_run_ex (every rank's trace is compared with rank 0's
        # inside the call: replicas out of step would be an error here) + 20 final ELBOs
        grp.reinit(psi0, None)
        t2 = time.perf_counter()
        trace = grp.run(None, 200, 1e-6)
        finals = grp.final_elbo(None, 20)
        fit_s = time.perf_counter() - t2
        rinfo = [grp.rank_info(r) for r in range(len(devices))]
        shard = [int(i["N"]) for i in rinfo]
        series = {"on": bool(rinfo[0]["fwd_series"]), "passes": [int(i["series_passes"]) for i in rinfo], "handed_to_the_sweeps": [int(i["series_fallbacks"]) for i in rinfo]}
        return {"series_form": series, "value": steps / dt, "unit": "iterations/s", "ms_per_step": dt / steps * 1e3, "steps": steps, "regions_ms_per_step": [t / steps * 1e3 for t in times],
                "devices": [int(d) for d in devices], "transport": gi["transport_name"], "p2p_status": gi["p2p_status"], "rccl_status": gi["rccl_status"],
                "rebuilds": gi["rebuilds"], "selftest_rounds": gi["selftest_rounds"], "note": gi["note"], "cells_per_rank": shard,
                "create_seconds": create_s, "final_elbo": last, "finite": bool(np.isfinite(last)),
                "fit_wallclock": {"seconds": fit_s, "iterations": int(len(trace) - 1), "final_elbo_mean": float(np.mean(finals))},
                "what": "the same workload as ONE fit sharded over the devices inside one process (ca_group_*: what inference_tflow(devices=) and "
                        "C_clonealign_fit(devices) run); host matrix in, loc0 made on the devices; regions timed on the host around ca_group_iterate"}
    finally:
        grp.close()


def sq_fractions(build_id, kernel_class):
    """VALU-active and MFMA-busy fractions of a kernel class from the SQ counter pass committed under profiles/ for THIS build
    (profiles/*_sq_counters.json); None when there is none.  The fp32 roof is soft for kernels whose contraction runs as bf16 MFMAs
    (the backward sweep sits above it): these two say what actually binds."""
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_sq_counters.json")), reverse=True):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        c = d.get("classes", {}).get(kernel_class)
        if d.get("build_id") == build_id and c and "valu_active_frac" in c:
            return {"valu_active_frac": c.get("valu_active_frac"), "mfma_busy_frac": c.get("mfma_busy_frac"), "file": os.path.basename(f)}
    return None


def visible_devices():
    """HIP devices this environment shows, counted in a CHILD process (the launcher itself must never touch the GPU runtime:
    its children are fresh processes, and nothing that initialised a GPU may be replaced or forked from)."""
    import subprocess
    r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True, timeout=900)
    try:
        return int(r.stdout.strip().splitlines()[-1])
    except (ValueError, IndexError):
        raise SystemExit(f"bench.py: cannot count the visible GPUs (rc {r.returncode}): {r.stderr.strip()[-300:]}")


def self_launch(n_gpus, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher around it (WORLD_SIZE unset): THIS process becomes the launcher -- the
    reference's only "many fits" entry is one call (run_clonealign(), R/clonealign.R:50-56), not a launcher recipe.  It imports
    neither torch nor the engine; it starts N fresh rank processes of this same script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*
    set, rendezvous on 127.0.0.1), lets rank 0's JSON line through on the inherited stdout, takes the others down when one fails,
    and exits with the worst rank's code.  Never a silent 1-GPU run: fewer visible devices than ranks is an error unless the
    plumbing-test override CLONEALIGN_BENCH_DEVICE puts every rank on one device."""
    import socket
    import subprocess
    if "CLONEALIGN_BENCH_DEVICE" not in os.environ:
        have = visible_devices()
        if have < n_gpus:
            print(f"bench.py: --gpus {n_gpus} but {have} GPU(s) visible; refusing to report a {n_gpus}-GPU number from fewer devices "
                  "(CLONEALIGN_BENCH_DEVICE=<ordinal> puts every rank on one device for plumbing tests)", file=sys.stderr, flush=True)
            return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n_gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_gpus), LOCAL_WORLD_SIZE=str(n_gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", CLONEALIGN_BENCH_SELF_LAUNCHED="1")
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n_gpus)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv], env=env))
    worst, t_fail = 0, None
    while any(p.poll() is None for p in procs):
        for p in procs:
            rc = p.poll()
            if rc is not None and rc != 0 and t_fail is None:
                worst, t_fail = rc, time.time()
        # a rank that failed leaves its peers in a barrier or in a device-side wait (bounded: comm_timeout_ms); give them that
        # long to say why and go, then end exactly the processes started here
        if t_fail is not None and time.time() - t_fail > 20.0:
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            time.sleep(3.0)
            for p in procs:
                if p.poll() is None:
                    p.kill()
        time.sleep(0.05)
    for r, p in enumerate(procs):
        if p.returncode != 0:
            print(f"bench.py launcher: rank {r} exited with code {p.returncode}", file=sys.stderr, flush=True)
            if worst == 0:
                worst = p.returncode
    return worst if worst >= 0 else 128 - worst   # (a signal's negative code as the shell would report it)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)   # the reference's default max_iter
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--repeats", type=int, default=5, help="the --steps long timed region is run this many times; value = median region")
    ap.add_argument("--cells", type=int, default=100_000)
    ap.add_argument("--genes", type=int, default=5_000)
    ap.add_argument("--clones", type=int, default=8)
    ap.add_argument("--latent", type=int, default=1)
    ap.add_argument("--y-storage", default="auto")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-cells", type=int, default=0,
                    help="cells handed to the CPU baseline; 0 = all of them when the host has 10x the matrix free, else 65536")
    ap.add_argument("--seed", type=int, default=20243)
    ap.add_argument("--collective", default="auto", choices=["auto", "p2p", "rccl", "host"],
                    help="data-path all-reduce at --gpus > 1: auto = one-shot peer-to-peer, else RCCL")
    ap.add_argument("--allow-host-fallback", action="store_true",
                    help="at --gpus > 1, let the run continue on the gloo host all-reduce when no device transport comes up "
                         "(the result is then NOT a measurement of the device data path)")
    ap.add_argument("--selftest-fail", default="", help="test hook: treat the known-answer all-reduce test of this transport (p2p | rccl) as failed on rank 0")
    ap.add_argument("--preheat-ms", type=float, default=60.0,
                    help="untimed iterations run for this long right before the timed regions (after the --warmup steps): the GPU's clocks need "
                         "~30 ms of load to ramp; 0 = none")
    ap.add_argument("--busy-seconds", type=float, default=4.0,
                    help="untimed iterations run for this long after the measurements (single GPU), so that a coarse GPU-utilisation sampler sees the device at work")
    ap.add_argument("--allow-foreign-lib", action="store_true",
                    help="run a library whose ca_build_id() is not the tree's (CLONEALIGN_HIP_LIB timing builds; labelled in the output)")
    ap.add_argument("--no-live-events", action="store_true",
                    help="A/B only: no HIP events around the dominant kernel in the timed region (roofline then comes from the warm-up)")
    ap.add_argument("--variant-off", default="", help="comma-separated engine variants to switch off (engine.VARIANTS), for A/B runs")
    ap.add_argument("--variant-on", default="", help="comma-separated opt-in engine variants (engine.VARIANTS_ON), for A/B runs")
    ap.add_argument("--tune", default="", help="comma-separated name=value decomposition overrides (engine.TUNE), for A/B runs")
    ap.add_argument("--steady-steps", type=int, default=-1,
                    help="length of the extra one-region steady-state measurement after the headline regions; -1 = 200 unless --steps is 200 "
                         "or a profiler is attached (its launches would mix into the per-launch statistics), 0 = none")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the 200-step regions of the other single-GPU BASELINE configurations (cfg-2, one cfg-5 restart, the 12.5k-cell shard) "
                         "and of the VALU fallback shapes that follow the headline measurement (single GPU only)")
    ap.add_argument("--through-api", action="store_true",
                    help="run the workload as ONE fit cell-sharded over --gpus devices INSIDE THIS PROCESS (the device group behind "
                         "inference_tflow(devices=) / C_clonealign_fit(devices)) instead of one process per GPU; no launcher is used or needed")
    ap.add_argument("--no-through-api-leg", action="store_true",
                    help="at --gpus > 1 (one process per GPU): skip the extra measurement in which rank 0, after the headline, runs the same workload "
                         "through the in-process device group over all the devices")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.through_api:
        if int(os.environ.get("WORLD_SIZE", "1")) != 1:
            raise SystemExit("bench.py --through-api is ONE process driving --gpus devices: run it bare, not under a launcher")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import torch   # (first: torch's bundled HIP runtime has to be the first one initialised)
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU fallback")
        from clonealign_amd import engine as eng_mod
        if eng_mod.build_id() != eng_mod.source_build_id() and not args.allow_foreign_lib:
            raise SystemExit("bench.py: the engine library is not built from this tree")
        if "CLONEALIGN_BENCH_DEVICE" in os.environ:   # plumbing test on a 1-GPU box: every rank on the same device (host transport)
            devices = [int(os.environ["CLONEALIGN_BENCH_DEVICE"])] * args.gpus
        else:
            if eng_mod.device_count() < args.gpus:
                raise SystemExit(f"bench.py --through-api: --gpus {args.gpus} but {eng_mod.device_count()} device(s) visible "
                                 "(CLONEALIGN_BENCH_DEVICE=<ordinal> repeats one device for plumbing tests)")
            devices = list(range(args.gpus))
        N, G, C, K = args.cells, args.genes, args.clones, args.latent
        r = through_api(N, G, C, K, devices, args.steps, args.warmup, max(args.repeats, 1), args.seed,
                        {"auto": "auto", "p2p": "p2p", "rccl": "rccl", "host": "host"}[args.collective])
        if not r["finite"]:
            raise SystemExit(f"non-finite ELBO after the timed steps: {r['final_elbo']}")
        print(json.dumps({
            "metric": "ELBO iterations/sec", "value": r["value"], "unit": "iterations/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": r["ms_per_step"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": ("f32 variables (contraction: float64 series over gene bins; count-matrix products: int8 MFMA, exact)" if r["series_form"]["on"] and r["series_form"]["passes"][0] > 0
                      else "f32 (bf16x3-split MFMA contraction, fp32 accumulate)"), "data": "synthetic",
            "config": {"workload": f"synthetic {N} cells x {G} genes x {C} clones, K={K}, S=1, ONE fit cell-sharded over {args.gpus} device(s) of one process "
                                   f"(BASELINE.json configs[{2 if args.gpus == 1 else 3}] through the drop-in's device group)",
                       "cells": N, "genes": G, "clones": C, "K": K, "parallelism": f"cells/{args.gpus}, one process (ca_group)", "collective": r["transport"],
                       "build_id": eng_mod.build_id()},
            "through_api": r, "roofline": None, "cpu_baseline": None}), flush=True)
        return

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "CLONEALIGN_BENCH_DEVICE" in os.environ:   # plumbing test on a 1-GPU box: every rank on the same device
        local_rank = int(os.environ["CLONEALIGN_BENCH_DEVICE"])
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher around this process: be the launcher (before torch or the engine are imported -- nothing here has touched a GPU)
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        # never a silent run at another width than the one asked for: `--gpus 8` under a 1-process launcher is an error, not a 1-GPU number
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch {args.gpus} ranks, or run `python bench.py --gpus {args.gpus}` "
                         "bare and let it start them")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # (CPU baseline placement: its float32 matrix is first touched inside the baseline's own OpenMP loop, static schedule, the same
    #  partition every pass uses -- oracle/c/clonealign_simd.c.  Pinning the team through OMP_PROC_BIND here was tried and removed:
    #  libgomp then pins THIS thread to one core, the engine's Philox generator threads inherit that mask, and the 200-iteration fit
    #  below took 0.091 s instead of 0.063.)

    import torch
    import torch.distributed as dist
    from clonealign_amd import engine as eng_mod
    from clonealign_amd import sharding
    import synth_data as synth
    from clonealign_amd.engine import HipEngine, comm_unique_id

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU fallback")
    # the number of record must come from the sources in this tree: a stale .so, or a timing build reached through
    # CLONEALIGN_HIP_LIB (tools/lab_ab.sh builds some that give wrong results on purpose), is refused
    foreign = eng_mod.build_id() != eng_mod.source_build_id()
    if foreign and not args.allow_foreign_lib:
        raise SystemExit(f"bench.py: the engine library's build id {eng_mod.build_id()} is not the tree's {eng_mod.source_build_id()} "
                         "(stale build or CLONEALIGN_HIP_LIB override); rebuild, or pass --allow-foreign-lib for a labelled A/B run")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)   # control plane only; the data path is in the engine

    N, G, C, K = args.cells, args.genes, args.clones, args.latent
    lo, hi = sharding.cell_range(N, rank, world)
    Yd, aux = synth.make_problem_torch(N, G, C, seed=args.seed, rows=(lo, hi), device=f"cuda:{local_rank}")
    n_loc = hi - lo
    # data-driven loc0 (R/inference-tflow.R:220-235,262) from GLOBAL column means; psi0 ~ N(0,1)
    rm = Yd.sum(1, keepdim=True).to(torch.float64) / G
    colacc = torch.zeros(G, dtype=torch.float64, device=Yd.device)
    for b0 in range(0, n_loc, 8192):
        colacc += (Yd[b0:b0 + 8192].to(torch.float64) / rm[b0:b0 + 8192]).sum(0)
    colacc = colacc.cpu()
    if world > 1:
        dist.all_reduce(colacc)
    from clonealign_amd.hostprep import safe_inverse_softplus
    loc0 = safe_inverse_softplus(np.maximum(colacc.numpy() / N, 1e-6))
    psi0 = np.random.default_rng(args.seed + 1).normal(size=(N, K))[lo:hi]
    torch.cuda.synchronize()

    def make_engine(**kw):
        return HipEngine(None, aux["L"], psi0, loc0, K, 1, y_device_ptr=Yd.data_ptr(), y_device_dtype=np.int32,
                         shape=(n_loc, G), device=local_rank, y_storage=args.y_storage, rank=rank, world=world, profile=0,
                         variant_off=tuple(v for v in args.variant_off.split(",") if v),
                         variant_on=tuple(v for v in args.variant_on.split(",") if v),
                         tune={k: (v if ":" in v else int(v)) for k, v in (kv.split("=") for kv in args.tune.split(",") if kv)}, **kw)

    collective, tried, selftest = "none", [], {}
    if world == 1:
        eng = make_engine()
    else:
        # Device data path: one-shot peer-to-peer all-reduce (IPC inbox slabs over xGMI), else RCCL.  Every transport is
        # brought up collectively -- all ranks succeed or all move on to the next one -- and the run FAILS when none of the
        # device transports comes up, unless the host fallback was asked for explicitly.
        def gloo_sum(buf):
            t = torch.from_numpy(buf.copy())
            dist.all_reduce(t)
            buf[:] = t.numpy()

        def exchange(hd):
            box = [None] * world
            dist.all_gather_object(box, hd)
            return box

        order = {"auto": ["p2p", "rccl"], "p2p": ["p2p"], "rccl": ["rccl"], "host": []}[args.collective]
        if args.allow_host_fallback or args.collective == "host":
            order.append("host")
        eng = None
        for tr in order:
            ok, e_new, why = 1, None, ""
            try:
                if tr == "p2p":
                    e_new = make_engine(p2p_exchange=exchange)
                elif tr == "rccl":
                    box = [None]
                    try:
                        if rank == 0:
                            box = [comm_unique_id()]
                    except Exception as ex:  # noqa: BLE001
                        why = str(ex)
                    dist.broadcast_object_list(box, src=0)
                    if box[0] is None:
                        raise RuntimeError("RCCL unavailable on rank 0: " + why)
                    e_new = make_engine(comm_id=box[0])
                else:
                    e_new = make_engine(host_allreduce=gloo_sum)
            except Exception as ex:  # noqa: BLE001
                ok, why = 0, str(ex)
                print(f"[rank {rank}] transport {tr} failed: {ex}", file=sys.stderr, flush=True)
            flag = torch.tensor([ok], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            tried.append(tr)
            if int(flag[0]) == 1:
                # it came up everywhere: before it carries a number, it must ADD -- known-answer all-reduces of the train pass's payload
                # (the peer-to-peer transport has never run across xGMI on any box this build has seen; every rank takes part, then the
                # ranks agree on the verdict)
                try:
                    bad = e_new.comm_selftest(48)
                except Exception as ex:  # noqa: BLE001
                    bad = -1
                    print(f"[rank {rank}] transport {tr}: known-answer test failed to run: {ex}", file=sys.stderr, flush=True)
                if args.selftest_fail == tr and rank == 0:
                    bad = 1
                if bad:
                    print(f"[rank {rank}] transport {tr}: known-answer all-reduce wrong ({bad} sums)", file=sys.stderr, flush=True)
                flag = torch.tensor([0 if bad else 1], dtype=torch.int32)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                selftest[tr] = bool(int(flag[0]))
            if int(flag[0]) == 1:
                eng, collective = e_new, tr
                break
            if e_new is not None:
                e_new.close()
        if eng is None:
            if rank == 0:
                print(f"bench.py: no device all-reduce transport came up on all {world} ranks (tried {tried}); refusing to report a "
                      "scaling number from a host fallback (pass --allow-host-fallback to run it anyway)", file=sys.stderr, flush=True)
            dist.barrier()
            raise SystemExit(3)
        if collective == "host":
            collective = "gloo-host-fallback"
    info = eng.info()
    if world > 1:
        assert info["transport_name"] == {"p2p": "p2p", "rccl": "rccl"}.get(collective, "host"), info["transport_name"]
        plan = sharding.reduce_plan(G, C, K, 0, 1, series=bool(info["fwd_series"]), world=world)
        assert info["red_n"] == plan["total"], (info["red_n"], plan)
    # The CPU baseline's copy of the counts is taken AFTER the GPU measurements (the generated matrix stays on the device until
    # then): fetching and converting 2 GB here left the GPU idle for seconds in front of the timed region, and W = 5 warm-up
    # iterations (1.7 ms) do not bring its clocks back -- the driver-style run read 3 % low whenever the baseline was on.
    want_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline
    # The generated int32 matrix (2 GB at cfg-3) is dropped before anything is timed, ALSO when the CPU baseline will want its counts: kept alive beside the
    # engine it made the 20-step regions read 2 % low (188.4 against 184.3 us per iteration on one box, with and without --no-cpu-baseline; round 6).  The
    # baseline's leg generates it again from the same seed after the measurements (its parity check against the engine would show any difference).
    del Yd
    torch.cuda.empty_cache()

    rng = np.random.default_rng(args.seed + 2 + 0)   # same eps on every rank
    eps0 = rng.normal(size=(1, G)).astype(np.float32)
    eng.gamma_init(eps0)
    # --- warmup (+ find the dominant kernel class with all classes timed)
    eng.set_profile(0x1F)
    eps_w = rng.normal(size=(2 * max(args.warmup, 1), 1, G)).astype(np.float32)
    eng.iterate(max(args.warmup, 1), eps_w)
    kt = eng.kernel_times(reset=True)
    dominant = max(("fwd", "bwd", "ypass"), key=lambda k: kt[k][0])
    if info.get("fwd_series") and kt["ypass"][0] > 0:
        # the series form: the count-matrix stream is the one launch whose work is proportional to the matrix (HBM-bound); the "fwd" class there is a handful of
        # small launches and the cell kernel, with no flop count of the reference's formulation to price them against -- at 2+ ranks (half the matrix per rank)
        # their summed warm-up time can edge past the stream's, and the sweeps' flop roofline would be quoted for kernels that do not execute those flops
        dominant = "ypass"
    kid = {"fwd": 0, "bwd": 1, "ypass": 2}[dominant]
    # live HIP events around every 8th launch of the dominant class (an event pair costs the stream 5-6 us: 1.6 % at cfg-3, 9 % at
    # cfg-2 when every launch is timed -- measured with --no-live-events; sampled, the timed region is left alone)
    eng.set_profile(0 if args.no_live_events else (1 << kid) | ((EVENT_STRIDE - 1) << 8))
    eps_t = draws_for_calls(rng, args.steps, 1, G)   # (2 K + 1 draws: every timed call carries the next call's forward half, K sweeps per K iterations)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        eng.synchronize()

    # --- clock pre-heat (untimed, after the W warm-up steps): the device needs ~30 ms of continuous load to reach its operating
    #     clocks -- kernel trace of a default run (profiles/r03_v1_timeline.txt): the iteration period falls from
    #     322 us right after an idle gap to 289 us 30 ms later, the merged forward launch from 151 to 135 us.  W = 5 warm-up steps are
    #     1.6 ms of work, and a 20-step timed region is 6 ms: without this the whole region sits on the ramp and measures the power
    #     management, not the kernels.  The same iterations, the same arguments as the timed call; `preheat_ms` is in the output.
    #     At world > 1 every ca_iterate call is collective (one all-reduce per pass), so the NUMBER of pre-heat calls must be the same
    #     on every rank: one call is timed, the slowest rank's time is agreed on (gloo MAX) and the call count follows from it --
    #     a loop that lets each rank's own clock decide ran one call more on one rank than on its peer (r03: CA_ERR_COMM after 10 s).
    pre_it, pre_calls, t_pre = 0, 0, time.perf_counter()
    if args.preheat_ms > 0:
        barrier()
        t_pre = time.perf_counter()
        pre_calls = run_agreed_calls(lambda: eng.iterate(args.steps, eps_t), args.preheat_ms, dist if world > 1 else None)
        pre_it = pre_calls * args.steps
    else:
        eng.iterate(args.steps, eps_t)   # (one untimed call all the same: the first timed region must start from a carried half like the others)
    pre_ms = (time.perf_counter() - t_pre) * 1e3
    # --- timed: `repeats` regions of exactly `steps` iterations, each bracketed by barrier + synchronize on both sides and
    #     maxed over the ranks; the quoted value is the MEDIAN region (a 20-step region at cfg-3 is only 6 ms long)
    regions, last = [], float("nan")
    for _ in range(max(args.repeats, 1)):
        barrier()
        t0 = time.perf_counter()
        last = eng.iterate(args.steps, eps_t)
        eng.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t[0])
        regions.append(dt)
    dt = float(np.median(regions))
    kt_timed = eng.kernel_times(reset=True)
    info = {**info, **{k: v for k, v in eng.info().items() if k in ("series_passes", "series_fallbacks")}}   # (how the passes so far were carried out)
    # the same call at the reference's default max_iter = 200 (one region): a ca_iterate call of k steps makes k + 1 forward sweeps
    # (first and last carry one draw), so a 20-step region pays 21/20 of the steady-state sweep cost -- this shows the difference
    steady = None
    profiled = "rocprof" in os.environ.get("LD_PRELOAD", "") or any(k.startswith("ROCPROF") for k in os.environ)
    steady_steps = args.steady_steps if args.steady_steps >= 0 else (0 if (args.steps == 200 or profiled) else 200)
    if steady_steps > 0:
        # (under a profiler this region's 2 x steady_steps iterations -- steady_steps + 1 sweeps per call, another call shape -- would mix into the
        #  per-launch statistics of the --steps regions: off by default there)
        eps_s = draws_for_calls(rng, steady_steps, 1, G)
        eng.set_profile(0)
        eng.iterate(steady_steps, eps_s)
        barrier()
        t0 = time.perf_counter()
        eng.iterate(steady_steps, eps_s)
        eng.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        ds = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([ds], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            ds = float(t[0])
        steady = {"steps": steady_steps, "value": steady_steps / ds, "ms_per_step": ds / steady_steps * 1e3,
                  "what": f"one ca_iterate({steady_steps}) region, same barriers; not the headline (the headline is the --steps region above)"}
    # the replicas after everything the timed regions did to them: every rank applied the same all-reduced sums in the same order, so the
    # replicated variables must be bit-identical across ranks (a transport that tears or reorders rarely would show here first)
    replicas_equal = None
    if world > 1:
        import hashlib
        hsh = hashlib.sha1()
        for nm in ("loc", "ls", "W", "alpha_unconstr", "v"):
            hsh.update(np.ascontiguousarray(eng.get(nm)).tobytes())
        box = [None] * world
        dist.all_gather_object(box, hsh.hexdigest())
        replicas_equal = len(set(box)) == 1
    # a monitor pass on its own (plain forward + its (3 + C)-double all-reduce + read-back): the latency floor of one collective
    mon_us = None
    if world > 1:
        eng.set_profile(0)
        barrier()
        t1 = time.perf_counter()
        for _ in range(20):
            eng.elbo(eps0)
        eng.synchronize()
        mon_us = (time.perf_counter() - t1) / 20 * 1e6
    # the collective on its own: 200 back-to-back all-reduces of the train pass's payload (15 011 doubles at cfg-4) on the
    # transport in use and -- when it also comes up on every rank -- on RCCL, between two HIP events on the engine's stream
    ar_us = None
    if world > 1 and collective in ("p2p", "rccl"):
        ar_us = {"payload_doubles": int(info["red_n"]), "calls": 200}
        try:
            ar_us[collective] = eng.comm_benchmark(collective, 200)
            if collective == "p2p":
                box, why = [None], ""
                try:
                    if rank == 0:
                        box = [comm_unique_id()]
                except Exception as ex:  # noqa: BLE001
                    why = str(ex)
                dist.broadcast_object_list(box, src=0)
                ok = 0
                if box[0] is not None:
                    try:
                        eng.comm_init(box[0])
                        ok = 1
                    except Exception as ex:  # noqa: BLE001
                        why = str(ex)
                flag = torch.tensor([ok], dtype=torch.int32)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                if int(flag[0]) == 1:
                    ar_us["rccl"] = eng.comm_benchmark("rccl", 200)
                else:
                    ar_us["rccl"] = None
                    ar_us["rccl_unavailable"] = why[:200]
        except Exception as ex:  # noqa: BLE001
            ar_us["error"] = str(ex)[:200]
    # second half of the metric ("wall-clock to convergence"): the reference's default fit FROM ITS INITIAL VALUES (ca_reinit:
    # variables, Adam slots and beta powers reset), max_iter = 200, rel_tol = 1e-6, through ca_run (host reads the ELBO every
    # iteration for the window-10 stop rule), then the 20 final ELBOs
    eng.set_profile(0)
    eng.reinit(psi0, loc0)
    barrier()
    t1 = time.perf_counter()
    trace = eng.run(None, 200, 1e-6)
    finals = eng.final_elbo(None, 20)
    eng.synchronize()
    fit_s = time.perf_counter() - t1
    # every other single-GPU BASELINE configuration on the same clock, after the headline and outside its regions: cfg-2, one
    # restart of cfg-5, the shards of cfg-4 at 8 and 4 GPUs as one-device problems; and the argument space outside the fused loop
    # (mc_samples > 2, more than 16 clones, D = K + P >= 3: vector sweeps up to round 5, matrix-core sweeps since round 6), so that their cost is a number.  Skipped under a profiler (their
    # launches would mix into the per-kernel statistics of the headline shape).
    other, fallbacks = None, None
    if world == 1 and rank == 0 and not args.no_other_configs and not profiled:
        other, fallbacks = {}, {}
        for nm, kw in (("cfg2", dict(N=10_000, G=2_000, C=4, what="BASELINE.json configs[1]")),
                       ("cfg5_one_restart", dict(N=50_000, G=3_000, C=6, what="one restart of BASELINE.json configs[4]")),
                       ("shard_12500", dict(N=12_500, G=5_000, C=8, what="an UNSHARDED problem of the size of one rank's shard of configs[3] at 8 GPUs; a rank of a sharded fit adds the collective's launch and picks series / sweeps by its own rule, DESIGN.md section 6")),
                       ("shard_25000", dict(N=25_000, G=5_000, C=8, what="an UNSHARDED problem of the size of one rank's shard of configs[3] at 4 GPUs; a rank of a sharded fit adds the collective's launch and picks series / sweeps by its own rule, DESIGN.md section 6")),
                       ("shard_50000", dict(N=50_000, G=5_000, C=8, what="an UNSHARDED problem of the size of one rank's shard of configs[3] at 2 GPUs; a rank of a sharded fit adds the collective's launch and picks series / sweeps by its own rule, DESIGN.md section 6"))):
            try:
                other[nm] = side_config(nm, device=local_rank, **kw)
            except Exception as ex:  # noqa: BLE001
                other[nm] = {"error": str(ex)[:200]}
        for nm, kw in (("S3", dict(N=N, G=G, C=C, S=3, what="mc_samples = 3: plain passes, since round 6 with matrix-core sweeps over pairs of samples")),
                       ("C20", dict(N=N, G=G, C=20, what="20 clones: plain passes, since round 6 with matrix-core sweeps over pairs of clone chunks")),
                       ("K2P1", dict(N=N, G=G, C=C, K=2, P=1, what="D = K + P = 3: matrix-core sweeps since round 6, the count-matrix stream a launch of its own")),
                       ("K1P2", dict(N=N, G=G, C=C, K=1, P=2, what="D = K + P = 3 with one latent dimension and two covariates"))):
            try:
                fallbacks[nm] = side_config(nm, device=local_rank, steps=40, regions=2, **kw)
            except Exception as ex:  # noqa: BLE001
                fallbacks[nm] = {"error": str(ex)[:200]}
    # keep the device demonstrably busy for a few seconds after the measurements (untimed): the timed regions of a 20-step run are
    # 35 ms inside a minute of CPU baseline, which a once-per-few-seconds utilisation sampler around the run never lands on
    busy_s, busy_it = 0.0, 0
    if world == 1 and args.busy_seconds > 0:
        tb = time.perf_counter()
        while time.perf_counter() - tb < args.busy_seconds:
            eng.iterate(args.steps, eps_t)
            eng.synchronize()
            busy_it += args.steps
        busy_s = time.perf_counter() - tb
    if not np.isfinite(last):
        raise SystemExit(f"non-finite ELBO after the timed steps: {last}")
    # The drop-in's own way to use several GPUs: after the headline, with every rank's engine closed, rank 0 ALONE runs the
    # same workload as one fit over all the devices through the in-process device group -- the driver's multi-GPU command thereby exercises the
    # path inference_tflow(devices=) / C_clonealign_fit(devices) take, next to the one-process-per-GPU number.  Never part of `value`; an error
    # in it is reported inside the line, it cannot fail the headline.
    api_leg = None
    if world > 1 and not args.no_through_api_leg:
        keep_info = info
        eng.close()
        dist.barrier()
        if rank == 0:
            # a fresh CHILD process (`bench.py --through-api`, bare: no launcher variables), under a time limit: a transport that has never run on
            # this hardware must not be able to hang or crash the process that holds the headline
            import subprocess
            env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                                                     "GROUP_RANK", "ROLE_RANK", "ROLE_WORLD_SIZE", "TORCHELASTIC_RUN_ID", "CLONEALIGN_BENCH_SELF_LAUNCHED")}
            cmd = [sys.executable, os.path.abspath(__file__), "--through-api", "--gpus", str(world), "--steps", str(args.steps), "--warmup", str(args.warmup),
                   "--repeats", "3", "--cells", str(N), "--genes", str(G), "--clones", str(C), "--latent", str(K), "--seed", str(args.seed)]
            try:
                r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=420)
                line = next((l for l in reversed(r.stdout.splitlines()) if l.startswith("{")), None)
                api_leg = json.loads(line)["through_api"] if (r.returncode == 0 and line) else {"error": f"rc {r.returncode}: {(r.stderr or r.stdout)[-400:]}"}
            except subprocess.TimeoutExpired:
                api_leg = {"error": "timed out after 420 s (child process ended)"}
            except Exception as ex:  # noqa: BLE001
                api_leg = {"error": f"{type(ex).__name__}: {str(ex)[:400]}"}
        dist.barrier()
        info = keep_info

    if rank == 0:
        build = eng_mod.build_id()
        ms, launches = kt_timed[dominant] if not args.no_live_events else kt[dominant]
        per_launch_s = ms / max(launches, 1) * 1e-3
        ride = bool(info.get("y_ride")) and dominant == "fwd"
        bound, work = algorithmic_work(dominant, n_loc, G, C, K, bool(info.get("fused_sweep")), args.steps, ride)
        canonical_work = work
        if dominant == "ypass":
            # the count-matrix stream is the dominant launch (series form of the contraction): its roof is HBM, and what it MOVES is the matrix at its
            # stored width -- the canonical 4 B per count of SURVEY.md section 8d would read as 2.6x the HBM peak (a fraction above one by
            # construction is not printed); the canonical figure is kept beside it, labelled
            work = n_loc * G * float(info["y_bytes_per_elem"]) + (n_loc + G) * K * 4.0 * 2
        if bound == "hbm":
            achieved, peak, unit = work / per_launch_s / 1e9, PEAK_HBM_GBS, "GB/s"
        else:
            achieved, peak, unit = work / per_launch_s / 1e12, PEAK_F32_TFLOPS, "TFLOP/s"
        same_workload = (N, G, C, K, world) == (100_000, 5_000, 8, 1, 1)
        traffic, traffic_src = pmc_traffic(build, dominant) if same_workload else (None, None)
        # the HBM-bound kernel of the iteration (SURVEY.md section 8d asks for both roofs): the Y stream, timed by HIP events on
        # its own stream during the warmup iterations, where every kernel class is timed
        ystream = None
        if K > 0 and (kt["ypass"][1] > 0 or ride):
            # riding: the stream has no launch of its own -- its bytes are spread over the merged launch's whole duration
            y_s = per_launch_s if ride else kt["ypass"][0] / kt["ypass"][1] * 1e-3
            canon = n_loc * G * 4.0 + (n_loc + G) * K * 4.0 * 2
            ytraffic, ysrc = pmc_traffic(build, "fwd" if ride else "ypass") if same_workload else (None, None)
            stored = n_loc * G * float(info["y_bytes_per_elem"])
            ystream = {"bound": "hbm", "kernel": "ypass (blocks inside the forward sweep's launch)" if ride else "ypass",
                       "achieved": stored / y_s / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": stored / y_s / 1e9 / PEAK_HBM_GBS,
                       "traffic": ytraffic, "traffic_source": ysrc, "launch_ms": y_s * 1e3, "canonical_bytes": canon, "stored_bytes": stored,
                       "note": f"physical rate: the {info['y_bytes_per_elem']} B per count the matrix is stored at, over the launch the stream rides in.  (The "
                               "canonical 4 B per count of SURVEY.md section 8d would read as more than the HBM peak here -- bytes this build does not move; "
                               "that figure is no longer printed.)  HBM is not what binds this launch: see roofline.binding"}
        step_s = dt / args.steps
        it_flops = N * G * (8.0 * C + 12.0 * K + 3.0)                       # SURVEY.md section 8d, whole iteration, all ranks
        it_bytes = N * G * 4.0 + N * (8.0 * C + 6.0 * K + 2.0) * 4.0
        it_bytes_stored = N * G * float(info["y_bytes_per_elem"]) + N * (8.0 * C + 6.0 * K + 2.0) * 4.0   # the same formula at the stored width
        meas = [pmc_traffic(build, k)[0] for k in (("ypass", "fwd", "bwd") if dominant == "ypass" else ("fwd", "bwd"))] if same_workload else [None]
        it_bytes_measured = float(sum(meas)) if all(m is not None for m in meas) else None               # PMC: forward (+ riding Y stream) + backward launch
        out = {
            "metric": "ELBO iterations/sec", "value": args.steps / dt, "unit": "iterations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": step_s * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": ("f32 variables (contraction: float64 series over gene bins; count-matrix products: int8 MFMA, exact)" if info.get("fwd_series") and info.get("series_passes", 0) > 0
                      else "f32 (bf16x3-split MFMA contraction, fp32 accumulate)" if info.get("fwd_mfma") else "f32"),
            "data": "synthetic",
            "config": {"workload": f"synthetic {N} cells x {G} genes x {C} clones, K={K}, S=1, cell-sharded over "
                                   f"{world} GPU(s) (BASELINE.json configs[{2 if world == 1 else 3}])",
                       "cells": N, "genes": G, "clones": C, "K": K, "mc_samples": 1, "learning_rate": 0.1,
                       "y_storage": info["y_storage_name"], "y_bytes_per_elem": info["y_bytes_per_elem"],
                       "fused_sweep": bool(info.get("fused_sweep")), "fwd_mfma": bool(info.get("fwd_mfma")), "series_form": bool(info.get("fwd_series")),
                       "bwd_mfma": bool(info.get("bwd_mfma")), "y_mfma": bool(info.get("y_mfma")),
                       "parallelism": f"cells/{world}" if world > 1 else "single", "collective": collective,
                       "collectives_tried": tried, "allreduce_selftest": selftest, "allreduce_doubles_per_train_pass": int(sharding.reduce_plan(G, C, K, 0, 1, series=bool(info["fwd_series"]), world=world).get("series_total", info["red_n"])) if world > 1 else int(info["red_n"]),
                       "build_id": build, **({"foreign_library": True} if foreign else {})},
            "repeats": {"n": len(regions), "ms_per_step_median": step_s * 1e3, "ms_per_step_min": min(regions) / args.steps * 1e3,
                        "ms_per_step_max": max(regions) / args.steps * 1e3,
                        "what": "the --steps long region timed this many times (barrier + synchronize on both sides each time); "
                                "value / ms_per_step are the median region"},
            "roofline": {"bound": bound, "kernel": dominant, "achieved": achieved, "peak": peak, "unit": unit,
                         "frac": achieved / peak, "traffic": traffic, "traffic_source": traffic_src,
                         **({"bytes_per_launch_stored": work, "bytes_per_launch_canonical_f32": canonical_work,
                             "series_form": {"passes": int(info.get("series_passes", 0)), "handed_to_the_sweeps": int(info.get("series_fallbacks", 0))}}
                            if dominant == "ypass" else {}),
                         "launch_ms": per_launch_s * 1e3, "launches": int(launches),
                         **({k: v for k, v in (sq_fractions(build, dominant) or {}).items() if k != "file"} if same_workload else {}),
                         "sq_source": (sq_fractions(build, dominant) or {}).get("file") if same_workload else None,
                         "event_stride": 1 if args.no_live_events else EVENT_STRIDE,
                         "binding": ("HBM latency / occupancy of the stream kernel (5.3-5.7 TB/s of the 8 TB/s peak with every CU streaming)" if dominant == "ypass" else
                                     "vector issue (VALU + MFMA share the SIMD's issue port; valu_active_frac / mfma_busy_frac from the SQ counter pass of this "
                                     "build; HBM: traffic per launch against launch_ms is ~0.5 of peak)"),
                         "note": ("the count-matrix stream (Y.W and Y^T.psi on the int8 matrix cores, one pass over the 1-byte matrix) is the iteration's longest launch: "
                                  "the cells x genes x clones contraction runs in its series form (ca_poly.hip: moments over gene bins, O(N + G) work), so the "
                                  "O(N G) work left is this stream; achieved = stored bytes / launch time"
                                  if dominant == "ypass" else
                                  "algorithmic fp32 flops of the fused two-eps sweep against the fp32 peak; the contraction "
                                  "itself runs as bf16 hi/lo MFMAs (fp32-accurate), the kernel is bound by VALU issue "
                                  "(v_exp_f32 + bf16 split)"
                                  + ("; the Y stream's blocks ride inside this launch (its 4K flop per count are counted, its "
                                     "512 MB of HBM reads are roofline_ystream's), so launch_ms is sweep + stream"
                                     if ride else " and shares the GPU with the Y-stream kernel on a side stream")
                                  + " (DESIGN.md sections 5 and 8); traffic is filled only from a PMC file of THIS build")
                         if dominant == "fwd" else "traffic is filled only from a PMC file of THIS build"},
            "roofline_iteration": (
                {"reference_flops": it_flops, "reference_bytes_canonical": it_bytes,
                 "reference_roofs_ms": {"fp32": it_flops / (PEAK_F32_TFLOPS * 1e12 * world) * 1e3, "hbm_canonical": it_bytes / (PEAK_HBM_GBS * 1e9 * world) * 1e3},
                 "measured_ms": step_s * 1e3,
                 "bytes_stored": it_bytes_stored, "frac_of_hbm_peak_stored": it_bytes_stored / step_s / 1e9 / (PEAK_HBM_GBS * world),
                 "bytes_measured_pmc": it_bytes_measured,
                 "frac_of_hbm_peak_measured": None if it_bytes_measured is None else it_bytes_measured / step_s / 1e9 / (PEAK_HBM_GBS * world),
                 "what": "whole iteration (train + monitor pass).  reference_flops / reference_bytes_canonical are what the REFERENCE's formulation costs (SURVEY.md section 8d: "
                         "N G (8C + 12K + 3) flop, float32 counts) and reference_roofs_ms the time either roof allows for it; the measured iteration is shorter than both "
                         "because the series form of the contraction (ca_poly.hip) executes O(N + G) work instead of the N G C flops and the matrix is held at one byte "
                         "per count -- so no fraction of those roofs is quoted (it would exceed one by construction).  What the iteration is bound by: one pass over the stored "
                         "matrix (frac_of_hbm_peak_stored) plus a chain of small launches (DESIGN.md section 5e)"}
                if (info.get("fwd_series") and info.get("series_passes", 0) > 0) else
                {"flops": it_flops, "bytes_canonical": it_bytes,
                 "achieved_TFLOPs": it_flops / step_s / 1e12,
                 "frac_of_fp32_peak": it_flops / step_s / 1e12 / (PEAK_F32_TFLOPS * world),
                 "achieved_GBps_canonical": it_bytes / step_s / 1e9,
                 "frac_of_hbm_peak_canonical": it_bytes / step_s / 1e9 / (PEAK_HBM_GBS * world),
                 "bytes_stored": it_bytes_stored,
                 "frac_of_hbm_peak_stored": it_bytes_stored / step_s / 1e9 / (PEAK_HBM_GBS * world),
                 "bytes_measured_pmc": it_bytes_measured,
                 "frac_of_hbm_peak_measured": None if it_bytes_measured is None else it_bytes_measured / step_s / 1e9 / (PEAK_HBM_GBS * world),
                 "what": "whole iteration (train + monitor pass) against both roofs: N G (8C + 12K + 3) flop and "
                         "N G 4 + N (8C + 6K + 2) 4 canonical bytes (SURVEY.md section 8d) over ms_per_step.  The canonical bytes are an "
                         "accounting convention (float32 counts): bytes_stored is the same formula at the width the matrix is held at, "
                         "bytes_measured_pmc the two sweeps' HBM traffic from the PMC pass of this build -- what the memory system actually moves"}),
            "kernel_ms_per_iter_warmup": {k: v[0] / max(args.warmup, 1) for k, v in kt.items()},
            "roofline_ystream": ystream,
            "final_elbo": last,
            "fit_wallclock": {"seconds": fit_s, "iterations": int(len(trace) - 1), "max_iter": 200, "rel_tol": 1e-6,
                              "final_elbo_mean": float(np.mean(finals)), "from_initial_values": True,
                              "what": "ca_reinit (initial values, fresh Adam state), ca_run + 20 final ELBOs, eps generated by the "
                                      "built-in Philox stream (inside the time)"},
        }
        out["preheat"] = {"ms": pre_ms, "iterations": pre_it, "calls": pre_calls,
                          "what": "untimed iterations between the --warmup steps and the timed regions (clock ramp; see --preheat-ms)"}
        if steady is not None:
            out["steady_state_200"] = steady
        if other is not None:
            out["other_configs"] = other
            out["fallbacks"] = fallbacks
        if busy_it:
            out["untimed_busy_tail"] = {"seconds": busy_s, "iterations": busy_it, "it_per_s": busy_it / busy_s,
                                        "what": "untimed ca_iterate calls after the measurements (see --busy-seconds); not part of value"}
        if replicas_equal is not None:
            out["replicas_bit_identical_after_timed_regions"] = replicas_equal
        if mon_us is not None:
            out["monitor_pass_us_with_collective"] = mon_us
        if ar_us is not None:
            out["allreduce_us"] = ar_us
        if api_leg is not None:
            out["through_api"] = api_leg
        if want_cpu:
            n_cpu = args.cpu_sample_cells
            if n_cpu <= 0:
                import psutil
                n_cpu = n_loc if psutil.virtual_memory().available > 10 * 8 * n_loc * G else 65536
            Yd, _aux2 = synth.make_problem_torch(N, G, C, seed=args.seed, rows=(lo, hi), device=f"cuda:{local_rank}")   # (the same counts as above: same seed)
            Ysample = Yd[:min(n_cpu, n_loc)].cpu().numpy().astype(np.float64)
            del Yd, _aux2
            torch.cuda.empty_cache()
            try:
                out["parity_check"] = parity_check(eng, Ysample, aux["L"], psi0[:Ysample.shape[0]], loc0, K) if Ysample.shape[0] == n_loc else {
                    "skipped": "the host has no room for the full matrix in float64; tests/test_gpu_scale.py holds the at-size check"}
            except Exception as ex:  # noqa: BLE001
                out["parity_check"] = {"error": str(ex)[:300]}
            out["cpu_baseline"] = cpu_baseline(Ysample, aux["L"], psi0, loc0, K, N)
            try:
                out["cpu_ref_dataflow"] = cpu_ref_dataflow()
            except Exception as ex:  # noqa: BLE001
                out["cpu_ref_dataflow"] = {"error": str(ex)}
        print(json.dumps(out), flush=True)
    eng.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    try:
        main()
    except SystemExit:
        raise
    except BaseException as ex:   # a launcher prints only its own traceback: say which rank failed and why, then fail the same way
        print(f"[rank {os.environ.get('RANK', '0')}] bench.py: {type(ex).__name__}: {ex}", file=sys.stderr, flush=True)
        raise
