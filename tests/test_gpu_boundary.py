"""The drop-in boundary as the R caller uses it: column-major matrices, the `.Call` shim, the per-iteration
interrupt hook and the row / column selection at upload.

R hands `Y_dat`, `L_dat`, `pcs`, `x` column-major (R/inference-tflow.R:190-191,355) and expects column-major results; the
shim (clonealign_amd/r_shim/src/clonealign_hip_shim.c) therefore sets CA_COL_MAJOR for every matrix.  Everything here is
compared with the row-major engine bit for bit: the layout only changes how the host buffers are indexed."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from tests import _golden
from tests._cases import eps_for, make_case

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

LAYOUT_CASES = {
    "k1": dict(N=300, G=130, C=3, K=1),
    "k2p1s2x": dict(N=200, G=90, C=4, K=2, P=1, S=2, extra=True),
    "k0p1": dict(N=100, G=40, C=3, K=0, P=1),
    "c11": dict(N=150, G=64, C=11, K=1),
    "fused": dict(N=2100, G=700, C=5, K=1),          # matrix-core sweeps, u8 storage
}


def _pair(case, ydt, **kw):
    from clonealign_amd.engine import HipEngine
    c = dict(case)
    c["Y"] = case["Y"].astype(ydt)
    return HipEngine(**c, layout="row", **kw), HipEngine(**c, layout="col", **kw)


@pytest.mark.parametrize("ydt", [np.float64, np.int32])
@pytest.mark.parametrize("name", list(LAYOUT_CASES))
def test_col_major_boundary_is_bitwise_the_row_major_engine(name, ydt):
    """Fortran-ordered Y (float64 AND int32, the two types the shim passes), L, psi0, X, extra_loglik in; every
    ca_get_param / ca_get_gradient / ca_set_param / ca_reinit matrix out and in -- equal to the row-major run."""
    case = make_case(seed=31, **LAYOUT_CASES[name])
    a, b = _pair(case, ydt)
    try:
        assert (a.layout, b.layout) == (0, 1)
        G, S = a.G, a.S
        e = [eps_for(S, G, 50 + i) for i in range(6)]
        for eng in (a, b):
            eng.gamma_init(e[0])
        assert a.elbo(e[1]) == b.elbo(e[1])
        for eng in (a, b):
            eng.step(e[2])
        ga, ea = a.gradients(e[3])
        gb, eb = b.gradients(e[3])
        assert ea == eb
        for n in a.VAR_NAMES:
            assert np.array_equal(ga[n], gb[n]), n
        pa, pb = a.get_params(), b.get_params()
        assert list(pa) == list(pb)
        for n in pa:
            assert pb[n].flags["F_CONTIGUOUS"] or pb[n].ndim == 1 or min(pb[n].shape) <= 1
            assert np.array_equal(pa[n], pb[n]), n
        sa, sb = a.get_state(), b.get_state()
        for n in sa:
            assert np.array_equal(sa[n], sb[n]), n
        # whole loop + final ELBOs
        eps = np.stack([eps_for(S, G, 200 + i) for i in range(2 + 2 * 6 + 4)])
        ta, tb = a.run(eps, 6, 1e-12), b.run(eps, 6, 1e-12)
        assert np.array_equal(ta, tb)
        assert np.array_equal(a.final_elbo(eps[14:], 4), b.final_elbo(eps[14:], 4))
        # ca_set_param with matrices, then ca_reinit
        rng = np.random.default_rng(5)
        for n in ("psi", "W", "gamma_logits", "beta"):
            v = rng.normal(size=a._shape(n)) * 0.1
            a.set(n, v); b.set(n, v)
        assert a.elbo(e[4]) == b.elbo(e[4])
        psi1 = rng.normal(size=(a.N, a.K))
        a.reinit(psi1); b.reinit(psi1)
        assert np.array_equal(a.get("psi"), b.get("psi")) and np.array_equal(a.get("psi"), psi1.astype(np.float32))
        for eng in (a, b):
            eng.gamma_init(e[0])
        assert a.elbo(e[5]) == b.elbo(e[5])
    finally:
        a.close(); b.close()


@pytest.mark.parametrize("layout", ["row", "col"])
def test_host_ingestion_of_every_dtype_and_layout_is_bitwise_the_device_pointer_upload(layout):
    """Round 5: host matrices go up through pinned double buffers in chunks, float64 narrowed to float32 by the host
    threads on the way (clonealign_hip.hip ingest_host_matrix) instead of one pageable copy into an N*G*8-byte staging buffer.  For all five
    ca_dtypes and both layouts -- on a matrix that spans several 16 MiB chunks, ragged at the end, with counts above 255 (overflow list)
    -- the engine must be bit for bit the one built from a DEVICE pointer to the same counts (the path that never touches host bytes)."""
    import torch
    from clonealign_amd.engine import HipEngine
    N, G = 9100, 1031                                    # 9.4 M counts: 5 chunks as float32, 3 as int32, ragged tails
    case = make_case(seed=91, N=N, G=G, C=5, K=1)
    rng = np.random.default_rng(4)
    Y = np.minimum(case["Y"], 250).astype(np.int64)
    idx = rng.integers(0, Y.size, size=300)
    Y.reshape(-1)[idx] += rng.integers(200, 3000, size=idx.size)    # entries above 255: u8 storage with the overflow list
    e = [eps_for(1, G, 70 + i) for i in range(4)]

    def fit(eng):
        try:
            eng.gamma_init(e[0])
            a = eng.elbo(e[1])
            eng.step(e[2])
            return a, eng.elbo(e[3]), eng.get_state(), eng.info()["y_storage_name"]
        finally:
            eng.close()

    order = "F" if layout == "col" else "C"
    kw = dict(L=case["L"], psi0=case["psi0"], loc0=case["loc0"], K=1, layout=layout)
    for dt, cap in ((np.float64, None), (np.float32, None), (np.int32, None), (np.uint16, None), (np.uint8, 255)):
        Yt = (np.minimum(Y, cap) if cap else Y).astype(dt, order=order)
        Yd = torch.from_numpy(np.ascontiguousarray(Yt.T if layout == "col" else Yt)).cuda()     # the same bytes on the device, same layout
        ref = fit(HipEngine(None, y_device_ptr=Yd.data_ptr(), y_device_dtype=dt, shape=(N, G), **kw))
        got = fit(HipEngine(Yt, **kw))
        assert got[0] == ref[0] and got[1] == ref[1] and got[3] == ref[3] == "u8", (dt, got[:2], ref[:2], got[3])
        for n in ref[2]:
            assert np.array_equal(got[2][n], ref[2][n]), (dt, n)
        del Yd


def test_host_ingestion_reports_the_errors_of_the_device_scan():
    """What the narrowing host pass must not change: NaN / negative counts and counts no storage type can hold are refused with the
    messages the device scan gives -- also when the offending double sits in the LAST ragged chunk; a double that is not exactly a float
    (0.1) is 'not exactly representable'; NaN wins over it; and with a selection (ca_problem.cell_index) a bad value OUTSIDE the
    selected cells is nobody's business."""
    from clonealign_amd.engine import EngineError, HipEngine
    N, G = 5000, 900
    case = make_case(seed=92, N=N, G=G, C=3, K=1)
    kw = dict(L=case["L"], psi0=case["psi0"], loc0=case["loc0"], K=1)
    Y = case["Y"].astype(np.float64)
    for val, msg in ((np.nan, "negative or NaN"), (-1.0, "negative or NaN"), (0.1, "not exactly representable"), (2.0 ** 40 + 1.0, "not exactly representable")):
        Yb = Y.copy()
        Yb[N - 1, G - 1] = val
        if val == 0.1:
            Yb[3, 3] = np.nan; msg = "negative or NaN"          # NaN takes precedence over inexactness
        with pytest.raises(EngineError) as ex:
            HipEngine(Yb, **kw)
        assert ex.value.code == 1 and msg in ex.value.msg, (val, ex.value.msg)
    Yb = Y.copy()
    Yb[N - 1, 5] = 0.1
    with pytest.raises(EngineError) as ex:
        HipEngine(Yb, **kw)
    assert "not exactly representable" in ex.value.msg
    # ... but outside the selection it does not matter: same engine as on the clean matrix cut the same way
    keep = np.arange(N - 1)
    kws = dict(L=case["L"], psi0=case["psi0"][:-1], loc0=case["loc0"], K=1, cell_index=keep)
    a, b = HipEngine(Yb, **kws), HipEngine(Y, **kws)
    try:
        e0 = eps_for(1, G, 5)
        a.gamma_init(e0); b.gamma_init(e0)
        assert a.elbo(e0) == b.elbo(e0)
    finally:
        a.close(); b.close()


def test_col_major_device_helpers_match_row_major():
    """ca_init_psi_pca (noise in, pcs out), ca_clone_gene_sums (T out), ca_preprocess, ca_allele_loglik in both layouts."""
    from clonealign_amd.engine import allele_loglik, preprocess_masks
    case = make_case(seed=8, N=900, G=260, C=4, K=2)
    case["Y"] = case["Y"] + np.random.default_rng(1).poisson(0.3, size=case["Y"].shape)   # no constant genes
    a, b = _pair(case, np.int32)
    try:
        noise = np.random.default_rng(2).normal(0, 0.05, size=(a.N, a.K))
        pa, pb = a.pca_init(noise, n_iter=30, seed=3), b.pca_init(noise, n_iter=30, seed=3)
        assert np.array_equal(pa, pb) and np.array_equal(a.get("psi"), b.get("psi"))
        ci = np.random.default_rng(3).integers(-1, a.C, size=a.N)
        (Ta, Sa), (Tb, Sb) = a.clone_gene_sums(ci), b.clone_gene_sums(ci)
        assert np.array_equal(Ta, Tb) and np.array_equal(Sa, Sb) and Tb.flags["F_CONTIGUOUS"]
    finally:
        a.close(); b.close()
    Y, L = case["Y"].astype(np.int32), case["L"].copy()
    L[::7] = 2.0                                       # same copy number everywhere
    L[3, 1] = 9.0                                      # above max_copy_number
    for ydt in (np.int32, np.float64):
        r = preprocess_masks(Y.astype(ydt), L, min_counts_per_gene=600, min_counts_per_cell=560, layout="row")
        c = preprocess_masks(Y.astype(ydt), L, min_counts_per_gene=600, min_counts_per_cell=560, layout="col")
        for x, y in zip(r, c):
            assert np.array_equal(x, y)
        assert 0 < r[0].sum() < r[0].size and 0 < r[1].sum() < r[1].size
    rng = np.random.default_rng(4)
    cov = rng.poisson(6, size=(70, 33)).astype(np.float64)
    ref = rng.binomial(cov.astype(int), 0.4).astype(np.float64)
    ca_ = rng.integers(1, 4, size=(33, 5)).astype(np.float64)
    assert np.array_equal(allele_loglik(ca_, cov, ref, layout="row"), allele_loglik(ca_, cov, ref, layout="col"))


@pytest.mark.parametrize("layout", ["row", "col"])
@pytest.mark.parametrize("ydt", [np.int32, np.float64, np.uint8])
def test_selection_lists_at_upload_equal_the_host_cut(layout, ydt):
    """ca_problem.cell_index / gene_index: the raw matrix goes up once and is cut on the device -- same fit as cutting
    Y[cells][:, genes] on the host (what R/preprocess.R:141-147 and R/inference-tflow.R:117-124 do with copies)."""
    from clonealign_amd.engine import EngineError, HipEngine
    rng = np.random.default_rng(12)
    raw = make_case(seed=4, N=700, G=420, C=4, K=1)
    Yraw = np.minimum(raw["Y"], 250).astype(ydt)
    cells = np.sort(rng.choice(700, size=523, replace=False))
    genes = np.sort(rng.choice(420, size=301, replace=False))
    sub = dict(L=raw["L"][genes], psi0=raw["psi0"][cells], loc0=raw["loc0"][genes], K=1, S=1)
    a = HipEngine(Y=np.ascontiguousarray(Yraw[cells][:, genes]), layout=layout, **sub)
    b = HipEngine(Y=Yraw, cell_index=cells, gene_index=genes, layout=layout, **sub)
    c = HipEngine(Y=np.ascontiguousarray(Yraw[:, genes]), cell_index=cells, layout=layout, **sub)
    try:
        assert (b.N, b.G) == (523, 301) == (a.N, a.G) == (c.N, c.G)
        eps = np.stack([eps_for(1, 301, 900 + i) for i in range(2 + 2 * 5)])
        ta = a.run(eps, 5, 1e-12)
        assert np.array_equal(ta, b.run(eps, 5, 1e-12)) and np.array_equal(ta, c.run(eps, 5, 1e-12))
        for n in ("mu", "clone_probs", "s", "psi", "W"):
            assert np.array_equal(a.get(n), b.get(n)), n
        assert a.info()["y_storage_name"] == b.info()["y_storage_name"]
    finally:
        a.close(); b.close(); c.close()
    with pytest.raises(EngineError, match="strictly increasing"):
        HipEngine(Y=Yraw, cell_index=cells[::-1].copy(), gene_index=genes, **sub)
    with pytest.raises(EngineError, match="G_src"):
        HipEngine(Y=Yraw, cell_index=cells, gene_index=np.append(genes[:-1], 420), **sub)


def test_poll_hook_cancels_between_iterations():
    """ca_run_ex: the reference's loop can be interrupted every iteration (R/inference-tflow.R:394-417).  Cancelling at
    iteration k returns k + 1 trace values -- the prefix of the uninterrupted trace -- and the variables after iteration k."""
    from clonealign_amd.engine import HipEngine
    case = make_case(seed=21, N=600, G=300, C=4, K=1)
    eps = np.stack([eps_for(1, 300, 400 + i) for i in range(2 + 2 * 12)])
    full, cut, ref = HipEngine(**case), HipEngine(**case), HipEngine(**case)
    try:
        seen = []
        t_full = full.run(eps, 12, 1e-12, poll=lambda i, v: seen.append((i, v)) and False)
        assert [i for i, _ in seen] == list(range(13)) and np.array_equal([v for _, v in seen], t_full)
        assert not full.interrupted
        k = 5
        t_cut = cut.run(eps, 12, 1e-12, poll=lambda i, v: i >= k)
        assert cut.interrupted and len(t_cut) == k + 1 and np.array_equal(t_cut, t_full[:k + 1])
        t_ref = ref.run(eps, k, 1e-12)                      # a plain run of exactly k iterations
        # (its LAST monitor pass has no train pass to share a sweep with and takes the plain fp32 kernel instead of the
        #  fused matrix-core one: same variables, ELBO equal to fp32 rounding)
        assert np.array_equal(t_ref[:-1], t_cut[:-1]) and abs(t_ref[-1] - t_cut[-1]) <= 2e-6 * abs(t_ref[-1])
        for n in ref.VAR_NAMES:
            assert np.array_equal(ref.get(n), cut.get(n)), n
        # cancelling at the initial ELBO; an exception in the callback stops the loop and is re-raised
        assert len(cut.run(eps, 12, 1e-12, poll=lambda i, v: True)) == 1
        with pytest.raises(KeyError):
            cut.run(eps, 12, 1e-12, poll=lambda i, v: {}["boom"])
        # the engine stays usable
        assert np.isfinite(cut.elbo(eps[0]))
    finally:
        full.close(); cut.close(); ref.close()


# ------------------------------------------------------------------------------------------------ the R shim itself
def _harness():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "r_stub")], stdout=subprocess.DEVNULL)
    lib = C.CDLL(os.path.join(ROOT, "tests", "r_stub", "libshim_harness.so"))
    lib.harness_fit.restype = C.c_int
    return lib


def _call_shim(lib, Y, L, psi0, loc0, K, S, max_iter, rel_tol, lr, eps, X=None, extra=None, interrupt_after=0, psi_noise=None):
    """psi0 None with K > 0: the shim initialises psi on the device (prcomp + scale) plus psi_noise (or nothing)."""
    N, G = Y.shape
    Cn = L.shape[1]
    P = 0 if X is None else X.shape[1]
    f = lambda a: None if a is None else np.asfortranarray(np.asarray(a, dtype=np.float64))   # noqa: E731  R matrices
    ptr = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)                       # noqa: E731
    Yd = None if Y.dtype == np.int32 else f(Y)
    Yi = np.asfortranarray(Y) if Y.dtype == np.int32 else None
    Lf, p0, Xf, exf, pn = f(L), f(psi0), f(X), f(extra), f(psi_noise)
    l0 = None if loc0 is None else np.ascontiguousarray(loc0, dtype=np.float64)
    ev = None if eps is None else np.ascontiguousarray(eps, dtype=np.float64).reshape(-1)
    out = dict(elbo=np.zeros(max_iter + 1), finals=np.zeros(20), mu=np.zeros(G), clone_probs=np.zeros((N, Cn), order="F"),
               s=np.zeros(N), alpha=np.zeros(Cn), psi=np.zeros((N, K), order="F"), W=np.zeros((G, K), order="F"),
               chi=np.zeros(K), beta=np.zeros((G, P), order="F"))
    n_elbo = C.c_long()
    err = C.create_string_buffer(1024)
    rc = lib.harness_fit(ptr(Yd), ptr(Yi), C.c_int(N), C.c_int(G), ptr(Lf), C.c_int(Cn), ptr(p0), ptr(pn), ptr(l0), ptr(Xf), C.c_int(P),
                         ptr(exf), C.c_int(K), C.c_int(S), C.c_int(max_iter), C.c_double(rel_tol), C.c_double(lr), ptr(ev),
                         C.c_long(0 if ev is None else ev.size), C.c_int(interrupt_after), ptr(out["elbo"]), C.byref(n_elbo),
                         ptr(out["finals"]), ptr(out["mu"]), ptr(out["clone_probs"]), ptr(out["s"]), ptr(out["alpha"]),
                         ptr(out["psi"]), ptr(out["W"]), ptr(out["chi"]), ptr(out["beta"]), err)
    out["elbo"] = out["elbo"][:max(n_elbo.value, 0)]
    return rc, err.value.decode(), out


@pytest.mark.parametrize("ydt", [np.float64, np.int32])
def test_r_shim_call_entry_point_on_example_sce(ydt):
    """C_clonealign_fit, compiled from the shim's own source against the stand-in R API, called from C with column-major
    example_sce inputs (config 1): same trace, final ELBOs and ml_params as the Python mirror of the same boundary."""
    from clonealign_amd import hostprep
    from clonealign_amd.engine import HipEngine
    lib = _harness()
    Y, L, *_ = _golden.example()
    g = _golden.load("cfg1")
    psi0, loc0 = g["psi0"], g["loc0"]
    max_iter = 25
    eps = g["eps"][:2 + 2 * max_iter + 20].astype(np.float32)
    rc, msg, out = _call_shim(lib, Y.astype(ydt), hostprep.saturate(L, 6), psi0, loc0, 1, 1, max_iter, 1e-12, 0.1, eps)
    assert rc == 0, msg
    eng = HipEngine(Y.astype(ydt), hostprep.saturate(L, 6), psi0, loc0, 1, 1, layout="col")
    try:
        t = eng.run(eps, max_iter, 1e-12)
        fin = eng.final_elbo(eps[2 + 2 * max_iter:], 20)
        assert np.array_equal(out["elbo"], t) and np.array_equal(out["finals"], fin)
        p = eng.get_params()
        for n in ("mu", "clone_probs", "s", "alpha", "psi", "W", "chi"):
            assert np.array_equal(out[n], p[n]), n
    finally:
        eng.close()
    # the oracle's golden trace for the same eps stream (200 iterations recorded; the first 25 here)
    np.testing.assert_allclose(out["elbo"], g["elbo_trace"][:max_iter + 1], rtol=1e-5)


def test_r_shim_single_fit_with_the_initial_values_made_on_the_device():
    """the drop-in inference_tflow() always called host prcomp() -- minutes at 100k x 5k in front of a 60 ms fit.
    C_clonealign_fit(psi0 = NULL, psi_noise, loc0 = NULL) makes both initial values on the device from the resident matrix
    (ca_init_psi_pca; ca_problem.loc0 = NULL).  On example_sce, against the fit started from the host's prcomp + scale + the same noise
    and the host's mu guess (hostprep = R/inference-tflow.R:204-235): the device PCs equal prcomp's to 2e-4 up to sign (asserted in
    tests/test_gpu_scale.py), so from the same eps the two fits give the same clone labels and the same ml_params to 1e-3 -- after the
    device components are given the host's sign, which prcomp itself does not define."""
    from clonealign_amd import hostprep
    from clonealign_amd.api import clone_assignment
    lib = _harness()
    Y, L, *_ = _golden.example()
    Ls = hostprep.saturate(L, 6)
    N, G = Y.shape
    max_iter = 40
    rng = np.random.default_rng(12)
    noise = rng.normal(0, 0.05, size=(N, 1))
    eps = rng.normal(size=(2 + 2 * max_iter + 20, G)).astype(np.float32)
    pcs = hostprep.pca_init(Y, 1, None)
    loc0 = hostprep.safe_inverse_softplus(hostprep.mu_guess(Y, True))
    rc, msg, dev = _call_shim(lib, Y, Ls, None, None, 1, 1, max_iter, 1e-12, 0.1, eps, psi_noise=noise)
    assert rc == 0, msg
    # the device component's sign (largest loading positive) may be the opposite of LAPACK's: start the host fit from the same one
    rc0, msg0, probe = _call_shim(lib, Y, Ls, None, None, 1, 1, 0, 1e-12, 0.1, eps[:22], psi_noise=np.zeros((N, 1)))
    assert rc0 == 0, msg0
    sign = 1.0 if np.abs(probe["psi"][:, 0] - pcs[:, 0]).max() < np.abs(probe["psi"][:, 0] + pcs[:, 0]).max() else -1.0
    assert np.abs(probe["psi"][:, 0] - sign * pcs[:, 0]).max() < 2e-4
    rc, msg, host = _call_shim(lib, Y, Ls, sign * pcs + noise, loc0, 1, 1, max_iter, 1e-12, 0.1, eps)
    assert rc == 0, msg
    assert len(dev["elbo"]) == len(host["elbo"]) == max_iter + 1
    np.testing.assert_allclose(dev["elbo"], host["elbo"], rtol=1e-4)
    la, lb = clone_assignment(dev["clone_probs"], list("ABC")), clone_assignment(host["clone_probs"], list("ABC"))
    assert list(la) == list(lb)
    for n in ("mu", "clone_probs", "alpha", "psi", "W", "chi", "s"):
        assert np.abs(dev[n] - host[n]).max() <= 1e-3 * max(np.abs(host[n]).max(), 1e-30), n


def test_r_shim_interrupt_and_error_paths_free_the_engine_first():
    lib = _harness()
    case = make_case(seed=2, N=120, G=50, C=3, K=1)
    eps = np.stack([eps_for(1, 50, i) for i in range(2 + 2 * 30 + 20)])
    rc, msg, _ = _call_shim(lib, case["Y"], case["L"], case["psi0"], case["loc0"], 1, 1, 30, 1e-12, 0.1, eps, interrupt_after=4)
    assert rc == 1 and "interrupted" in msg
    # a cell without counts: "Initial elbo is NA" is not it -- the library rejects nothing here, R does (:212-214); but a
    # negative count is refused by ca_create and surfaces as an R error with the library's message
    Ybad = case["Y"].copy(); Ybad[3, 4] = -1
    rc, msg, _ = _call_shim(lib, Ybad, case["L"], case["psi0"], case["loc0"], 1, 1, 5, 1e-12, 0.1, None)
    assert rc == 1 and "negative" in msg
    # built-in eps stream (eps = NULL) works through the shim as well
    rc, msg, out = _call_shim(lib, case["Y"], case["L"], case["psi0"], None, 1, 1, 5, 1e-12, 0.1, None)
    assert rc == 0 and len(out["elbo"]) == 6 and np.all(np.isfinite(out["elbo"]))


def _call_multifit(lib, Y, L, psi, by_noise, loc0, K, max_iter, eps, devices, want_sums, interrupt_after=0):
    """psi: [R, N, K] (psi0 per restart, or the PCA noise per restart); eps: [R, draws, S*G] or None."""
    N, G = Y.shape
    Cn, R = L.shape[1], psi.shape[0]
    ptr = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)                       # noqa: E731
    Yd = None if Y.dtype == np.int32 else np.asfortranarray(Y, dtype=np.float64)
    Yi = np.asfortranarray(Y) if Y.dtype == np.int32 else None
    Lf = np.asfortranarray(L, dtype=np.float64)
    pf = np.ascontiguousarray(np.stack([np.asfortranarray(psi[r]).reshape(-1, order="F") for r in range(R)]))   # R column-major matrices
    l0 = None if loc0 is None else np.ascontiguousarray(loc0, dtype=np.float64)
    ev = None if eps is None else np.ascontiguousarray(eps, dtype=np.float64).reshape(R, -1)
    dv = np.ascontiguousarray(devices, dtype=np.int32)
    o = dict(elbo=np.zeros((R, max_iter + 1)), finals=np.zeros((R, 20)), mu=np.zeros((R, G)), clone_probs=np.zeros((R, N * Cn)),
             alpha=np.zeros((R, Cn)), psi=np.zeros((R, N * K)), W=np.zeros((R, G * K)), T=np.zeros((R, G * Cn)), Syy=np.zeros((R, G)))
    n_elbo = (C.c_long * R)()
    err = C.create_string_buffer(1024)
    lib.harness_multifit.restype = C.c_int
    rc = lib.harness_multifit(ptr(Yd), ptr(Yi), C.c_int(N), C.c_int(G), ptr(Lf), C.c_int(Cn), ptr(pf), C.c_int(int(by_noise)), ptr(l0),
                              C.c_int(K), C.c_int(1), C.c_int(max_iter), C.c_double(1e-12), C.c_double(0.1), ptr(ev),
                              C.c_long(0 if ev is None else ev.shape[1]), C.c_int(R), ptr(dv), C.c_int(len(dv)), C.c_int(int(want_sums)),
                              C.c_int(interrupt_after), ptr(o["elbo"]), n_elbo, ptr(o["finals"]), ptr(o["mu"]), ptr(o["clone_probs"]),
                              ptr(o["alpha"]), ptr(o["psi"]), ptr(o["W"]), ptr(o["T"]), ptr(o["Syy"]), err)
    o["n_elbo"] = [int(v) for v in n_elbo]
    o["clone_probs"] = o["clone_probs"].reshape(R, Cn, N).transpose(0, 2, 1)     # column-major N x C per restart
    o["psi"] = o["psi"].reshape(R, K, N).transpose(0, 2, 1)
    o["W"] = o["W"].reshape(R, K, G).transpose(0, 2, 1)
    o["T"] = o["T"].reshape(R, Cn, G).transpose(0, 2, 1)
    return rc, err.value.decode(), o


@pytest.mark.parametrize("ydt", [np.float64, np.int32])
def test_r_shim_multifit_runs_the_restart_loop_on_resident_engines(ydt):
    """C_clonealign_multifit (run_clonealign's restart loop, R/clonealign.R:50-56, as ONE .Call): five restarts dealt over two
    worker threads (both on device 0 here; one per device on a node), the first restart of a worker uploads, the others are
    ca_reinit()s.  Every restart must equal the Python mirror's separate fit with the same psi0 and eps bit for bit, and the
    correlation sums (compute_correlations, :318-334) the mirror's device pass."""
    from clonealign_amd.engine import HipEngine
    lib = _harness()
    case = make_case(seed=5, N=900, G=260, C=4, K=1)
    N, G, R, max_iter = 900, 260, 5, 8
    rng = np.random.default_rng(1)
    psi = np.stack([case["psi0"] + rng.normal(0, 0.05, size=(N, 1)) for _ in range(R)])
    eps = np.stack([np.stack([eps_for(1, G, 1000 * r + i) for i in range(2 + 2 * max_iter + 20)]) for r in range(R)])
    Y = case["Y"].astype(ydt)
    rc, msg, out = _call_multifit(lib, Y, case["L"], psi, False, case["loc0"], 1, max_iter, eps, [0, 0], True)
    assert rc == 0, msg
    for r in range(R):
        eng = HipEngine(Y, case["L"], psi[r], case["loc0"], 1, 1, layout="col")
        try:
            t = eng.run(eps[r].astype(np.float32), max_iter, 1e-12)
            fin = eng.final_elbo(eps[r, 2 + 2 * max_iter:].astype(np.float32), 20)
            assert out["n_elbo"][r] == max_iter + 1 and np.array_equal(out["elbo"][r], t), r
            assert np.array_equal(out["finals"][r], fin), r
            p = eng.get_params()
            for n in ("mu", "clone_probs", "alpha", "psi", "W"):
                assert np.array_equal(out[n][r], np.asarray(p[n]).reshape(out[n][r].shape)), (r, n)
            cp = np.asarray(p["clone_probs"])
            call = np.where(cp.max(1) >= 0.95, cp.argmax(1), -1)
            T, Syy = eng.clone_gene_sums(call)
            assert np.array_equal(out["T"][r], T) and np.array_equal(out["Syy"][r], Syy), r
        finally:
            eng.close()
    assert len({tuple(out["elbo"][r]) for r in range(R)}) == R          # five different restarts
    # psi initialised on the device per restart (prcomp + scale of :204-208 by subspace iteration, plus the restart's noise)
    noise = rng.normal(0, 0.05, size=(2, N, 1))
    rc, msg, o2 = _call_multifit(lib, Y, case["L"], noise, True, None, 1, 4, None, [0], False)
    assert rc == 0, msg
    assert np.all(np.isfinite(o2["elbo"][:, :5])) and not np.array_equal(o2["psi"][0], o2["psi"][1])
    # Ctrl-C while the workers run: they stop between iterations, the engines are freed, then R gets its error
    rc, msg, _ = _call_multifit(lib, Y, case["L"], psi, False, case["loc0"], 1, 4000, None, [0, 0], False, interrupt_after=2)
    assert rc == 1 and "interrupted" in msg


def test_r_shim_preprocess_and_allele_entry_points_match_the_python_mirror():
    """C_clonealign_preprocess (R/preprocess.R:93-147) and C_clonealign_allele_loglik (R/allele-specific.R:17-58) through the
    stand-in R API, column-major, integer and double count matrices: identical to the ctypes mirror of the same C ABI."""
    from clonealign_amd import engine as E
    from clonealign_amd import preprocess as pre
    lib = _harness()
    rng = np.random.default_rng(9)
    N, G, Cn = 700, 300, 4
    L = rng.integers(1, 8, size=(G, Cn)).astype(np.float64)
    Y = rng.poisson(rng.lognormal(0.0, 1.5, G)[None, :] * 0.6, size=(N, G)).astype(np.int32)
    Y[:, :5] = 0
    Y[:7] = 0
    ptr = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)                       # noqa: E731
    for ydt in (np.int32, np.float64):
        Yc = Y.astype(ydt)
        kg, kc = np.zeros(G, dtype=np.int32), np.zeros(N, dtype=np.int32)
        gs, cs = np.zeros(G), np.zeros(N)
        err = C.create_string_buffer(1024)
        Yf = np.asfortranarray(Yc)
        lib.harness_preprocess.restype = C.c_int
        rc = lib.harness_preprocess(ptr(Yf) if ydt == np.float64 else None, ptr(Yf) if ydt == np.int32 else None, C.c_int(N), C.c_int(G),
                                    ptr(np.asfortranarray(L)), C.c_int(Cn), C.c_double(20.0), C.c_double(100.0), C.c_int(1), C.c_double(10.0),
                                    C.c_double(6.0), C.c_int(1), ptr(kg), ptr(kc), ptr(gs), ptr(cs), err)
        assert rc == 0, err.value
        mg, mc, mgs, mcs = E.preprocess_masks(Yc, L, min_counts_per_gene=20, min_counts_per_cell=100, remove_outlying_genes=True,
                                              nmads=10, max_copy_number=6, remove_genes_same_copy_number=True, layout="col")
        assert np.array_equal(kg.astype(bool), mg) and np.array_equal(kc.astype(bool), mc)
        assert np.array_equal(gs, mgs) and np.array_equal(cs, mcs)
        assert 0 < kg.sum() < G and 0 < kc.sum() < N
    V = 40
    ca_ = rng.integers(1, 4, size=(V, Cn)).astype(np.float64)
    cov = rng.integers(0, 30, size=(N, V)).astype(np.float64)
    ref = np.floor(cov * rng.random((N, V)))
    out = np.zeros((N, Cn), order="F")
    err = C.create_string_buffer(1024)
    lib.harness_allele.restype = C.c_int
    rc = lib.harness_allele(ptr(np.asfortranarray(ca_)), C.c_int(V), C.c_int(Cn), ptr(np.asfortranarray(cov)), ptr(np.asfortranarray(ref)),
                            C.c_int(N), ptr(out), err)
    assert rc == 0, err.value
    assert np.array_equal(out, E.allele_loglik(ca_, cov, ref, layout="col"))
