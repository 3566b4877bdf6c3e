"""ctypes binding of ``libclonealign_hip.so`` (C ABI: ``include/clonealign_hip.h``).

This is the product path: there is NO CPU fallback.  Importing works anywhere (so the
symbol table can be checked without a GPU); constructing a :class:`HipEngine` fails loudly
when the library is missing or no MI355X is visible.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libclonealign_hip.so")

CA_OK = 0
CA_ERR_NAN = 4
CA_INTERRUPTED = 7
CA_ABI_VERSION = 6
P2P_HANDLE_BYTES = 128
CA_F64, CA_F32, CA_I32, CA_U16, CA_U8 = 0, 1, 2, 3, 4
CA_ROW_MAJOR, CA_COL_MAJOR = 0, 1
YSTORE = {"auto": 0, "f32": 1, "u16": 2, "u8": 3}
YSTORE_NAME = {v: k for k, v in YSTORE.items()}
KERNEL_NAMES = ("fwd", "bwd", "ypass", "cell", "other")
# ca_variant bits (ca_options.variant_off: a set bit switches the variant OFF) and ca_tune_id slots
VARIANTS = {"fused": 1 << 0, "fwd_mfma": 1 << 1, "fwd_cell": 1 << 2, "bwd_mfma": 1 << 3, "tail_fuse": 1 << 4, "async_y": 1 << 5,
            "pre": 1 << 6, "pair_elbo": 1 << 7, "prep_fast": 1 << 8, "update_merge": 1 << 9, "p2p": 1 << 10, "fold_gsum": 1 << 11, "y_ride": 1 << 12, "ride_seq": 1 << 13, "y_mfma1": 1 << 14, "yfin_ride": 1 << 15, "p2p_ride": 1 << 16, "run_gate": 1 << 17, "s2_fuse": 1 << 18, "fwd_bal": 1 << 19, "bwd_tl3": 1 << 20, "series": 1 << 21}
VARIANTS_ON = {"y_mfma2": 1 << 0, "async_small": 1 << 1, "y_mfma1": 1 << 2, "fold_always": 1 << 3, "ride_seq": 1 << 4, "p2p_same_device": 1 << 5, "run_fwd": 1 << 6, "bal_tiles": 1 << 7, "series": 1 << 8}      # ca_variant_on: opt-in variants
OPT_VERBOSE = 0x80000000
TUNE = {"gsplit": 0, "fsplit": 1, "fc_tl": 2, "fc_nbig": 3, "csplit": 4, "csplit_m": 5, "tr": 6, "rg": 7}
TRANSPORT_NAME = {0: "none", 1: "rccl", 2: "host", 3: "p2p"}

EXPORTS = (
    "ca_abi_version", "ca_build_id", "ca_device_count", "ca_default_options", "ca_create", "ca_destroy", "ca_last_error", "ca_get_info",
    "ca_synchronize", "ca_stream_busy", "ca_comm_unique_id", "ca_comm_init", "ca_p2p_export", "ca_p2p_connect", "ca_p2p_commit", "ca_comm_benchmark", "ca_comm_selftest", "ca_set_host_allreduce", "ca_gamma_init", "ca_elbo", "ca_elbo_terms",
    "ca_step", "ca_gradients", "ca_run", "ca_run_ex", "ca_iterate", "ca_final_elbo", "ca_init_psi_pca", "ca_clone_gene_sums", "ca_get_param", "ca_set_param",
    "ca_get_gradient", "ca_reinit", "ca_get_kernel_times", "ca_reset_kernel_times", "ca_set_profile", "ca_eps_draw", "ca_allele_loglik", "ca_preprocess",
    # ABI 6: one fit cell-sharded over several devices of one process
    "ca_group_create", "ca_group_destroy", "ca_group_last_error", "ca_group_get_info", "ca_group_rank_handle", "ca_group_init_psi_pca", "ca_group_gamma_init",
    "ca_group_elbo", "ca_group_step", "ca_group_run_ex", "ca_group_iterate", "ca_group_final_elbo", "ca_group_get_param", "ca_group_reinit",
    "ca_group_clone_gene_sums",
)


class CaProblem(C.Structure):
    _fields_ = [("N", C.c_int64), ("G", C.c_int32), ("C", C.c_int32), ("K", C.c_int32), ("P", C.c_int32),
                ("S", C.c_int32), ("layout", C.c_int32), ("y_dtype", C.c_int32), ("y_on_device", C.c_int32),
                ("Y", C.c_void_p), ("L", C.c_void_p), ("psi0", C.c_void_p), ("loc0", C.c_void_p),
                ("X", C.c_void_p), ("extra_loglik", C.c_void_p),
                ("N_src", C.c_int64), ("G_src", C.c_int32), ("cell_index", C.c_void_p), ("gene_index", C.c_void_p),
                ("y_ld", C.c_int64)]


class CaOptions(C.Structure):
    _fields_ = [("learning_rate", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double),
                ("adam_eps", C.c_double), ("seed", C.c_uint64), ("device", C.c_int32),
                ("y_storage", C.c_int32), ("rank", C.c_int32), ("world", C.c_int32), ("profile", C.c_int32),
                ("variant_off", C.c_uint32), ("tune", C.c_int32 * 8), ("variant_on", C.c_uint32),
                ("ride_pattern", C.c_int32), ("comm_timeout_ms", C.c_int32), ("gate_timeout_us", C.c_int32), ("reserved", C.c_int32 * 2)]


class CaInfo(C.Structure):
    _fields_ = [("N", C.c_int64), ("G", C.c_int32), ("C", C.c_int32), ("K", C.c_int32), ("P", C.c_int32),
                ("S", C.c_int32), ("y_storage", C.c_int32), ("y_bytes_per_elem", C.c_int32),
                ("y_device_bytes", C.c_int64), ("device_bytes", C.c_int64), ("gsplit", C.c_int32),
                ("csplit", C.c_int32), ("n_cu", C.c_int32), ("fused_sweep", C.c_int32), ("fwd_mfma", C.c_int32),
                ("bwd_mfma", C.c_int32), ("fsplit", C.c_int32), ("fwd_cell", C.c_int32), ("y_mfma", C.c_int32),
                ("transport", C.c_int32), ("y_ride", C.c_int32), ("red_n", C.c_int64),
                ("fwd_block_cells", C.c_int32), ("fwd_blocks_big", C.c_int32), ("fold_gsum", C.c_int32), ("yfin_split", C.c_int32),
                ("update_merge", C.c_int32), ("fwd_balanced", C.c_int32), ("fwd_series", C.c_int32), ("reserved_", C.c_int32),
                ("series_passes", C.c_int64), ("series_fallbacks", C.c_int64)]


class CaGroupInfo(C.Structure):
    _fields_ = [("world", C.c_int32), ("transport", C.c_int32), ("p2p_status", C.c_int32), ("rccl_status", C.c_int32),
                ("rebuilds", C.c_int32), ("selftest_rounds", C.c_int32), ("N", C.c_int64), ("note", C.c_char * 384)]


class CaPreprocessParams(C.Structure):
    _fields_ = [("min_counts_per_gene", C.c_double), ("min_counts_per_cell", C.c_double),
                ("remove_outlying_genes", C.c_int32), ("remove_genes_same_copy_number", C.c_int32),
                ("nmads", C.c_double), ("max_copy_number", C.c_double)]


HOST_ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int64)
POLL_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.c_double)

_lib = None


def load_library(path=None):
    """dlopen the engine; raises (never falls back) when it is not built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("CLONEALIGN_HIP_LIB") or LIB_PATH     # (the override is for A/B builds of the same ABI)
    if not os.path.exists(p):
        raise RuntimeError(
            f"{p} not found: build the HIP engine first (python -c 'import __graft_entry__ as g; g.build()' "
            "or make -C clonealign_amd/csrc). clonealign_amd has no CPU fallback.")
    lib = C.CDLL(p)
    lib.ca_last_error.restype = C.c_char_p
    lib.ca_last_error.argtypes = [C.c_void_p]
    lib.ca_build_id.restype = C.c_char_p
    lib.ca_build_id.argtypes = []
    lib.ca_group_last_error.restype = C.c_char_p
    lib.ca_group_last_error.argtypes = [C.c_void_p]
    for name in EXPORTS:
        if name not in ("ca_last_error", "ca_build_id", "ca_group_last_error"):
            getattr(lib, name).restype = C.c_int
    lib.ca_device_count.argtypes = [C.POINTER(C.c_int32)]
    lib.ca_create.argtypes = [C.POINTER(CaProblem), C.POINTER(CaOptions), C.POINTER(C.c_void_p)]
    lib.ca_destroy.argtypes = [C.c_void_p]
    lib.ca_get_info.argtypes = [C.c_void_p, C.POINTER(CaInfo)]
    lib.ca_synchronize.argtypes = [C.c_void_p]
    lib.ca_stream_busy.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
    lib.ca_comm_unique_id.argtypes = [C.c_char_p]
    lib.ca_comm_init.argtypes = [C.c_void_p, C.c_char_p]
    lib.ca_p2p_export.argtypes = [C.c_void_p, C.c_char_p]
    lib.ca_p2p_connect.argtypes = [C.c_void_p, C.c_char_p]
    lib.ca_p2p_commit.argtypes = [C.c_void_p, C.c_int32]
    lib.ca_comm_benchmark.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.POINTER(C.c_double)]
    lib.ca_comm_selftest.argtypes = [C.c_void_p, C.c_int32, C.c_int64, C.POINTER(C.c_int64)]
    lib.ca_set_host_allreduce.argtypes = [C.c_void_p, HOST_ALLREDUCE_FN, C.c_void_p]
    lib.ca_gamma_init.argtypes = [C.c_void_p, C.c_void_p]
    lib.ca_elbo.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]
    lib.ca_elbo_terms.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]
    lib.ca_step.argtypes = [C.c_void_p, C.c_void_p]
    lib.ca_gradients.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]
    lib.ca_run.argtypes = [C.c_void_p, C.c_int32, C.c_double, C.c_void_p, C.c_int64, C.c_void_p,
                           C.POINTER(C.c_int32)]
    lib.ca_run_ex.argtypes = [C.c_void_p, C.c_int32, C.c_double, C.c_void_p, C.c_int64, C.c_void_p,
                              C.POINTER(C.c_int32), POLL_FN, C.c_void_p]
    lib.ca_iterate.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.POINTER(C.c_double)]
    lib.ca_final_elbo.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p,
                                  C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.ca_init_psi_pca.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_uint64, C.c_void_p]
    lib.ca_clone_gene_sums.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ca_get_param.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p]
    lib.ca_set_param.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p]
    lib.ca_get_gradient.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p]
    lib.ca_reinit.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ca_get_kernel_times.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ca_reset_kernel_times.argtypes = [C.c_void_p]
    lib.ca_set_profile.argtypes = [C.c_void_p, C.c_int32]
    lib.ca_eps_draw.argtypes = [C.c_uint64, C.c_uint64, C.c_int64, C.c_void_p]
    lib.ca_preprocess.argtypes = [C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                  C.POINTER(CaPreprocessParams), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p]
    lib.ca_allele_loglik.argtypes = [C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                     C.c_void_p, C.c_char_p]
    lib.ca_group_create.argtypes = [C.POINTER(CaProblem), C.POINTER(CaOptions), C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
    lib.ca_group_destroy.argtypes = [C.c_void_p]
    lib.ca_group_get_info.argtypes = [C.c_void_p, C.POINTER(CaGroupInfo)]
    lib.ca_group_rank_handle.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p)]
    lib.ca_group_init_psi_pca.argtypes = lib.ca_init_psi_pca.argtypes
    lib.ca_group_gamma_init.argtypes = lib.ca_gamma_init.argtypes
    lib.ca_group_elbo.argtypes = lib.ca_elbo.argtypes
    lib.ca_group_step.argtypes = lib.ca_step.argtypes
    lib.ca_group_run_ex.argtypes = lib.ca_run_ex.argtypes
    lib.ca_group_iterate.argtypes = lib.ca_iterate.argtypes
    lib.ca_group_final_elbo.argtypes = lib.ca_final_elbo.argtypes
    lib.ca_group_get_param.argtypes = lib.ca_get_param.argtypes
    lib.ca_group_reinit.argtypes = lib.ca_reinit.argtypes
    lib.ca_group_clone_gene_sums.argtypes = lib.ca_clone_gene_sums.argtypes
    # initialise this library's HIP runtime NOW: torch bundles its own, and whichever runtime is loaded first must also be
    # initialised first (loaded first but initialised second it reports "no ROCm-capable device is detected")
    lib.ca_device_count(None)
    if path is None:
        _lib = lib
    return lib


def device_count():
    n = C.c_int32()
    load_library().ca_device_count(C.byref(n))
    return n.value


def _sources():
    """The library's sources in the order csrc/Makefile hashes them: every .hip / .h / .inc / .cpp of csrc/ but ca_build_id.cpp, sorted, then the C ABI."""
    import glob
    d = os.path.join(_HERE, "csrc")
    fs = sorted(f for ext in ("*.hip", "*.h", "*.inc", "*.cpp") for f in glob.glob(os.path.join(d, ext)) if os.path.basename(f) != "ca_build_id.cpp")
    return fs + [os.path.join(os.path.dirname(_HERE), "include", "clonealign_hip.h")]


def source_build_id():
    """What ca_build_id() of a library built from the sources in this tree returns (same recipe as csrc/Makefile)."""
    import hashlib
    h = hashlib.sha1()
    for f in _sources():
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def build_id():
    return load_library().ca_build_id().decode()


class EngineError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"clonealign_hip error {code}: {msg}")
        self.code = code
        self.msg = msg


_Y_DTYPES = {np.dtype(np.float64): CA_F64, np.dtype(np.float32): CA_F32, np.dtype(np.int32): CA_I32,
             np.dtype(np.uint16): CA_U16, np.dtype(np.uint8): CA_U8}


def comm_unique_id():
    lib = load_library()
    buf = C.create_string_buffer(128)
    rc = lib.ca_comm_unique_id(buf)
    if rc != CA_OK:
        raise EngineError(rc, (lib.ca_last_error(None) or b"").decode())
    return bytes(buf.raw)


class HipEngine:
    """One fit on one MI355X: same constructor/method protocol as the oracle models."""

    VAR_NAMES = ("W", "v", "psi", "beta", "alpha_unconstr", "loc", "ls", "gamma_logits")
    DEVICE_MU_INIT = True   # loc0=None: mu_guess / loc0 of R/inference-tflow.R:220-235,262 from the resident matrix

    def __init__(self, Y, L, psi0, loc0, K, S=1, X=None, extra_loglik=None, learning_rate=0.1,
                 device=0, y_storage="auto", seed=0x5EED5EED, rank=0, world=1, profile=False,
                 y_device_ptr=None, y_device_dtype=None, shape=None, comm_id=None, host_allreduce=None, p2p_exchange=None,
                 layout="row", cell_index=None, gene_index=None, variant_off=(), variant_on=(), tune=None, verbose=False,
                 comm_timeout_ms=0, defer_transport=False, gate_timeout_us=0):
        """``layout``: "row" (C / numpy order) or "col" -- every matrix is then handed over column-major (Fortran order),
        which is what the R caller has (R/inference-tflow.R:190-191,355) and what r_shim/src/clonealign_hip_shim.c passes; the
        ``get``/``set`` matrices use the same layout.  ``cell_index`` / ``gene_index``: Y is the RAW matrix and the fit uses
        these rows / columns of it (ca_problem.cell_index / gene_index); L, psi0, loc0, X, extra_loglik are for the selection.
        ``p2p_exchange(handle: bytes) -> list[bytes]`` (world > 1): an all-gather of the ranks' P2P_HANDLE_BYTES-byte handles in
        rank order (torch.distributed / MPI); selects the one-shot peer-to-peer all-reduce (ca_p2p_export / ca_p2p_connect).
        ``comm_timeout_ms``: bound of the peer-to-peer all-reduce's device-side wait for its peers (0 = 10 s; ca_options).
        ``gate_timeout_us``: how long ``run``'s queued-ahead update waits on the device for the host's decision before it gives up and is
        queued again afterwards (0 = 1000 us; ca_options).
        ``variant_off``: names from VARIANTS (or a bitmask) to switch off; ``variant_on``: names from VARIANTS_ON to switch on;
        ``tune``: {name from TUNE: value}."""
        prob, opt = self._prepare(Y, L, psi0, loc0, K, S, X, extra_loglik, learning_rate, device, y_storage, seed, rank, world, profile,
                                  y_device_ptr, y_device_dtype, shape, layout, cell_index, gene_index, variant_off, variant_on, tune, verbose,
                                  comm_timeout_ms, gate_timeout_us)
        rc = self.lib.ca_create(C.byref(prob), C.byref(opt), C.byref(self.h))
        if rc != CA_OK:
            msg = (self.lib.ca_last_error(None) or b"").decode()
            self.h = C.c_void_p()
            raise EngineError(rc, msg)
        if world > 1:
            if host_allreduce is not None:
                # host_allreduce(np.ndarray float64 view) must sum the array in place over all ranks
                def _cb(_user, buf, n, _f=host_allreduce):
                    try:
                        _f(np.ctypeslib.as_array(buf, shape=(n,)))
                        return 0
                    except Exception:          # never unwind through C
                        import traceback
                        traceback.print_exc()
                        return 1
                self._cb = HOST_ALLREDUCE_FN(_cb)
                self._ck(self.lib.ca_set_host_allreduce(self.h, self._cb, None))
            elif p2p_exchange is not None:
                try:
                    self._p2p_setup(p2p_exchange, world)
                except BaseException:
                    self.close()          # (frees the device resources now, not whenever the half-built object is collected)
                    raise
            elif comm_id is not None:
                self._ck(self.lib.ca_comm_init(self.h, comm_id))
            elif not defer_transport:   # (defer_transport: the caller brings the transport up itself -- _p2p_setup / comm_init)
                raise ValueError("world > 1 needs p2p_exchange, comm_id (bytes from comm_unique_id(), broadcast from rank 0) "
                                 "or host_allreduce")

    # names of the C entry points this object drives (HipGroupEngine: the ca_group_* family on a group handle)
    _PREFIX = "ca_"
    _LAST_ERROR = "ca_last_error"

    def _fn(self, name):
        return getattr(self.lib, self._PREFIX + name)

    def _prepare(self, Y, L, psi0, loc0, K, S, X, extra_loglik, learning_rate, device, y_storage, seed, rank, world, profile,
                 y_device_ptr, y_device_dtype, shape, layout, cell_index, gene_index, variant_off, variant_on, tune, verbose,
                 comm_timeout_ms, gate_timeout_us):
        """The ca_problem / ca_options of the constructor's arguments (inputs kept alive on self)."""
        self.lib = load_library()
        self.h = C.c_void_p()
        if layout not in ("row", "col"):
            raise ValueError("layout must be 'row' or 'col'")
        self.layout = CA_COL_MAJOR if layout == "col" else CA_ROW_MAJOR
        self._order = "F" if layout == "col" else "C"
        mat = lambda a, shape=None: np.require(  # noqa: E731  (a contiguous float64 matrix in the problem's layout)
            np.asarray(a, dtype=np.float64) if shape is None else np.asarray(a, dtype=np.float64).reshape(shape),
            requirements=[self._order, "A"])
        if y_device_ptr is not None:
            Ns, Gs = shape
            y_dt = _Y_DTYPES[np.dtype(y_device_dtype)]
            y_ptr = C.c_void_p(int(y_device_ptr))
            self._keep = []
        else:
            Y = np.asarray(Y)
            if Y.dtype not in _Y_DTYPES:
                Y = Y.astype(np.float64)
            Y = np.require(Y, requirements=[self._order, "A"])
            Ns, Gs = Y.shape
            y_dt = _Y_DTYPES[Y.dtype]
            y_ptr = Y.ctypes.data_as(C.c_void_p)
            self._keep = [Y]
        ci = None if cell_index is None else np.ascontiguousarray(np.asarray(cell_index, dtype=np.int64))
        gi = None if gene_index is None else np.ascontiguousarray(np.asarray(gene_index, dtype=np.int32))
        N = Ns if ci is None else ci.shape[0]
        G = Gs if gi is None else gi.shape[0]
        L = mat(L)
        self.N, self.G, self.C = int(N), int(G), int(L.shape[1])
        self.K, self.S = int(K), int(S)
        loc0 = None if loc0 is None else np.ascontiguousarray(np.asarray(loc0, dtype=np.float64))
        psi0 = mat(psi0, (self.N, self.K))
        Xc = None if X is None else mat(X, (self.N, -1))
        self.P = 0 if Xc is None else Xc.shape[1]
        ex = None if extra_loglik is None else mat(extra_loglik)
        if L.shape[0] != self.G or (loc0 is not None and loc0.shape[0] != self.G):
            raise ValueError("L / loc0 do not match the number of genes")
        self._keep += [L, loc0, psi0, Xc, ex, ci, gi]
        ptr = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)  # noqa: E731
        prob = CaProblem(N=self.N, G=self.G, C=self.C, K=self.K, P=self.P, S=self.S, layout=self.layout,
                         y_dtype=y_dt, y_on_device=int(y_device_ptr is not None), Y=y_ptr, L=ptr(L),
                         psi0=ptr(psi0) if self.K > 0 else None, loc0=ptr(loc0), X=ptr(Xc), extra_loglik=ptr(ex),
                         N_src=int(Ns), G_src=int(Gs), cell_index=ptr(ci), gene_index=ptr(gi))
        opt = CaOptions()
        self.lib.ca_default_options(C.byref(opt))
        opt.learning_rate = float(learning_rate)
        opt.device = int(device)
        opt.y_storage = YSTORE[y_storage] if isinstance(y_storage, str) else int(y_storage)
        opt.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        opt.rank, opt.world = int(rank), int(world)
        opt.profile = 0x1F if profile is True else int(profile)
        opt.comm_timeout_ms = int(comm_timeout_ms)
        opt.gate_timeout_us = int(gate_timeout_us)
        voff = int(variant_off) if isinstance(variant_off, int) else sum(VARIANTS[v] for v in variant_off)
        opt.variant_off = (voff | (OPT_VERBOSE if verbose else 0)) & 0xFFFFFFFF
        opt.variant_on = int(variant_on) if isinstance(variant_on, int) else sum(VARIANTS_ON[v] for v in variant_on)
        for k, v in (tune or {}).items():
            if k in ("series_blocks", "series_side"):   # lab knobs of the series form (ca_options.reserved[0 / 1])
                opt.reserved[0 if k == "series_blocks" else 1] = int(v)
            elif k == "ride_pattern":       # (a << 8) | b, or "a:b"
                opt.ride_pattern = (lambda a, b: (int(a) << 8) | int(b))(*str(v).split(":")) if ":" in str(v) else int(v)
            else:
                opt.tune[TUNE[k]] = int(v)
        return prob, opt

    def _p2p_setup(self, p2p_exchange, world):
        """Two-phase bring-up of the one-shot peer-to-peer all-reduce (include/clonealign_hip.h): export, all-gather the handles,
        map the peers, AGREE over the control plane that every rank got that far, and only then commit -- the first call that
        waits for peers on the device.  A failure on any rank fails every rank the same way, with nobody left waiting."""
        err = None
        buf = C.create_string_buffer(P2P_HANDLE_BYTES)
        exported = self.lib.ca_p2p_export(self.h, buf) == CA_OK
        if not exported:
            err = (self.lib.ca_last_error(self.h) or b"").decode()
        allh = p2p_exchange(bytes(buf.raw) if exported else bytes(P2P_HANDLE_BYTES))
        if len(allh) != world or any(len(x) != P2P_HANDLE_BYTES for x in allh):
            raise ValueError("p2p_exchange must return one handle per rank, in rank order")
        ok = exported
        if ok and self.lib.ca_p2p_connect(self.h, b"".join(allh)) != CA_OK:
            ok, err = False, (self.lib.ca_last_error(self.h) or b"").decode()
        flags = p2p_exchange(bytes([1 if ok else 0]) + bytes(P2P_HANDLE_BYTES - 1))
        all_ok = len(flags) == world and all(len(f) > 0 and f[0] == 1 for f in flags)
        if exported:
            self._ck(self.lib.ca_p2p_commit(self.h, 1 if all_ok else 0))
        if not all_ok:
            raise EngineError(5, err or "peer-to-peer transport: another rank failed to export or map its peers")

    def comm_init(self, comm_id):
        """Join the RCCL communicator whose id rank 0 made (comm_unique_id()); collective.  Beside a committed peer-to-peer
        transport it only serves comm_benchmark("rccl"): the engine keeps reducing over peer-to-peer."""
        self._ck(self.lib.ca_comm_init(self.h, comm_id))

    def comm_benchmark(self, transport, n_calls=200, n_doubles=None):
        """us per all-reduce of ``n_doubles`` (default: the train pass's payload) on ``transport`` ("p2p" | "rccl"); collective."""
        us = C.c_double()
        n = int(self.info()["red_n"] if n_doubles is None else n_doubles)
        self._ck(self.lib.ca_comm_benchmark(self.h, {"rccl": 1, "p2p": 3}[transport], int(n_calls), n, C.byref(us)))
        return us.value

    def comm_selftest(self, n_rounds=4, n_doubles=None):
        """Known-answer test of the active all-reduce (ca_comm_selftest); returns the number of wrong sums (0 = good); collective."""
        bad = C.c_int64()
        n = int(self.info()["red_n"] if n_doubles is None else n_doubles)
        self._ck(self.lib.ca_comm_selftest(self.h, int(n_rounds), n, C.byref(bad)))
        return bad.value

    # -------------------------------------------------------------- plumbing
    def _ck(self, rc):
        if rc != CA_OK:
            raise EngineError(rc, (getattr(self.lib, self._LAST_ERROR)(self.h) or b"").decode())

    def _eps(self, eps):
        if eps is None:
            return None, None
        e = np.ascontiguousarray(np.asarray(eps, dtype=np.float32)).reshape(-1)
        if e.size != self.S * self.G:
            raise ValueError("eps must have S*G elements")
        return e, e.ctypes.data_as(C.c_void_p)

    def _stream(self, eps_stream, need):
        """eps_stream: None (built-in), ndarray [draws,S,G], or an object with .block(n)."""
        if eps_stream is None:
            return None, None, 0
        if hasattr(eps_stream, "block"):
            arr = eps_stream.block(need)
        else:
            arr = np.asarray(eps_stream, dtype=np.float32)
        arr = np.ascontiguousarray(arr, dtype=np.float32).reshape(-1, self.S * self.G)
        return arr, arr.ctypes.data_as(C.c_void_p), arr.shape[0]

    def info(self):
        i = CaInfo()
        self._ck(self.lib.ca_get_info(self.h, C.byref(i)))
        return {f[0]: getattr(i, f[0]) for f in CaInfo._fields_ if f[0] != "reserved"} | {
            "y_storage_name": YSTORE_NAME[i.y_storage], "transport_name": TRANSPORT_NAME.get(i.transport, "?")}

    # -------------------------------------------------------------- sess$run equivalents
    def gamma_init(self, eps):
        _k, p = self._eps(eps)
        self._ck(self._fn("gamma_init")(self.h, p))

    def elbo(self, eps):
        _k, p = self._eps(eps)
        out = C.c_double()
        self._ck(self._fn("elbo")(self.h, p, C.byref(out)))
        return out.value

    def elbo_terms(self, eps):
        _k, p = self._eps(eps)
        out = (C.c_double * 3)()
        self._ck(self.lib.ca_elbo_terms(self.h, p, out))
        return tuple(out)

    def step(self, eps):
        _k, p = self._eps(eps)
        self._ck(self._fn("step")(self.h, p))

    def gradients(self, eps):
        _k, p = self._eps(eps)
        out = C.c_double()
        self._ck(self.lib.ca_gradients(self.h, p, C.byref(out)))
        return {n: self._get(self.lib.ca_get_gradient, n) for n in self.VAR_NAMES}, out.value

    def run(self, eps_stream, max_iter, rel_tol, poll=None):
        """Whole loop of R/inference-tflow.R:368-417 in one call; returns the ELBO trace.

        ``poll(iteration, elbo)`` (optional) is called once per ELBO value as the host learns it (0 = the initial
        ELBO); a truthy return stops the loop after that iteration (ca_run_ex, CA_INTERRUPTED): the trace so far is
        returned and ``self.interrupted`` is set.  The hook may be slow and may call the read-only methods (``get``,
        ``get_params``, ``info``, ``synchronize``, ``stream_busy``): it sees the variables after that iteration, and the fit is
        the same bit for bit (include/clonealign_hip.h, ca_poll_fn); methods that change the engine's state raise from inside it."""
        need = 2 + 2 * int(max_iter)
        start = getattr(eps_stream, "draw", None)
        _k, p, n = self._stream(eps_stream, need)
        trace = np.zeros(int(max_iter) + 1, dtype=np.float64)
        cnt = C.c_int32()
        self.interrupted = False
        if poll is None:
            rc = self._fn("run_ex")(self.h, int(max_iter), float(rel_tol), p, n, trace.ctypes.data_as(C.c_void_p),
                                    C.byref(cnt), POLL_FN(0), None)
        else:
            err = []

            def _cb(_user, it, val, _f=poll):
                try:
                    return 1 if _f(int(it), float(val)) else 0
                except BaseException as e:       # never unwind through C: stop the loop, re-raise afterwards
                    err.append(e)
                    return 1
            cb = POLL_FN(_cb)
            rc = self._fn("run_ex")(self.h, int(max_iter), float(rel_tol), p, n, trace.ctypes.data_as(C.c_void_p),
                                    C.byref(cnt), cb, None)
            if err:
                raise err[0]
        if rc == CA_ERR_NAN:
            raise FloatingPointError((getattr(self.lib, self._LAST_ERROR)(self.h) or b"").decode())
        if rc == CA_INTERRUPTED:
            self.interrupted = True
        else:
            self._ck(rc)
        if start is not None:      # a stream object advances only by what the loop consumed
            eps_stream.draw = start + 2 * cnt.value
        return trace[:cnt.value].copy()

    def iterate(self, n_iter, eps_stream=None, want_elbo=True):
        _k, p, n = self._stream(eps_stream, 2 * int(n_iter))
        out = C.c_double()
        self._ck(self._fn("iterate")(self.h, int(n_iter), p, n, C.byref(out) if want_elbo else None))
        return out.value

    def final_elbo(self, eps_stream, n_rep=20):
        _k, p, n = self._stream(eps_stream, int(n_rep))
        vals = np.zeros(int(n_rep), dtype=np.float64)
        self._ck(self._fn("final_elbo")(self.h, int(n_rep), p, n, vals.ctypes.data_as(C.c_void_p), None, None))
        return vals

    def pca_init(self, noise=None, n_iter=40, seed=0):
        """psi <- scale(first K PCs of standardised log2(Y+1)) + noise, computed on the device (R/inference-tflow.R:204-208)."""
        out = np.zeros((self.N, self.K), dtype=np.float64, order=self._order)
        nz = None if noise is None else np.require(np.asarray(noise, dtype=np.float64).reshape(self.N, self.K),
                                                   requirements=[self._order, "A"])
        if self.K > 0:
            self._ck(self._fn("init_psi_pca")(self.h, None if nz is None else nz.ctypes.data_as(C.c_void_p), int(n_iter),
                                              int(seed) & 0xFFFFFFFFFFFFFFFF, out.ctypes.data_as(C.c_void_p)))
        return out

    def clone_gene_sums(self, clone_idx):
        """(T[G,C], Syy[G]): per-gene count sums by assigned clone and sums of squares over assigned cells (-1 = unassigned)."""
        ci = np.ascontiguousarray(np.asarray(clone_idx, dtype=np.int32).reshape(self.N))
        T = np.zeros((self.G, self.C), dtype=np.float64, order=self._order)
        Syy = np.zeros(self.G, dtype=np.float64)
        self._ck(self._fn("clone_gene_sums")(self.h, ci.ctypes.data_as(C.c_void_p), T.ctypes.data_as(C.c_void_p),
                                             Syy.ctypes.data_as(C.c_void_p)))
        return T, Syy

    def synchronize(self):
        self._ck(self.lib.ca_synchronize(self.h))

    def stream_busy(self):
        """True while work queued on the engine's stream has not completed (ca_stream_busy; never blocks)."""
        b = C.c_int32()
        self._ck(self.lib.ca_stream_busy(self.h, C.byref(b)))
        return bool(b.value)

    # -------------------------------------------------------------- fetch / poke
    def _shape(self, name):
        N, G, Cn, K, P = self.N, self.G, self.C, self.K, self.P
        return {"mu": (G,), "loc": (G,), "ls": (G,), "clone_probs": (N, Cn), "gamma_logits": (N, Cn), "s": (N,),
                "alpha": (Cn,), "alpha_unconstr": (Cn,), "beta": (G, P), "psi": (N, K), "W": (G, K),
                "chi": (K,), "v": (K,)}[name]

    def _get(self, fn, name):
        out = np.zeros(self._shape(name), dtype=np.float64, order=self._order)   # the library writes the problem's layout
        if out.size:
            self._ck(fn(self.h, name.encode(), out.ctypes.data_as(C.c_void_p)))
        return out

    def get(self, name):
        return self._get(self._fn("get_param"), name)

    def set(self, name, value):
        v = np.require(np.asarray(value, dtype=np.float64).reshape(self._shape(name)), requirements=[self._order, "A"])
        if v.size:
            self._ck(self.lib.ca_set_param(self.h, name.encode(), v.ctypes.data_as(C.c_void_p)))

    def reinit(self, psi0, loc0=None):
        """A new restart on the resident data: variables back to their initial values (psi = psi0, loc = loc0 or what the
        constructor started from), fresh Adam state (ca_reinit)."""
        p0 = None
        if self.K > 0:
            p0 = np.require(np.asarray(psi0, dtype=np.float64).reshape(self.N, self.K), requirements=[self._order, "A"])
        l0 = None if loc0 is None else np.ascontiguousarray(np.asarray(loc0, dtype=np.float64).reshape(self.G))
        self._ck(self._fn("reinit")(self.h, None if p0 is None else p0.ctypes.data_as(C.c_void_p),
                                    None if l0 is None else l0.ctypes.data_as(C.c_void_p)))

    def get_params(self):
        """R/inference-tflow.R:424-434."""
        out = {n: self.get(n) for n in ("mu", "clone_probs", "s", "alpha")}
        if self.P > 0:
            out["beta"] = self.get("beta")
        if self.K > 0:
            for n in ("psi", "W", "chi"):
                out[n] = self.get(n)
        return out

    def get_state(self):
        return {n: self.get(n) for n in self.VAR_NAMES}

    def kernel_times(self, reset=False):
        ms = (C.c_double * 5)()
        cnt = (C.c_int64 * 5)()
        self._ck(self.lib.ca_get_kernel_times(self.h, ms, cnt))
        if reset:
            self._ck(self.lib.ca_reset_kernel_times(self.h))
        return {k: (ms[i], cnt[i]) for i, k in enumerate(KERNEL_NAMES)}

    def set_profile(self, mask):
        self._ck(self.lib.ca_set_profile(self.h, int(mask)))

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            self._fn("destroy")(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HipGroupEngine(HipEngine):
    """ONE fit cell-sharded over several devices of this process (ca_group_*, include/clonealign_hip.h ABI 6): the engine behind
    ``inference_tflow(..., devices=[...])``.  Same constructor arguments as :class:`HipEngine` for ALL cells, plus ``devices`` (HIP
    ordinals in rank order; an ordinal may repeat on a one-GPU rig) and ``transport`` ("auto": peer-to-peer by address -> RCCL -> host
    reduction between the rank threads, each after a known-answer test; or one of "p2p" / "rccl" / "host" to insist on it).  Every method returns
    what the one-handle method returns for all cells.  Calls that have no group form (gradients, set, kernel_times ...) raise."""

    _PREFIX = "ca_group_"
    _LAST_ERROR = "ca_group_last_error"
    _TRANSPORT = {"auto": 0, "rccl": 1, "host": 2, "p2p": 3}

    def __init__(self, Y, L, psi0, loc0, K, S=1, X=None, extra_loglik=None, learning_rate=0.1, devices=(0,), transport="auto",
                 y_storage="auto", seed=0x5EED5EED, profile=False, y_device_ptr=None, y_device_dtype=None, shape=None, layout="row",
                 cell_index=None, gene_index=None, variant_off=(), variant_on=(), tune=None, verbose=False, comm_timeout_ms=0, gate_timeout_us=0):
        prob, opt = self._prepare(Y, L, psi0, loc0, K, S, X, extra_loglik, learning_rate, 0, y_storage, seed, 0, 1, profile,
                                  y_device_ptr, y_device_dtype, shape, layout, cell_index, gene_index, variant_off, variant_on, tune, verbose,
                                  comm_timeout_ms, gate_timeout_us)
        self.devices = [int(d) for d in devices]
        dev = (C.c_int32 * len(self.devices))(*self.devices)
        rc = self.lib.ca_group_create(C.byref(prob), C.byref(opt), dev, len(self.devices), self._TRANSPORT[transport], C.byref(self.h))
        if rc != CA_OK:
            msg = (self.lib.ca_group_last_error(None) or b"").decode()
            self.h = C.c_void_p()
            raise EngineError(rc, msg)

    def group_info(self):
        i = CaGroupInfo()
        self._ck(self.lib.ca_group_get_info(self.h, C.byref(i)))
        return {"world": i.world, "transport": i.transport, "transport_name": TRANSPORT_NAME.get(i.transport, "?"), "p2p_status": i.p2p_status,
                "rccl_status": i.rccl_status, "rebuilds": i.rebuilds, "selftest_rounds": i.selftest_rounds, "N": i.N, "note": i.note.decode()}

    def rank_info(self, rank):
        """ca_get_info of one rank's engine (its shard's cells, its kernel picks)."""
        h = C.c_void_p()
        self._ck(self.lib.ca_group_rank_handle(self.h, int(rank), C.byref(h)))
        i = CaInfo()
        rc = self.lib.ca_get_info(h, C.byref(i))
        if rc != CA_OK:
            raise EngineError(rc, "ca_get_info")
        return {f[0]: getattr(i, f[0]) for f in CaInfo._fields_} | {"y_storage_name": YSTORE_NAME[i.y_storage],
                                                                    "transport_name": TRANSPORT_NAME.get(i.transport, "?")}

    def info(self):
        return self.rank_info(0) | {"N": self.N, "group": self.group_info()}

    def _no(self, *a, **k):
        raise NotImplementedError("not available on a device group (use the rank engines through rank_info / a single HipEngine)")

    elbo_terms = gradients = set = kernel_times = set_profile = synchronize = stream_busy = _no
    comm_init = comm_benchmark = comm_selftest = _no


def eps_draw(seed, draw, n):
    """The library's built-in eps stream (host side), for checking against rng.normal_draw."""
    lib = load_library()
    out = np.zeros(int(n), dtype=np.float32)
    rc = lib.ca_eps_draw(int(seed), int(draw), int(n), out.ctypes.data_as(C.c_void_p))
    if rc != CA_OK:
        raise EngineError(rc, "ca_eps_draw")
    return out


def _lay(layout):
    if layout not in ("row", "col"):
        raise ValueError("layout must be 'row' or 'col'")
    return (CA_COL_MAJOR, "F") if layout == "col" else (CA_ROW_MAJOR, "C")


def allele_loglik(clone_allele, cov, ref, device=0, layout="row"):
    """The allele-specific [N, C] addend of the log-likelihood on the device (ca_allele_loglik; R/allele-specific.R:17-58
    with alt = cov - ref as at R/inference-tflow.R:173).  clone_allele [V, C]; cov, ref [N, V].  ``layout="col"`` hands the
    matrices over column-major, as the R caller would."""
    lib = load_library()
    lay, order = _lay(layout)
    req = [order, "A"]
    ca = np.require(np.asarray(clone_allele, dtype=np.float64), requirements=req)
    cv = np.require(np.asarray(cov, dtype=np.float64), requirements=req)
    rf = np.require(np.asarray(ref, dtype=np.float64), requirements=req)
    V, Cn = ca.shape
    N = cv.shape[0]
    if cv.shape != (N, V) or rf.shape != (N, V):
        raise ValueError("cov and ref must be cells x variants, clone_allele variants x clones")
    out = np.zeros((N, Cn), dtype=np.float64, order=order)
    err = C.create_string_buffer(256)
    rc = lib.ca_allele_loglik(N, V, Cn, lay, ca.ctypes.data_as(C.c_void_p), cv.ctypes.data_as(C.c_void_p),
                              rf.ctypes.data_as(C.c_void_p), int(device), out.ctypes.data_as(C.c_void_p), err)
    if rc != CA_OK:
        raise EngineError(rc, err.value.decode() or "ca_allele_loglik")
    return out


_NP_DTYPE = {np.dtype(np.float64): 0, np.dtype(np.float32): 1, np.dtype(np.int32): 2, np.dtype(np.uint16): 3, np.dtype(np.uint8): 4}


def preprocess_masks(Y, L, min_counts_per_gene=20, min_counts_per_cell=100, remove_outlying_genes=True, nmads=10,
                     max_copy_number=6, remove_genes_same_copy_number=True, device=0, layout="row"):
    """Gene / cell retention masks of preprocess_for_clonealign() with the two O(N G) statistics taken on the device
    (ca_preprocess; R/preprocess.R:93-147).  Y [N, G] in float64/float32/int32/uint16/uint8 (other dtypes are converted
    to float64), L [G, C].  Returns (keep_gene bool[G], keep_cell bool[N], gene_sums[G], cell_sums[N])."""
    lib = load_library()
    Y = np.asarray(Y)
    if Y.dtype not in _NP_DTYPE:
        Y = Y.astype(np.float64)
    lay, order = _lay(layout)
    Y = np.require(Y, requirements=[order, "A"])
    L = np.require(np.asarray(L, dtype=np.float64), requirements=[order, "A"])
    N, G = Y.shape
    if L.shape[0] != G:
        raise ValueError("copy_number_data must have same number of genes (rows) as gene_expression_data")
    pp = CaPreprocessParams(float(min_counts_per_gene), float(min_counts_per_cell), int(bool(remove_outlying_genes)),
                            int(bool(remove_genes_same_copy_number)), float(nmads), float(max_copy_number))
    kg = np.zeros(G, dtype=np.uint8)
    kc = np.zeros(N, dtype=np.uint8)
    gs = np.zeros(G, dtype=np.float64)
    cs = np.zeros(N, dtype=np.float64)
    err = C.create_string_buffer(256)
    rc = lib.ca_preprocess(N, G, L.shape[1], lay, _NP_DTYPE[Y.dtype], 0, Y.ctypes.data_as(C.c_void_p),
                           L.ctypes.data_as(C.c_void_p), C.byref(pp), int(device), kg.ctypes.data_as(C.c_void_p),
                           kc.ctypes.data_as(C.c_void_p), gs.ctypes.data_as(C.c_void_p), cs.ctypes.data_as(C.c_void_p), err)
    if rc != CA_OK:
        raise EngineError(rc, err.value.decode() or "ca_preprocess")
    return kg.astype(bool), kc.astype(bool), gs, cs
