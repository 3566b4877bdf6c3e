import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


# Order of the GPU suite under `pytest -x`: the oracle-parity evidence first, process plumbing last, so that one flaky
# multi-process launch can never hide the numerical record (r03: test_gpu_multi stopped the run before parity was collected).
_FILE_ORDER = ["test_gpu_parity.py", "test_gpu_scale.py", "test_gpu_boundary.py", "test_gpu_group.py", "test_gpu_sharding.py", "test_gpu_multi.py"]
# inside test_gpu_scale.py the BASELINE.json configurations come first (cfg-2, cfg-3, cfg-5), then the rest in file order
_SCALE_FIRST = ["test_cfg2_iterations_match_c_oracle", "test_fused_loop_at_shard_size_matches_c_oracle", "test_full_size_cfg3_matches_c_oracle",
                "test_full_size_run_is_bit_reproducible_and_storage_is_u8", "test_full_size_gradient_matches_finite_difference_of_the_elbo",
                "test_full_size_two_shards_equal_one_engine", "test_config5_run_clonealign_eight_restarts_at_size"]


def pytest_collection_modifyitems(session, config, items):
    def key(ix_item):
        ix, item = ix_item
        fname = os.path.basename(str(item.fspath))
        f = _FILE_ORDER.index(fname) if fname in _FILE_ORDER else -1          # CPU files keep their place in front
        base = item.name.split("[")[0]
        w = _SCALE_FIRST.index(base) if (fname == "test_gpu_scale.py" and base in _SCALE_FIRST) else len(_SCALE_FIRST)
        return (f, w if fname == "test_gpu_scale.py" else 0, ix)
    items[:] = [it for _, it in sorted(enumerate(items), key=key)]


def pytest_sessionstart(session):
    """(Re)build the in-tree native libraries: the gfx950 engine (hipcc cross-compiles without a GPU), the C oracle and
    the R-shim harness.  ALWAYS runs the makefiles -- they are incremental and carry the header dependencies, so an
    up-to-date tree costs a fraction of a second, and a stale .so left from older sources (the .so files are git-ignored
    but travel to the GPU box) can never be what the tests validate (tests/test_abi.py also compares the build id compiled
    into the library with the hash of the sources).  Same makefiles as ``__graft_entry__.build()``."""
    if os.environ.get("PYTEST_XDIST_WORKER"):
        return
    # GPU runs: torch's bundled HIP runtime has to be the FIRST one initialised in this process (see below); a run that starts
    # with tests which only load the engine (pytest tests/test_gpu_parity.py tests/test_gpu_scale.py) otherwise leaves the
    # torch-using tests after them with "No HIP GPUs are available"
    if "not gpu" not in (session.config.option.markexpr or ""):
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.init()
        except Exception:
            pass
    import subprocess
    # (makefiles only, in child processes: loading the engine here would bring the system HIP runtime into this process
    #  before the tests that import torch load torch's bundled one, and the second runtime then sees no device)
    for d in (os.path.join(ROOT, "clonealign_amd", "csrc"), os.path.join(ROOT, "oracle", "c"), os.path.join(ROOT, "tests", "r_stub")):
        try:
            subprocess.check_call(["make", "-C", d], stdout=subprocess.DEVNULL)
        except Exception as e:  # the tests that need the libraries then fail with their own message
            print(f"[conftest] make -C {d} failed: {e}", file=sys.stderr)
