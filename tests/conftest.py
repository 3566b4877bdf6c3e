import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


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
