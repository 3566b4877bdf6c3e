"""The R-side `.Call` shim (clonealign_amd/r_shim/clonealign_hip_shim.c) compiles cleanly against include/clonealign_hip.h.

The image has no R toolchain, so the R API is a minimal stand-in (tests/r_stub/Rinternals.h: declarations only for the calls
the shim makes).  This is the CPU half; tests/test_gpu_boundary.py calls the compiled entry point on the GPU box."""
import ctypes
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "clonealign_amd", "r_shim", "clonealign_hip_shim.c")


def test_shim_is_valid_c_against_the_header():
    r = subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "tests", "r_stub"),
                        "-I", os.path.join(ROOT, "include"), SHIM], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_shim_uses_the_column_major_boundary_and_the_interrupt_hook():
    src = open(SHIM).read()
    assert "p.layout = CA_COL_MAJOR" in src
    assert "ca_run_ex(" in src and "R_CheckUserInterrupt" in src and "R_ToplevelExec" in src
    assert src.index("ca_destroy(h)") < src.index("Rf_error(\"%s: %s\"")     # device memory is freed before the longjmp


def test_harness_links_against_the_engine_library():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "r_stub")], stdout=subprocess.DEVNULL)
    lib = ctypes.CDLL(os.path.join(ROOT, "tests", "r_stub", "libshim_harness.so"))
    assert hasattr(lib, "harness_fit") and hasattr(lib, "C_clonealign_fit") and hasattr(lib, "R_init_clonealign")
