"""The R-side `.Call` shim (clonealign_amd/r_shim/src/clonealign_hip_shim.c + src/init.c) compiles cleanly against include/clonealign_hip.h.

The image has no R toolchain, so the R API is a minimal stand-in (tests/r_stub/Rinternals.h: declarations only for the calls
the shim makes).  This is the CPU half; tests/test_gpu_boundary.py calls the compiled entry point on the GPU box."""
import ctypes
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC_DIR = os.path.join(ROOT, "clonealign_amd", "r_shim", "src")
SHIM = os.path.join(SRC_DIR, "clonealign_hip_shim.c")
INIT = os.path.join(SRC_DIR, "init.c")


def test_shim_is_valid_c_against_the_header():
    for src in (SHIM, INIT):
        r = subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "tests", "r_stub"),
                            "-I", os.path.join(ROOT, "include"), src], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr


def test_registration_table_matches_the_entry_points_and_the_r_calls():
    """src/init.c registers every SEXP entry point of the shim with the number of arguments its definition takes, and every
    .Call in R/inference-hip.R names a registered routine and passes that many arguments."""
    import re
    shim, init = open(SHIM).read(), open(INIT).read()
    rsrc = open(os.path.join(ROOT, "clonealign_amd", "r_shim", "R", "inference-hip.R")).read()
    defs = {m.group(1): m.group(2).count("SEXP") for m in re.finditer(r"^SEXP (C_clonealign_\w+)\(([^)]*)\)\s*\{", shim, re.S | re.M)}
    table = {m.group(1): int(m.group(2)) for m in re.finditer(r'\{"(C_clonealign_\w+)", \(DL_FUNC\)&\1, (\d+)\}', init)}
    assert defs == table and len(table) == 4, (defs, table)
    assert "R_init_clonealign" in init and "R_useDynamicSymbols(dll, FALSE)" in init and "R_init_clonealign" not in shim

    def n_args(text, start):                    # arguments of the call whose '(' is at `start`, commas at depth 1 only
        depth, n, i = 0, 1, start
        while True:
            c = text[i]
            if c in "([":
                depth += 1
            elif c in ")]":
                depth -= 1
                if depth == 0:
                    return n
            elif c == "," and depth == 1:
                n += 1
            i += 1
    calls = [(m.group(1), n_args(rsrc, m.start(0) + len(".Call"))) for m in re.finditer(r'\.Call\("(C_clonealign_\w+)"', rsrc)]
    assert {c for c, _ in calls} == set(table), calls
    for name, n in calls:                       # .Call(name, <args...>, PACKAGE = "clonealign")
        assert n - 2 == table[name], (name, n - 2, table[name])


def test_makevars_links_the_engine_library_by_its_c_abi_only():
    mk = open(os.path.join(SRC_DIR, "Makevars")).read()
    assert "-lclonealign_hip" in mk and "-I$(CLONEALIGN_HIP)/include" in mk and "init.o" in mk and "hipcc" not in mk.split("PKG_CPPFLAGS")[1]


def test_shim_uses_the_column_major_boundary_and_the_interrupt_hook():
    src = open(SHIM).read()
    assert "p.layout = CA_COL_MAJOR" in src
    assert "ca_run_ex(" in src and "R_CheckUserInterrupt" in src and "R_ToplevelExec" in src
    assert src.index("ca_group_destroy(g)") < src.index("Rf_error(\"%s: %s\"")     # device memory (every rank's) is freed before the longjmp
    # round 6: the single fit goes through the device-group entry points (one device = a group of one), the hook runs on the R thread
    assert "ca_group_create(&p, &o, devs, n_dev, 0, &h)" in src and "ca_group_run_ex(" in src and "poll_interrupt" in src


def test_harness_links_against_the_engine_library():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "r_stub")], stdout=subprocess.DEVNULL)
    lib = ctypes.CDLL(os.path.join(ROOT, "tests", "r_stub", "libshim_harness.so"))
    assert hasattr(lib, "harness_fit") and hasattr(lib, "C_clonealign_fit") and hasattr(lib, "R_init_clonealign")
    assert hasattr(lib, "harness_fit_devices") and hasattr(lib, "rstub_violations") and hasattr(lib, "rstub_protect_depth")


def test_the_r_stub_enforces_the_protect_rules_it_claims_to():
    """Round 6: the stand-in R runtime collects at EVERY allocation and keeps a protect stack, so the shim's PROTECT discipline is
    tested on the GPU box (tests/test_gpu_boundary.py), not only compiled.  Here, without a GPU, the stub itself is held to its word: an
    unprotected object is dead after the next allocation and touching it is recorded; a protected one, and one reachable from a protected list,
    survive; UNPROTECT underflow is recorded; the depth is what PROTECT / UNPROTECT left."""
    prog = r'''
#include <stdio.h>
#include "Rinternals.h"
int main(void) {
  char m[256];
  rstub_begin_call();
  SEXP a = Rf_allocVector(REALSXP, 4);            /* never protected */
  SEXP b = PROTECT(Rf_allocVector(REALSXP, 4));   /* this allocation collects a */
  REAL(b)[0] = 1.0;
  SEXP l = PROTECT(Rf_allocVector(VECSXP, 2));
  SET_VECTOR_ELT(l, 0, Rf_allocVector(REALSXP, 3));   /* reachable from l as soon as it is stored */
  SEXP c = Rf_allocVector(REALSXP, 2);
  (void)c;
  if (rstub_violations(m, 256) != 0) { printf("early violation: %s\n", m); return 1; }
  REAL(VECTOR_ELT(l, 0))[0] = 2.0;
  if (rstub_violations(m, 256) != 0) { printf("a reachable object was collected: %s\n", m); return 1; }
  REAL(a)[0] = 3.0;                               /* use after collection */
  if (rstub_violations(m, 256) != 1) { printf("use of a collected object went unnoticed\n"); return 1; }
  if (rstub_protect_depth() != 2) { printf("depth %d\n", rstub_protect_depth()); return 1; }
  UNPROTECT(2);
  UNPROTECT(1);                                   /* underflow */
  if (rstub_violations(m, 256) != 2 || rstub_protect_depth() != 0) { printf("underflow unnoticed\n"); return 1; }
  rstub_free_all();
  printf("ok\n");
  return 0;
}
'''
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "t.c")
        open(src, "w").write(prog)
        exe = os.path.join(d, "t")
        subprocess.check_call(["gcc", "-std=c11", "-I", os.path.join(ROOT, "tests", "r_stub"), "-o", exe, src, os.path.join(ROOT, "tests", "r_stub", "rstub.c")])
        out = subprocess.run([exe], capture_output=True, text=True)
        assert out.returncode == 0 and out.stdout.strip() == "ok", out.stdout + out.stderr
