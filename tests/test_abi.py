"""The C-ABI library loads (no GPU needed) and exports every symbol include/*.h declares."""
import ctypes
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    syms = []
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        txt = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        syms += re.findall(r"^\s*(?:const\s+char\s*\*|int)\s+(ca_\w+)\s*\(", txt, flags=re.M)
    return sorted(set(syms))


def test_library_exports_every_declared_symbol():
    from clonealign_amd import engine
    lib = engine.load_library()
    syms = declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), s
    assert set(syms) == set(engine.EXPORTS)
    assert lib.ca_abi_version() == 1


def test_structs_match_header_sizes():
    from clonealign_amd import engine
    assert ctypes.sizeof(engine.CaProblem) == 8 + 4 * 8 + 6 * 8
    assert ctypes.sizeof(engine.CaOptions) == 4 * 8 + 8 + 4 * 5 + 4 * 7
    opt = engine.CaOptions()
    engine.load_library().ca_default_options(ctypes.byref(opt))
    assert (opt.learning_rate, opt.beta1, opt.beta2, opt.adam_eps, opt.world) == (0.1, 0.9, 0.999, 1e-8, 1)


def test_create_fails_loudly_without_gpu_or_with_bad_args():
    import numpy as np
    import pytest
    from clonealign_amd.engine import EngineError, HipEngine
    with pytest.raises(EngineError):      # K + P > 8 is rejected before any HIP call
        HipEngine(np.ones((4, 3)), np.ones((3, 2)), np.zeros((4, 9)), np.zeros(3), K=9)
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if not has_gpu:
        with pytest.raises(EngineError):  # no silent CPU fallback
            HipEngine(np.ones((4, 3)), np.ones((3, 2)), np.zeros((4, 1)), np.zeros(3), K=1)
