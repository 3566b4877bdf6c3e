"""clonealign_amd -- MI355X-native variational-inference engine for clonealign.

Drop-in for the hot path of kieranrcampbell/clonealign (``inference_tflow`` and its
callers ``clonealign`` / ``run_clonealign``): the TensorFlow ELBO loop is replaced by
hand-written HIP kernels for gfx950 behind the C ABI in ``include/clonealign_hip.h``.
"""
from .api import (ClonealignFit, clone_assignment, clonealign, compute_correlations,  # noqa: F401
                  recompute_clone_assignment, run_clonealign)
from .hostprep import inverse_softplus, safe_inverse_softplus, saturate, softplus  # noqa: F401
from .inference import inference_tflow  # noqa: F401
from .preprocess import preprocess_for_clonealign  # noqa: F401

__version__ = "0.1.0"
