"""exemplars_vc_amd - MI355X-native exemplar-NMF activation solver.

One hot path of entn-at/exemplars_vc, rebuilt for gfx950: given a fixed parallel exemplar
dictionary (A source, B target) and non-negative source frames X, run the Frobenius
multiplicative update  H <- H (.) A^T X (/) (A^T A H + eps)  and synthesise Y = B H.
Hand-written HIP kernels behind a C ABI (include/evc.h); this package is the host-side
mirror of the reference's three Python call surfaces (see `compat`).
"""
from .solver import (PreparedDictionary, cached_dictionary, convert, dtw_align, dtw_dictionary, frame_residuals, griffin_lim,
                     griffin_lim_batch, prepare_dictionary, release_workspaces, require_device, solve_activations, stft,
                     synthesize, workspace_bytes)
from . import compat, shard  # noqa: F401

__all__ = ["prepare_dictionary", "cached_dictionary", "PreparedDictionary", "solve_activations", "convert", "synthesize", "griffin_lim", "griffin_lim_batch", "stft", "dtw_align", "dtw_dictionary", "frame_residuals", "workspace_bytes",
           "require_device", "release_workspaces", "compat", "shard"]
__version__ = "0.1.0"
