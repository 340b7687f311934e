"""On-disk artefacts of the reference pipeline: read what its scripts wrote, write what
they can read back, so the GPU solver can be swapped into an existing `data/vc/exem_dict/` tree.

Files (all `pickle`, protocol 3, numpy arrays inside plain lists/dicts):

  exemplar_W_A, exemplar_W_B          list (one entry per parallel file) of DTW path index arrays
                                      01_make_dict_parallel.py:325-339, read at 04_align_n_nmf.py:65-70
  <spk>_feat_sp_ap_f0.pkl             list of {'sp','ap','f0','fs','sr'}     03_a_b_r_parallel.py:98,122-135
  <spk>_feat_stft.pkl                 list of {'stft','fs'}                   03_a_b_r_parallel.py:101-104,139-153
  H_test_sp_ap_f0_<n>.pkl             {'H_sp','H_ap','H_f0'}  (N x T each)    04_align_n_nmf.py:251-260,296-297
  R_test_sp_ap_f0_<n>.pkl             {'r_sp','r_ap','r_f0'}                  04_align_n_nmf.py:256-258,299-301
  H_test_stft_<n>.pkl                 {'H_stft'}                              04_align_n_nmf.py:305-308,330-331

Reading never executes anything from the file: `safe_load` is a pickle reader whose only admissible
globals are numpy's array/dtype/scalar reconstructors (both the numpy 1.x and 2.x module spellings) and
`collections.OrderedDict`; anything else - including object arrays - raises `UnsafeArtifactError`.
No pickle ships with the reference; the tests exercise files written by `dump` below, byte-compatible
with what `pickle.dump(obj, f, protocol=3)` produces there.

The reference keys its activation caches by the NUMBER of dictionary files only (`<n>`), so converting a
second utterance silently returns the first one's H.  `activation_cache_path(...,
key="content")` (the default of `compat.factorize.factorize(cache_dir=...)`) appends a digest of the
utterance and the dictionary; `key="reference"` reproduces the reference's file name so existing caches
are found.
"""
from __future__ import annotations

import hashlib
import io
import os
import pickle

import numpy as np

EXEM_DICT = os.path.join("data", "vc", "exem_dict")        # the reference's hard-coded directory
PROTOCOL = 3                                               # 03_a_b_r_parallel.py:133, 01_make_dict_parallel.py:325


class UnsafeArtifactError(pickle.UnpicklingError):
    pass


# numpy moved numpy.core -> numpy._core in 2.0; files written by either generation must load
def _resolve(module, name):
    allowed = {
        ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
        ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
        ("numpy", "ndarray"), ("numpy", "dtype"),
        ("collections", "OrderedDict"),
    }
    if (module, name) not in allowed:
        raise UnsafeArtifactError(f"global {module}.{name} is not admissible in a reference artefact")
    if module == "collections":
        import collections
        return collections.OrderedDict
    if module == "numpy":
        return getattr(np, name)
    try:
        from numpy._core import multiarray as ma          # numpy >= 2
    except ImportError:                                   # pragma: no cover
        from numpy.core import multiarray as ma
    return getattr(ma, name)


class _ArtifactUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        return _resolve(module, name)

    def persistent_load(self, pid):
        raise UnsafeArtifactError("persistent ids are not admissible in a reference artefact")


def _reject_object_arrays(obj, depth=0):
    if depth > 16:
        raise UnsafeArtifactError("artefact nesting too deep")
    if isinstance(obj, np.ndarray):
        if obj.dtype.hasobject:
            raise UnsafeArtifactError("object arrays are not admissible in a reference artefact")
    elif isinstance(obj, dict):
        for v in obj.values():
            _reject_object_arrays(v, depth + 1)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            _reject_object_arrays(v, depth + 1)


def safe_loads(data: bytes):
    obj = _ArtifactUnpickler(io.BytesIO(data)).load()
    _reject_object_arrays(obj)
    return obj


def safe_load(path):
    with open(path, "rb") as f:
        return safe_loads(f.read())


def dump(obj, path, protocol=PROTOCOL):
    """`pickle.dump(obj, f, protocol=3)` as the reference writes it (creates the directory)."""
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "wb") as f:
        pickle.dump(obj, f, protocol=protocol)


# ------------------------------------------------------------------ dictionaries and features

def feature_path(root, speaker, use_stft):
    name = f"{speaker}_feat_stft.pkl" if use_stft else f"{speaker}_feat_sp_ap_f0.pkl"
    return os.path.join(root, EXEM_DICT, name)


def _check_features(feats, use_stft, path):
    if not isinstance(feats, list):
        raise ValueError(f"{path}: expected a list of per-file feature dicts, got {type(feats).__name__}")
    need = ("stft", "fs") if use_stft else ("sp", "ap", "f0", "fs")
    for i, f in enumerate(feats):
        if not isinstance(f, dict) or any(k not in f for k in need):
            raise ValueError(f"{path}: entry {i} lacks one of {need}")
    return feats


def read_features(root, speaker, use_stft):
    """`<spk>_feat_stft.pkl` / `<spk>_feat_sp_ap_f0.pkl`: list of per-file dicts (frames as rows)."""
    path = feature_path(root, speaker, use_stft)
    return _check_features(safe_load(path), use_stft, path)


def write_features(root, speaker, feats, use_stft):
    path = feature_path(root, speaker, use_stft)
    dump(_check_features(list(feats), use_stft, path), path)
    return path


def read_exemplar_paths(root):
    """(`exemplar_W_A`, `exemplar_W_B`): per parallel file the DTW path into the source / target
    utterance; frame k of the dictionary built from file i is (src[i][W_A[i][k]], tar[i][W_B[i][k]])."""
    out = []
    for name in ("exemplar_W_A", "exemplar_W_B"):
        path = os.path.join(root, EXEM_DICT, name)
        W = safe_load(path)
        if not isinstance(W, (list, tuple)):
            raise ValueError(f"{path}: expected a list of index arrays, got {type(W).__name__}")
        W = [np.asarray(w) for w in W]
        for i, w in enumerate(W):
            if w.ndim != 1 or not np.issubdtype(w.dtype, np.integer):
                raise ValueError(f"{path}: entry {i} is not a 1-D integer index array")
        out.append(W)
    if len(out[0]) != len(out[1]) or any(len(a) != len(b) for a, b in zip(*out)):
        raise ValueError("exemplar_W_A and exemplar_W_B do not describe the same alignment")
    return out[0], out[1]


def write_exemplar_paths(root, W_A, W_B):
    """io_save_exemplar_dictionaries({'exemplar_W_A': ..., 'exemplar_W_B': ...}) (01_make_dict_parallel.py:325-339)."""
    for name, W in (("exemplar_W_A", W_A), ("exemplar_W_B", W_B)):
        dump([np.asarray(w) for w in W], os.path.join(root, EXEM_DICT, name))


def io_load_from_pickle(root=".", speakerA="SF1", speakerB="TF1", use_stft=True):
    """`io_load_from_pickle()` of 04_align_n_nmf.py:65-85 with its module-level settings as arguments:
    -> (src_feat, tar_feat, src_W, tar_W)."""
    src_W, tar_W = read_exemplar_paths(root)
    return read_features(root, speakerA, use_stft), read_features(root, speakerB, use_stft), src_W, tar_W


# ------------------------------------------------------------------ activation / residual caches

def content_digest(*arrays):
    h = hashlib.sha1()
    for a in arrays:
        a = np.ascontiguousarray(a)
        h.update(str((a.shape, a.dtype.str)).encode())
        h.update(a.tobytes())
    return h.hexdigest()[:16]


def activation_cache_path(root, use_stft, n_files, kind="H", key="reference", digest=None):
    """`data/vc/exem_dict/H_test_stft_<n>.pkl` etc.; key='content' adds `_<digest>` before `.pkl`."""
    if kind not in ("H", "R"):
        raise ValueError("kind must be 'H' or 'R'")
    if kind == "R" and use_stft:
        raise ValueError("the STFT branch has no residual file")
    stem = f"{kind}_test_stft_{n_files}" if use_stft else f"{kind}_test_sp_ap_f0_{n_files}"
    if key == "content":
        if not digest:
            raise ValueError("key='content' needs a digest")
        stem += "_" + digest
    elif key != "reference":
        raise ValueError("key must be 'reference' or 'content'")
    return os.path.join(root, EXEM_DICT, stem + ".pkl")


def _check_named_matrices(d, names, path):
    if not isinstance(d, dict) or any(n not in d for n in names):
        raise ValueError(f"{path}: expected a dict with keys {names}")
    for n in names:
        if not isinstance(d[n], np.ndarray) or d[n].ndim != 2:
            raise ValueError(f"{path}: {n} is not a 2-D array")
    return d


def read_activations(path, use_stft):
    names = ("H_stft",) if use_stft else ("H_sp", "H_ap", "H_f0")
    return _check_named_matrices(safe_load(path), names, path)


def read_residuals(path):
    return _check_named_matrices(safe_load(path), ("r_sp", "r_ap", "r_f0"), path)


def write_activations(path, H):
    # default protocol in the reference (pickle.dump(H, f), Python 3.6 => 3)
    dump({k: np.asarray(v) for k, v in H.items()}, path)


write_residuals = write_activations
