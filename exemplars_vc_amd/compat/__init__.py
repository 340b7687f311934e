"""Drop-in mirrors of the reference's call surfaces for the activation-solve path.

  factorize  - `_factorize` / `convert` of 04_align_n_nmf.py (the scikit-learn based live path)
  pymf       - `pymf.nmf.NMF(data, num_bases).factorize(compute_w=False)`
  nmf_tool   - `nmf_tool.nmf.NMF(...).fit_transform(X, r, initW=True, givenW=A)`
  griffin_lim - `zz_audio_utilities.reconstruct_signal_griffin_lim` (the STFT back end)
  make_dict  - `_dtw_alignment` / `dtw_alignment` / `align_sp_ap_f0` (dictionary construction)
"""
from . import factorize, griffin_lim, make_dict, nmf_tool, pymf  # noqa: F401
