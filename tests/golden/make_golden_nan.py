"""Golden vectors for NaN rows through the scaling step (`preprocessing.scale` at /root/reference/GMM_UBM.py:93,99).

A digitally silent frame of the sidekit dialect has ln 0 = -inf filterbank energies and NaN cepstra (no floor); `GMM_UBM.delta`
(GMM_UBM.py:53-69) spreads them +-2 frames, and the reference then hands the matrix to sklearn's `scale`, which takes its statistics
over the entries that are not NaN and leaves the NaN entries in place (and raises on +-inf).  This script calls that very library
function — installed in the build container — on seeded inputs and stores inputs + outputs; nothing of the reference's source is
read or stored.  Run:  python tests/golden/make_golden_nan.py
"""
import os
import warnings

import numpy as np
from sklearn import preprocessing

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    rng = np.random.default_rng(2024)
    out = {}
    # (T, D) with NaN rows in a delta-like pattern: cepstra NaN on rows 40..49, deltas on 38..51
    X = rng.normal(size=(120, 26)) * rng.uniform(0.5, 3.0, size=26) + rng.normal(size=26) * 4.0
    X[40:50, :13] = np.nan
    X[38:52, 13:] = np.nan
    out["nanrows_in"] = X
    out["nanrows_out"] = preprocessing.scale(X)
    # one column entirely NaN, one with a single finite entry, one constant among NaNs
    Y = rng.normal(size=(30, 5))
    Y[:, 1] = np.nan
    Y[:, 2] = np.nan
    Y[7, 2] = 1.5
    Y[:, 3] = 2.0
    Y[::3, 3] = np.nan
    out["nancols_in"] = Y
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        out["nancols_out"] = preprocessing.scale(Y)
    # +-inf: the library raises (recorded as a fact, the message is not stored)
    Z = rng.normal(size=(10, 3))
    Z[4, 1] = -np.inf
    try:
        preprocessing.scale(Z)
        out["inf_raises"] = np.array(0)
    except ValueError:
        out["inf_raises"] = np.array(1)
    import sklearn
    out["sklearn_version"] = np.array(sklearn.__version__)
    np.savez_compressed(os.path.join(HERE, "scale_nan.npz"), **out)
    print({k: getattr(v, "shape", None) for k, v in out.items()}, "inf_raises", int(out["inf_raises"]))


if __name__ == "__main__":
    main()
