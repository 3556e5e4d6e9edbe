"""Golden vectors for GMM training: sklearn.mixture.GaussianMixture(covariance_type='diag').fit — the very call the
reference trains with (GMM_UBM.py:158-160,169-170) — run from explicit initial parameters (weights_init / means_init /
precisions_init), so that the EM iterates are deterministic.  Run in the build container:

    python tests/golden/make_golden_em.py        ->  tests/golden/gmm_em.npz
"""
import os
import warnings

import numpy as np
from sklearn.mixture import GaussianMixture

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    out = {}
    cases = [("a", 8, 13, 3000, 5, 0.0), ("b", 16, 26, 5000, 3, 0.0), ("c", 5, 7, 777, 100, 1e-3), ("d", 70, 39, 6000, 2, 0.0)]
    for tag, K, D, n, max_iter, tol in cases:
        rng = np.random.default_rng(ord(tag) + 17)
        centres = 2.0 * rng.standard_normal((K, D))
        lab = rng.integers(0, K, n)
        X = (centres[lab] + rng.uniform(0.5, 1.5, (K, D))[lab] * rng.standard_normal((n, D))).astype(np.float32)
        w0 = rng.dirichlet(5 * np.ones(K))
        mu0 = centres + 0.5 * rng.standard_normal((K, D))
        cov0 = rng.uniform(0.8, 2.0, (K, D))
        g = GaussianMixture(n_components=K, covariance_type="diag", tol=tol, max_iter=max_iter, reg_covar=1e-6,
                            weights_init=w0, means_init=mu0, precisions_init=1.0 / cov0)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")  # ConvergenceWarning of the fixed-iteration cases
            g.fit(X.astype(np.float64))
        out.update({"%s_X" % tag: X, "%s_w0" % tag: w0, "%s_mu0" % tag: mu0, "%s_cov0" % tag: cov0,
                    "%s_cfg" % tag: np.array([K, D, n, max_iter, tol]), "%s_w" % tag: g.weights_, "%s_mu" % tag: g.means_,
                    "%s_cov" % tag: g.covariances_, "%s_lb" % tag: np.array(g.lower_bound_), "%s_niter" % tag: np.array(g.n_iter_),
                    "%s_conv" % tag: np.array(g.converged_)})
    np.savez_compressed(os.path.join(HERE, "gmm_em.npz"), **out)
    print("wrote gmm_em.npz", {k: v.shape for k, v in out.items() if k.endswith("_X")})


if __name__ == "__main__":
    main()
