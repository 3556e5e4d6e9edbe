"""GPU-backed EM training of diagonal GMMs with the call surface of the class the reference trains with:
``sklearn.mixture.GaussianMixture(n_components, covariance_type='diag')`` (GMM_UBM.py:158-170).

The O(frames x K x D) work of every EM iteration — E step and the resp.T @ X / resp.T @ X^2 sums — runs in HIP kernels
(``ssp_gmm_em_stats``); the O(K x D) closing arithmetic of the M step and the convergence test are the float64 lines of
sklearn's ``_m_step`` / ``fit_predict`` restated here on the host.  Trained objects duck-type a fitted sklearn model
(``weights_``, ``means_``, ``covariances_``, ``precisions_cholesky_``, ``converged_``, ``n_iter_``, ``lower_bound_``),
so ``GMM_UBM.score_matrix`` / ``api.GmmScorer.from_sklearn`` take them as they take sklearn's.

Initialisation.  With ``weights_init`` / ``means_init`` / ``precisions_init`` given, the start is exactly sklearn's and
so is every iterate (parity tests).  Otherwise ``init_params='kmeans'`` (sklearn's default too): k-means++ seeding on a
host-side subsample (sequential sampling, as sklearn does it on the CPU), Lloyd iterations on the GPU — the E/M statistics
kernels at a small shared spherical variance are the hard-assignment limit of the same sums — and the component weights,
means and variances of the resulting clusters as the start, like sklearn's ``_initialize_parameters``.  The random streams
differ from sklearn's, so trained models agree in quality, not bit for bit.  ``init_params='random_from_data'``: K distinct
frames as means, the global variance as every covariance, uniform weights.
"""
from __future__ import annotations

import numpy as np

from . import api


class GaussianMixture:
    def __init__(self, n_components=1, covariance_type='diag', tol=1e-3, reg_covar=1e-6, max_iter=100, n_init=1,
                 weights_init=None, means_init=None, precisions_init=None, random_state=None, init_params='kmeans', ctx=None):
        if covariance_type != 'diag':
            raise ValueError("only covariance_type='diag' is supported (what GMM_UBM.py:158,169 uses)")
        if n_components < 1 or max_iter < 1 or n_init < 1 or tol < 0 or reg_covar < 0:
            raise ValueError("invalid GaussianMixture parameter")
        self.n_components = int(n_components)
        self.covariance_type = covariance_type
        self.tol, self.reg_covar, self.max_iter, self.n_init = float(tol), float(reg_covar), int(max_iter), int(n_init)
        self.weights_init, self.means_init, self.precisions_init = weights_init, means_init, precisions_init
        self.random_state = random_state
        if init_params not in ('kmeans', 'random_from_data'):
            raise ValueError("init_params must be 'kmeans' or 'random_from_data'")
        self.init_params = init_params
        self._ctx = ctx

    def __getstate__(self):  # picklable like the sklearn object it stands in for (GMM_UBM.py:173-179): no device handles
        st = dict(self.__dict__)
        st["_ctx"] = None
        return st

    # ---- sklearn's M step (mixture/_gaussian_mixture.py:_estimate_gaussian_parameters, _m_step), float64
    def _m_step(self, st, n):
        nk = st["nk"] + 10 * np.finfo(np.float64).eps
        means = st["sx"] / nk[:, None]
        avg_X2 = st["sxx"] / nk[:, None]
        covars = avg_X2 - 2 * (means * st["sx"] / nk[:, None]) + means ** 2 + self.reg_covar
        weights = nk / n
        return weights / weights.sum(), means, covars

    def _rows(self, X, idx):
        return np.asarray(X[idx].cpu() if api._is_torch(X) else X[idx], dtype=np.float64)

    def _kmeans(self, ctx, X, n, D, rng, gvar):
        """k-means++ seeds (host, on <= 20000 sampled frames) + Lloyd on the GPU; returns the clusters' (nk, sx, sxx)."""
        K = self.n_components
        sub = self._rows(X, np.sort(rng.choice(n, size=min(n, max(20000, 50 * K)), replace=False)))
        centres = np.empty((K, D))
        centres[0] = sub[rng.randint(len(sub))]
        d2 = ((sub - centres[0]) ** 2).sum(1)
        sq = (sub * sub).sum(1)
        for k in range(1, K):  # D^2 sampling, best of 2 + log K candidates (sklearn's kmeans_plusplus)
            cand = np.searchsorted(np.cumsum(d2), rng.uniform(size=2 + int(np.log(K))) * d2.sum())
            cand = np.clip(cand, 0, len(sub) - 1)
            # |s - c|^2 = |s|^2 + |c|^2 - 2 s.c as one small matrix product (the broadcast difference cost 0.5 s at K = 64)
            dc = np.maximum(sq[:, None] + sq[cand][None, :] - 2.0 * (sub @ sub[cand].T), 0.0)
            pot = np.minimum(d2[:, None], dc).sum(0)
            b = int(np.argmin(pot))
            centres[k] = sub[cand[b]]
            d2 = np.minimum(d2, dc[:, b])
        tau = np.full((K, D), max(1e-2 * float(gvar.mean()), 1e-12))  # hard-assignment limit of the E step
        w = np.full(K, 1.0 / K)
        st = None
        for _ in range(30):
            st = api.gmm_em_stats(ctx, w, centres, tau, X)
            nk = np.maximum(st["nk"], 1e-12)
            new = np.where(st["nk"][:, None] > 0.5, st["sx"] / nk[:, None], centres)  # an empty cluster keeps its centre
            shift = float(((new - centres) ** 2).sum())
            centres = new
            if shift <= 1e-4 * float(gvar.sum()):
                break
        return st, centres

    def _initial(self, ctx, X, n, D, rng):
        K = self.n_components
        need = self.means_init is None or self.precisions_init is None or self.weights_init is None
        gvar = None
        st = centres = None
        if need:
            # global variance through the same kernels: one component with unit responsibilities
            g = api.gmm_em_stats(ctx, np.ones(1), np.zeros((1, D)), np.ones((1, D)), X)
            mu = g["sx"][0] / n
            gvar = np.maximum(g["sxx"][0] / n - mu * mu, 0.0) + self.reg_covar
            if self.init_params == 'kmeans' and K > 1:
                st, centres = self._kmeans(ctx, X, n, D, rng, gvar)
        if self.means_init is not None:
            means = np.array(self.means_init, dtype=np.float64).reshape(K, D)
        elif centres is not None:
            means = centres
        elif K == 1:
            means = (g["sx"][0] / n)[None]
        else:
            means = self._rows(X, np.sort(rng.choice(n, size=K, replace=False)))
        if self.precisions_init is not None:
            covars = 1.0 / np.array(self.precisions_init, dtype=np.float64).reshape(K, D)
        elif st is not None:  # per-cluster variances (sklearn: _estimate_gaussian_covariances_diag of the one-hot resp)
            nk = st["nk"] + 10 * np.finfo(np.float64).eps
            covars = np.maximum(st["sxx"] / nk[:, None] - (st["sx"] / nk[:, None]) ** 2, 0.0) + self.reg_covar
            covars = np.where(st["nk"][:, None] > 1.5, covars, gvar[None])
        else:
            covars = np.tile(gvar, (K, 1))
        if self.weights_init is not None:
            weights = np.array(self.weights_init, dtype=np.float64).reshape(K)
        elif st is not None:
            weights = np.maximum(st["nk"], 1.0) / np.maximum(st["nk"], 1.0).sum()
        else:
            weights = np.full(K, 1.0 / K)
        return weights, means, covars

    def fit(self, X, y=None):
        """EM until |delta lower bound| < tol or max_iter (sk:mixture/_base.py fit_predict), best of n_init starts."""
        ctx = self._ctx or api.default_context()
        if not api._is_torch(X):
            X = np.ascontiguousarray(X, dtype=np.float32)
        if X.ndim != 2:
            raise ValueError("X must be (n_samples, n_features)")
        n, D = int(X.shape[0]), int(X.shape[1])
        if n < self.n_components:
            raise ValueError("Expected n_samples >= n_components but got n_components = %d, n_samples = %d"
                             % (self.n_components, n))
        if not api._is_torch(X):  # one upload for every EM / Lloyd iteration instead of one per ssp_gmm_em_stats call
            import torch
            X = torch.from_numpy(X).to("cuda:%d" % ctx.device)
            torch.cuda.synchronize()
        rng = np.random.RandomState(self.random_state) if not isinstance(self.random_state, np.random.RandomState) else self.random_state
        best = None
        for _ in range(self.n_init):
            weights, means, covars = self._initial(ctx, X, n, D, rng)
            lower, converged, n_iter = -np.inf, False, 0
            for n_iter in range(1, self.max_iter + 1):
                prev = lower
                st = api.gmm_em_stats(ctx, weights, means, covars, X)
                weights, means, covars = self._m_step(st, n)
                lower = st["loglik_sum"] / n
                if abs(lower - prev) < self.tol:
                    converged = True
                    break
            if best is None or lower > best[0]:
                best = (lower, weights, means, covars, n_iter, converged)
        self.lower_bound_, self.weights_, self.means_, self.covariances_, self.n_iter_, self.converged_ = best
        self.precisions_cholesky_ = 1.0 / np.sqrt(self.covariances_)
        self.precisions_ = self.precisions_cholesky_ ** 2
        return self

    # ---- scoring through the MFMA scorer (api.GmmScorer), same numbers as sklearn's methods
    def _scorer(self):
        ctx = self._ctx or api.default_context()
        return api.GmmScorer(ctx, self.weights_[None], self.means_[None], self.covariances_[None], has_ubm=False), ctx

    def score_samples(self, X):
        sc, ctx = self._scorer()
        X = np.ascontiguousarray(X, dtype=np.float32)
        seg = api.Segments.from_lengths(ctx, [X.shape[0]])
        return np.asarray(sc.score(X, seg, loglik=True, scores=False, argmax=False)["loglik"], dtype=np.float64)[0]

    def score(self, X, y=None):
        sc, ctx = self._scorer()
        X = np.ascontiguousarray(X, dtype=np.float32)
        seg = api.Segments.from_lengths(ctx, [X.shape[0]])
        return float(np.asarray(sc.score(X, seg, loglik=False, scores=True, argmax=False)["scores"])[0, 0])
