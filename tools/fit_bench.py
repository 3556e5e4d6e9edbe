"""Wall time of gmm_train.GaussianMixture.fit on host (numpy) features: EM with the features uploaded once."""
import sys, time, numpy as np
sys.path.insert(0, '.')
from speech_signal_processing_amd.gmm_train import GaussianMixture
rng = np.random.default_rng(0)
K, D, n = 64, 39, 1000000
mu = rng.standard_normal((K, D)) * 2
X = (mu[rng.integers(0, K, n)] + rng.standard_normal((n, D))).astype(np.float32)
for init in ("random_from_data", "kmeans"):
    g = GaussianMixture(n_components=K, random_state=0, init_params=init, max_iter=20, tol=0.0)
    g.fit(X[:50000])
    t0 = time.perf_counter(); g.fit(X); dt = time.perf_counter() - t0
    print("fit K=%d n=%d init=%s: %.1f ms for %d EM iterations (%.2f ms each incl. init)" % (K, n, init, dt * 1e3, g.n_iter_, dt * 1e3 / g.n_iter_))
from speech_signal_processing_amd import api
ctx = api.default_context()
w = np.full(K, 1.0 / K); cov = np.ones((K, D))
api.gmm_em_stats(ctx, w, mu, cov, X)
t0 = time.perf_counter(); api.gmm_em_stats(ctx, w, mu, cov, X); dt = time.perf_counter() - t0
print("one ssp_gmm_em_stats call on the same HOST array (what every iteration cost before): %.1f ms" % (dt * 1e3))
