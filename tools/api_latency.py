"""Per-call latency of the reference-shaped single-utterance APIs (the reference calls them in Python loops)."""
import sys, time, numpy as np
sys.path.insert(0, '.')
from speech_signal_processing_amd.utils import processing as P
from speech_signal_processing_amd import GMM_UBM, MFCC_DTW, d_vector, api
from speech_signal_processing_amd.gmm_train import GaussianMixture
rng = np.random.default_rng(0)
x8 = (0.3 * rng.standard_normal(24000)).astype(np.float32)
x16 = (0.3 * rng.standard_normal(48000)).astype(np.float32)
feat = rng.standard_normal((298, 13))
def t(name, fn, n=30, warm=40):
    # (40 untimed calls first: the first few dozen calls of an entry point on a context that has already run larger batches cost
    #  about 1 ms each -- GMM_UBM.delta showed it -- and then settle; the table quotes the settled figure)
    for _ in range(warm): fn()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    print("%-52s %8.3f ms / call" % (name, (time.perf_counter() - t0) / n * 1e3))
t("utils.processing.MFCC(3 s @ 8 kHz)", lambda: P.MFCC(x8))
t("utils.processing.MFCC(frameSize=400, step=160)", lambda: P.MFCC(x8, 8000, 400, 160))
t("utils.processing.enframe", lambda: P.enframe(x8))
t("MFCC_DTW._MFCC", lambda: MFCC_DTW._MFCC(x8))
t("MFCC_DTW.MFCC_lib", lambda: MFCC_DTW.MFCC_lib(x8))
t("GMM_UBM.mfcc (sidekit)", lambda: GMM_UBM.mfcc(x16))
t("GMM_UBM.plp (sidekit)", lambda: GMM_UBM.plp(x16))
t("GMM_UBM.delta", lambda: GMM_UBM.delta(feat))
t("GMM_UBM.extract_feature([1 utt])", lambda: GMM_UBM.extract_feature([x16], [0]))
t("GMM_UBM.extract_feature([64 utt])", lambda: GMM_UBM.extract_feature([x16] * 64, [0] * 64), n=5, warm=5)
g = GaussianMixture(n_components=16, random_state=0).fit(rng.standard_normal((5000, 26)).astype(np.float32))
f26 = rng.standard_normal((298, 26)).astype(np.float32)
t("GaussianMixture.score (298 x 26, K=16)", lambda: g.score(f26))
t("GMM_UBM.score_matrix(10 models, 1 utt)", lambda: GMM_UBM.score_matrix([g] * 10, g, [f26]))
t("GMM_UBM.score_matrix(10 models, 100 utt)", lambda: GMM_UBM.score_matrix([g] * 10, g, [f26] * 100), n=5, warm=5)
a, b = rng.standard_normal(1222), rng.standard_normal(1100)
t("MFCC_DTW.distance_dtw (1222 x 1100)", lambda: MFCC_DTW.distance_dtw(a, b))
C = rng.standard_normal((50, 256)).astype(np.float32); Xe = rng.standard_normal((1, 256)).astype(np.float32)
t("d_vector.cosine_scores(1 x 50)", lambda: d_vector.cosine_scores(Xe, C))
