"""GPU-backed mirror of the inference half of the reference's ``d_vector.py``: Data_gen's feature front end
(d_vector.py:80-98), the forward pass of the fully connected speaker network (``DenseNet.predict`` = spkModel.predict of
the Sequential built at d_vector.py:171-189) and nn_model.test / enroll / eval (d_vector.py:296-361).  Training the
Keras networks (and the GRU / LSTM variants) is out of scope: weights are inputs."""
from __future__ import annotations

import functools
import os
import pickle as pkl

import numpy as np

from . import api, frontend
from .GMM_UBM import delta as _delta


def cosine_scores(X, centroids):
    """(N, S) float64 matrix of scipy.spatial.distance.cosine(X[i], centroids[j]) — d_vector.py:315-318."""
    r = api.cosine_identify(api.default_context(), X, np.asarray(centroids, dtype=np.float32), dist=True, argmin=False, minval=False)
    return np.asarray(r["dist"], dtype=np.float64)


def identify(X, centroids, precision=0):
    """argmin_j cosine(X[i], centroids[j]) (first index on ties) — d_vector.py:319.  precision 1 / 2: the split-precision sweeps of
    ssp_cosine_identify2 (the fp32 path's index on every row through proven error bands, 3 - 5 x faster; embeddings of at most 256 dims)."""
    r = api.cosine_identify(api.default_context(), X, np.asarray(centroids, dtype=np.float32), dist=False, argmin=True, minval=False,
                            precision=int(precision))
    return np.asarray(r["argmin"]).astype(np.int64)


class DenseNet:
    """Forward pass of the reference's fully connected d-vector network (d_vector.py:171-189): Dense(256)+ReLU x 3, then
    Dense(256) — what ``load_model('feature/d_vector/d_vector_nn.h5').predict`` computes (dropout is the identity at
    inference).  ``layers``: list of (kernel (d_in, units), bias (units,) or None, activation in {'relu', 'linear', None}) in
    Keras' own layout, e.g. ``[(l.get_weights()[0], l.get_weights()[1], 'relu') ...]``.  Weights go to the GPU once."""

    def __init__(self, layers, device: int = 0):
        import torch
        self._ctx = api.default_context(device, torch_stream=True)
        self.layers = []
        d_prev = None
        for W, b, act in layers:
            W = np.asarray(W, dtype=np.float32)
            if W.ndim != 2 or (d_prev is not None and W.shape[0] != d_prev):
                raise ValueError("layer kernels must be (d_in, units) and chain")
            if act not in ('relu', 'linear', None):
                raise ValueError("activation must be 'relu' or 'linear'")
            d_prev = W.shape[1]
            Wt = torch.from_numpy(np.ascontiguousarray(W.T)).cuda(device)
            bt = None if b is None else torch.from_numpy(np.asarray(b, dtype=np.float32).reshape(-1)).cuda(device)
            self.layers.append((Wt, bt, act == 'relu'))
        self.input_dim = int(self.layers[0][0].shape[1])
        self.output_dim = int(d_prev)
        # the whole network as one packed object: hidden / output layers chained in registers (ssp_dnn)
        self._net = api.DnnForward(self._ctx, [(np.ascontiguousarray(np.asarray(W, dtype=np.float32).T),
                                                None if b is None else np.asarray(b, dtype=np.float32), act == 'relu') for W, b, act in layers])

    def predict(self, X, batch_size=None):
        """X (N, input_dim) numpy or torch CUDA tensor -> (N, output_dim) of the same kind (numpy in: float32 out)."""
        import torch
        is_t = api._is_torch(X)
        h = X if is_t else torch.from_numpy(np.ascontiguousarray(X, dtype=np.float32)).cuda(self.layers[0][0].device)
        if h.ndim != 2 or h.shape[1] != self.input_dim:
            raise ValueError("X must be (N, %d)" % self.input_dim)
        h = self._net.forward(h)
        return h if is_t else h.cpu().numpy()


# ---- model registry: the reference addresses its networks by name — load_model('feature/d_vector/d_vector_{}.h5'.format(model_name)),
# d_vector.py:297,329,347.  Keras / h5py are not part of this path: a DenseNet is registered under the name (or saved next to where
# the .h5 would be, as d_vector_{name}.npz) and `model_name=` resolves to it.
_MODELS = {}
MODEL_DIR = os.path.join('feature', 'd_vector')


def register_model(name, net):
    """Make ``net`` (a DenseNet, or any object with .predict) the model that ``model_name=name`` refers to."""
    _MODELS[str(name)] = net


def save_model(net: "DenseNet", name, model_dir=None):
    """Store a DenseNet's weights as {model_dir}/d_vector_{name}.npz (the .h5's place, d_vector.py:297)."""
    model_dir = MODEL_DIR if model_dir is None else model_dir
    os.makedirs(model_dir, exist_ok=True)
    arrs = {}
    for i, (Wt, bt, relu) in enumerate(net.layers):
        arrs["W%d" % i] = Wt.cpu().numpy().T
        arrs["b%d" % i] = np.zeros(0, np.float32) if bt is None else bt.cpu().numpy()
        arrs["a%d" % i] = np.array(1 if relu else 0)
    np.savez(os.path.join(model_dir, "d_vector_%s.npz" % name), **arrs)


def load_model(name, model_dir=None):
    """The network ``model_name`` refers to: a registered one, else {model_dir}/d_vector_{name}.npz; OSError when neither exists (as
    keras.models.load_model raises for a missing file)."""
    name = str(name)
    if name in _MODELS:
        return _MODELS[name]
    path = os.path.join(MODEL_DIR if model_dir is None else model_dir, "d_vector_%s.npz" % name)
    if not os.path.exists(path):
        raise OSError("no d-vector model %r: register_model(%r, net) or save one as %s" % (name, name, path))
    z = np.load(path)
    n = len([k for k in z.files if k.startswith("W")])
    net = DenseNet([(z["W%d" % i], z["b%d" % i] if z["b%d" % i].size else None, 'relu' if int(z["a%d" % i]) else 'linear') for i in range(n)])
    _MODELS[name] = net
    return net


_UNSET = object()


def _resolve_model(model_name, spk_model, default_name):
    """spk_model= (an object) wins; an explicit model_name= must resolve (OSError otherwise, like the reference); with neither given the
    reference's default name is tried and, when no such model exists, the inputs are taken to be embeddings already."""
    if spk_model is not None:
        return spk_model
    if model_name is _UNSET:
        try:
            return load_model(default_name)
        except OSError:
            return None
    if model_name is None:
        return None
    return model_name if hasattr(model_name, "predict") else load_model(model_name)


@functools.lru_cache(maxsize=8)
def _feature_plan(feature_type, sample_rate):
    if feature_type == 'PLP':  # d_vector.py:92-93: plp(x, fs)[0]
        return api.MfccPlan(api.default_context(), frontend.preset_sidekit_plp(fs=sample_rate))
    if feature_type != 'MFCC':
        raise NameError  # d_vector.py:94-95
    return api.MfccPlan(api.default_context(), frontend.preset_sidekit(fs=sample_rate))


class Data_gen:
    """Feature front end of d_vector.Data_gen: 1-second chunks -> sidekit mfcc(x, fs)[0] -> (98, 13) (d_vector.py:80-98)."""

    def __init__(self, sample_rate=16000):
        self.sample_rate = sample_rate
        self.path = None  # the dataset directory of _load() (the reference's load_data sets it)

    @staticmethod
    def delta(feat, N=2):
        """d_vector.py:143-160 (identical to GMM_UBM.delta)."""
        return _delta(feat, N)

    def _plan(self, feature_type):
        return _feature_plan(feature_type, int(self.sample_rate))  # (keyed on the rate too: _load() takes it from the files)

    def _load(self):
        """d_vector.py:34-57 — every wav under ``self.path/<speaker>/<dir>/``: (list of audio arrays, list of speaker names); the sample
        rate of the files becomes ``self.sample_rate``."""
        from .utils.tools import read
        x, y = [], []
        for speaker in os.listdir(self.path):
            path1 = os.path.join(self.path, speaker)
            for _dir in os.listdir(path1):
                path2 = os.path.join(path1, _dir)
                for _wav in os.listdir(path2):
                    self.sample_rate, audio = read(os.path.join(path2, _wav))
                    y.append(speaker)
                    x.append(audio)
        return x, y

    @staticmethod
    def save(data, file_name):
        """d_vector.py:133-136"""
        with open('feature/{}.pkl'.format(file_name), 'wb') as f:
            pkl.dump(data, f)

    @staticmethod
    def load(file_name):
        """d_vector.py:138-141"""
        with open('feature/{}.pkl'.format(file_name), 'rb') as f:
            return pkl.load(f)

    def extract_feature(self, *args, feature_type='MFCC', datatype='dev'):
        """Two call shapes.  The reference's (d_vector.py:59-118): ``extract_feature(feature_type='MFCC', datatype='dev')`` — audio from
        ``self._load()`` (``self.path`` set by the caller, as ``load_data`` does), features cached under ``feature/<datatype>_<type>_*.pkl``
        exactly as the reference caches them.  And the in-memory one: ``extract_feature(x, y, feature_type='MFCC')`` with x a list of audio
        arrays and y their labels.  Either way every audio is cut into 1 s chunks (d_vector.py:80-83), all chunks go through ONE kernel
        launch, and chunks whose features contain NaN are dropped (d_vector.py:97-98).  Returns (feature, label)."""
        if args and not isinstance(args[0], str):
            if len(args) < 2 or len(args) > 3:
                raise TypeError("extract_feature(x, y, feature_type='MFCC') or extract_feature(feature_type='MFCC', datatype='dev')")
            if len(args) == 3:
                feature_type = args[2]
            return self._extract(args[0], args[1], feature_type)
        if len(args) > 2:
            raise TypeError("extract_feature(feature_type='MFCC', datatype='dev')")
        if len(args) >= 1:
            feature_type = args[0]
        if len(args) == 2:
            datatype = args[1]
        if not os.path.exists('feature'):
            os.mkdir('feature')
        if os.path.exists('feature/{}_{}_feature.pkl'.format(datatype, feature_type)):
            return self.load('{}_{}_feature'.format(datatype, feature_type)), self.load('{}_{}_label'.format(datatype, feature_type))
        x, y = self._load()
        feature, label = self._extract(x, y, feature_type)
        self.save(feature, '{}_{}_feature'.format(datatype, feature_type))
        self.save(label, '{}_{}_label'.format(datatype, feature_type))
        return feature, label

    def _extract(self, x, y, feature_type='MFCC'):
        plan = self._plan(feature_type)
        sr = self.sample_rate
        chunks, labels = [], []
        for xi, yi in zip(x, y):
            xi = np.asarray(xi).reshape(-1)   # (int16 PCM stays int16: api.flatten_signals below)
            for j in range(xi.shape[0] // sr):
                chunks.append(xi[j * sr:(j + 1) * sr])
                labels.append(yi)
        if not chunks:
            return [], []
        seg = api.Segments.from_lengths(plan.ctx, [sr] * len(chunks))
        fseg = plan.frame_segments(seg)
        feats = plan.run(api.flatten_signals(chunks)[0], seg, fseg)
        if feature_type == 'PLP':
            feats = api.plp_post(plan.ctx, feats, fseg, sr / 2.0)
        feats = np.asarray(feats, dtype=np.float64)
        feature, label = [], []
        for i, lab in enumerate(labels):
            f = feats[fseg.offsets[i]:fseg.offsets[i + 1]]
            if np.isnan(f).sum() > 0:
                continue
            feature.append(f)
            label.append(lab)
        return feature, label


class nn_model:
    """Inference half of d_vector.nn_model.  ``X_*`` are embeddings, or network inputs when ``spk_model`` (a DenseNet, the
    stand-in for load_model('feature/d_vector/d_vector_{}.h5')) is given.  ``store``: path of the enrolment dictionary pickle
    (the reference always uses 'feature/d_vector/d_vector.pkl', d_vector.py:333-344,350-351); None keeps it in memory."""

    def __init__(self, n_class=40, store=None):
        # (n_class: the reference's only constructor argument, d_vector.py:165 — the width of the training network's softmax; the inference
        #  half kept here never reads it, but nn_model(n_class=40) and nn_model(40) must construct)
        if isinstance(n_class, (str, os.PathLike)) and store is None:  # (round 4's signature: nn_model('path.pkl'))
            n_class, store = 40, n_class
        self.n_class = n_class
        self.store = store
        self.d_vector = {}  # name -> mean embedding, the dict the reference pickles (d_vector.py:333-344)

    def _load(self):
        if self.store is not None:
            try:
                with open(self.store, 'rb') as f:
                    self.d_vector = pkl.load(f)
            except Exception:  # d_vector.py:336-337: a missing / unreadable file starts an empty dictionary
                self.d_vector = {}

    def _save(self):
        if self.store is not None:
            d = os.path.dirname(self.store)
            if d and not os.path.exists(d):
                os.makedirs(d)
            with open(self.store, 'wb') as f:
                pkl.dump(self.d_vector, f)

    def test(self, X_train, Y_train, X_val, Y_val, model_name=_UNSET, spk_model=None):
        """d_vector.py:296-320, same positional signature: X = spkModel.predict(X) with the network ``model_name`` names (default
        'nn'; see _resolve_model), per-speaker centroids of X_train (one-hot Y_train), cosine distance of every X_val row to every
        centroid, accuracy of the arg-min.  The centroids stay available as ``self.centroids_`` (float64 (num, d) like the reference's
        ``avg``, d_vector.py:310)."""
        spk_model = _resolve_model(model_name, spk_model, 'nn')
        if spk_model is not None:
            X_train, X_val = spk_model.predict(X_train), spk_model.predict(X_val)
        num = Y_train.shape[1]
        lab = np.argmax(Y_train, axis=1)  # decoding the one-hot labels is index bookkeeping, not arithmetic
        avg = np.asarray(api.centroids(api.default_context(), np.asarray(X_train, dtype=np.float32), lab, num))
        self.centroids_ = avg.astype(np.float64)
        pred = identify(np.asarray(X_val, dtype=np.float32), avg)
        return (np.argmax(Y_val, axis=1) == pred).sum() / X_val.shape[0]

    def enroll(self, X_train, name, model_name=_UNSET, spk_model=None):
        """d_vector.py:322-344, same positional signature (default model 'lstm'): store the mean embedding under ``name`` (overwrites,
        like the reference)."""
        spk_model = _resolve_model(model_name, spk_model, 'lstm')
        if spk_model is not None:
            X_train = spk_model.predict(X_train)
        self._load()
        if name in self.d_vector:
            print("sample already exists")
        X = np.asarray(X_train, dtype=np.float32)
        self.d_vector[name] = np.asarray(api.centroids(api.default_context(), X, np.zeros(len(X), np.int32), 1))[0]
        self._save()

    def eval(self, target, model_name=_UNSET, spk_model=None):
        """d_vector.py:346-361, same positional signature (default model 'lstm'): the distances to every enrolled vector come from the
        GPU scorer; the decision is the reference's own scan in dict order — the running minimum starts at 1 and only a strictly
        smaller distance replaces it, so a NaN distance (zero-norm embedding or enrolment) is never selected and ties keep the first
        name.  Returns the name or None."""
        spk_model = _resolve_model(model_name, spk_model, 'lstm')
        if spk_model is not None:
            target = spk_model.predict(np.asarray(target, dtype=np.float32).reshape(1, -1))
        self._load()
        if not self.d_vector:
            return None
        names = list(self.d_vector.keys())
        C = np.stack([self.d_vector[n] for n in names]).astype(np.float32)
        r = api.cosine_identify(api.default_context(), np.asarray(target, dtype=np.float32).reshape(1, -1), C,
                                dist=True, argmin=False, minval=False)
        dist = np.asarray(r["dist"], dtype=np.float64).reshape(-1)
        min_distance, target_name = 1, None
        for name, dv in zip(names, dist):
            if min_distance > dv:
                min_distance, target_name = dv, name
        return target_name
