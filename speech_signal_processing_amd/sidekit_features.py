"""GPU-backed stand-ins for the two sidekit feature functions the reference imports into GMM_UBM.py / d_vector.py / the GUIs
(``from sidekit.frontend.features import plp, mfcc``, GMM_UBM.py:20, d_vector.py:18).  Same call shape, same return value: a list
whose first item is the (frames, 13) cepstra - the only item the reference uses.  sidekit's source is absent from the reference
tree; both follow its published algorithms (parity unpinned, see oracle/ref_cpu.py)."""
from __future__ import annotations

import functools

import numpy as np

from . import api, frontend


@functools.lru_cache(maxsize=8)
def _mfcc_plan(fs, nwin, shift, nceps, prefac):
    return api.MfccPlan(api.default_context(), frontend.preset_sidekit(fs=fs, nwin=nwin, shift=shift, nceps=nceps, prefac=prefac))


@functools.lru_cache(maxsize=8)
def _plp_plan(fs, nwin, shift, prefac):
    return api.MfccPlan(api.default_context(), frontend.preset_sidekit_plp(fs=fs, nwin=nwin, shift=shift, prefac=prefac))


def mfcc(input_sig, fs=16000, nwin=0.025, shift=0.01, nceps=13, prefac=0.97):
    """sidekit mfcc at the reference's call sites (GMM_UBM.py:89, d_vector.py:91): [cepstra (T, nceps), None, None, None]."""
    plan = _mfcc_plan(int(fs), float(nwin), float(shift), int(nceps), float(prefac))
    x, lens = api.flatten_signals([input_sig])
    seg = api.Segments.from_lengths(plan.ctx, lens)
    return [np.asarray(plan.run(x, seg), dtype=np.float64), None, None, None]


def plp_batch(signals, fs=16000, nwin=0.025, shift=0.01, plp_order=13, prefac=0.97, rasta=True):
    """PLP cepstra of a list of utterances in two launches: -> (feats (sum T_i, plp_order) float32 array, frame Segments)."""
    plan = _plp_plan(int(fs), float(nwin), float(shift), float(prefac))
    flat, lens = api.flatten_signals(signals)
    seg = api.Segments.from_lengths(plan.ctx, lens)
    fseg = plan.frame_segments(seg)
    logspec = plan.run(flat, seg, fseg)
    return api.plp_post(plan.ctx, logspec, fseg, fs / 2.0, plp_order, rasta), fseg


def plp(input_sig, nwin=0.025, fs=16000, plp_order=13, shift=0.01, get_spec=False, get_mspec=False, prefac=0.97, rasta=True):
    """sidekit plp (call sites GMM_UBM.py:95, d_vector.py:93, UI/GMM_UBM_GUI.py:93): [cepstra (T, plp_order), None, None, None].
    (log-energy / spectra, items 1-3 of sidekit's list, are not used by the reference and not computed.)"""
    if get_spec or get_mspec:
        raise NotImplementedError("plp: get_spec / get_mspec are not used by the reference and not provided")
    feats, _ = plp_batch([input_sig], fs, nwin, shift, plp_order, prefac, rasta)
    return [np.asarray(feats, dtype=np.float64), None, None, None]
