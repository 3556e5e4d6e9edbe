"""ORACLE — TEST INFRASTRUCTURE ONLY.

CPU restatement (numpy / scipy / scikit-learn, float64) of the reference hot path
MFCC -> GMM-UBM log-likelihood scoring / d-vector cosine scoring.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this package, and only as the *checker*.  The product
package ``speech_signal_processing_amd`` never imports it and has no CPU
fallback: it fails loudly when the HIP library is missing.
"""
