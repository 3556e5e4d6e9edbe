"""Algorithmic FLOP count of one MFCC frame — a COUNT of the algorithm's real additions and multiplications, not a counter reading.

    python tools/flop_count.py [--win 400 --nfft 512 --nfilt 24 --nceps 13 --order 2 --preemph 1] [--json]

The transform is counted by running the factorisation symbolically: every complex value knows whether it is structurally zero (the
zero padding of a 400-sample window inside the 512-point transform), every constant whether it is trivial (+-1, +-i: free), an
eighth root ((+-1 +- i) / sqrt 2: two multiplications + two additions) or general (four multiplications + two additions).  Operations
on structural zeros are free, so the pruning of the padded transform is counted, not assumed.  The factorisation is the standard one for
a real input of even length: N-point real DFT = N/2-point complex DFT of z[n] = x[2n] + i x[2n+1] (radix 4 x 4 x 4 x 4 for N/2 = 256,
any power of two by radix-4 / radix-2 steps) + the split step.  The total uses the smaller of that count and the best published one
for the un-pruned transform (real split-radix, Sorensen et al. 1987): the bound is not to be flattered by the kernel's own factorisation.

Two figures per stage:
  flop  real additions + real multiplications (an FMA counts 2)
  ops   the fewest add / mul / fma lane-operations that perform them (a multiplication whose only consumer is an addition fuses)
The vector ALU's peak is one packed-FMA lane pair per cycle: 2 ops = 4 flop per lane-cycle = 157.3 TFLOP/s on 256 CUs at 2.4 GHz
(MI355X_MICROARCH.md); code whose operations are not all FMAs can reach at most  flop / ops / 2  of it — the mix-weighted peak.
bench.py prints both next to the HBM roofline (`roofline_flop`).
"""
from __future__ import annotations

import argparse
import cmath
import json
import math


class Count:
    def __init__(self):
        self.add = 0
        self.mul = 0
        self.fma = 0  # fused pairs (each also counted once in add and once in mul)

    @property
    def flop(self):
        return self.add + self.mul

    @property
    def ops(self):
        return self.add + self.mul - self.fma

    def take(self):
        out = (self.add, self.mul, self.fma)
        return out


class Z:
    """a complex value: only whether it is structurally zero, and whether it is purely real, is tracked"""
    __slots__ = ("zero", "real")

    def __init__(self, zero=False, real=False):
        self.zero = zero
        self.real = real


def cadd(c: Count, a: Z, b: Z) -> Z:
    if a.zero:
        return b
    if b.zero:
        return a
    if a.real and b.real:
        c.add += 1
        return Z(real=True)
    # (a real plus a complex: only the real parts meet)
    c.add += 1 if (a.real or b.real) else 2
    return Z()


def classify(w: complex):
    r, i = round(w.real, 12), round(w.imag, 12)
    if (abs(r), abs(i)) in ((1.0, 0.0), (0.0, 1.0)):
        return "trivial"
    if abs(abs(r) - abs(i)) < 1e-12:
        return "eighth"
    return "general"


def cmul_const(c: Count, a: Z, w: complex) -> Z:
    if a.zero:
        return a
    k = classify(w)
    if k == "trivial":
        return Z(real=a.real and abs(w.imag) < 1e-12)
    if a.real:  # real times complex constant: two multiplications
        c.mul += 2
        return Z()
    if k == "eighth":  # ((x - y) + i (x + y)) / sqrt 2
        c.add += 2
        c.mul += 2
        return Z()
    c.mul += 4
    c.add += 2
    c.fma += 2  # x wr - y wi, x wi + y wr: one plain multiplication + one fused pair each
    return Z()


def dft(c: Count, x: list) -> list:
    """decimation-in-time, radix 4 while the length allows it, else radix 2; structural zeros propagate"""
    n = len(x)
    if n == 1:
        return x
    r = 4 if n % 4 == 0 else 2
    m = n // r
    subs = [dft(c, x[k::r]) for k in range(r)]
    out = [None] * n
    for k in range(m):
        t = [cmul_const(c, subs[q][k], cmath.exp(-2j * math.pi * q * k / n)) for q in range(r)]
        if r == 2:
            out[k] = cadd(c, t[0], t[1])
            out[k + m] = cadd(c, t[0], t[1])  # (a subtraction: same cost)
        else:
            # radix-4 butterfly: 8 complex additions, the multiplications by -i are free
            a0, a1 = cadd(c, t[0], t[2]), cadd(c, t[0], t[2])
            b0, b1 = cadd(c, t[1], t[3]), cadd(c, t[1], t[3])
            out[k] = cadd(c, a0, b0)
            out[k + m] = cadd(c, a1, b1)
            out[k + 2 * m] = cadd(c, a0, b0)
            out[k + 3 * m] = cadd(c, a1, b1)
    return out


def real_dft(c: Count, n_fft: int, win: int):
    """N-point DFT of a real frame of `win` samples (zero padded): N/2-point complex DFT + split step; returns bins 0 .. N/2"""
    h = n_fft // 2
    z = []
    for n in range(h):
        lo, hi = 2 * n < win, 2 * n + 1 < win
        z.append(Z(zero=not lo, real=lo and not hi))
    zf = dft(c, z)
    # split: X[k] = E[k] + W^k O[k], E = (Z[k] + conj Z[h-k]) / 2, O = -i (Z[k] - conj Z[h-k]) / 2  (the halves fold into the filterbank)
    for k in range(1, h // 2):
        # one pair of bins (k, h - k): E (2 adds), D (2 adds), -i D W (complex multiplication), E +- O (4 adds)
        c.add += 4
        cmul_const(c, Z(), cmath.exp(-2j * math.pi * k / n_fft))
        c.add += 4
    c.add += 2  # bins 0 and N/2: re + im, re - im; bin N/4: a conjugation
    return h + 1


def count(win=400, n_fft=512, n_filt=24, fb_nonzero=454, n_ceps=13, order=2, delta_N=2, preemph=True, power=2):
    stages = {}

    def stage(name, fn):
        c = Count()
        fn(c)
        stages[name] = {"flop": c.flop, "ops": c.ops, "add": c.add, "mul": c.mul}

    def s_pre(c):
        if preemph:  # y[n] = x[n] - a x[n-1]: one fused pair per sample
            c.add += win
            c.mul += win
            c.fma += win
    stage("pre-emphasis", s_pre)
    stage("window", lambda c: setattr(c, "mul", c.mul + win))
    nb = [0]
    stage("real FFT (pruned, N/2 complex + split)", lambda c: nb.__setitem__(0, real_dft(c, n_fft, win)))
    nb = nb[0]

    def s_pow(c):  # re^2 + im^2: a multiplication + a fused pair; bins 0 and N/2 are real
        c.mul += 2 * (nb - 2) + 2
        c.add += nb - 2
        c.fma += nb - 2
        if power == 1:
            c.mul += nb  # (a square root counted as one operation)
    stage("power spectrum", s_pow)

    def s_mel(c):  # one fused pair per non-zero weight, the first of each filter a plain multiplication
        c.mul += fb_nonzero
        c.add += fb_nonzero - n_filt
        c.fma += fb_nonzero - n_filt
    stage("filterbank (%d non-zero weights)" % fb_nonzero, s_mel)
    stage("log (one operation per filter)", lambda c: setattr(c, "mul", c.mul + n_filt))

    def s_dct(c):
        c.mul += n_ceps * n_filt
        c.add += n_ceps * (n_filt - 1)
        c.fma += n_ceps * (n_filt - 1)
    stage("DCT-II (%d x %d)" % (n_ceps, n_filt), s_dct)

    def s_delta(c):  # sum_n n (c[t+n] - c[t-n]) / denom per coefficient: N subtractions, N - 1 multiplications (n = 1 is free), N - 1 additions, 1 scale
        per = delta_N + (delta_N - 1) + (delta_N - 1) + 1
        c.add += order * n_ceps * (2 * delta_N - 1)
        c.mul += order * n_ceps * delta_N
        c.fma += order * n_ceps * (delta_N - 1)
        assert per == (2 * delta_N - 1) + delta_N
    stage("delta x %d" % order, s_delta)
    # the transform is priced at the SMALLER of the counted factorisation and the best published count for this length (un-pruned
    # real split-radix, Sorensen et al. 1987 = half of the complex 4 N log2 N - 6 N + 8): the bound must not be flattered by the
    # factorisation the kernel happens to use.  Its add / mul mix is the counted one.
    lg = math.log2(n_fft)
    split_radix_real = (4 * n_fft * lg - 6 * n_fft + 8) / 2
    fkey = "real FFT (pruned, N/2 complex + split)"
    fs = stages[fkey]
    fs["flop_counted_radix4"] = fs["flop"]
    if split_radix_real < fs["flop"]:
        fs["ops"] = int(round(fs["ops"] * split_radix_real / fs["flop"]))
        fs["flop"] = int(split_radix_real)
    tot_f = sum(s["flop"] for s in stages.values())
    tot_o = sum(s["ops"] for s in stages.values())
    return {"stages": stages, "flop_per_frame": tot_f, "ops_per_frame": tot_o, "fma_fraction_of_peak": tot_f / tot_o / 2.0,
            "split_radix_real_fft_unpruned": split_radix_real,
            "cfg": {"win": win, "n_fft": n_fft, "n_filt": n_filt, "fb_nonzero": fb_nonzero, "n_ceps": n_ceps, "order": order, "delta_N": delta_N,
                    "preemph": bool(preemph), "power": power}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--win", type=int, default=400)
    ap.add_argument("--nfft", type=int, default=512)
    ap.add_argument("--nfilt", type=int, default=24)
    ap.add_argument("--fb-nonzero", type=int, default=454)
    ap.add_argument("--nceps", type=int, default=13)
    ap.add_argument("--order", type=int, default=2)
    ap.add_argument("--preemph", type=int, default=1)
    ap.add_argument("--power", type=int, default=2)
    ap.add_argument("--json", action="store_true")
    a = ap.parse_args()
    r = count(a.win, a.nfft, a.nfilt, a.fb_nonzero, a.nceps, a.order, 2, bool(a.preemph), a.power)
    if a.json:
        print(json.dumps(r))
        return
    print("%-46s %8s %8s" % ("stage", "flop", "ops"))
    for k, s in r["stages"].items():
        print("%-46s %8d %8d" % (k, s["flop"], s["ops"]))
    print("%-46s %8d %8d" % ("total per frame", r["flop_per_frame"], r["ops_per_frame"]))
    print("(transform: counted radix-4 factorisation %d flop, published un-pruned real split-radix %d flop; the smaller is used)"
          % (r["stages"]["real FFT (pruned, N/2 complex + split)"]["flop_counted_radix4"], r["split_radix_real_fft_unpruned"]))
    print("mix-weighted share of the packed-FMA peak: %.3f (every operation an FMA would be 1.0)" % r["fma_fraction_of_peak"])


if __name__ == "__main__":
    main()
