"""CPU-only: the argument behind the folded column sums of the normalisation (adapter.hip,
colsum_*_kernel), restated in plain Python integers and checked against the float32 chain it replaces.

numpy's column means are row-ordered float32 sums, s = fl(s + x) row after row.  fl rounds the exact
sum E to a multiple of Q = 2^(floor(log2|E|) - 23), ties to even.  While s and E stay in one binade, Q
is constant, s = T Q with an integer T, and a row only needs the PARITY of T:
    x = a Q + b (0 <= b < Q),  T' = T + a + c,  c = 0 / 1 for b below / above Q/2, parity(T + a) on a tie.
A chunk of rows is then a function parity -> (increment, parity) and chunks compose.  The device builds
that function for the binade a double-precision prefix EXPECTS and applies it only where the true
running sum is in that binade and cannot leave it during the chunk; everything else is added row by
row.  This file is that procedure, slowly."""
import math

import numpy as np
import pytest


def seq_sum32(x):
    s = np.float32(0)
    for v in x:
        s = np.float32(s + v)
    return s


def fexp(v):
    return math.frexp(abs(float(v)))[1] - 1


def elem(bits, kq):
    """(a, cmp) of one float32 (its bits) relative to Q = 2^kq: a = floor(x / Q); cmp = -1 / 0 / +1 for
    b = x - a Q below / equal to / above Q / 2 (-1 also for b = 0)."""
    sign, ef, man = bits >> 31, (bits >> 23) & 0xFF, bits & 0x7FFFFF
    if ef == 0 and man == 0:
        return 0, -1
    m = man | (0x800000 if ef else 0)
    sh = kq - (max(ef, 1) - 127 - 23)
    if sh <= 0:
        a = m << -sh
        return (-a if sign else a), -1
    if sh >= 26:
        return (-1, 1) if sign else (0, -1)
    apos, rem, half = m >> sh, m & ((1 << sh) - 1), 1 << (sh - 1)
    if not sign:
        a, b = apos, rem
    elif rem == 0:
        a, b = -apos, 0
    else:
        a, b = -apos - 1, (1 << sh) - rem
    return a, (-1 if b == 0 or b < half else (0 if b == half else 1))


def chunk_function(bits, kq):
    D, q = [0, 0], [0, 1]
    for xb in bits:
        a, cmp_ = elem(int(xb), kq)
        for p in (0, 1):
            par = q[p] ^ (a & 1)
            c = 0 if cmp_ < 0 else (1 if cmp_ > 0 else par)
            D[p] += a + c
            q[p] = par ^ c
    return D, q


def folded_sum(x, chunk):
    x = np.asarray(x, np.float32)
    S, P = np.float32(0), 0.0
    folded = walked = 0
    for j0 in range(0, len(x), chunk):
        xc = x[j0:j0 + chunk]
        pre = np.concatenate([[0.0], np.cumsum(xc.astype(np.float64))])
        pmin, pmax = pre.min(), pre.max()
        lo, hi = P + pmin, P + pmax
        guess = None
        if np.isfinite(pre).all() and lo != 0 and hi != 0 and (lo > 0) == (hi > 0) and fexp(lo) == fexp(hi):
            guess = fexp(lo)
        P += pre[-1]
        done = False
        Sd = float(S)
        if guess is not None and -100 < guess < 100 and Sd != 0 and math.isfinite(Sd) and fexp(Sd) == guess:
            kq = guess - 23
            Q = math.ldexp(1.0, kq)
            T = int(Sd / Q)
            tlo = math.floor(pmin / Q) - len(xc)
            thi = math.ceil(pmax / Q) + len(xc)
            inside = (T + tlo >= 1 << 23 and T + thi < 1 << 24) if T > 0 else (T + tlo > -(1 << 24) and T + thi <= -(1 << 23))
            if inside:
                D, _ = chunk_function(xc.view(np.uint32), kq)
                T += D[T & 1]
                S = np.float32(T * Q)
                assert float(S) == T * Q
                done = True
        if done:
            folded += 1
        else:
            walked += 1
            for v in xc:
                S = np.float32(S + v)
    return S, folded, walked


CASES = {
    "integers 0..255 (sums cross 2^24: every odd add is a tie)": lambda r, n: r.integers(0, 256, n) * 3,
    "integers 0..65535": lambda r, n: r.integers(0, 65536, n),
    "odd integers": lambda r, n: 2 * r.integers(0, 5000, n) + 1,
    "x.5 values": lambda r, n: r.integers(0, 2000, n) + 0.5,
    "pixel coordinates": lambda r, n: r.uniform(0, 1280, n),
    "angles, a random walk around zero": lambda r, n: r.uniform(-np.pi, np.pi, n),
    "mixed-sign integers": lambda r, n: r.integers(-1000, 1001, n),
    "negative drift": lambda r, n: -r.uniform(0, 300, n),
    "tiny among huge": lambda r, n: np.where(r.random(n) < 0.01, 1e7, 1e-3),
    "powers of two": lambda r, n: 2.0 ** r.integers(-10, 20, n),
    "exact cancellation": lambda r, n: np.tile(np.array([1e6, -1e6, 3.25, -3.0]), n // 4 + 1)[:n],
    "subnormals": lambda r, n: r.integers(0, 1000, n) * 1e-45,
    "signed zeros": lambda r, n: np.where(r.random(n) < 0.5, 0.0, -0.0),
    "gaussian, large offset": lambda r, n: r.standard_normal(n) * 1e4 + 1e6,
    "gaussian, small": lambda r, n: r.standard_normal(n) * 3e-2 - 1e-2,
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_folded_sum_equals_the_float32_chain(name):
    rng = np.random.default_rng(sum(map(ord, name)))
    with np.errstate(all='ignore'):
        x = np.asarray(CASES[name](rng, 12000), dtype=np.float64).astype(np.float32)
        for chunk in (64, 256):
            want = seq_sum32(x)
            got, folded, walked = folded_sum(x, chunk)
            assert got == want or (np.isnan(got) and np.isnan(want)), (name, chunk, got, want)
            # the point of the exercise: monotone data folds nearly everywhere
            if name in ("integers 0..65535", "pixel coordinates", "negative drift", "gaussian, large offset"):
                assert folded > 0.8 * (folded + walked), (name, chunk, folded, walked)


def test_inf_and_nan_are_walked():
    rng = np.random.default_rng(3)
    with np.errstate(all='ignore'):
        for bad in (np.inf, -np.inf, np.nan):
            x = rng.uniform(0, 100, 6000).astype(np.float32)
            x[3000] = bad
            want = seq_sum32(x)
            got, _, _ = folded_sum(x, 128)
            assert got == want or (np.isnan(got) and np.isnan(want))
