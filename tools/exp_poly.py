#!/usr/bin/env python3
"""Coefficients of fm::exp_neg_ll (erm_rng.hpp): the polynomial of degree `deg` that interpolates e^t at the Chebyshev nodes of [-a, a], a = 0.3466 (|t| <= ln2 / 2 after
the range reduction), computed with 60 decimal digits, rounded to fp64 and checked on 4001 points.  usage: python tools/exp_poly.py [deg ...]"""
import sys
from decimal import Decimal as D, getcontext
getcontext().prec = 60
PI = D('3.14159265358979323846264338327950288419716939937510582097494')
A = D('0.34660')


def dcos(x):
    getcontext().prec += 5
    x = x % (2 * PI)
    s, term, n = D(0), D(1), 0
    while abs(term) > D(10) ** -(getcontext().prec):
        s += term
        n += 2
        term = -term * x * x / (n * (n - 1))
    getcontext().prec -= 5
    return +s


def cheb_poly(deg):
    n = deg + 1
    nodes = [dcos(PI * D(2 * k + 1) / (2 * n)) for k in range(n)]
    f = [(A * x).exp() for x in nodes]
    c = [sum(f[k] * dcos(PI * D(j) * D(2 * k + 1) / (2 * n)) for k in range(n)) * 2 / n for j in range(n)]
    c[0] /= 2
    T = [[D(1)], [D(0), D(1)]]
    for j in range(2, n):
        cur = [D(0)] + [2 * v for v in T[-1]]
        for i, v in enumerate(T[-2]):
            cur[i] -= v
        T.append(cur)
    mono = [D(0)] * n
    for j in range(n):
        for i, v in enumerate(T[j]):
            mono[i] += c[j] * v
    return [mono[i] / A ** i for i in range(n)]


for deg in [int(v) for v in sys.argv[1:]] or [11]:
    md = [float(v) for v in cheb_poly(deg)]
    err = D(0)
    for k in range(-2000, 2001):
        t = A * D(k) / 2000
        pv = D(0)
        for cc in reversed(md):
            pv = pv * t + D(cc)
        err = max(err, abs(pv - t.exp()))
    print(f"degree {deg}: max abs error {float(err):.3e} with fp64 coefficients")
    print("  " + ", ".join(v.hex() for v in md))
