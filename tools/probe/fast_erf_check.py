#!/usr/bin/env python3
"""CPU check of csrc/common.h::ssv_erf (the erf every GELU of the library uses): the two fp32 polynomials, emulated operation by operation in numpy float32, against
scipy's fp64 erf on 6 million arguments - maximum absolute error of erf, of gelu and of gelu'.  The coefficients were fitted by iteratively re-weighted least squares
(towards the minimax fit): x * P5(x^2) on |x| < 1 weighted for RELATIVE error, and erfc(t) exp(t^2) / u = P4(u), u = 1 / (1 + p t), on 1 <= t <= 4.3 weighted for
ABSOLUTE error of erf, p chosen by a bounded scalar search.        python tools/probe/fast_erf_check.py"""
import numpy as np
from scipy.special import erf

f32 = np.float32
CA = [1.1283791065216064, -0.37612324953079224, 0.11280179768800735, -0.02671131119132042, 0.004917551297694445, -0.0005631421809084713]
CB = [0.529606282711029, 0.47274520993232727, 0.5967519283294678, -0.9805029630661011, 0.3885830342769623]
PB = 0.932012140750885


def horner(c, s):
    acc = f32(c[-1]) * np.ones_like(s)
    for k in range(len(c) - 2, -1, -1):
        acc = (acc * s + f32(c[k])).astype(f32)          # numpy has no fused multiply-add: one more rounding per step than the device's fmaf
    return acc


def erf32(x):
    x = x.astype(f32)
    t, z = np.abs(x), (x * x).astype(f32)
    a = (x * horner(CA, z)).astype(f32)
    u = (f32(1) / (f32(1) + f32(PB) * t)).astype(f32)
    e = np.exp2((z * f32(-1.4426950408889634)).astype(f32)).astype(f32)
    b = (f32(1) - (u * horner(CB, u)).astype(f32) * e).astype(f32)
    return np.where(t < f32(1), a, np.copysign(b, x))


if __name__ == "__main__":
    xs = np.concatenate([np.linspace(-6, 6, 4000001), np.random.default_rng(0).normal(size=2000000) * 1.5]).astype(f32)
    v = xs.astype(np.float64)
    err = np.abs(erf32(xs).astype(np.float64) - erf(v))
    print(f"erf  : max abs error {err.max():.3e} at x = {xs[err.argmax()]:.4f}; max relative error on |x| < 1: {(err / np.maximum(np.abs(erf(v)), 1e-30))[np.abs(xs) < 1].max():.3e}")
    cdf32 = (f32(0.5) * (f32(1) + erf32((xs * f32(0.70710678118654752440)).astype(f32)))).astype(f32)
    gelu = 0.5 * v * (1 + erf(v / np.sqrt(2)))
    print(f"gelu : max abs error {np.abs((xs * cdf32).astype(np.float64) - gelu).max():.3e} (|v| <= 6)")
    pdf32 = (f32(0.39894228040143267794) * np.exp2(((xs * xs).astype(f32) * f32(-0.72134752044448170368)).astype(f32))).astype(f32)
    dg = 0.5 * (1 + erf(v / np.sqrt(2))) + v * np.exp(-0.5 * v * v) / np.sqrt(2 * np.pi)
    print(f"gelu': max abs error {np.abs((cdf32 + xs * pdf32).astype(np.float64) - dg).max():.3e}")
