"""Coefficients of mx_gelu.h::gelu_erf2_precise (CPU, dev container): Chebyshev fits of erf(a)/a on [0, 1) and of log2(erfc t)/t on
[1, 4.2], turned into monomials, then the whole GELU evaluated in emulated f32 Horner against scipy over [-9, 9]."""
import numpy as np, math
from numpy.polynomial import chebyshev as C, polynomial as P
from scipy.special import erfc, erf
def cheb_to_mono(c, a, b):
    pu = P.Polynomial(C.cheb2poly(c))
    return pu(P.Polynomial([-(a + b) / (b - a), 2 / (b - a)])).coef
f32 = np.float32
def fma(a, b, c):  # emulate fused multiply-add in f32 via f64 (exact product, one rounding)
    return (a.astype(np.float64) * np.float64(b) + np.float64(c)).astype(np.float32) if np.isscalar(b) else (a.astype(np.float64) * b.astype(np.float64) + (c.astype(np.float64) if not np.isscalar(c) else np.float64(c))).astype(np.float32)
SPLIT, TMAX = 1.0, 4.2
k = np.arange(4000)
s = 0.5 * SPLIT**2 * (1 - np.cos(np.pi * (k + .5) / 4000)); a = np.sqrt(s)
g = np.where(a > 1e-9, erf(a) / np.maximum(a, 1e-300), 2 / math.sqrt(math.pi))
PS = cheb_to_mono(C.chebfit(2 * s / SPLIT**2 - 1, g, 5), 0, SPLIT**2).astype(np.float32)
t = SPLIT + 0.5 * (TMAX - SPLIT) * (1 - np.cos(np.pi * (k + .5) / 4000))
q = (np.log(erfc(t)) / math.log(2)) / t
QT = cheb_to_mono(C.chebfit(2 * (t - SPLIT) / (TMAX - SPLIT) - 1, q, 6), SPLIT, TMAX).astype(np.float32)
print("PS =", [float(x) for x in PS]); print("QT =", [float(x) for x in QT])
def gelu32(x):
    x = x.astype(np.float32)
    a = (x * f32(0.70710678118654752)).astype(np.float32)
    tt = np.minimum(np.abs(a), f32(TMAX)).astype(np.float32)
    ss = (a * a).astype(np.float32)
    p = np.full_like(ss, PS[-1])
    for c in PS[-2::-1]: p = fma(p, ss, f32(c))
    es = (a * p).astype(np.float32)                       # erf, small branch
    qq = np.full_like(tt, QT[-1])
    for c in QT[-2::-1]: qq = fma(qq, tt, f32(c))
    e2 = np.exp2((tt * qq).astype(np.float32).astype(np.float64)).astype(np.float32)   # erfc(|a|)
    half_tail = np.where(a < 0, f32(0.5) * e2, f32(1.0) - f32(0.5) * e2).astype(np.float32)     # Phi on the tail branch
    phi = np.where(np.abs(a) < f32(SPLIT), fma(es, f32(0.5), f32(0.5)), half_tail).astype(np.float32)
    return (x * phi).astype(np.float32)
xs = np.concatenate([np.linspace(-9, 9, 2000001), np.random.default_rng(0).standard_normal(1000000) * 2]).astype(np.float32)
ref = xs.astype(np.float64) * 0.5 * erfc(-xs.astype(np.float64) / math.sqrt(2))
got = gelu32(xs).astype(np.float64)
err = np.abs(got - ref)
rel = err / np.maximum(np.abs(ref), 1e-30)
print("max abs err", err.max(), "at x =", xs[err.argmax()])
m = np.abs(ref) > 1e-6
print("max rel err where |gelu| > 1e-6:", rel[m].max(), "at x =", xs[m][rel[m].argmax()])
print("max err / (ulp-ish 6e-8 * max(|gelu|, 1e-3))", (err / (6e-8 * np.maximum(np.abs(ref), 1e-3))).max())
