"""The GELU the HIP epilogues use (memex_amd/csrc/mx_gelu.h: x * (1/2 + xc P(xc^2)), a degree-8 polynomial for the erf term,
no quarter-rate instruction) against the exact erf form of the reference (rust-bert `gelu`, oracle/bert_oracle.py): the
constants are read out of the header, evaluated the way the kernel does (f32 Horner), and the claims of its comment checked."""
import math
import os
import re

import numpy as np

HDR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "memex_amd", "csrc", "mx_gelu.h")


def _header_constants():
    src = open(HDR).read()
    body = src[src.index("gelu_f32x2 gelu_erf2("):]
    clamp = float(re.search(r"fmed3f\(x\[0\], -([0-9.]+)f, ([0-9.]+)f\)", body).group(2))
    first = re.search(r"fma\(t, \(v2\)(-?[0-9.e+-]+)f, \(v2\)(-?[0-9.e+-]+)f\)", body)
    coeffs = [float(first.group(1)), float(first.group(2))]
    coeffs += [float(m) for m in re.findall(r"fma\(P, t, \(v2\)(-?[0-9.e+-]+)f\)", body)]
    assert len(coeffs) == 9, coeffs
    return clamp, coeffs  # highest degree first


def gelu_kernel_form(x, clamp, coeffs):
    x = x.astype(np.float32)
    xc = np.clip(x, np.float32(-clamp), np.float32(clamp))
    t = (xc * xc).astype(np.float32)
    p = np.full_like(t, np.float32(coeffs[0]))
    for c in coeffs[1:]:
        p = (p * t + np.float32(c)).astype(np.float32)
    phi = (xc * p + np.float32(0.5)).astype(np.float32)
    return (x * phi).astype(np.float32)


def gelu_exact(x):
    x = x.astype(np.float64)
    return 0.5 * x * (1.0 + np.vectorize(math.erf)(x / math.sqrt(2.0)))


def test_polynomial_gelu_error_bounds():
    clamp, coeffs = _header_constants()
    x = np.concatenate([np.linspace(-12.0, 12.0, 480001), np.array([0.0, -0.0, 1e-30, -1e-30, 1e4, -1e4, clamp, -clamp])])
    got = gelu_kernel_form(x, clamp, coeffs).astype(np.float64)
    ref = gelu_exact(x)
    err = np.abs(got - ref)
    assert (err <= 5e-5 + 1e-7 * np.abs(x)).all(), (err.max(), x[err.argmax()])
    assert err[np.abs(x) <= 64.0].max() <= 5e-5
    inside = np.abs(x) <= clamp
    assert (err[inside] <= 1.3e-5 * np.abs(x[inside]) + 1e-7).all()
    pos = x > 1e-3
    assert (err[pos] / ref[pos]).max() <= 2.4e-5
    assert np.isfinite(got).all()


def test_polynomial_gelu_moves_no_embedding():
    """swapping the oracle's exact erf for the kernel's polynomial moves its f64 embeddings by far less than bf16 does"""
    from memex_amd.weights import EncoderConfig, checkpoint_like_weights, synthetic_weights
    from oracle import bert_oracle
    clamp, coeffs = _header_constants()

    def poly(x):
        x = np.asarray(x, dtype=np.float64)
        xc = np.clip(x, -clamp, clamp)
        t = xc * xc
        p = np.full_like(t, coeffs[0])
        for c in coeffs[1:]:
            p = p * t + c
        return x * (0.5 + xc * p)

    exact = bert_oracle.gelu_erf
    try:
        for make in (synthetic_weights, checkpoint_like_weights):
            cfg = EncoderConfig(layers=3, hidden=384, heads=12, ffn=1536, vocab=3000)
            w = make(cfg, 5)
            rng = np.random.default_rng(5)
            ids = rng.integers(1000, cfg.vocab, size=(4, 48)).astype(np.int32)
            lens = np.array([48, 17, 3, 40], dtype=np.int32)
            bert_oracle.gelu_erf = exact
            a = bert_oracle.encode(w, cfg.as_dict(), ids, lens)
            bert_oracle.gelu_erf = poly
            b = bert_oracle.encode(w, cfg.as_dict(), ids, lens)
            cos = (a * b).sum(1) / np.linalg.norm(a, axis=1) / np.linalg.norm(b, axis=1)
            assert (1.0 - cos).max() <= 1e-9, (make.__name__, cos)
    finally:
        bert_oracle.gelu_erf = exact


def test_precise_gelu_of_the_bf16x3_mode():
    """mx_gelu.h::gelu_erf2_precise (MX_PREC_BF16X3's W1 epilogue): constants read out of the header, evaluated the way the
    kernel does -- f32 Horner with fused multiply-adds, v_exp_f32 as exp2 -- against the exact erf form: |err| <= 4e-7 everywhere,
    one part in 2^-18 relative wherever the value is not tiny (the mode keeps 16 significant bits per operand)."""
    src = open(HDR).read()
    body = src[src.index("gelu_f32x2 gelu_erf2_precise("):]
    pfirst = re.search(r"v2 p = __builtin_elementwise_fma\(s, \(v2\)(-?[0-9.e+-]+)f, \(v2\)(-?[0-9.e+-]+)f\)", body)
    P = [float(pfirst.group(1)), float(pfirst.group(2))] + [float(m) for m in re.findall(r"p = __builtin_elementwise_fma\(p, s, \(v2\)(-?[0-9.e+-]+)f\)", body)]
    qfirst = re.search(r"v2 q = __builtin_elementwise_fma\(t, \(v2\)(-?[0-9.e+-]+)f, \(v2\)(-?[0-9.e+-]+)f\)", body)
    Q = [float(qfirst.group(1)), float(qfirst.group(2))] + [float(m) for m in re.findall(r"q = __builtin_elementwise_fma\(q, t, \(v2\)(-?[0-9.e+-]+)f\)", body)]
    tmax = float(re.search(r"fabsf\(a\[0\]\), ([0-9.]+)f\)", body).group(1))
    assert len(P) == 6 and len(Q) == 7 and tmax == 4.2, (P, Q, tmax)
    f32 = np.float32

    def fma(a, b, c):
        return (a.astype(np.float64) * np.float64(b) + np.float64(c)).astype(np.float32) if np.isscalar(b) else \
            (a.astype(np.float64) * b.astype(np.float64) + np.float64(c)).astype(np.float32)
    x = np.concatenate([np.linspace(-12.0, 12.0, 960001), np.random.default_rng(0).standard_normal(200000) * 3,
                        np.array([0.0, -0.0, 1e-30, -1e-30, 1e4, -1e4, math.sqrt(2.0), -math.sqrt(2.0)])]).astype(np.float32)
    a = (x * f32(0.70710678118654752)).astype(np.float32)
    t = np.minimum(np.abs(a), f32(tmax)).astype(np.float32)
    s = (a * a).astype(np.float32)
    p = fma(s, f32(P[0]), f32(P[1]))
    for c in P[2:]:
        p = fma(p, s, f32(c))
    phi_small = fma((a * p).astype(np.float32), f32(0.5), f32(0.5))
    q = fma(t, f32(Q[0]), f32(Q[1]))
    for c in Q[2:]:
        q = fma(q, t, f32(c))
    he = (f32(0.5) * np.exp2((t * q).astype(np.float32).astype(np.float64)).astype(np.float32)).astype(np.float32)
    phi = np.where(t < f32(1.0), phi_small, np.where(a < 0, he, f32(1.0) - he)).astype(np.float32)
    got = (x * phi).astype(np.float64)
    ref = gelu_exact(x)
    err = np.abs(got - ref)
    assert np.isfinite(got).all() and err[np.abs(x) <= 64].max() <= 4e-7, (err.max(), x[err.argmax()])
    big = np.abs(ref) > 1e-3
    assert (err[big] / np.abs(ref[big])).max() <= 2.0 ** -18, (err[big] / np.abs(ref[big])).max()
