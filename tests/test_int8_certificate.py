"""The certificate of the int8 filter copy (memex_amd/csrc/scan8.hip), restated in numpy and
checked on the CPU: with  c8 = rint((c/|c|) / s_h),  s_h = max |c_i/|c|| / 127  per 32-row half tile,  q8 likewise
with the query's own step, the filter score  s_q * s_h * sum q8_i c8_i  (an exact integer sum) differs from the cosine
by at most  Ec + Eq + Ec*Eq  with the MEASURED residual norms  Ec = |c/|c| - s_h c8|,  Eq = |q/|q| - s_q q8|.
The GPU test of the same statement is tests/test_search_gpu.py::test_approximation_error_bound_holds."""
import numpy as np
import pytest


def _quantise_half_tiles(X):
    """rows [n, d] f32 -> (c8 int32 [n, d], step per row f32 [n], residual norm per row f64 [n]); 32-row groups."""
    n, d = X.shape
    nrm = np.linalg.norm(X.astype(np.float64), axis=1)
    unit = np.divide(X, nrm[:, None], out=np.zeros((n, d)), where=nrm[:, None] > 0).astype(np.float32)
    c8 = np.zeros((n, d), dtype=np.int32)
    step = np.zeros(n, dtype=np.float32)
    for h0 in range(0, n, 32):
        blk = unit[h0:h0 + 32]
        mx = np.float32(np.abs(blk).max())
        sh = np.float32(mx / np.float32(127.0))
        inv = np.float32(127.0) / mx if mx > 0 else np.float32(0.0)
        c8[h0:h0 + 32] = np.clip(np.rint(blk * inv), -127, 127).astype(np.int32)
        step[h0:h0 + 32] = sh
    resid = np.linalg.norm(unit.astype(np.float64) - step[:, None].astype(np.float64) * c8, axis=1)
    return unit, c8, step, resid


def _quantise_query(q):
    qn = (q / np.linalg.norm(q.astype(np.float64))).astype(np.float32)
    mx = np.float32(np.abs(qn).max())
    sq = np.float32(mx / np.float32(127.0))
    q8 = np.clip(np.rint(qn * (np.float32(127.0) / mx)), -127, 127).astype(np.int32)
    return qn, q8, sq, float(np.linalg.norm(qn.astype(np.float64) - float(sq) * q8))


def _corpora(rng):
    yield "gaussian 384", rng.standard_normal((512, 384)).astype(np.float32), rng.standard_normal((8, 384)).astype(np.float32)
    yield "gaussian 3", rng.standard_normal((256, 3)).astype(np.float32), rng.standard_normal((8, 3)).astype(np.float32)
    yield "gaussian 1536", rng.standard_normal((128, 1536)).astype(np.float32), rng.standard_normal((4, 1536)).astype(np.float32)
    X = rng.standard_normal((256, 384)).astype(np.float32)
    X[40] = 0
    X[40, 7] = 3.0                                   # a one-hot row inflates its half tile's step
    X[41] = 0                                        # a zero-norm row
    X[100:110] *= 1e-20                              # tiny norms
    X[200:210] = np.sign(X[200:210])                 # constant magnitude: every element sits at the extreme code
    Q = rng.standard_normal((8, 384)).astype(np.float32)
    Q[0] = X[40]                                     # one-hot query: its error is 0, the row's is not
    Q[1] = X[5] * 7.0                                # a query that IS a row
    Q[2] = np.sign(Q[2])
    yield "adversarial 384", X, Q
    yield "heavy tails 768", rng.standard_t(2, (256, 768)).astype(np.float32), rng.standard_t(2, (8, 768)).astype(np.float32)


@pytest.mark.parametrize("seed", [0, 1])
def test_measured_residuals_bound_the_filter_score(seed):
    rng = np.random.default_rng(seed)
    for name, X, Q in _corpora(rng):
        unit, c8, step, ec = _quantise_half_tiles(X)
        for q in Q:
            qn, q8, sq, eq = _quantise_query(q)
            acc = c8.astype(np.int64) @ q8.astype(np.int64)                  # exact integer sums, as the MFMA produces them
            score = (acc.astype(np.float32) * step) * sq                     # ((float)sum * s_h) * s_q
            cos = unit.astype(np.float64) @ qn.astype(np.float64)
            bound = ec + eq + ec * eq + 2.7e-4                               # kAccSlack: f32 normalisation and scaling
            err = np.abs(score.astype(np.float64) - cos)
            assert (err <= bound).all(), (name, float((err - bound).max()))
            assert int(np.abs(acc).max()) < 2 ** 31 and int(np.abs(c8).max()) <= 127
    # what the certificate costs: on dense rows the worst half tile of a big corpus is ~0.014, a query ~0.007-0.009
    _, _, _, ec = _quantise_half_tiles(rng.standard_normal((4096, 384)).astype(np.float32))
    assert 0.007 < ec.max() < 0.016 and 0.007 < np.median(ec) < 0.011


def test_centred_copy_accumulator_initial_value():
    """The centred int8 copy (scan8_kernel<..., CEN = true>): a_q a_c enters as the int32 the MFMA accumulators
    start from, I = trunc(f32(a_c * f32(f32(a_q / s_q) * f32(1 / s_h)))), and the score is ((float)(I + sum) * s_h) * s_q.  Restated
    in numpy with the kernel's f32 roundings: for steps at or above kMinStep8 = 2^-15 (what the builder and prep_queries_kernel keep)
    the integer never leaves int32, and the score differs from a_q a_c + s_h s_q sum by at most s_h s_q + 5e-7 -- inside the
    2.7e-4 the certificate carries as slack."""
    f = np.float32
    rng = np.random.default_rng(3)
    k_min = f(2.0 ** -15)
    n = 200_000
    a_c = rng.uniform(-1.0, 1.0, n).astype(f)
    a_c[:8] = f(1.0) + f(1e-6), f(-1.0) - f(1e-6), 0.0, 1e-30, 0.98, -0.98, 1.0, -1.0      # the extremes a unit row can reach
    a_q = rng.uniform(-1.0, 1.0, n).astype(f)
    a_q[:8] = 1.0, 1.0, 1.0, 1.0, -1.0, 0.97, f(1.0) + f(1e-6), f(-1.0) - f(1e-6)
    # steps from the floor up to a plain unit vector's (1/127), log-uniform; the first entries sit ON the floor
    s_h = np.exp(rng.uniform(np.log(float(k_min)), np.log(1.0 / 127.0), n)).astype(f)
    s_q = np.exp(rng.uniform(np.log(float(k_min)), np.log(1.0 / 127.0), n)).astype(f)
    s_h[:16] = k_min
    s_q[:16] = k_min
    inv_sh = (f(1.0) / s_h).astype(f)
    kq = (a_q / s_q).astype(f)
    ku = (kq * inv_sh).astype(f)
    t = (a_c * ku).astype(f)
    assert np.abs(t.astype(np.float64)).max() < 2.0 ** 30 * (1 + 4e-6)                       # |a| <= 1 + 1e-6 on both sides
    init = np.trunc(t.astype(np.float64)).astype(np.int64)
    # integer sums: what two residual vectors no longer than a unit vector can produce (|s_h s_q sum| <= 1.05), capped at the widest row
    lim = np.minimum(127.0 * 127.0 * 1536.0, 1.05 / (s_h.astype(np.float64) * s_q.astype(np.float64)))
    sums = np.rint(rng.uniform(-1.0, 1.0, n) * lim).astype(np.int64)
    sums[:4] = np.rint(lim[:4]).astype(np.int64)                                             # steps on the floor: 2^30 + the largest sum
    total = init + sums
    assert np.abs(total).max() < 2 ** 31                                                     # the MFMA accumulates in int32
    score = ((total.astype(f) * s_h).astype(f) * s_q).astype(f).astype(np.float64)
    exact = a_q.astype(np.float64) * a_c.astype(np.float64) + s_h.astype(np.float64) * s_q.astype(np.float64) * sums
    err = np.abs(score - exact)
    bound = s_h.astype(np.float64) * s_q.astype(np.float64) + 5e-7 * np.maximum(1.0, np.abs(exact))
    assert (err <= bound).all(), (err.max(), (err / bound).max())
    assert err.max() < 2.7e-4 / 4      # (the truncation costs up to one unit s_h s_q: 6.2e-5 when both steps are a plain unit vector's)
