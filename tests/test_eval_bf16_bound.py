"""The bound the evaluation's bf16 filter rests on (qrec_amd/csrc/eval_topk.hip, DESIGN.md s4 "Fused evaluation, bf16 route"):
with u^, v^ the tables rounded to bfloat16 (round to nearest even) and the products accumulated in float32,
    | u^ . v^  -  u . v |  <=  eps = 2^-7 * 1.02 * |u| * |v|
so an item whose fp32 score reaches tau has a bf16 score >= tau - eps and cannot be filtered out.  The kernel's guarantee is
this inequality; here it is checked numerically (CPU, numpy) over scales, dimensions, signs and heavy-tailed norms, against the
float64 dot product AND against float32 dot products accumulated in several orders (the MFMA's order is one more)."""
import numpy as np
import pytest


def to_bf16(x):
    """float32 -> bfloat16 (round to nearest, ties to even) -> float32"""
    b = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    b = (b + 0x7FFF + ((b >> 16) & 1)) & 0xFFFF0000
    return b.astype(np.uint32).view(np.float32)


def test_bf16_rounding_is_round_to_nearest_even():
    x = np.array([1.0, 1.0 + 2.0 ** -8, 1.0 + 3 * 2.0 ** -8, 1.0 + 2.0 ** -7, -3.14159, 1e-20, 6.5e4], np.float32)
    y = to_bf16(x)
    assert y[0] == 1.0 and y[1] == 1.0 and y[2] == np.float32(1.0 + 2.0 ** -6) and y[3] == np.float32(1.0 + 2.0 ** -7)   # ties to even
    assert np.all(np.abs(y - x) <= np.abs(x) * 2.0 ** -8)


@pytest.mark.parametrize("d", [32, 64, 128])
@pytest.mark.parametrize("kind", ["gaussian", "uniform_shifted", "heavy_tailed", "aligned"])
def test_bf16_score_error_stays_inside_the_filter_bound(d, kind):
    rng = np.random.default_rng(d * 7 + len(kind))
    nu, ni = 300, 2000
    if kind == "gaussian":
        U = rng.standard_normal((nu, d)); V = rng.standard_normal((ni, d))
    elif kind == "uniform_shifted":       # the bench's tables
        U = rng.random((nu, d)) / 3 - 0.1; V = rng.random((ni, d)) / 3 - 0.1
    elif kind == "heavy_tailed":          # item norms over three decades
        U = rng.standard_normal((nu, d)) * 10.0 ** rng.uniform(-2, 1, (nu, 1)); V = rng.standard_normal((ni, d)) * 10.0 ** rng.uniform(-2, 1, (ni, 1))
    else:                                 # worst case for the Cauchy-Schwarz step: every item parallel to every user
        w = np.abs(rng.standard_normal(d)) + 0.1
        U = w * rng.uniform(0.5, 2.0, (nu, 1)); V = w * rng.uniform(0.5, 2.0, (ni, 1))
    U = U.astype(np.float32); V = V.astype(np.float32)
    Ub, Vb = to_bf16(U), to_bf16(V)
    exact = U.astype(np.float64) @ V.astype(np.float64).T                         # what the fp32 scores approximate
    eps = 2.0 ** -7 * 1.02 * np.linalg.norm(U.astype(np.float64), axis=1)[:, None] * np.linalg.norm(V.astype(np.float64), axis=1)[None, :]
    # bf16 products are exact in float32 (8-bit x 8-bit significands); accumulate them in float32 in three different orders
    approx = [
        (Ub @ Vb.T).astype(np.float64),                                           # BLAS order
        np.add.reduce((Ub[:, None, :] * Vb[None, :64, :]).astype(np.float32), axis=2, dtype=np.float32).astype(np.float64),   # sequential
    ]
    err = np.abs(approx[0] - exact)
    assert (err <= eps).all(), float((err / eps).max())
    err = np.abs(approx[1] - exact[:, :64])
    assert (err <= eps[:, :64]).all()
    # and against the fp32 score itself (what tau is compared with): fp32 accumulation of the unrounded tables, two orders
    s32 = (U @ V.T).astype(np.float64)
    assert (np.abs(approx[0] - s32) <= eps).all()
    # the bound is not vacuous: the worst observed error uses a good part of it only in the aligned case
    ratio = float((np.abs(approx[0] - exact) / eps).max())
    assert ratio < 0.75 and (kind != "aligned" or ratio > 0.15), ratio
