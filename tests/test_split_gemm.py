"""The distance GEMM's three-term split (harmonypy_amd/csrc/hmx_device.h: bf16_split3, MFMA_BF16), restated in NumPy.

The kernels multiply fp32 operands on the bf16 matrix pipe: x = h + m + l with h, m, l bf16 (8 significant bits each,
round to nearest even), six of the nine partial products, fp32 accumulation.  These tests pin the two claims the design
rests on -- on the CPU, with the instruction's arithmetic emulated exactly (bf16 x bf16 products are exact in fp32; a
k-step's products are summed and added to the fp32 accumulator with one rounding):

  * the split is EXACT for every fp32 value in the range the kernels see (and everywhere else short of underflow),
  * the six-product sum is as close to the float64 result as the f32-input MFMA it replaces, and the three dropped
    products change nothing.

No GPU, no engine calls; the GPU side of the same statement is the parity suite (tests/test_parity_gpu.py)."""
import numpy as np
import pytest


def bf16_rne(x):
    """float32 -> nearest bf16 (ties to even), returned as float32."""
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    return ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32).view(np.float32)


def split3(x):
    x = np.asarray(x, np.float32)
    h = bf16_rne(x)
    r1 = (x - h).astype(np.float32)
    m = bf16_rne(r1)
    r2 = (r1 - m).astype(np.float32)
    lo = bf16_rne(r2)
    return h, m, lo


def test_three_bf16_terms_hold_an_fp32_value_exactly():
    rng = np.random.default_rng(1)
    bits = rng.integers(0, 2 ** 32, size=2_000_000, dtype=np.uint64).astype(np.uint32)
    x = bits.view(np.float32)
    x = x[np.isfinite(x) & (np.abs(x) > 1e-30) & (np.abs(x) < 1e30)]          # away from underflow of the low terms
    x = np.concatenate([x, rng.standard_normal(500_000).astype(np.float32), np.float32([0.0, 1.0, -1.0, 28.853901, 1 - 2 ** -24])])
    h, m, lo = split3(x)
    total = h.astype(np.float64) + m.astype(np.float64) + lo.astype(np.float64)
    assert np.array_equal(total, x.astype(np.float64))
    nz = x != 0
    assert np.all(np.abs(m[nz]) <= 2.0 ** -8 * np.abs(x[nz])) and np.all(np.abs(lo[nz]) <= 2.0 ** -16 * np.abs(x[nz]))


def _workload(N=20000, K=112, d=52, sigma=0.1, seed=0):
    """Unit rows and centroid rows scaled by c = 2 log2(e) / sigma, some centroids close to cells (arguments near 0)."""
    rng = np.random.default_rng(seed)
    Z = rng.standard_normal((N, d)).astype(np.float32)
    Z /= np.linalg.norm(Z, axis=1, keepdims=True)
    Y = rng.standard_normal((K, d)).astype(np.float32)
    Y[: K // 2] = Z[: K // 2] + 0.3 * rng.standard_normal((K // 2, d)).astype(np.float32)
    Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    c = np.float32(2 * np.log2(np.e) / sigma)
    return Z, (Y * c).astype(np.float32), c


def _accumulate(acc, A, B):
    """one MFMA: exact products, their sum added to the fp32 accumulator with one rounding"""
    return (acc.astype(np.float64) + A.astype(np.float64) @ B.astype(np.float64).T).astype(np.float32)


def f32_mfma(Z, Ys, c):
    acc = np.full((Z.shape[0], Ys.shape[0]), -c, np.float32)
    for k in range(0, Z.shape[1], 4):                                    # v_mfma_f32_16x16x4_f32: k-steps of 4
        acc = _accumulate(acc, Z[:, k:k + 4], Ys[:, k:k + 4])
    return acc


SIX = [("l", "h"), ("h", "l"), ("m", "m"), ("m", "h"), ("h", "m"), ("h", "h")]     # (centroid term, cell term), small first


def bf16_mfma(Z, Ys, c, products):
    d = Z.shape[1]
    dpad = -(-d // 32) * 32
    Zp = np.zeros((Z.shape[0], dpad), np.float32); Zp[:, :d] = Z
    Yp = np.zeros((Ys.shape[0], dpad), np.float32); Yp[:, :d] = Ys
    z = dict(zip("hml", split3(Zp)))
    y = dict(zip("hml", split3(Yp)))
    acc = np.full((Z.shape[0], Ys.shape[0]), -c, np.float32)
    for k in range(0, dpad, 32):                                         # v_mfma_f32_16x16x32_bf16: k-steps of 32
        for py, pz in products:
            acc = _accumulate(acc, z[pz][:, k:k + 32], y[py][:, k:k + 32])
    return acc


@pytest.mark.parametrize("d,K,sigma", [(52, 112, 0.1), (32, 48, 0.1), (64, 112, 0.05), (200, 208, 0.1)])
def test_six_products_are_as_good_as_the_f32_matrix_instruction(d, K, sigma):
    Z, Ys, c = _workload(N=6000, K=K, d=d, sigma=sigma)
    exact = Z.astype(np.float64) @ Ys.astype(np.float64).T - float(c)
    err = lambda a: np.abs(a.astype(np.float64) - exact)
    e32, e6 = err(f32_mfma(Z, Ys, c)), err(bf16_mfma(Z, Ys, c, SIX))
    e9 = err(bf16_mfma(Z, Ys, c, [("l", "l"), ("m", "l"), ("l", "m")] + SIX))
    rms = lambda e: float(np.sqrt((e ** 2).mean()))
    print(f"d={d} K={K} c={float(c):.1f}: |err| vs float64  f32 MFMA max {e32.max():.2e} rms {rms(e32):.2e} | six bf16 products max {e6.max():.2e} "
          f"rms {rms(e6):.2e} | all nine max {e9.max():.2e} rms {rms(e9):.2e}")
    assert rms(e6) <= 1.02 * rms(e32) and e6.max() <= 1.25 * e32.max()
    assert rms(e6) <= 1.02 * rms(e9)                                      # the dropped products buy nothing
    # and the two sets of exponent arguments are no further from each other than the f32 instruction's own error allows:
    # a few 1e-5 in an exp2 argument -- nothing the 1e-4 bar on R could see
    assert np.abs(bf16_mfma(Z, Ys, c, SIX) - f32_mfma(Z, Ys, c)).max() <= 2 * e32.max() <= 1e-4


def test_fewer_products_would_not_do():
    """Why six: without the m x m product, or with the h/m terms alone, the error is 6-9x the fp32 one."""
    Z, Ys, c = _workload(N=4000)
    exact = Z.astype(np.float64) @ Ys.astype(np.float64).T - float(c)
    rms = lambda a: float(np.sqrt(((a.astype(np.float64) - exact) ** 2).mean()))
    r32 = rms(f32_mfma(Z, Ys, c))
    assert rms(bf16_mfma(Z, Ys, c, [p for p in SIX if p != ("m", "m")])) > 3 * r32
    assert rms(bf16_mfma(Z, Ys, c, [("m", "h"), ("h", "m"), ("h", "h")])) > 5 * r32
