"""CPU: the split-fp16 ("S16") operand format of precision mode f16x3 (csrc/igemm.hip header), stated in
numpy: x = h + l with h = f16(x), l = f16(x - h).  These properties are what the f16x3 kernels rely on;
the kernels themselves are tested on the GPU (tests/test_gpu_f16x3.py)."""
import numpy as np


def split(x):
    x = np.asarray(x, np.float32)
    h = x.astype(np.float16)
    l = (x - h.astype(np.float32)).astype(np.float16)
    return h, l


def test_split_is_exact_to_22_bits():
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(200000) * np.exp(rng.uniform(-6, 6, 200000))).astype(np.float32)
    x = x[np.abs(x) < 6.0e4]
    h, l = split(x)
    back = h.astype(np.float32) + l.astype(np.float32)
    # relative 2^-22 where the low half is a normal fp16 number, absolute 2^-25 below (subnormal step 2^-24)
    err = np.abs(back.astype(np.float64) - x.astype(np.float64))
    assert (err <= np.maximum(np.abs(x) * 2.0 ** -22, 2.0 ** -25)).all()
    # x - h is exact in fp32 (Sterbenz), so the only rounding is f16(x - h)
    r = x - h.astype(np.float32)
    assert (r.astype(np.float64) == x.astype(np.float64) - h.astype(np.float64)).all()


def test_three_term_product_drops_only_the_low_low_term():
    rng = np.random.default_rng(1)
    x = np.maximum(rng.standard_normal(2304), 0).astype(np.float32)
    w = (rng.standard_normal(2304) * np.sqrt(2.0 / 2304)).astype(np.float32)
    sh = 8 - int(np.floor(np.log2(np.abs(w).max())))        # power-of-two weight scale of pack_conv (api.hip)
    xh, xl = split(x)
    wh, wl = split(np.ldexp(w, sh))
    f = lambda a: a.astype(np.float64)
    three = (f(xh) * f(wh) + f(xh) * f(wl) + f(xl) * f(wh)).sum() * 2.0 ** -sh
    exact = (f(x) * f(w)).sum()
    chain = np.float32(0)
    for a, b in zip(x, w):                                   # the oracle's arithmetic: one fp32 fma chain
        chain = np.float32(np.float64(chain) + np.float64(a) * np.float64(b))
    assert abs(three - exact) <= 2e-6 * np.abs(f(x) * f(w)).sum()
    assert abs(three - exact) <= max(abs(float(chain) - exact), 1e-7) * 4 + 1e-6
    # every term is exact in fp32: 11-bit x 11-bit significands
    p = f(xh) * f(wh)
    assert (p.astype(np.float32).astype(np.float64) == p).all()


def test_range_and_overflow():
    h, l = split(np.float32([65504.0, 65519.0, 1e-8, 0.0, -3.0]))
    assert np.isfinite(h.astype(np.float32)[[0, 1, 3, 4]]).all() and h[0] == np.float16(65504)
    assert np.isinf(np.float32(65520.0).astype(np.float16))  # beyond: the kernels clamp and raise the status bit
    assert h[2] == 0 or abs(float(h[2]) - 1e-8) <= 2.0 ** -25
