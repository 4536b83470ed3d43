"""-m gpu: every HIP kernel behind the C ABI against the CPU oracle on the same seeded
inputs.  Integer / index results must be identical; fp32 values are compared with the
tolerance BASELINE.json's north_star states (1e-4), and -- because the kernels accumulate
in the oracle's k-order on the exact-fp32 matrix cores -- additionally reported / asserted
bit-for-bit where noted."""
import os
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-4


def dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def bn_params(rng, c):
    g = rng.uniform(0.5, 1.5, c).astype(np.float32)
    b = rng.normal(0, 0.1, c).astype(np.float32)
    m = rng.normal(0, 0.1, c).astype(np.float32)
    v = rng.uniform(0.5, 1.5, c).astype(np.float32)
    return g, b, m, v


def close(got, ref, what):
    got = np.asarray(got)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    err = np.abs(got - ref).max() if ref.size else 0.0
    scale = max(1.0, float(np.abs(ref).max()) if ref.size else 1.0)
    exact = float((got == ref).mean()) if ref.size else 1.0
    print("%s: max abs err %.3g (scale %.3g), bit-equal fraction %.6f" % (what, err, scale, exact))
    assert err <= TOL * scale, what
    return exact


CONV_CASES = [
    # B, H, W, Cin, Cout, k, stride, mode, bn, bias, act, up
    (1, 10, 14, 256, 256, 3, 1, "SAME", True, False, "relu", False),     # head tower / fpn p conv
    (2, 7, 5, 256, 256, 3, 1, "SAME", True, False, "relu", False),       # tiny level, M tail
    (1, 20, 28, 256, 480, 3, 1, "SAME", False, True, None, False),       # class_net/logits
    (1, 20, 28, 256, 24, 3, 1, "SAME", False, True, None, False),        # box_net/encoded_boxes
    (2, 20, 28, 64, 256, 3, 2, "EXPLICIT", True, False, "relu", False),  # fpn p6/p7 form
    (1, 10, 14, 1024, 256, 3, 2, "EXPLICIT", False, False, None, False), # fpn p6 Cin 1024
    (2, 40, 56, 32, 64, 1, 1, "SAME", True, False, "relu6", False),      # Conv2d_1_pointwise
    (1, 20, 28, 512, 512, 1, 1, "SAME", True, False, "relu6", False),    # Conv2d_7..11_pointwise
    (1, 5, 7, 1024, 1024, 1, 1, "SAME", True, False, "relu6", False),    # Conv2d_13_pointwise (tile tail)
    (2, 20, 28, 512, 256, 1, 1, "SAME", False, False, None, True),       # fpn lateral + upsample-add
    (1, 16, 16, 24, 58, 1, 1, "SAME", True, False, "relu", False),       # shufflenet odd channels
    (1, 16, 16, 116, 116, 1, 1, "SAME", True, False, "relu", False),
    (3, 33, 17, 40, 72, 3, 1, "SAME", False, False, None, False),        # ragged everything
    # odd numbers of K-steps (the K loop runs its steps in stage pairs plus one): 3, 9, 27
    (1, 12, 12, 96, 64, 1, 1, "SAME", True, False, "relu6", False),
    (1, 9, 11, 32, 64, 3, 1, "SAME", True, False, "relu", False),
    (2, 8, 8, 96, 128, 3, 2, "EXPLICIT", True, False, "relu", False),
    (1, 16, 16, 64, 64, 1, 2, "SAME", False, False, None, False),        # 1x1 with a stride: rows of A are not the rows of the input
    (1, 8, 8, 64, 64, 3, 1, "SAME", True, False, None, False),           # batch norm without an activation
]


# (28, 29, 31, 32 exist in the diagnostics build only; the shipped library refuses them, test_gpu_multishape.py)
@pytest.mark.parametrize("tile", ["128", "64", "20", "21", "22", "23", "24", "25", "26", "27", "30"])
@pytest.mark.parametrize("case", CONV_CASES, ids=[str(i) for i in range(len(CONV_CASES))])
def test_conv2d(cuda, ssd, oracle_ops, case, tile, libopt):
    # the library picks 64x64 tiles for small problems and 128x128 for large ones: pin each
    # in turn so both kernels see every shape (narrow outputs keep their 128x64 / 128x32 tiles);
    # 20 .. 23: the four wave tiles of the latency form (igemm_lat.hip, v_mfma_f32_16x16x4_f32), which
    # takes every case whose output rows are 16-byte aligned; 24 .. 27: its blocks of two / four waves that share the
    # positions through LDS (where the padded output width is a multiple of the block's 32 / 64 / 128 channels) -- the
    # same bits from all of them
    libopt(igemm_tile=int(tile, 0))
    B, H, W, Cin, Cout, k, stride, mode, use_bn, use_bias, act, use_up = case
    rng = np.random.default_rng(100 + CONV_CASES.index(case))
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((k, k, Cin, Cout)) * np.sqrt(2.0 / (k * k * Cin))).astype(np.float32)
    ref = oracle_ops.conv2d(x, w, stride, mode)
    bn = None
    if use_bn:
        g, b, m, v = bn_params(rng, Cout)
        ref = oracle_ops.bn_act(ref, g, b, m, v, None)
        bn = (m, oracle_ops.bn_scale(g, v), b)
    bias = None
    if use_bias:
        bias = rng.standard_normal(Cout).astype(np.float32)
        ref = oracle_ops.bias_add(ref, bias)
    up = None
    if use_up:
        coarse = rng.standard_normal((B, ref.shape[1] // 2, ref.shape[2] // 2, Cout)).astype(np.float32)
        ref = oracle_ops.upsample2_add(coarse, ref)
        up = dev(cuda, coarse)
    if act == "relu":
        ref = np.maximum(ref, 0)
    elif act == "relu6":
        ref = np.clip(ref, 0, 6)
    got = ssd.ssd.conv2d(dev(cuda, x), w, stride, mode, bn=bn, bias=bias, up=up, act=act).cpu().numpy()
    exact = close(got, ref, "conv2d %s" % (case,))
    # same accumulation order on the exact-fp32 MFMA as the oracle's fmaf chain
    assert exact == 1.0, "conv2d result is within tolerance but not bit-identical to the oracle"


@pytest.mark.parametrize("act", ["relu", "relu6"])
@pytest.mark.parametrize("shape", [(2, 20, 28, 64, 128, 3), (1, 12, 12, 64, 64, 1)])
def test_conv2d_nan_inf(cuda, ssd, oracle_ops, shape, act, libopt):
    """Non-finite inputs through the batch-norm + activation epilogues (the wide-tile form clamps with v_med3_f32, the
    others with two selects): a NaN must come out as 0, +inf as the upper bound, exactly as the oracle's act_apply
    (`v > 0 ? v : 0`, then `v < 6 ? v : 6`) produces them."""
    B, H, W, Cin, Cout, k = shape
    rng = np.random.default_rng(Cin + Cout + k)
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
    x[0, 3, 4, 5] = np.nan
    x[0, 7, 7, 9] = np.inf
    x[B - 1, 9, 2, 1] = -np.inf
    w = (rng.standard_normal((k, k, Cin, Cout)) * np.sqrt(2.0 / (k * k * Cin))).astype(np.float32)
    g, b, m, v = bn_params(rng, Cout)
    with np.errstate(all="ignore"):
        ref = oracle_ops.bn_act(oracle_ops.conv2d(x, w, 1, "SAME"), g, b, m, v, act)
    assert not np.isnan(ref).any() and (ref == 0).any()      # (ReLU keeps +inf; ReLU6 turns it into 6)
    for tile in (128, 64, 20, 23, 25):
        libopt(igemm_tile=tile)
        got = ssd.ssd.conv2d(dev(cuda, x), w, 1, "SAME", bn=(m, oracle_ops.bn_scale(g, v), b), act=act).cpu().numpy()
        assert np.array_equal(got, ref), "tile %d: %d of %d values differ" % (tile, int((got != ref).sum()), ref.size)


@pytest.mark.parametrize("cout", [96, 192, 480])
@pytest.mark.parametrize("use96", ["0", "1"])
def test_conv2d_96_wide_tiles(cuda, ssd, oracle_ops, cout, use96, libopt):
    # output widths that 96 divides and 128 does not (class logits 480, ShuffleNet 96/192) run on
    # 128x96 tiles; option igemm_96 = 0 keeps the zero-padded 128-wide tiles.  Same bits either way.
    libopt(igemm_96=int(use96, 0))
    rng = np.random.default_rng(cout)
    x = rng.standard_normal((2, 19, 23, 64)).astype(np.float32)
    w = (rng.standard_normal((3, 3, 64, cout)) * np.sqrt(2.0 / (9 * 64))).astype(np.float32)
    bias = rng.standard_normal(cout).astype(np.float32)
    ref = oracle_ops.bias_add(oracle_ops.conv2d(x, w, 1, "SAME"), bias)
    got = ssd.ssd.conv2d(dev(cuda, x), w, 1, "SAME", bias=bias).cpu().numpy()
    assert close(got, ref, "conv2d Cout=%d use96=%s" % (cout, use96)) == 1.0


def test_conv2d_empty_and_errors(cuda, ssd):
    x = cuda.zeros((1, 4, 4, 8), dtype=cuda.float32, device="cuda")
    with pytest.raises(ValueError):
        ssd.ssd.conv2d(x, np.zeros((3, 3, 16, 8), np.float32))
    with pytest.raises(ssd.SsdError):
        ssd.ssd.conv2d(x, np.zeros((5, 5, 8, 8), np.float32))         # only k = 1 or 3
    with pytest.raises(TypeError):
        ssd.ssd.conv2d(x.cpu(), np.zeros((3, 3, 8, 8), np.float32))   # no CPU path


DWPW_CASES = [
    # B, H, W, C, Cout, stride, dw_act, pw_act
    (2, 16, 24, 32, 64, 1, "relu6", "relu6"),       # Conv2d_1: 128x64 block shape
    (1, 24, 16, 64, 128, 2, "relu6", "relu6"),      # Conv2d_2: stride 2, 64x128 blocks
    (2, 12, 20, 128, 128, 1, "relu6", "relu6"),     # Conv2d_3
    (1, 16, 16, 128, 256, 2, "relu6", "relu6"),     # Conv2d_4: two column tiles
    (1, 8, 12, 256, 256, 1, "relu6", "relu6"),      # Conv2d_5: 8 K-steps resident
    (3, 4, 4, 24, 58, 1, None, "relu"),             # ShuffleNet unit: odd channels, no act after dw, M tail
    (1, 8, 8, 116, 116, 2, None, "relu"),           # ShuffleNet down-sampling branch
]
# K > 256 (Conv2d_6..13), sizes that the kernel's 8x8 / 4x8 position tiles do not divide (20x28, 10x14, 5x7 ...), odd
# sizes, several tiles per block and several n-tiles
DWPW_STREAM_CASES = [
    (2, 20, 28, 512, 512, 1, "relu6", "relu6"),     # Conv2d_7..11 at 640x896 / 2: 16 slices, ragged tiles, 4 n-tiles
    (1, 20, 28, 512, 1024, 2, "relu6", "relu6"),    # Conv2d_12: stride 2, 8 n-tiles
    (1, 10, 14, 1024, 1024, 1, "relu6", "relu6"),   # Conv2d_13: 32 slices
    (3, 9, 13, 96, 192, 1, "relu6", "relu"),        # odd sizes, width 192 = 1.5 n-tiles
    (2, 18, 6, 64, 24, 2, None, None),              # narrow output, no activations, OW = 3
    (5, 40, 56, 32, 64, 1, "relu6", "relu6"),       # many tiles per block sequence (persistent loop), BN = 64
    (1, 80, 80, 64, 58, 1, None, "relu"),           # ShuffleNet Stage2 unit at 640x640
]


# batch-1 Conv2d_5 .. 13 at 640x896 / 4, position tails, a single slice, activation-free depthwise
DWPW_LAT_CASES = [
    (1, 20, 28, 256, 256, 1, "relu6", "relu6"),     # Conv2d_5
    (1, 20, 28, 256, 512, 2, "relu6", "relu6"),     # Conv2d_6
    (2, 7, 9, 64, 128, 1, None, "relu"),            # one slice, M = 126: a ragged last position tile
    (1, 6, 10, 192, 320, 2, "relu", None),          # three slices (odd count), width 320 -> padded 384
]


@pytest.mark.parametrize("case", DWPW_CASES + DWPW_STREAM_CASES + DWPW_LAT_CASES,
                         ids=[str(i) for i in range(len(DWPW_CASES) + len(DWPW_STREAM_CASES) + len(DWPW_LAT_CASES))])
def test_dw_pw_fused(cuda, ssd, oracle_ops, case):
    # dwpw_stream.hip: LDS-DMA staged input patches, K streamed in 32-channel slices
    B, H, W, C, Cout, stride, dact, pact = case
    rng = np.random.default_rng(500 + (DWPW_CASES + DWPW_STREAM_CASES + DWPW_LAT_CASES).index(case))
    x = rng.standard_normal((B, H, W, C)).astype(np.float32)
    wd = rng.standard_normal((3, 3, C, 1)).astype(np.float32)
    wp = (rng.standard_normal((1, 1, C, Cout)) * np.sqrt(2.0 / C)).astype(np.float32)
    g1, b1, m1, v1 = bn_params(rng, C)
    g2, b2, m2, v2 = bn_params(rng, Cout)
    mid = oracle_ops.bn_act(oracle_ops.depthwise3x3(x, wd, stride), g1, b1, m1, v1, dact)
    ref = oracle_ops.bn_act(oracle_ops.conv2d(mid, wp, 1, "SAME"), g2, b2, m2, v2, pact)
    got = ssd.ssd.dw_pw(dev(cuda, x), wd, stride, (m1, oracle_ops.bn_scale(g1, v1), b1), dact,
                        wp, (m2, oracle_ops.bn_scale(g2, v2), b2), pact).cpu().numpy()
    assert close(got, ref, "dw_pw %s" % (case,)) == 1.0
    # and identical to the two separate kernels
    sep = ssd.ssd.conv2d(ssd.ssd.depthwise3x3(dev(cuda, x), wd, stride, bn=(m1, oracle_ops.bn_scale(g1, v1), b1), act=dact),
                         wp, 1, "SAME", bn=(m2, oracle_ops.bn_scale(g2, v2), b2), act=pact).cpu().numpy()
    assert np.array_equal(got, sep)


@pytest.mark.parametrize("shape", [(8, 160, 224, 32, 64, 1), (4, 80, 112, 256, 512, 2), (6, 40, 56, 512, 512, 1), (16, 40, 40, 128, 116, 1)])
def test_dw_pw_fused_repeatable(cuda, ssd, oracle_ops, shape):
    """Race screen for the LDS-DMA pipeline of dwpw_stream.hip (patch / weight / B slices are ordered for their readers only by
    the issuing wave's counted vmcnt wait plus a barrier, two iterations ahead of their use): the same launch repeated must give
    the same bits, on shapes where a block walks many tiles (persistent loop), with 1, 8 and 16 slices per tile, both strides and
    both column-tile widths -- and beside another stream that keeps the memory system busy."""
    B, H, W, C, Cout, stride = shape
    rng = np.random.default_rng(sum(shape))
    x = dev(cuda, rng.standard_normal((B, H, W, C)).astype(np.float32))
    wd = rng.standard_normal((3, 3, C, 1)).astype(np.float32)
    wp = (rng.standard_normal((1, 1, C, Cout)) * np.sqrt(2.0 / C)).astype(np.float32)
    g1, b1, m1, v1 = bn_params(rng, C)
    g2, b2, m2, v2 = bn_params(rng, Cout)
    bn1, bn2 = (m1, oracle_ops.bn_scale(g1, v1), b1), (m2, oracle_ops.bn_scale(g2, v2), b2)
    first = ssd.ssd.dw_pw(x, wd, stride, bn1, "relu6", wp, bn2, "relu6")
    sep = ssd.ssd.conv2d(ssd.ssd.depthwise3x3(x, wd, stride, bn=bn1, act="relu6"), wp, 1, "SAME", bn=bn2, act="relu6")
    assert cuda.equal(first, sep)
    side = cuda.cuda.Stream()
    junk = cuda.empty((64 << 20,), dtype=cuda.float32, device="cuda")
    for rep in range(20):
        if rep % 2:
            with cuda.cuda.stream(side):
                junk.add_(1.0)                      # 512 MB of traffic beside the kernel
        again = ssd.ssd.dw_pw(x, wd, stride, bn1, "relu6", wp, bn2, "relu6")
        assert cuda.equal(first, again), rep
    cuda.cuda.synchronize()


@pytest.mark.parametrize("stride", [1, 2])
@pytest.mark.parametrize("act", ["relu", "relu6"])
def test_dw_pw_nan_inf(cuda, ssd, oracle_ops, stride, act):
    """Non-finite inputs through dwpw_stream.hip, whose two activations are v_med3_f32(x, 0, hi) on uniform bounds: a NaN
    must come out of an activation as 0 and +inf as the upper bound, exactly as the oracle's act_apply (`v > 0 ? v : 0`,
    then `v < 6 ? v : 6`) and the two-kernel pair produce them -- a compiler or IEEE-mode change of med3's NaN behaviour
    would show here."""
    B, H, W, C, Cout = 2, 16, 24, 64, 96
    rng = np.random.default_rng(40 + stride)
    x = rng.standard_normal((B, H, W, C)).astype(np.float32)
    x[0, 3, 4, 5] = np.nan
    x[0, 8, 7, 9] = np.inf
    x[1, 9, 2, 1] = -np.inf
    x[1, 0, 0, 33] = np.nan                           # a corner: the zero padding of the patch beside it
    wd = rng.standard_normal((3, 3, C, 1)).astype(np.float32)
    wp = (rng.standard_normal((1, 1, C, Cout)) * np.sqrt(2.0 / C)).astype(np.float32)
    g1, b1, m1, v1 = bn_params(rng, C)
    g2, b2, m2, v2 = bn_params(rng, Cout)
    bn1, bn2 = (m1, oracle_ops.bn_scale(g1, v1), b1), (m2, oracle_ops.bn_scale(g2, v2), b2)
    with np.errstate(all="ignore"):
        mid = oracle_ops.bn_act(oracle_ops.depthwise3x3(x, wd, stride), g1, b1, m1, v1, act)
        ref = oracle_ops.bn_act(oracle_ops.conv2d(mid, wp, 1, "SAME"), g2, b2, m2, v2, act)
    assert not np.isnan(mid).any() and (mid == 0).any()          # the depthwise activation already removed every NaN
    if act == "relu":
        assert np.isinf(mid).any()                                # ... ReLU keeps +inf (ReLU6 turns it into 6)
    got = ssd.ssd.dw_pw(dev(cuda, x), wd, stride, bn1, act, wp, bn2, act).cpu().numpy()
    sep = ssd.ssd.conv2d(ssd.ssd.depthwise3x3(dev(cuda, x), wd, stride, bn=bn1, act=act), wp, 1, "SAME", bn=bn2, act=act).cpu().numpy()
    assert np.array_equal(got, sep, equal_nan=True), int((got != sep).sum())
    assert np.array_equal(got, ref, equal_nan=True), int((got != ref).sum())


def test_dw_pw_unsupported_shapes_fail_loudly(cuda, ssd):
    x = cuda.zeros((1, 7, 6, 32), dtype=cuda.float32, device="cuda")       # stride 2 on an odd height
    bn32 = (np.zeros(32, np.float32), np.ones(32, np.float32), np.zeros(32, np.float32))
    with pytest.raises(ssd.SsdError):
        ssd.ssd.dw_pw(x, np.zeros((3, 3, 32, 1), np.float32), 2, bn32, None, np.zeros((1, 1, 32, 32), np.float32), bn32, None)


@pytest.mark.parametrize("B,H,W,C,stride,act", [(2, 20, 28, 32, 1, "relu6"), (1, 40, 56, 64, 2, "relu6"),
                                                (2, 16, 16, 24, 2, None), (1, 10, 10, 58, 1, None),
                                                (1, 6, 8, 1024, 1, "relu6"),
                                                # odd output sizes: the last row pair / pixel pair of a thread is partly outside
                                                (1, 7, 9, 32, 1, "relu6"), (2, 14, 10, 16, 2, None), (1, 5, 3, 64, 1, "relu")])
def test_depthwise(cuda, ssd, oracle_ops, B, H, W, C, stride, act):
    rng = np.random.default_rng(C * 7 + stride)
    x = rng.standard_normal((B, H, W, C)).astype(np.float32)
    w = rng.standard_normal((3, 3, C, 1)).astype(np.float32)
    g, b, m, v = bn_params(rng, C)
    ref = oracle_ops.bn_act(oracle_ops.depthwise3x3(x, w, stride), g, b, m, v, act)
    got = ssd.ssd.depthwise3x3(dev(cuda, x), w, stride, bn=(m, oracle_ops.bn_scale(g, v), b), act=act).cpu().numpy()
    assert close(got, ref, "depthwise C=%d s=%d" % (C, stride)) == 1.0


# widths 32 / 24: the pixel-per-lane kernel's compile-time instances; 8 / 16 / 64: its run-time width; 12: the 4-channel-per-thread
# kernel (Cout % 8 != 0); 10x14 and 18x22: pixel counts that leave the last wave partly filled (35 and 297 pixels)
@pytest.mark.parametrize("B,H,W,Cout,act", [(2, 128, 128, 32, "relu6"), (1, 64, 96, 24, "relu"), (1, 10, 14, 32, "relu6"),
                                            (3, 18, 22, 24, "relu"), (2, 32, 32, 8, "relu6"), (1, 32, 64, 16, None),
                                            (1, 16, 16, 64, "relu6"), (2, 32, 32, 12, "relu")])
def test_first_conv(cuda, ssd, oracle_ops, B, H, W, Cout, act):
    rng = np.random.default_rng(Cout)
    img = rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)
    w = (rng.standard_normal((3, 3, 3, Cout)) * 0.3).astype(np.float32)
    g, b, m, v = bn_params(rng, Cout)
    ref = oracle_ops.bn_act(oracle_ops.conv2d(oracle_ops.preprocess(img), w, 2, "SAME"), g, b, m, v, act)
    got = ssd.ssd.first_conv(dev(cuda, img), w, bn=(m, oracle_ops.bn_scale(g, v), b), act=act).cpu().numpy()
    assert close(got, ref, "first conv") == 1.0


# front.hip: first convolution + Conv2d_1 in one launch.  640x896 / 4 (tiles 12 x 16 of the 80 x 112 output: ragged last tile row),
# a frame smaller than one tile, sizes that leave ragged tiles on both axes, more tiles than resident blocks (persistent loop),
# every activation combination
@pytest.mark.parametrize("B,H,W,acts", [(1, 160, 224, ("relu6", "relu6", "relu6")), (2, 10, 14, ("relu6", "relu6", "relu6")),
                                        (3, 50, 70, ("relu", None, "relu6")), (1, 26, 34, (None, "relu6", None)),
                                        (2, 640, 896, ("relu6", "relu6", "relu6")), (5, 128, 160, ("relu6", "relu", "relu"))])
def test_front_block(cuda, ssd, oracle_ops, B, H, W, acts):
    rng = np.random.default_rng(H * 7 + W)
    img = rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)
    if H >= 50:
        img[0, :3] = 0; img[0, -3:] = 255; img[-1, :, :2] = 255; img[-1, :, -2:] = 0       # edges: pad taps next to extreme pixels
    w0 = (rng.standard_normal((3, 3, 3, 32)) * 0.3).astype(np.float32)
    wd = rng.standard_normal((3, 3, 32, 1)).astype(np.float32)
    wp = (rng.standard_normal((1, 1, 32, 64)) * np.sqrt(2.0 / 32)).astype(np.float32)
    g0, b0, m0, v0 = bn_params(rng, 32)
    g1, b1, m1, v1 = bn_params(rng, 32)
    g2, b2, m2, v2 = bn_params(rng, 64)
    bn0, bn1, bn2 = (m0, oracle_ops.bn_scale(g0, v0), b0), (m1, oracle_ops.bn_scale(g1, v1), b1), (m2, oracle_ops.bn_scale(g2, v2), b2)
    x = dev(cuda, img)
    got = ssd.ssd.front_block(x, w0, bn0, acts[0], wd, bn1, acts[1], wp, bn2, acts[2])
    # bit-identical to the two launches it replaces ...
    sep = ssd.ssd.dw_pw(ssd.ssd.first_conv(x, w0, bn=bn0, act=acts[0]), wd, 1, bn1, acts[1], wp, bn2, acts[2])
    assert cuda.equal(got, sep), int((got != sep).sum())
    # ... and the oracle's three layers
    if B * H * W <= 3 * 50 * 70 or (H, W) == (160, 224):
        c0 = oracle_ops.bn_act(oracle_ops.conv2d(oracle_ops.preprocess(img), w0, 2, "SAME"), g0, b0, m0, v0, acts[0])
        mid = oracle_ops.bn_act(oracle_ops.depthwise3x3(c0, wd, 1), g1, b1, m1, v1, acts[1])
        ref = oracle_ops.bn_act(oracle_ops.conv2d(mid, wp, 1, "SAME"), g2, b2, m2, v2, acts[2])
        assert close(got.cpu().numpy(), ref, "front block") == 1.0
    # the persistent loop and the one-tile-ahead frame fetch: the same launch again, the same bits
    for _ in range(3):
        assert cuda.equal(got, ssd.ssd.front_block(x, w0, bn0, acts[0], wd, bn1, acts[1], wp, bn2, acts[2]))


def test_front_kernels_random_shapes(cuda, ssd, oracle_ops):
    """front.hip on 60 random frame sizes (even H, W for the MobileNet block; multiples of 4 for ShuffleNet's convolution + max
    pool), batch sizes and activation combinations -- tile grids ragged on either axis, frames smaller than a tile, grids beyond
    the resident blocks: every output bit equals the separate launches'."""
    rng = np.random.default_rng(2024)
    acts = [None, "relu", "relu6"]
    w0 = (rng.standard_normal((3, 3, 3, 32)) * 0.3).astype(np.float32)
    wd = rng.standard_normal((3, 3, 32, 1)).astype(np.float32)
    wp = (rng.standard_normal((1, 1, 32, 64)) * 0.25).astype(np.float32)
    w24 = (rng.standard_normal((3, 3, 3, 24)) * 0.3).astype(np.float32)
    bns = {c: tuple(np.ascontiguousarray(v) for v in (lambda g, b, m, v: (m, oracle_ops.bn_scale(g, v), b))(*bn_params(rng, c))) for c in (24, 32, 64)}
    bn32b = tuple(np.ascontiguousarray(v) for v in (lambda g, b, m, v: (m, oracle_ops.bn_scale(g, v), b))(*bn_params(rng, 32)))
    for case in range(60):
        B = int(rng.integers(1, 4))
        if case % 2 == 0:
            H, W = 2 * int(rng.integers(1, 90)), 2 * int(rng.integers(1, 90))
            a = [acts[int(rng.integers(0, 3))] for _ in range(3)]
            x = dev(cuda, rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8))
            got = ssd.ssd.front_block(x, w0, bns[32], a[0], wd, bn32b, a[1], wp, bns[64], a[2])
            sep = ssd.ssd.dw_pw(ssd.ssd.first_conv(x, w0, bn=bns[32], act=a[0]), wd, 1, bn32b, a[1], wp, bns[64], a[2])
        else:
            H, W = 4 * int(rng.integers(1, 60)), 4 * int(rng.integers(1, 60))
            a = [acts[int(rng.integers(0, 3))]]
            x = dev(cuda, rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8))
            got = ssd.ssd.first_conv_maxpool(x, w24, bns[24], a[0])
            sep = ssd.ssd.maxpool3x3s2(ssd.ssd.first_conv(x, w24, bn=bns[24], act=a[0]))
        assert cuda.equal(got, sep), (case, B, H, W, a, int((got != sep).sum()))


def test_front_block_unsupported_widths_fail_loudly(cuda, ssd):
    img = cuda.zeros((1, 16, 16, 3), dtype=cuda.uint8, device="cuda")
    bn = lambda c: (np.zeros(c, np.float32), np.ones(c, np.float32), np.zeros(c, np.float32))
    with pytest.raises(Exception):
        ssd.ssd.front_block(img, np.zeros((3, 3, 3, 24), np.float32), bn(24), "relu", np.zeros((3, 3, 24, 1), np.float32), bn(24), "relu",
                            np.zeros((1, 1, 24, 64), np.float32), bn(64), "relu")
    with pytest.raises(Exception):
        ssd.ssd.front_block(img, np.zeros((3, 3, 3, 32), np.float32), bn(32), "relu", np.zeros((3, 3, 32, 1), np.float32), bn(32), "relu",
                            np.zeros((1, 1, 32, 128), np.float32), bn(128), "relu")


# front.hip, ShuffleNet: first convolution + max pool in one launch.  640x640 / 4 (tiles of 7 x 8 pooled positions: ragged last
# tile row), a frame smaller than one tile, sizes with ragged tiles on both axes, more tiles than resident blocks, no activation
# (negative values under the maximum)
@pytest.mark.parametrize("B,H,W,act", [(1, 160, 160, "relu"), (2, 12, 20, "relu"), (3, 52, 76, None), (2, 640, 640, "relu"),
                                       (9, 128, 96, "relu6")])
def test_first_conv_maxpool(cuda, ssd, oracle_ops, B, H, W, act):
    rng = np.random.default_rng(H * 5 + W)
    img = rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)
    w = (rng.standard_normal((3, 3, 3, 24)) * 0.3).astype(np.float32)
    g, b, m, v = bn_params(rng, 24)
    bn = (m, oracle_ops.bn_scale(g, v), b)
    x = dev(cuda, img)
    got = ssd.ssd.first_conv_maxpool(x, w, bn, act)
    sep = ssd.ssd.maxpool3x3s2(ssd.ssd.first_conv(x, w, bn=bn, act=act))
    assert cuda.equal(got, sep), int((got != sep).sum())
    if B * H * W <= 3 * 160 * 160:
        ref = oracle_ops.maxpool3x3s2(oracle_ops.bn_act(oracle_ops.conv2d(oracle_ops.preprocess(img), w, 2, "SAME"), g, b, m, v, act))
        assert close(got.cpu().numpy(), ref, "first conv + max pool") == 1.0
    for _ in range(3):
        assert cuda.equal(got, ssd.ssd.first_conv_maxpool(x, w, bn, act))


def test_first_conv_maxpool_unsupported_shapes_fail_loudly(cuda, ssd):
    bn = lambda c: (np.zeros(c, np.float32), np.ones(c, np.float32), np.zeros(c, np.float32))
    with pytest.raises(Exception):       # 32 channels
        ssd.ssd.first_conv_maxpool(cuda.zeros((1, 16, 16, 3), dtype=cuda.uint8, device="cuda"), np.zeros((3, 3, 3, 32), np.float32), bn(32), "relu")
    with pytest.raises(Exception):       # H not a multiple of 4
        ssd.ssd.first_conv_maxpool(cuda.zeros((1, 18, 16, 3), dtype=cuda.uint8, device="cuda"), np.zeros((3, 3, 3, 24), np.float32), bn(24), "relu")


def test_maxpool_and_shuffle(cuda, ssd, oracle_ops):
    rng = np.random.default_rng(11)
    x = rng.standard_normal((2, 16, 24, 24)).astype(np.float32)
    assert np.array_equal(ssd.ssd.maxpool3x3s2(dev(cuda, x)).cpu().numpy(), oracle_ops.maxpool3x3s2(x))
    a = rng.standard_normal((3, 5, 7, 58)).astype(np.float32)
    b = rng.standard_normal((3, 5, 7, 58)).astype(np.float32)
    xo, yo = ssd.ssd.concat_shuffle_split(dev(cuda, a), dev(cuda, b))
    rx, ry = oracle_ops.concat_shuffle_split(a, b)
    assert np.array_equal(xo.cpu().numpy(), rx) and np.array_equal(yo.cpu().numpy(), ry)


@pytest.mark.parametrize("shape,D,Cout,act", [((2, 40, 40), 116, 116, "relu"), ((3, 5, 7), 58, 58, "relu"), ((1, 1, 2), 24, 24, "relu"),
                                              ((1, 13, 5), 232, 232, "relu"), ((2, 9, 9), 58, 40, None), ((1, 1, 67), 32, 8, "relu6"),
                                              ((1, 3, 43), 244, 244, "relu"), ((4, 80, 80), 58, 58, "relu"), ((1, 1, 1), 58, 58, "relu"), ((1, 1, 3), 116, 116, "relu")])
def test_shuffle_conv1x1(cuda, ssd, oracle_ops, shape, D, Cout, act):
    """sn_pw.hip through its stage entry point: concat_shuffle_split folded into conv1x1_before's loads (shufflenet_v2.py:94-115,119).
    Row counts that are no multiple of the 64-row tile (2, 65, 129, 105, 162, 67), one / two / four / eight K slices, channel
    counts that end inside an octet (58, 116, 244), narrow outputs -- bit-identical to the oracle's shuffle -> conv -> batch norm and
    to the library's own two-step form."""
    rng = np.random.default_rng(D * 1000 + Cout + shape[1])
    x = rng.standard_normal(shape + (D,)).astype(np.float32)
    y = rng.standard_normal(shape + (D,)).astype(np.float32)
    w = (rng.standard_normal((1, 1, D, Cout)) * np.sqrt(2.0 / D)).astype(np.float32)
    g, b, m, v = bn_params(rng, Cout)
    sf = oracle_ops.bn_scale(g, v)
    xs, _ = oracle_ops.concat_shuffle_split(x, y)
    ref = oracle_ops.bn_act(oracle_ops.conv2d(xs, w, 1, "SAME"), g, b, m, v, act)
    got = ssd.ssd.shuffle_conv1x1(dev(cuda, x), dev(cuda, y), w, (m, sf, b), act).cpu().numpy()
    assert close(got, ref, "shuffle_conv1x1 %s D=%d" % (shape, D)) == 1.0
    xo, _ = ssd.ssd.concat_shuffle_split(dev(cuda, x), dev(cuda, y))
    two = ssd.ssd.conv2d(xo, w, 1, "SAME", bn=(m, sf, b), act=act).cpu().numpy()
    assert np.array_equal(got, two)
    # NaN / inf in the inputs propagate exactly as through the two-step form (the zero-filled pad channels must not turn them into NaN elsewhere)
    x[0, 0, 0, 1] = np.inf
    y[-1, -1, -1, 0] = np.nan
    got = ssd.ssd.shuffle_conv1x1(dev(cuda, x), dev(cuda, y), w, (m, sf, b), act).cpu().numpy()
    xo, _ = ssd.ssd.concat_shuffle_split(dev(cuda, x), dev(cuda, y))
    two = ssd.ssd.conv2d(xo, w, 1, "SAME", bn=(m, sf, b), act=act).cpu().numpy()
    assert np.array_equal(got, two, equal_nan=True)


# --------------------------------------------------------------------------- post-processing
def synth_heads(rng, B, N, C, frac=0.002, base=-4.6):
    """codes ~ N(0,1); logits = base + sparse positives U[-1.5, 3]."""
    codes = rng.standard_normal((B, N, 4)).astype(np.float32)
    logits = np.full((B, N, C), base, np.float32)
    npos = max(1, int(frac * N))
    for b in range(B):
        rows = rng.choice(N, npos, replace=False)
        cls = rng.integers(0, C, npos)
        logits[b, rows, cls] = rng.uniform(-1.5, 3.0, npos).astype(np.float32)
    return codes, logits


def run_post(cuda, ssd, oracle_ops, codes, logits, anc, thr=0.15, iou=0.6, m=25, scaler=None):
    ref = oracle_ops.postprocess(logits, codes, anc, thr, iou, m, scaler)
    boxes, scores, classes, num = ssd.batch_multiclass_non_max_suppression(
        dev(cuda, codes), dev(cuda, anc), dev(cuda, logits), thr, iou, m, box_scaler=scaler)
    got = (boxes.cpu().numpy(), classes.cpu().numpy(), scores.cpu().numpy(), num.cpu().numpy())
    rb, rl, rs, rn = ref
    assert np.array_equal(got[3], rn), ("num_boxes", got[3], rn)
    assert np.array_equal(got[1], rl), "labels differ"
    assert np.abs(got[2] - rs).max() <= TOL and np.abs(got[0] - rb).max() <= TOL
    print("postprocess: num", rn.tolist()[:8], "scores bit-equal", float((got[2] == rs).mean()),
          "boxes bit-equal", float((got[0] == rb).mean()))
    assert np.array_equal(got[2], rs) and np.array_equal(got[0], rb)
    return got


def test_postprocess_config3(cuda, ssd, oracle_ops):
    rng = np.random.default_rng(0)
    anc = oracle_ops.anchors(640, 896)
    codes, logits = synth_heads(rng, 4, anc.shape[0], 80)
    got = run_post(cuda, ssd, oracle_ops, codes, logits, anc)
    assert got[3].min() > 20


def test_postprocess_edge_cases(cuda, ssd, oracle_ops):
    rng = np.random.default_rng(1)
    anc = oracle_ops.anchors(128, 128)
    N = anc.shape[0]
    # (a) nothing above threshold -> all zero outputs
    codes, logits = synth_heads(rng, 2, N, 80)
    logits[:] = -4.6
    got = run_post(cuda, ssd, oracle_ops, codes, logits, anc)
    assert not got[3].any() and not got[0].any()
    # (b) one class with > 512 candidates (global-memory path of the NMS kernel) and
    #     heavy overlap (many suppressions), other classes sparse
    codes, logits = synth_heads(rng, 2, N, 80, frac=0.01)
    logits[0, :, 7] = rng.uniform(-1.0, 4.0, N).astype(np.float32)
    codes[0] *= 0.1
    run_post(cuda, ssd, oracle_ops, codes, logits, anc)
    # (c) exact ties in score: lower anchor index first
    codes, logits = synth_heads(rng, 1, N, 80)
    logits[0, 100:140, 3] = 2.0
    run_post(cuda, ssd, oracle_ops, codes, logits, anc)
    # (d) score exactly at the threshold (0.5 = sigmoid(0)) is not selected; just above is
    codes, logits = synth_heads(rng, 1, N, 80)
    logits[:] = -9.0
    logits[0, 10, 0] = 0.0
    logits[0, 900, 1] = np.float32(1e-6)
    got = run_post(cuda, ssd, oracle_ops, codes, logits, anc, thr=0.5)
    assert got[3][0] == 1 and got[1][0][0] == 1
    # (e) fully clipped (zero-area) boxes never suppress each other: cap at max_per_class
    codes = np.zeros((1, N, 4), np.float32)
    codes[..., 0] = 200.0                      # cy far below the image -> clipped to y = 1
    logits = np.full((1, N, 80), -9.0, np.float32)
    logits[0, :60, 5] = rng.uniform(0, 3, 60).astype(np.float32)
    got = run_post(cuda, ssd, oracle_ops, codes, logits, anc, m=25)
    assert got[3][0] == 25
    # (f) box_scaler division (model.py:67-68) and other thresholds / caps
    codes, logits = synth_heads(rng, 3, N, 80, frac=0.02)
    run_post(cuda, ssd, oracle_ops, codes, logits, anc, thr=0.3, iou=0.4, m=7,
             scaler=np.array([0.8, 1.0, 0.8, 1.0], np.float32))
    # (h) register path of the NMS kernel (<= 512 candidates) with heavy suppression
    codes, logits = synth_heads(rng, 2, N, 80, frac=0.005)
    codes *= 0.1
    logits[0, 200:500, 9] = rng.uniform(-1.0, 4.0, 300).astype(np.float32)
    logits[1, 0:480, 79] = rng.uniform(-1.0, 4.0, 480).astype(np.float32)
    got = run_post(cuda, ssd, oracle_ops, codes, logits, anc)
    assert 0 < np.bincount(got[1][0][:got[3][0]], minlength=80)[9] < 300
    # (g) non-multiple-of-4 class count takes the scalar scan path
    codes, logits = synth_heads(rng, 2, N, 3, frac=0.05)
    run_post(cuda, ssd, oracle_ops, codes, logits, anc, m=5)


@pytest.mark.parametrize("seed", range(10))
def test_postprocess_fuzz(cuda, ssd, oracle_ops, seed):
    """Random class counts, thresholds, caps, densities, box scalers and batch sizes."""
    rng = np.random.default_rng(1000 + seed)
    H, W = [(128, 128), (128, 256), (256, 128)][seed % 3]
    anc = oracle_ops.anchors(H, W)
    N = anc.shape[0]
    C = int(rng.choice([1, 2, 3, 4, 7, 20, 80]))
    B = int(rng.integers(1, 4))
    codes = (rng.standard_normal((B, N, 4)) * rng.choice([0.1, 0.5, 1.5])).astype(np.float32)
    logits = (rng.standard_normal((B, N, C)) * rng.choice([0.5, 1.5, 3.0]) + rng.choice([-6.0, -3.0, -1.0])).astype(np.float32)
    thr = float(rng.choice([0.05, 0.15, 0.5, 0.9]))
    iou = float(rng.choice([0.3, 0.5, 0.6, 0.9]))
    m = int(rng.choice([1, 5, 25, 40]))
    scaler = None if seed % 2 else rng.uniform(0.5, 1.0, 4).astype(np.float32)
    run_post(cuda, ssd, oracle_ops, codes, logits, anc, thr=thr, iou=iou, m=m, scaler=scaler)


def test_postprocess_long_lists(cuda, ssd, oracle_ops, libopt):
    """Candidate lists longer than the one-wave path: > 512 (block kernel, registers) and > 8192 (block kernel, keys in
    global memory); and the block kernel forced on shorter lists."""
    rng = np.random.default_rng(7)
    anc = oracle_ops.anchors(640, 896)
    N = anc.shape[0]
    codes, logits = synth_heads(rng, 2, N, 80, frac=0.001)
    codes *= 0.2
    logits[0, :3000, 11] = rng.uniform(-1.0, 4.0, 3000).astype(np.float32)       # 512 < n <= 8192: block kernel, registers
    logits[0, 5000:6400, 12] = rng.uniform(-1.0, 4.0, 1400).astype(np.float32)
    logits[0, 9000:9600, 13] = 4.0                                                # ... all ties (anchor order decides)
    logits[1, 20000:45000, 42] = rng.uniform(-1.5, 4.0, 25000).astype(np.float32)  # n > 8192
    logits[1, :, 3] = 5.0                                                          # every anchor, all ties
    run_post(cuda, ssd, oracle_ops, codes, logits, anc)
    libopt(nms_fast_max=200)          # a lower hand-over point: same results
    run_post(cuda, ssd, oracle_ops, codes, logits, anc)
    libopt(nms_fast_max=0)
    codes, logits = synth_heads(rng, 2, N, 80, frac=0.002)
    codes *= 0.3
    run_post(cuda, ssd, oracle_ops, codes, logits, anc)


def test_postprocess_many_classes_and_failing_trials(cuda, ssd, oracle_ops):
    """(a) More classes than the pack kernel has threads (its prefix of the per-class counts runs in chunks of 256).
    (b) Long lists built so that the top-score trials of post_nms_kernel FAIL at either level: heavy overlap among the
    best-scoring candidates (the top 128 / top 512 keep fewer than max_boxes_per_class boxes) sends the list on to the
    next trial and then to the full-list path -- the detections must be the oracle's in every case."""
    rng = np.random.default_rng(99)
    anc = oracle_ops.anchors(128, 128)
    N = anc.shape[0]
    C = 300
    codes = (rng.standard_normal((2, N, 4)) * 0.3).astype(np.float32)
    logits = (rng.standard_normal((2, N, C)) * 1.5 - 5.0).astype(np.float32)
    run_post(cuda, ssd, oracle_ops, codes, logits, anc, m=7)
    anc = oracle_ops.anchors(640, 896)
    N = anc.shape[0]
    codes, logits = synth_heads(rng, 1, N, 80, frac=0.0005)
    codes[:] = 0.0                                    # decoded box = its anchor: neighbours of one cell overlap heavily
    # class 5: 3 000 candidates; the 600 best are the six anchors of 100 neighbouring cells of level 3 (one cluster: NMS keeps few)
    logits[0, :, 5] = -9.0
    logits[0, 1000:4000, 5] = rng.uniform(-1.0, 1.0, 3000).astype(np.float32)
    logits[0, 1200:1800, 5] = rng.uniform(3.0, 4.0, 600).astype(np.float32)
    # class 6: the 100 best in one cluster (first trial fails), the next 300 spread out (second trial succeeds)
    logits[0, :, 6] = -9.0
    logits[0, 20000:23000, 6] = rng.uniform(-1.0, 0.5, 3000).astype(np.float32)
    logits[0, 20000:20100, 6] = rng.uniform(3.0, 4.0, 100).astype(np.float32)
    logits[0, 21000:22800:6, 6] = rng.uniform(1.0, 2.0, 300).astype(np.float32)
    run_post(cuda, ssd, oracle_ops, codes, logits, anc)


def _clustered(rng, anc, N, C, cls, sizes, lo=0.0, hi=4.0, spread=0.0, codes=None, logits=None, img=0, grid0=0):
    """A candidate list of class `cls` made of len(sizes) clusters of mutually overlapping boxes (cluster k: sizes[k] level-3
    anchors decoded onto one 0.08 x 0.08 box around a grid point; the clusters are far apart): greedy NMS keeps one box per
    cluster.  Scores ~ U(lo, hi) logits, or (lo, hi) per cluster."""
    if codes is None:
        codes = np.zeros((1, N, 4), np.float32)
        logits = np.full((1, N, C), -9.0, np.float32)
    total = int(sum(sizes))
    free = np.flatnonzero(logits[img, :53760].max(axis=1) < -8.0)         # level-3 anchors no other list of this image uses
    idx = rng.permutation(free)[:total]
    cl = np.repeat(np.arange(len(sizes)), sizes)
    ha, wa = anc[idx, 2] - anc[idx, 0], anc[idx, 3] - anc[idx, 1]
    cya, cxa = anc[idx, 0] + 0.5 * ha, anc[idx, 1] + 0.5 * wa
    g = cl + grid0
    cy = 0.06 + 0.88 * (g // 9) / 8.0 + spread * rng.standard_normal(total) * 0.002
    cx = 0.06 + 0.88 * (g % 9) / 9.0 + spread * rng.standard_normal(total) * 0.002
    codes[img, idx, 0] = 10.0 * (cy - cya) / ha
    codes[img, idx, 1] = 10.0 * (cx - cxa) / wa
    codes[img, idx, 2] = 5.0 * np.log(0.08 / ha)
    codes[img, idx, 3] = 5.0 * np.log(0.08 / wa)
    if np.ndim(lo) == 0:
        logits[img, idx, cls] = rng.uniform(lo, hi, total).astype(np.float32)
    else:
        logits[img, idx, cls] = rng.uniform(np.asarray(lo)[cl], np.asarray(hi)[cl]).astype(np.float32)
    return codes, logits


def test_postprocess_long_lists_go_on_behind_the_trial(cuda, ssd, oracle_ops):
    """Lists beyond the block's registers whose top-score trials keep K < max_boxes_per_class boxes (massive suppression): the
    kernel keeps those K as an exact prefix, drops the trial's candidates and what the K boxes suppress in one pass, and
    finishes the shorter list from K on -- in one wave's registers, the block's, or from global memory.  Every case bit-equal
    to the oracle (nms.py:28-45: one greedy NMS per class)."""
    rng = np.random.default_rng(123)
    anc = oracle_ops.anchors(640, 896)
    N, C = anc.shape[0], 80
    # (a) the timing script's cases: clusters of equal size, uniform scores -- the trial finds every cluster, nothing survives the pass
    for sizes in ([340] * 24, [680] * 12, [8160], [1500] * 24, [272] * 30):
        codes, logits = _clustered(rng, anc, N, C, 7, sizes)
        got = run_post(cuda, ssd, oracle_ops, codes, logits, anc)
        assert got[3][0] == min(len(sizes), 25), (sizes[:2], got[3])
    # (b) one dominant cluster owns the top scores (the trial keeps ONE box), 40 small clusters lie below the cut: the survivors of
    # the pass are finished from K = 1 -- by one wave (<= 512 left), by the block (<= 2 048) and from global memory (more)
    for small in (8, 40, 200):
        sizes = [3000] + [small] * 40
        lo = [2.0] + [-1.0] * 40
        hi = [4.0] + [1.5] * 40
        codes, logits = _clustered(rng, anc, N, C, 11, sizes, lo, hi)
        got = run_post(cuda, ssd, oracle_ops, codes, logits, anc)
        assert got[3][0] == 25, (small, got[3])
    # (c) slightly spread clusters (partial overlaps: some members survive their cluster's best box), several classes and images
    # at once, another cap and threshold; and a cap beyond the kernel's LDS box store (64): the full-list rounds as before
    codes = np.zeros((2, N, 4), np.float32)
    logits = np.full((2, N, C), -9.0, np.float32)
    _clustered(rng, anc, N, C, 3, [500] * 10, spread=6.0, codes=codes, logits=logits, img=0)
    _clustered(rng, anc, N, C, 4, [2500, 30, 30, 30, 900], [1.0, -1, -1, -1, 0.0], [4.0, 0, 0, 0, 0.5], spread=3.0, codes=codes, logits=logits, img=0, grid0=20)
    _clustered(rng, anc, N, C, 3, [6000, 3000], spread=8.0, codes=codes, logits=logits, img=1)
    _clustered(rng, anc, N, C, 79, [100] * 60, codes=codes, logits=logits, img=1, grid0=10)
    run_post(cuda, ssd, oracle_ops, codes, logits, anc)
    run_post(cuda, ssd, oracle_ops, codes, logits, anc, thr=0.3, iou=0.45, m=40)
    run_post(cuda, ssd, oracle_ops, codes, logits, anc, m=100)
    # (d) massive ties at the top (no score cut isolates <= 512 candidates: no trial runs) in front of clusters
    codes, logits = _clustered(rng, anc, N, C, 9, [400] * 12, lo=-1.0, hi=1.0)
    free = np.flatnonzero(logits[0, :53760].max(axis=1) < -8.0)[:3000]
    logits[0, free, 9] = 3.0
    run_post(cuda, ssd, oracle_ops, codes, logits, anc)


def test_postprocess_batch_properties(cuda, ssd, oracle_ops):
    """Full BASELINE size (config 3: B=32, N=71610): images are independent, so a
    permutation of the batch permutes the outputs and every image equals its B=1 run."""
    rng = np.random.default_rng(2)
    anc = oracle_ops.anchors(640, 896)
    codes, logits = synth_heads(rng, 32, anc.shape[0], 80)
    d_anc, d_codes, d_logits = dev(cuda, anc), dev(cuda, codes), dev(cuda, logits)
    full = [t.cpu().numpy() for t in ssd.batch_multiclass_non_max_suppression(d_codes, d_anc, d_logits, 0.15, 0.6, 25)]
    perm = rng.permutation(32)
    pl, pc = dev(cuda, logits[perm]), dev(cuda, codes[perm])
    permd = [t.cpu().numpy() for t in ssd.batch_multiclass_non_max_suppression(pc, d_anc, pl, 0.15, 0.6, 25)]
    for a, b in zip(full, permd):
        assert np.array_equal(a[perm], b)
    one = [t.cpu().numpy() for t in ssd.batch_multiclass_non_max_suppression(
        d_codes[5:6].contiguous(), d_anc, d_logits[5:6].contiguous(), 0.15, 0.6, 25)]
    for a, b in zip(full, one):
        assert np.array_equal(a[5:6], b)
    # the oracle on EVERY image of the batch (nms.py:96-101 maps over images; ~0.1 s of CPU per image)
    ref = oracle_ops.postprocess(logits, codes, anc, 0.15, 0.6, 25)
    assert np.array_equal(full[3], ref[3]) and np.array_equal(full[2], ref[1])
    assert np.array_equal(full[0], ref[0]) and np.array_equal(full[1], ref[2])
    assert ref[3].min() > 0
    # structure: class-major, scores descending inside a class, zero padding
    boxes, scores, classes, num = full
    for b in range(32):
        n = num[b]
        assert (np.diff(classes[b][:n]) >= 0).all()
        same = np.diff(classes[b][:n]) == 0
        assert (np.diff(scores[b][:n])[same] <= 0).all()
        assert not scores[b][n:].any() and not boxes[b][n:].any()
        assert np.bincount(classes[b][:n], minlength=80).max() <= 25
