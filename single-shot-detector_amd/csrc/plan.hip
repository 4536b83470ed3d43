// The layer plan of one network shape (batch, resized + padded height and width): the reference graph (create_pb.py +
// model.py PREDICT) as a flat list of kernel launches on a few streams with explicit dependencies, no host synchronisation.
// Built once per shape by ssd_forward (abi.hip) and KEPT (select_plans below: one set of plans per shape a handle has served,
// least recently used out first when the arena budget is passed), enqueued per call.  What a call takes from its SOURCE frames
// -- their size, the resize's target, box_scaler -- are launch arguments read from the handle (SrcGeom), not part of a plan.
#include "host.h"
#include <chrono>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>

Op make_conv_op(const ssd_handle *h, const ConvW &cw, const float *in, float *out, float *out2, const float *res, int B, int stride,
                int pad, int act, const std::vector<LevelDesc> &lv, bool dense, int in_fmt, int out_fmt,
                int res_fmt, int *flags, unsigned *scan_bits, float scan_lo, bool *scan_marked)
{
    IgemmArgs a;
    memset(&a, 0, sizeof(a));
    a.in = in; a.wt = in_fmt ? cw.wt16 : cw.wt; a.out = out; a.out2 = out2;
    a.wt_lat = in_fmt ? nullptr : cw.wlat;
    a.in_fmt = in_fmt; a.out_fmt = out_fmt; a.res_fmt = res ? res_fmt : 0;
    a.acc_scale = in_fmt ? cw.scale16 : 1.0f;
    a.flags = flags;
    a.mean = cw.mean; a.sf = cw.sf; a.beta = cw.beta; a.bias = cw.bias; a.res = res;
    a.B = B; a.Cin = cw.CinP; a.Cout = cw.CoutP; a.CoutPad = cw.CoutPad; a.taps = cw.taps;
    a.stride = stride; a.pad = pad; a.act = act;
    a.nlevels = (int)lv.size();
    a.ts = g_dbg_ts;
    // Small problems (batch 1, coarse pyramid levels): 128x128 tiles would leave most of the
    // 256 CUs with one wave per SIMD or idle; 64x64 tiles give 4x the blocks.
    int tile = cw.tile;
    if (tile == IGEMM_128x128 && g_force_tile < 0) {
        long long t128 = 0;
        for (size_t i = 0; i < lv.size(); ++i) t128 += ((long long)B * lv[i].OH * lv[i].OW + 127) / 128;
        if (t128 * (cw.CoutPad / 128) < 2 * 256) tile = IGEMM_64x64;
        else if (cw.taps == 1 && !in_fmt && cw.CinP >= 256) {    // (K < 256: 2 .. 4 K-steps per tile, epilogue-dominated: left on 128x128)
            // 1x1 convolutions (8 .. 32 K-steps per tile): a launch is a few rounds of tiles over the 512 block slots and
            // the last, partly filled round costs a whole tile time.  Estimated time = rounds x tile area / efficiency of the
            // shape (measured, scripts/bench_conv.py at 16 images: 512->512 at 40x56 128x128 / 128x64 / 64x64 = 0.184 / 0.173 /
            // 0.180 ms; 1024->1024 at 20x28 0.207 / 0.185 / 0.179 ms).  Inside the network, where two backbone chains run side by
            // side, the step time does not move (A/B on one box: 41.4 / 41.4 ms, ShuffleNet 55.3 / 55.4 ms).
            long long t64 = 0;
            for (size_t i = 0; i < lv.size(); ++i) t64 += ((long long)B * lv[i].OH * lv[i].OW + 63) / 64;
            const double c128 = ceil((double)t128 * (cw.CoutPad / 128) / 512.0) * 16384.0;
            const double c12864 = ceil((double)t128 * (cw.CoutPad / 64) / 512.0) * 8192.0 / 0.97;
            const double c64 = ceil((double)t64 * (cw.CoutPad / 64) / 512.0) * 4096.0 / 0.93;
            if (c12864 < c128 && c12864 <= c64) tile = IGEMM_128x64;
            else if (c64 < c128) tile = IGEMM_64x64;
        }
        // tests: option igemm_tile = 128 / 64 pins the choice so both variants see every shape
        const int pin = ssd_opt(h, OPT_IGEMM_TILE, 0);
        if (pin == 128) tile = IGEMM_128x128;
        else if (pin == 64) tile = IGEMM_64x64;
    }
    // Large S16 -> S16 batch-norm launches (head towers, FPN outputs at serving batch sizes) take the
    // 256 x 256-tile kernel of igemm16.hip once there are at least two full rounds of tiles for the 256 CUs;
    // option igemm16 = 0 / 1 pins the choice (tests, A/B runs).
    // Its second epilogue form (bias, fp32 rows: the class logits, 6 * num_classes wide) pads the width to a multiple of 256.
    const bool bnform = out_fmt && dense && cw.mean && !cw.bias && cw.CoutPad % 256 == 0 && cw.CoutP == cw.CoutPad;
    bool biasform = !out_fmt && !out2 && cw.bias && !cw.mean && act == SSD_ACT_NONE && cw.CoutP >= 256 && cw.CoutP % 8 == 0 &&
                    (cw.wt16w || cw.CoutPad % 256 == 0);
    for (size_t i = 0; i < lv.size(); ++i)
        if ((lv[i].out_rstride | lv[i].out_bstride | lv[i].out_off) & 3 || lv[i].OH * lv[i].OW < 4) biasform = false;
    if (in_fmt && !res && (bnform || biasform) && cw.taps * (cw.CinP / 32) >= 3 && g_force_tile < 0) {
        long long t256 = 0;
        for (size_t i = 0; i < lv.size(); ++i) t256 += ((long long)B * lv[i].OH * lv[i].OW + 255) / 256;
        bool use16 = t256 * (cw.CoutPad16 / 256) >= 2 * 256;
        const int pin16 = ssd_opt(h, OPT_IGEMM16, -1);
        if (pin16 >= 0) use16 = pin16 != 0;
        if (use16) {
            tile = IGEMM16_TILE;
            a.CoutPad = cw.CoutPad16;
            if (cw.wt16w) a.wt = cw.wt16w;
            if (scan_bits && biasform) { a.scan_lo = scan_lo; a.scan_bits = scan_bits; }
        }
    }
    // Exact-fp32 kernel, bias form with 16-byte stores (the class logits): its epilogue marks the candidate octets too, so
    // the post-processing's scan reads the bitmap instead of every logit in BOTH precision modes (batch 1: 52 -> ~10 us)
    if (scan_bits && !a.scan_bits && tile != IGEMM16_TILE && !in_fmt && !out_fmt && !out2 && !res && cw.bias && !cw.mean &&
        act == SSD_ACT_NONE && cw.CoutP % 4 == 0) {
        bool aligned = true;
        for (size_t i = 0; i < lv.size(); ++i)
            if ((lv[i].out_rstride | lv[i].out_bstride | lv[i].out_off) & 3) aligned = false;
        if (aligned) { a.scan_lo = scan_lo; a.scan_bits = scan_bits; }
    }
    if (scan_marked) *scan_marked = a.scan_bits != nullptr;
    a.dense_out = dense ? 1 : 0;
    double rows = 0, inb = 0;
    for (size_t i = 0; i < lv.size(); ++i) {
        IgemmLevel &L = a.lv[i];
        L.H = lv[i].H; L.W = lv[i].W; L.OH = lv[i].OH; L.OW = lv[i].OW;
        L.M = B * L.OH * L.OW;
        L.dP = ssd_udiv_make((unsigned)(L.OH * L.OW));
        L.dOW = ssd_udiv_make((unsigned)L.OW);
        L.param_off = lv[i].param_off;
        L.out_rstride = lv[i].out_rstride;
        L.in_off = lv[i].in_off; L.out_off = lv[i].out_off; L.out_bstride = lv[i].out_bstride;
        L.res_off = lv[i].res_off;
        L.wt_off = lv[i].wt_off;
        L.stride = lv[i].stride > 0 ? lv[i].stride : stride;
        L.pad = lv[i].pad >= 0 ? lv[i].pad : pad;
        rows += L.M;
        inb += (double)B * L.H * L.W * cw.Cin_l * 4.0;
    }
    // Small launches in exact fp32 (batch 1-2): the latency form of igemm_lat.hip -- v_mfma_f32_16x16x4_f32 (a quarter of the
    // 32x32x2 accumulator's chain latency, 16x16 tiles: sixteen times the independent chains of a 64x64 tile).  Same bits.
    //   one wave per block   fpn p6 / p7 / lateral5 (140, 35, 560 positions per image): alone p6 141 -> 57 us, p7 38 -> 17,
    //                        lateral5 20 -> 10 (profiles/r03_conv_latency_form.log)
    //   four waves per block 1x1 launches of one or two 64x64 tiles per CU, below
    // Option igemm_lat = 0 switches the form off, igemm_tile = 20 .. 27 pins a tile wherever the form applies (tests run
    // every shape on all of them).
    if (igemm_is_lat(tile)) {            // (diagnostics: ssd_bench_conv asked for this wave tile)
        if (!igemm_lat_supports(a) || cw.CoutPad % igemm_lat_bn(tile) || !a.wt_lat) tile = IGEMM_128x128;
    } else if (tile != IGEMM16_TILE && g_force_tile < 0 && igemm_lat_supports(a)) {
        const int pin = ssd_opt(h, OPT_IGEMM_TILE, 0);
        if (igemm_is_lat(pin) && a.wt_lat && cw.CoutPad % igemm_lat_bn(pin) == 0) tile = pin;
        else if (pin == 0 && ssd_opt(h, OPT_IGEMM_LAT, 1)) {
            long long waves = 0, b64 = 0;
            for (size_t i = 0; i < lv.size(); ++i) {
                waves += (((long long)a.lv[i].M + 15) / 16) * (cw.CoutPad / 16);
                b64 += (((long long)a.lv[i].M + 63) / 64) * ((cw.CoutPad + 63) / 64);
            }
            if ((waves <= 320 || b64 <= 40) && a.wt_lat) {
                // the one-accumulator wave with its K-step interleaved and 16 K-steps of operands in flight (igemm_lat.hip;
                // batch-1 forward 1.652 -> 1.637 ms against the plain K-step, profiles/r04_batch1_option_ab.log)
                tile = IGEMM_LAT_1x1_D16;
            }
            // 1x1 launches of one or two 64x64 tiles per CU (batch 1-2: MobileNet pointwise Conv2d_5 .. 13, laterals 3 and 4): the
            // tiles do not divide over the 256 CUs -- 280 of a 512 -> 512 layer at 40x56: 24 CUs run two, the launch takes two
            // tile times -- and a CU's one block per SIMD covers none of its own waits (0.75 of the matrix pipe in its K loop,
            // profiles/r03_tower_phases_b1.log).  The four-wave block form of igemm_lat.hip, 16 x 64 per block: 16x16 granularity,
            // 4.4 waves per SIMD.  Alone 22 -> 18 us per launch (profiles/r03_conv_latency_form.log); batch-1 forward
            // 1.735 -> 1.690 ms, batch 2 3.10 -> 3.06 ms (profiles/r03_batch1_option_ab.log).
            else if (a.wt_lat && cw.CoutPad % 64 == 0 && cw.taps == 1 && cw.CinP >= 256 && b64 <= (ssd_opt(h, OPT_IGEMM_LAT, 1) == 2 ? 1280 : 640))
                tile = IGEMM_LAT_W4_1x1;
            // fpn p6 / p7 (3x3 stride 2 on c5 / p6) at the batches where their launches are fewer than two 64x64 tiles per CU and more
            // than the one-wave form takes: the four-wave form.  Alone at 32 images (scripts/bench_conv.py 32 p67): p6 0.288 -> 0.228 ms,
            // p7 0.040 -> 0.028 ms; the 32-image step 39.03 / 39.06 -> 38.97 / 39.01 ms, same box.  Option igemm_lat = 3 keeps them on
            // the 64x64 tiles.
            else if (a.wt_lat && cw.CoutPad % 64 == 0 && cw.taps == 9 && stride == 2 && b64 <= 512 && ssd_opt(h, OPT_IGEMM_LAT, 1) != 3)
                tile = IGEMM_LAT_W4_1x1;
            // ... and the other single-level 3x3 launches of a small serving batch (fpn p4 / p5 from 4 images on: up to 1 280 tiles
            // of 64x64): 8 images 10.19 / 10.20 -> 10.13 / 10.14 ms per step, 32 images (p5 only) no difference, same box.  Not the
            // multi-level launches (towers, the grouped fpn launch of batch 1-2: +110 us per forward on this form).
            else if (a.wt_lat && cw.CoutPad % 64 == 0 && cw.taps == 9 && lv.size() == 1 && B >= 4 && b64 <= 1280 && ssd_opt(h, OPT_IGEMM_LAT, 1) != 3)
                tile = IGEMM_LAT_W4_1x1;
            // (The box head -- 3x3, 24 of 32 columns, 96 tiles of 128x32 at batch 1 -- is NOT such a launch: 1 492 / 746 waves of
            //  16x16 / 16x32 took 73 / 66 us against the 128x32 tiles' 41 us, profiles/r03_conv_latency_form.log.)
        }
    }
    // 64x64 tiles of a launch of a few blocks per CU (the FPN, pointwise and tower layers of a batch-1 / batch-2 forward):
    // the instance with two register sets, loads three K-steps ahead.  In the network these launches'
    // operands come from HBM / the Infinity Cache (57 MB of weights pass between two uses of a layer's), and a K-step of
    // 0.43 us of MFMA work with loads issued 3/4 of a step ahead waits for them.  Option igemm_deep64 = 0 / 1 pins it.
    if (tile == IGEMM_64x64 && !in_fmt && g_force_tile < 0) {
        long long b64 = 0;
        for (size_t i = 0; i < lv.size(); ++i) b64 += (((long long)a.lv[i].M + 63) / 64) * (cw.CoutPad / 64);
        bool deep = b64 <= 2048;     // (up to the tower launches of a batch-2 forward: those of batch 1 gain 1 %)
        const int pin = ssd_opt(h, OPT_IGEMM_DEEP64, -1);
        if (pin >= 0) deep = pin != 0;
        if (deep) tile = IGEMM_64x64D;
    }
    const bool lat = igemm_is_lat(tile);
    a.n_tiles_n = a.CoutPad / (tile == IGEMM16_TILE ? 256 : (lat ? igemm_lat_bn(tile) : igemm_tile_bn(tile)));
    a.dN = ssd_udiv_make((unsigned)a.n_tiles_n);
    int tiles = 0;
    const int BM = tile == IGEMM16_TILE ? 256 : (lat ? igemm_lat_bm(tile) : igemm_tile_bm(tile));
    for (size_t i = 0; i < lv.size(); ++i) {
        a.lv[i].tile_begin = tiles;
        tiles += (a.lv[i].M + BM - 1) / BM;
    }
    a.tiles_m = tiles;
    a.dM = ssd_udiv_make((unsigned)tiles);
    a.n_major = lat && igemm_lat_n_major(tile) ? 1 : 0;
    Op op;
    op.cls = cw.taps == 9 ? (tile == IGEMM16_TILE ? 7 : 0) : 1;
    op.flops = 2.0 * rows * cw.taps * cw.Cin_l * cw.Cout_l;
    op.bytes = inb + rows * cw.Cout_l * 4.0 + (double)cw.taps * cw.Cin_l * cw.Cout_l * 4.0;
    op.run = [a, tile, tiles](hipStream_t s) {
        if (tile == IGEMM16_TILE) return launch_igemm16(a, tiles, s);
        return igemm_is_lat(tile) ? launch_igemm_lat(tile, a, tiles, s) : launch_igemm(tile, a, tiles, s);
    };
    return op;
}

static void free_planset(PlanSet *ps)
{
    for (Plan *pl : ps->plans) {
        for (Op &op : pl->ops)
            if (op.done) (void)hipEventDestroy(op.done);
        pl->pool.free_all();
        if (pl->ev_join) (void)hipEventDestroy(pl->ev_join);
        if (pl->ev_begin) (void)hipEventDestroy(pl->ev_begin);
        delete pl;
    }
    delete ps;
}

void free_plans(ssd_handle *h)
{
    for (PlanSet *ps : h->cache) free_planset(ps);
    h->cache.clear();
    h->cur = nullptr;
}

// ----------------------------------------------------------------------------- plan
struct Pyr { int h[5], w[5]; long long off[5]; long long total; };

static Pyr make_pyr(int B, int H, int W, int C)
{
    Pyr p;
    long long o = 0;
    for (int l = 0; l < 5; ++l) {
        p.h[l] = (H + A_STRIDES[l] - 1) / A_STRIDES[l];
        p.w[l] = (W + A_STRIDES[l] - 1) / A_STRIDES[l];
        p.off[l] = o;
        o += (long long)B * p.h[l] * p.w[l] * C;
    }
    p.total = o;
    return p;
}

Op make_dw_op(const DwW &d, const float *in, int B, int H, int W, int stride, int act, float *out, int Cl, int out16, int *flags)
{
    const int OH = H / stride, OW = W / stride, pad = stride == 1 ? 1 : 0;
    Op op;
    op.cls = 2;
    op.flops = 2.0 * 9 * (double)B * OH * OW * Cl;
    op.bytes = ((double)B * H * W + (double)B * OH * OW) * Cl * 4.0;
    const DwW dd = d;
    op.run = [=](hipStream_t s) {
        return launch_depthwise(in, B, H, W, dd.Cp, dd.w, stride, pad, OH, OW, dd.mean, dd.sf, dd.beta, act, out, s, out16, flags);
    };
    return op;
}

// depthwise + pointwise on the streaming kernel (dwpw_stream.hip): any K % 32 == 0, any image size; dense output rows
// [M][out_rs] (out_rs: a ShuffleNet stage's last unit writes into the x half of the stage output's rows)
bool dwpws_eligible(const DwW &d, const ConvW &cw, int B, int H, int W, int stride)
{
    const int OH = H / stride, OW = W / stride;
    if (cw.taps != 1 || d.Cp != cw.CinP || d.Cp % 32 != 0 || !d.pack || !cw.mean || cw.bias) return false;
    if ((stride != 1 && stride != 2) || (stride == 2 && ((H | W) & 1))) return false;
    if ((long long)B * H * W * d.Cp * 4 >= (1LL << 31) || (long long)B * OH * OW * cw.CoutP * 4 >= (1LL << 31)) return false;
    return (cw.CoutP + dwpws_tile_n(stride, cw.CoutP) - 1) / dwpws_tile_n(stride, cw.CoutP) <= 64;
}

Op make_dwpws_op(const DwW &d, const ConvW &cw, const float *in, int B, int H, int W, int stride, int dact, int act,
                 float *out, int out_rs)
{
    DwPwSArgs a;
    memset(&a, 0, sizeof(a));
    a.in = in; a.dwpack = d.pack; a.wt = cw.wt; a.mean = cw.mean; a.sf = cw.sf; a.beta = cw.beta; a.out = out;
    a.B = B; a.H = H; a.W = W; a.K = d.Cp; a.OH = H / stride; a.OW = W / stride;
    a.Cout = cw.CoutP; a.wt_rows = cw.CoutPad;
    a.out_rs = out_rs > 0 ? out_rs : cw.CoutP;
    a.pad = stride == 1 ? 1 : 0;
    a.dact = dact; a.act = act;
    const int TY = stride == 1 ? 8 : 4, BN = dwpws_tile_n(stride, cw.CoutP);
    a.tiles_y = (a.OH + TY - 1) / TY; a.tiles_x = (a.OW + 7) / 8;
    a.m_tiles = B * a.tiles_y * a.tiles_x;
    a.n_tiles = (cw.CoutP + BN - 1) / BN;
    a.out_bytes = (int)(((long long)B * a.OH * a.OW - 1) * a.out_rs * 4 + (long long)cw.CoutP * 4);
    a.ts = g_dbg_ts;
#ifdef SSD_DIAG
    if (const char *e = getenv("SSD_DWPWS_ABL")) a.abl = atoi(e);
#endif
    Op op;
    op.cls = 6;
    const double M = (double)B * a.OH * a.OW;
    op.flops = (2.0 * 9 * cw.Cin_l + 2.0 * cw.Cin_l * cw.Cout_l) * M;
    op.bytes = ((double)B * H * W * cw.Cin_l + M * cw.Cout_l) * 4.0 + (double)cw.Cin_l * cw.Cout_l * 4.0;
    op.run = [a, stride](hipStream_t s) { return launch_dwpw_stream(stride, a, s); };
    return op;
}

// ShuffleNet's conv1x1_before on gathered rows (sn_pw.hip): `src` (device, cw.CinP entries) names the column of `base` that
// holds input channel k of row 0; M rows of `rs` bytes each
Op make_pw_gather_op(const ConvW &cw, const float *base, long long base_bytes, const int *src, int rs, long long M, int act, float *out)
{
    PwGArgs a;
    memset(&a, 0, sizeof(a));
    a.base = base; a.base_bytes = (int)base_bytes; a.src = src; a.rs = rs;
    a.wt = cw.wt; a.mean = cw.mean; a.sf = cw.sf; a.beta = cw.beta;
    a.out = out; a.out_rs = cw.CoutP; a.out_bytes = (int)(M * cw.CoutP * 4);
    a.M = (int)M; a.K = cw.CinP; a.Cout = cw.CoutP; a.wt_rows = cw.CoutPad; a.act = act;
    a.m_tiles = (int)((M + 63) / 64);
    const int BN = pw_gather_tile_n(cw.CoutP);
    a.n_tiles = (cw.CoutP + BN - 1) / BN;
    Op op;
    op.cls = 1;
    op.flops = 2.0 * (double)M * cw.Cin_l * cw.Cout_l;
    op.bytes = (double)M * (cw.Cin_l + cw.Cout_l) * 4.0 + (double)cw.Cin_l * cw.Cout_l * 4.0;
    op.run = [a](hipStream_t s) { return launch_pw_gather(a, s); };
    return op;
}

// MobileNet's first convolution + Conv2d_1 as one launch (front.hip): the frames at the network's input size, 32 -> 32 -> 64
Op make_front_op(ssd_handle *h, int img_index, const DwW &f, int act0, const DwW &d, const ConvW &cw, int B, int H, int W, int dact, int act, float *out)
{
    FrontArgs q;
    memset(&q, 0, sizeof(q));
    q.w0 = f.w; q.m0 = f.mean; q.s0 = f.sf; q.b0 = f.beta; q.dwpack = d.pack;
    q.wt = cw.wt; q.mean = cw.mean; q.sf = cw.sf; q.beta = cw.beta; q.out = out;
    q.B = B; q.H = H; q.W = W; q.act0 = act0; q.dact = dact; q.act = act;
    q.tiles_y = (H / 2 + front_tile_y() - 1) / front_tile_y();
    q.tiles_x = (W / 2 + front_tile_x() - 1) / front_tile_x();
    Op op;
    op.cls = 6;
    const double M = (double)B * (H / 2) * (W / 2);
    op.flops = (2.0 * 27 * cw.Cin_l + 2.0 * 9 * cw.Cin_l + 2.0 * cw.Cin_l * cw.Cout_l) * M;
    op.bytes = (double)B * H * W * 3 + M * cw.Cout_l * 4.0;
    op.run = [q, h, img_index, H, W](hipStream_t s) {
        FrontArgs r = q;
        r.img = h->cur_images + (size_t)img_index * H * W * 3;       // (a plan with this op runs frames of the network's own size only)
        return launch_front(r, s);
    };
    return op;
}

LevelDesc dense_level(int H, int W, int OH, int OW, int CoutP, long long in_off, long long out_off, int param_off, long long res_off)
{
    LevelDesc d;
    d.H = H; d.W = W; d.OH = OH; d.OW = OW;
    d.in_off = in_off; d.out_off = out_off;
    d.out_bstride = (long long)OH * OW * CoutP;
    d.out_rstride = CoutP;
    d.param_off = param_off;
    d.res_off = res_off;
    d.wt_off = 0;
    return d;
}

// H, W: the network's input size (multiples of 128); ident: the source frames already have it (no resize, no pad band)
static int build_plan(ssd_handle *h, Plan &pl, int B, int H, int W, bool ident, int img0)
{
    pl.B = B;
    pl.img0 = img0;
    DevPool &ap = pl.pool;
    auto falloc = [&](float **p, long long nfloats) { return ap.alloc((void **)p, (size_t)nfloats * sizeof(float)); };

    // precision mode f16x3: FPN + heads run on split-fp16 operands (igemm.hip "S16"); the backbone stays exact
    // fp32 and hands over c5 in S16 rows, c3 / c4 (which the next depthwise layer also reads) in fp32.
    const int X16 = h->precision == SSD_PRECISION_F16X3 ? 1 : 0;
    int *const FL = h->flags_dev;
    // ---------------- backbone
    float *C3 = nullptr, *C4 = nullptr, *C5 = nullptr;
    unsigned char *scratch = nullptr;           // backbone-only memory that later stages of the same forward may reuse (MobileNet)
    size_t scratch_bytes = 0;
    // a tensor's slot inside a shared block: its bytes + the 256 bytes of slack every tensor has behind it (DevPool::alloc), 256-aligned
    auto scratch_slot = [](size_t bytes) { return (bytes + 256 + 255) & ~(size_t)255; };
    const int h2 = H / 2, w2 = W / 2;
    int id_bb_last[4] = {-1, -1, -1, -1};     // last backbone op of each chain (MobileNet split), -1: no such chain
    int id_c4 = -1;                            // the op that completes c4 when the backbone is ONE chain (then c3 precedes it on the same stream)
    if (h->cfg.backbone == SSD_BACKBONE_MOBILENET) {
        // The backbone is a chain of ~30 short, latency-bound kernels (two blocks per CU each waiting for one
        // round of loads).  From 4 images on it runs as two half-batch chains on the plan's two streams, so that
        // one chain's memory phases sit under the other's compute; FPN and heads stay full-batch launches.
        // Measured (f16x3, same box): +2.1 % at 32 images, +3.8 % at 16, +3.2 % at 8, +4.5 % at 4; mode f32 (round 2):
        // +1.3 % at 4, +1.5 % at 8, +0.8 % at 16, none at 32.
        // option backbone_split = 1 keeps one chain.
        int nhalf = B >= 4 ? 2 : 1;
        { const int v = ssd_opt(h, OPT_BACKBONE_SPLIT, 0); if (v >= 1 && v <= 4 && v <= B) nhalf = v; }
        // retained outputs c3 / c4 / c5: full-batch tensors, each half writes its images
        {
            int hh = h2, ww = w2;
            for (int i = 0; i < 13; ++i) {
                hh /= MB_STRIDE[i]; ww /= MB_STRIDE[i];
                if (i == 4 || i == 10 || i == 12) {
                    float *t;
                    SSDCHK(falloc(&t, (long long)B * hh * ww * h->pw[i].CoutP));
                    if (i == 4) C3 = t; else if (i == 10) C4 = t; else C5 = t;
                    const char *nm = i == 4 ? "c3" : (i == 10 ? "c4" : "c5");
                    pl.retained[nm] = Retained{t, B, hh, ww, h->pw[i].Cout_l, h->pw[i].CoutP, true, (i == 12 && X16) ? 1 : 0};
                }
            }
        }
        // depthwise -> pointwise pairs that run as one launch (bit i = Conv2d_{i+1}); option fuse_dw overrides
        // (streaming kernel: every pair in mode f32; in mode f16x3 Conv2d_5..13 keep their f16x3 pointwise products)
        // Measured with the streaming kernel (B = 32, mode f32, one box): masks 0xf / 0x1f / 0x3f / 0x1fff -> 770.7 / 769.4 /
        // 765.4 / 758.9 img/s: from Conv2d_6 on the pointwise product is MFMA-bound, the exact-fp32 MFMA and the depthwise
        // VALU work do not overlap on a SIMD, and the two-kernel pair wins.
        unsigned fuse_mask = SSD_FUSE_DW_DEFAULT;
        if (ssd_opt(h, OPT_FUSE_DW, -1) >= 0) fuse_mask = (unsigned)ssd_opt(h, OPT_FUSE_DW, -1);
        std::vector<Op> half_ops[4];
        // The chains' ping-pong buffers (2 x the largest backbone tensor per chain: 73 MB per 640 x 896 frame, a third of the
        // arena) as ONE block: nothing reads them once c3 / c4 / c5 exist, so the head towers' buffers -- whose first writer runs
        // behind the FPN, i.e. behind every backbone launch -- and after them the NMS keys are carved from the same bytes (`scratch`).
        // The next forward's backbone starts behind this forward's last kernel (one caller stream; another stream waits for
        // ev_last), and every plan has a block of its own.
        auto chain_floats = [&](int nb) {
            long long maxf = (long long)nb * h2 * w2 * h->firstCp;
            int hh = h2, ww = w2;
            for (int i = 0; i < 13; ++i) {
                hh /= MB_STRIDE[i]; ww /= MB_STRIDE[i];
                maxf = std::max(maxf, std::max((long long)nb * hh * ww * h->dw[i].Cp, (long long)nb * hh * ww * h->pw[i].CoutP));
            }
            return maxf;
        };
        {
            size_t total = 0;
            for (int hf = 0; hf < nhalf; ++hf) {
                const int nb = (int)((long long)B * (hf + 1) / nhalf) - (int)((long long)B * hf / nhalf);
                total += 2 * scratch_slot((size_t)chain_floats(nb) * sizeof(float));
            }
            SSDCHK(ap.alloc((void **)&scratch, total));
            scratch_bytes = total;
        }
        size_t scratch_used = 0;
        for (int hf = 0; hf < nhalf; ++hf) {
            const int b0 = (int)((long long)B * hf / nhalf), nb = (int)((long long)B * (hf + 1) / nhalf) - b0;
            std::vector<Op> &ops = half_ops[hf];
            const long long maxf = chain_floats(nb);
            float *X = (float *)(scratch + scratch_used), *Y = (float *)(scratch + scratch_used + scratch_slot((size_t)maxf * sizeof(float)));
            scratch_used += 2 * scratch_slot((size_t)maxf * sizeof(float));
            // first convolution + Conv2d_1 as ONE launch (front.hip) when the frames arrive at the network's input size and
            // the three layers have MobileNet-1.0's widths; option front_fuse = 0 / 1 pins it
            bool front = ident && ((fuse_mask >> 0) & 1) && h->first.mean && h->dw[0].pack &&
                         h->pw[0].taps == 1 && h->pw[0].mean && !h->pw[0].bias && front_supports(nb, H, W, h->firstCp, h->dw[0].Cp, h->pw[0].CoutP) &&
                         h->pw[0].CinP == 32 && MB_STRIDE[0] == 1;
            { const int pin = ssd_opt(h, OPT_FRONT_FUSE, -1); if (pin >= 0) front = front && pin != 0; }
            // RESIZED frames whose width does not shrink (every COCO image at min_dimension 640) take the same fused launch with the
            // gather in its loads (front.hip, GEN).  Decided per CALL: the source geometry is a launch argument and a plan serves any
            // source size that lands on its network shape, so such a plan keeps both forms -- the first-convolution op launches the
            // fused kernel and the Conv2d_1 op does nothing, or (a frame that is reduced in width) the two run as themselves.  A
            // batch of frames of DIFFERENT sizes takes the fused launch too (per-frame geometry from its arguments) when none of its
            // frames is reduced in width.  Same conditions as the static form, plus Conv2d_1 being ONE op (the fused depthwise + pointwise).
            bool front_rt = !ident && ((fuse_mask >> 0) & 1) && h->first.mean && h->dw[0].pack &&
                            h->pw[0].taps == 1 && h->pw[0].mean && !h->pw[0].bias && front_supports(nb, H, W, h->firstCp, h->dw[0].Cp, h->pw[0].CoutP) &&
                            h->pw[0].CinP == 32 && MB_STRIDE[0] == 1 && dwpws_eligible(h->dw[0], h->pw[0], nb, h2, w2, MB_STRIDE[0]);
            { const int pin = ssd_opt(h, OPT_FRONT_FUSE, -1); if (pin >= 0) front_rt = front_rt && pin != 0; }
            ssd_handle *const hrt = h;
            const int rt_first = img0 + b0;
            auto fused_now = [hrt, nb, front_rt, rt_first, H, W]() {
                if (!front_rt) return false;
                if (hrt->mixed) return front_mixed_supports(hrt->mixed->geom, rt_first, nb, H, W);      // every frame of this chain's share
                const SrcGeom &g = hrt->src;
                return front_gen_supports(nb, g.srcH, g.srcW, g.nh, g.nw);
            };
            if (!front) {
                Op op;
                op.cls = 3;
                op.flops = 2.0 * 27 * (double)nb * h2 * w2 * h->pw[0].Cin_l;
                op.bytes = (double)nb * H * W * 3 + (double)nb * h2 * w2 * h->pw[0].Cin_l * 4.0;
                ssd_handle *hh = h;
                const DwW f = h->first;
                const int act = h->firstAct;
                const int first_img = img0 + b0;
                const int variant = ssd_opt(h, OPT_FIRST_CONV_PX, 1) == 0 ? 1 : 0;
                FrontArgs fq;
                memset(&fq, 0, sizeof(fq));
                if (front_rt) {
                    const ConvW &c0 = h->pw[0];
                    fq.w0 = f.w; fq.m0 = f.mean; fq.s0 = f.sf; fq.b0 = f.beta; fq.dwpack = h->dw[0].pack;
                    fq.wt = c0.wt; fq.mean = c0.mean; fq.sf = c0.sf; fq.beta = c0.beta; fq.out = Y;      // (Conv2d_1's output: `dwo` of layer 0 below)
                    fq.B = nb; fq.H = H; fq.W = W; fq.act0 = act; fq.dact = SSD_ACT_RELU6; fq.act = SSD_ACT_RELU6;
                    fq.tiles_y = (H / 2 + front_tile_y() - 1) / front_tile_y();
                    fq.tiles_x = (W / 2 + front_tile_x() - 1) / front_tile_x();
                }
                op.run = [=](hipStream_t s) {      // the source's size and the resize's target: this call's (SrcGeom), any that lands on H x W
                    if (hh->mixed) {               // ... or every frame's own (a batch of frames of different sizes)
                        if (fused_now()) {
                            FrontArgs r = fq;
                            r.img = hh->cur_images;
                            r.mixed = &hh->mixed->geom;
                            r.mixed_first = first_img;
                            return launch_front(r, s);
                        }
                        return launch_first_conv_mixed(hh->cur_images, hh->mixed->geom, first_img, nb, H, W, f.w, f.Cp, f.mean, f.sf, f.beta, act, X, s, variant);
                    }
                    const SrcGeom &g = hh->src;
                    if (fused_now()) {
                        FrontArgs r = fq;
                        r.img = hh->cur_images + (size_t)first_img * g.srcH * g.srcW * 3;
                        r.resized = 1; r.srcH = g.srcH; r.srcW = g.srcW; r.nh = g.nh; r.nw = g.nw;
                        return launch_front(r, s);
                    }
                    return launch_first_conv(hh->cur_images + (size_t)first_img * g.srcH * g.srcW * 3, nb, g.srcH, g.srcW, g.nh, g.nw, H, W, f.w, f.Cp,
                                             f.mean, f.sf, f.beta, act, X, s, variant);
                };
                ops.push_back(op);
            }
            float *cur = X;
            int ch = h2, cwid = w2;
            for (int i = 0; i < 13; ++i) {
                const int s = MB_STRIDE[i];
                float *dwo = (cur == X) ? Y : X;
                const ConvW &cw = h->pw[i];
                if (i == 0 && front) {
                    ops.push_back(make_front_op(h, img0 + b0, h->first, h->firstAct, h->dw[0], cw, nb, H, W,
                                                SSD_ACT_RELU6, SSD_ACT_RELU6, dwo));
                    cur = dwo;
                    continue;
                }
                const bool fuse = ((fuse_mask >> i) & 1) && dwpws_eligible(h->dw[i], cw, nb, ch, cwid, s);
                // f16x3: an unfused pair hands the depthwise result (exact fp32, in [0, 6]) to its pointwise
                // convolution in split-fp16 rows, and the pointwise product runs as 3 x f16 MFMA
                const int pw16 = X16 && !fuse && (h->dw[i].Cp % 32 == 0) ? 1 : 0;
                if (!fuse) ops.push_back(make_dw_op(h->dw[i], cur, nb, ch, cwid, s, SSD_ACT_RELU6, dwo, h->pw[i].Cin_l, pw16, FL));
                const int dh = ch, dwid = cwid;
                ch /= s; cwid /= s;
                float *pwo;
                if (i == 4 || i == 10 || i == 12) {
                    float *full = i == 4 ? C3 : (i == 10 ? C4 : C5);
                    pwo = full + (long long)b0 * ch * cwid * cw.CoutP;
                } else {
                    pwo = fuse ? dwo : ((dwo == X) ? Y : X);    // fused: input `cur` is live until the launch ends
                }
                if (fuse) {
                    Op o = make_dwpws_op(h->dw[i], cw, cur, nb, dh, dwid, s, SSD_ACT_RELU6, SSD_ACT_RELU6, pwo);
                    if (i == 0 && front_rt) {        // (its work was done by the fused launch of the op before it when fused_now())
                        const auto inner = o.run;
                        o.run = [inner, fused_now](hipStream_t st) { return fused_now() ? hipSuccess : inner(st); };
                    }
                    ops.push_back(o);
                } else   // c5 feeds only the FPN (lateral5, p6): in f16x3 mode it is written in split-fp16 rows
                    ops.push_back(make_conv_op(h, cw, dwo, pwo, nullptr, nullptr, nb, 1, 0, SSD_ACT_RELU6,
                                               {dense_level(ch, cwid, ch, cwid, cw.CoutP)}, true, pw16, (i == 12 && X16) ? 1 : 0, 0, h->flags_dev));
                if (i == 10 && nhalf == 1) id_c4 = (int)ops.size() - 1;          // Conv2d_11_pointwise = c4 (one chain: pl.ops keeps this index)
                cur = pwo;
            }
        }
        // enqueue order interleaved so that both queues are fed
        for (size_t i = 0; i < half_ops[0].size(); ++i)
            for (int hf = 0; hf < nhalf; ++hf)
                if (i < half_ops[hf].size()) {
                    Op op = half_ops[hf][i];
                    op.stream = hf;
                    pl.ops.push_back(op);
                    id_bb_last[hf] = (int)pl.ops.size() - 1;
                }
    } else {
        // ---------------- ShuffleNet v2 (shufflenet_v2.py:50-69,79-137)
        const int units[3] = {4, 8, 4};
        const int fc = h->firstCp;
        // depthwise -> 1x1 pairs of the units as one launch each (option fuse_dw = 0 keeps them apart)
        bool sn_fuse = SSD_FUSE_SHUFFLE_DEFAULT;
        if (ssd_opt(h, OPT_FUSE_DW, -1) >= 0) sn_fuse = ssd_opt(h, OPT_FUSE_DW, -1) != 0;
        const int h4 = h2 / 2, w4 = w2 / 2;
        // first convolution + max pool as ONE launch (front.hip) when the frames arrive at the network's input size; option
        // front_fuse = 0 / 1 pins it
        bool sn_front = ident && h->first.mean && front_pool_supports(B, H, W, fc);
        { const int pin = ssd_opt(h, OPT_FRONT_FUSE, -1); if (pin >= 0) sn_front = sn_front && pin != 0; }
        // ---- One or two half-batch chains (two from 4 images on, on the plan's two streams, as MobileNet's backbone above: the
        // backbone is ~45 short kernels far from any bound, one chain's load / store phases sit under the other's arithmetic);
        // FPN and heads stay full-batch launches.  Option backbone_split = 1 keeps one chain.
        //
        // concat_shuffle_split (shufflenet_v2.py:94-115) and the stage concat (:89) run as NO kernel of their own:
        //   * every producer of a stage -- unit_1's two branches, unit j's conv1x1_after -- stores its D channels DENSE, one
        //     contiguous run per position, into a tensor of its own (16-byte stores from dwpw_stream.hip's accumulators, every
        //     line written whole by one launch);
        //   * unit j's conv1x1_before GATHERS its input row (sn_pw.hip): the host traces input channel k through the interleave-
        //     and-split of the reference back to (producer tensor, column) and gives the kernel that table; the k order -- and
        //     with it bit-identity with the oracle -- is untouched;
        //   * the stage output is kept in two-part rows [x half | y half] (weights.hip packs its consumers for that): the last
        //     unit stores its channels straight into the x half, the y half (channels no later unit touched) is one row gather.
        // Each stage is ONE allocation [producer tensors of chain 0 | ... of chain 1 | stage output S of the whole batch] under
        // one buffer resource (select_plans keeps it below 2 GiB).
        int nhalf = B >= 4 ? 2 : 1;
        { const int v = ssd_opt(h, OPT_BACKBONE_SPLIT, 0); if (v >= 1 && v <= 2 && v <= B) nhalf = v; }
        const int nb_of[2] = {nhalf == 2 ? B / 2 : B, nhalf == 2 ? B - B / 2 : 0};
        struct StageGeo { int ch, cw, oh, ow, D, Dp, n_units, ipw, idw; long long tbytes[2], toff[2], soff, total; };
        StageGeo geo[3];
        {
            int ch = h4, cw = w4, ipw = 0, idw = 0;
            for (int st = 0; st < 3; ++st) {
                StageGeo &g = geo[st];
                g.ch = ch; g.cw = cw; g.oh = ch / 2; g.ow = cw / 2; g.ipw = ipw; g.idw = idw; g.n_units = units[st];
                const ConvW &after = h->pw[ipw + 1];
                g.Dp = after.CoutP; g.D = after.Cout_l;
                long long off = 0;
                for (int hf = 0; hf < nhalf; ++hf) {
                    g.tbytes[hf] = (long long)nb_of[hf] * g.oh * g.ow * g.Dp * 4;       // one producer tensor of this chain
                    g.toff[hf] = off;
                    off += g.tbytes[hf] * g.n_units;                                  // x1, y1, o_2 .. o_{n-1}
                }
                g.soff = off;
                g.total = off + (long long)B * g.oh * g.ow * 2 * g.Dp * 4;
                if (g.total >= (1LL << 31)) return ssd_fail(SSD_ERR_INVALID, "ssd_forward: ShuffleNet stage allocation past 2 GiB (sub-batch split failed)");
                for (int hf = 0; hf < nhalf; ++hf)
                    for (int j = 2; j <= g.n_units; ++j) {
                        const ConvW &b2 = h->pw[ipw + 3 + 2 * (j - 2)];
                        if (b2.CinP != g.Dp || b2.taps != 1 || !b2.mean || b2.bias ||
                            !pw_gather_supports(b2.CinP, b2.CoutP, (long long)nb_of[hf] * g.oh * g.ow, g.Dp * 4, g.total, (long long)nb_of[hf] * g.oh * g.ow * b2.CoutP * 4))
                            return ssd_fail(SSD_ERR_INVALID, "ssd_forward: a ShuffleNet unit's conv1x1_before is not a shape of the gathering kernel (sn_pw.hip)");
                    }
                ipw += 3 + 2 * (g.n_units - 1); idw += 2 + (g.n_units - 1);
                ch = g.oh; cw = g.ow;
            }
        }
        float *stage[3];
        for (int st = 0; st < 3; ++st) SSDCHK(falloc(&stage[st], geo[st].total / 4));
        const StageGeo &g2 = geo[2];
        const ConvW &c5w = h->pw[g2.ipw + 3 + 2 * (g2.n_units - 1)];
        SSDCHK(falloc(&C5, (long long)B * g2.oh * g2.ow * c5w.CoutP));
        std::vector<Op> half_ops[2];
        for (int hf = 0; hf < nhalf; ++hf) {
            const int b0 = hf == 0 ? 0 : nb_of[0], nb = nb_of[hf];
            std::vector<Op> &ops = half_ops[hf];
            float *F = nullptr, *MP, *MID = nullptr;
            SSDCHK(falloc(&MP, (long long)nb * h4 * w4 * fc));
            // depthwise -> 1x1 (+ batch norms, ReLU behind the 1x1) into dense rows [M][out_rs]: one launch of the streaming kernel, or
            // (option fuse_dw = 0, shapes it does not take) the depthwise kernel and the implicit-GEMM kernel with a tensor between them
            auto pair = [&](const DwW &d, const ConvW &cw, const float *in, int hh, int ww, int stride, float *out, int out_rs) -> int {
                if (sn_fuse && dwpws_eligible(d, cw, nb, hh, ww, stride)) {
                    ops.push_back(make_dwpws_op(d, cw, in, nb, hh, ww, stride, SSD_ACT_NONE, SSD_ACT_RELU, out, out_rs));
                    return SSD_OK;
                }
                if (!MID) SSDCHK(falloc(&MID, (long long)nb * h4 * w4 * std::max(fc, 32)));       // (the largest depthwise output: Stage2 unit_1 reads h4 x w4)
                const int oh = hh / stride, ow = ww / stride;
                if ((long long)nb * oh * ow * d.Cp > (long long)nb * h4 * w4 * std::max(fc, 32)) return ssd_fail(SSD_ERR_INVALID, "ssd_forward: depthwise scratch too small");
                ops.push_back(make_dw_op(d, in, nb, hh, ww, stride, SSD_ACT_NONE, MID, cw.Cin_l));
                LevelDesc lv = dense_level(oh, ow, oh, ow, cw.CoutP);
                lv.out_rstride = out_rs > 0 ? out_rs : cw.CoutP;
                lv.out_bstride = (long long)oh * ow * lv.out_rstride;
                ops.push_back(make_conv_op(h, cw, MID, out, nullptr, nullptr, nb, 1, 0, SSD_ACT_RELU, {lv}, true));
                return SSD_OK;
            };
            {
                ssd_handle *hh = h;
                const DwW f = h->first;
                const int act = h->firstAct;
                const int first_img = img0 + b0;
                Op op;
                op.cls = 3;
                op.flops = 2.0 * 27 * (double)nb * h2 * w2 * 24;
                if (sn_front) {
                    // first convolution + max pool as one launch (front.hip): the half-resolution tensor stays in LDS
                    op.bytes = (double)nb * H * W * 3 + (double)nb * h4 * w4 * 24 * 4.0;
                    op.run = [=](hipStream_t s) { return launch_front_pool(hh->cur_images + (size_t)first_img * H * W * 3, nb, H, W, f.w, f.Cp, f.mean, f.sf, f.beta, act, MP, s); };
                    ops.push_back(op);
                } else {
                    SSDCHK(falloc(&F, (long long)nb * h2 * w2 * fc));
                    op.bytes = (double)nb * H * W * 3 + (double)nb * h2 * w2 * 24 * 4.0;
                    const int variant = ssd_opt(h, OPT_FIRST_CONV_PX, 1) == 0 ? 1 : 0;
                    // resized frames whose width does not shrink: the fused launch with the gather in its loads, chosen per call (as
                    // MobileNet's above): this op then writes the pooled tensor and the max-pool op does nothing
                    bool front_rt = !ident && h->first.mean && front_pool_supports(nb, H, W, fc);
                    { const int pin = ssd_opt(h, OPT_FRONT_FUSE, -1); if (pin >= 0) front_rt = front_rt && pin != 0; }
                    auto fused_now = [hh, nb, front_rt, first_img, H, W]() {
                        if (!front_rt) return false;
                        if (hh->mixed) return front_mixed_supports(hh->mixed->geom, first_img, nb, H, W);
                        const SrcGeom &g = hh->src;
                        return front_gen_supports(nb, g.srcH, g.srcW, g.nh, g.nw);
                    };
                    op.run = [=](hipStream_t s) {
                        if (hh->mixed) {
                            if (fused_now())
                                return launch_front_pool(hh->cur_images, nb, H, W, f.w, f.Cp, f.mean, f.sf, f.beta, act, MP, s, nullptr, &hh->mixed->geom, first_img);
                            return launch_first_conv_mixed(hh->cur_images, hh->mixed->geom, first_img, nb, H, W, f.w, f.Cp, f.mean, f.sf, f.beta, act, F, s, variant);
                        }
                        const SrcGeom &g = hh->src;
                        if (fused_now()) {
                            const int src[4] = {g.srcH, g.srcW, g.nh, g.nw};
                            return launch_front_pool(hh->cur_images + (size_t)first_img * g.srcH * g.srcW * 3, nb, H, W, f.w, f.Cp, f.mean, f.sf, f.beta, act, MP, s, src);
                        }
                        return launch_first_conv(hh->cur_images + (size_t)first_img * g.srcH * g.srcW * 3, nb, g.srcH, g.srcW, g.nh, g.nw, H, W, f.w, f.Cp,
                                                 f.mean, f.sf, f.beta, act, F, s, variant);
                    };
                    ops.push_back(op);
                    Op mp;
                    mp.cls = 5; mp.flops = 0;
                    mp.bytes = ((double)nb * h2 * w2 + (double)nb * h4 * w4) * 24 * 4.0;
                    mp.run = [=](hipStream_t s) { return fused_now() ? hipSuccess : launch_maxpool(F, nb, h2, w2, fc, MP, s); };
                    ops.push_back(mp);
                }
            }
            const float *cur = MP;
            for (int st = 0; st < 3; ++st) {
                const StageGeo &g = geo[st];
                const ConvW &before = h->pw[g.ipw], &after = h->pw[g.ipw + 1], &after2 = h->pw[g.ipw + 2];
                const DwW &d1 = h->dw[g.idw], &d2 = h->dw[g.idw + 1];
                const int D = g.D, Dp = g.Dp, n_units = g.n_units;
                const long long rows = (long long)nb * g.oh * g.ow;
                float *t1, *U;
                SSDCHK(falloc(&t1, (long long)nb * g.ch * g.cw * before.CoutP));
                SSDCHK(falloc(&U, rows * Dp));
                float *sb = stage[st];
                // producer p's tensor of this chain (0: unit_1's second branch = x, 1: its main branch = y, j: unit j's output)
                auto tensor_off = [&](int p) { return g.toff[hf] + (long long)p * g.tbytes[hf]; };
                // the chain's rows of S start b0 images into the stage output
                float *S_chain = sb + g.soff / 4 + (long long)b0 * g.oh * g.ow * 2 * Dp;
                struct Src { int prod, col; };
                std::vector<Src> x(D), y(D);
                for (int d = 0; d < D; ++d) { x[d] = Src{0, d}; y[d] = Src{1, d}; }
                auto table = [&](const std::vector<Src> &v) {      // physical channel p of a D-channel row -> byte offset of its source
                    std::vector<int> t(Dp, -1);
                    for (int d = 0; d < D; ++d) t[ssd_phys_of_logical(d)] = (int)(tensor_off(v[d].prod) + (long long)ssd_phys_of_logical(v[d].col) * 4);
                    return t;
                };
                std::vector<const int *> src_dev(n_units + 1, nullptr);
                for (int j = 2; j <= n_units; ++j) {
                    std::vector<Src> z(2 * D);
                    for (int d = 0; d < D; ++d) { z[2 * d] = x[d]; z[2 * d + 1] = y[d]; }
                    std::vector<Src> xin(z.begin(), z.begin() + D);
                    int *dv;
                    SSDCHK(ap.upload(&dv, table(xin)));
                    src_dev[j] = dv;
                    for (int d = 0; d < D; ++d) { x[d] = Src{j, d}; y[d] = z[D + d]; }
                }
                int *ysrc;
                SSDCHK(ap.upload(&ysrc, table(y)));          // the stage output's y half (x = unit n's own channels)
                ops.push_back(make_conv_op(h, before, cur, t1, nullptr, nullptr, nb, 1, 0, SSD_ACT_RELU,
                                           {dense_level(g.ch, g.cw, g.ch, g.cw, before.CoutP)}, true));
                SSDCHK(pair(d1, after, t1, g.ch, g.cw, 2, sb + tensor_off(1) / 4, 0));
                SSDCHK(pair(d2, after2, cur, g.ch, g.cw, 2, sb + tensor_off(0) / 4, 0));
                for (int j = 2; j <= n_units; ++j) {
                    const ConvW &b2 = h->pw[g.ipw + 3 + 2 * (j - 2)], &a2 = h->pw[g.ipw + 3 + 2 * (j - 2) + 1];
                    const DwW &dd = h->dw[g.idw + j];
#ifdef SSD_DIAG
                    // ablation (scripts/experiments/sn_unit_fusion_bound.py): drop conv1x1_before of the stages in the mask -- WRONG
                    // results, the time of a step whose first 1x1 of every unit is free: what no whole-unit fusion can beat
                    if (const char *e = getenv("SSD_ABL_SKIP_PWG")) { if ((atoi(e) >> st) & 1) goto skip_before; }
#endif
                    ops.push_back(make_pw_gather_op(b2, sb, g.total, src_dev[j], Dp * 4, rows, SSD_ACT_RELU, U));
#ifdef SSD_DIAG
                skip_before:
#endif
                    if (j < n_units) SSDCHK(pair(dd, a2, U, g.oh, g.ow, 1, sb + tensor_off(j) / 4, 0));
                    else SSDCHK(pair(dd, a2, U, g.oh, g.ow, 1, S_chain, 2 * Dp));
                }
                {
                    Op gop;
                    gop.cls = 5; gop.flops = 0; gop.bytes = 2.0 * rows * D * 4.0;
                    const int rs = Dp * 4, ors = 2 * Dp;
                    float *ydst = S_chain + Dp;
                    gop.run = [=](hipStream_t s) { return launch_gather_rows(sb, ysrc, rs, rows, Dp, ydst, ors, s); };
                    ops.push_back(gop);
                }
                cur = S_chain;
            }
            ops.push_back(make_conv_op(h, c5w, cur, C5 + (long long)b0 * g2.oh * g2.ow * c5w.CoutP, nullptr, nullptr, nb, 1, 0, SSD_ACT_RELU,
                                       {dense_level(g2.oh, g2.ow, g2.oh, g2.ow, c5w.CoutP)}, true, 0, X16, 0, FL));
        }
        for (int st = 0; st < 2; ++st) {
            float *S = stage[st] + geo[st].soff / 4;
            if (st == 0) C3 = S; else C4 = S;
            pl.retained[st == 0 ? "c3" : "c4"] = Retained{S, B, geo[st].oh, geo[st].ow, 2 * geo[st].D, 2 * geo[st].Dp, true, 0, geo[st].D};
        }
        pl.retained["c5"] = Retained{C5, B, g2.oh, g2.ow, c5w.Cout_l, c5w.CoutP, true, X16};
        for (size_t i = 0; i < std::max(half_ops[0].size(), half_ops[1].size()); ++i)
            for (int hf = 0; hf < nhalf; ++hf)
                if (i < half_ops[hf].size()) {
                    Op op = half_ops[hf][i];
                    op.stream = hf;
                    pl.ops.push_back(op);
                    id_bb_last[hf] = (int)pl.ops.size() - 1;
                }
    }

    // ---------------- FPN (feature_extractor.py:40-76)
    const Pyr py = make_pyr(B, H, W, 256);
    float *P, *X5, *X4, *X3, *T6;
    SSDCHK(falloc(&P, py.total));
    SSDCHK(falloc(&X5, (long long)B * py.h[2] * py.w[2] * 256));
    SSDCHK(falloc(&X4, (long long)B * py.h[1] * py.w[1] * 256));
    SSDCHK(falloc(&X3, (long long)B * py.h[0] * py.w[0] * 256));
    SSDCHK(falloc(&T6, (long long)B * py.h[3] * py.w[3] * 256));
    auto lvl = [&](int l, int CoutP) { return dense_level(py.h[l], py.w[l], py.h[l], py.w[l], CoutP); };
    // Three streams, explicit dependencies.  Serving batches: main: lateral5 -> lateral4 (+up) -> lateral3 (+up) -> p3 (the
    // critical path); second stream: p5 (needs x5) -> p4 (needs x4); third stream: p6 -> p7 (need only c5).
    // All of them are the same 3x3 kernel, and two such kernels side by side fill each other's tails.
    auto push = [&](Op op, int stream, std::vector<int> deps = {}) {
        op.stream = stream;
        std::sort(deps.begin(), deps.end());                     // (one wait per producer: with p7 inside the grouped launch the towers
        deps.erase(std::unique(deps.begin(), deps.end()), deps.end());      //  name that launch twice)
        op.deps = deps;
        pl.ops.push_back(op);
        return (int)pl.ops.size() - 1;
    };
    // last backbone op on the main stream (produces c5, or its first half); the second half, if any, ends on the
    // second stream: the main stream's first FPN op waits for it
    const int id_c5 = id_bb_last[0] >= 0 ? id_bb_last[0] : (int)pl.ops.size() - 1;
    // Batch 1-2 in exact fp32: p3, p4, p5 (the same 3x3 256 -> 256 + batch norm + ReLU on x3, x4, x5) and p7 as ONE launch
    // behind lateral3, each level with its own kernel (IgemmLevel::wt_off into h->pgroup) and batch norm: 736 tiles -- a
    // tower-sized launch -- instead of three launches that stretch each other.  Option fpn_group = 0 / 1 pins it.
    bool grouped = false;
    if (!X16 && h->pgroup.wt) {
        const long long b64 = (((long long)B * py.h[0] * py.w[0] + 63) / 64) * (256 / 64);
        grouped = b64 <= 640;
        const int pin = ssd_opt(h, OPT_FPN_GROUP, -1);
        if (pin >= 0) grouped = pin != 0;
    }
    // ... and then p6 (-> p7) runs on the MAIN stream right behind c5 with the lateral chain on the third stream beside it: the
    // grouped launch waits for lateral3 (done long before p6) and nothing waits for p6 across streams.
    const bool swap67 = grouped && B <= 2;
    // fpn p7 (3x3 stride 2 on ReLU(p6 pre-BN), 35 positions per image) as a fourth level of the grouped launch behind p6: its
    // launch and one kernel boundary leave the critical path c5 -> p6 -> p7 -> grouped -> towers.  Same k-ordered chain, same bits.
    const bool p7grouped = swap67 && ssd_opt(h, OPT_FPN_P7_GROUP, 1) != 0;
    const int s_lat = swap67 ? 2 : 0;
    // Batch <= 2 (swap67), one backbone chain: lateral4(c4) and lateral3(c3) -- WITHOUT their upsampled operands -- on the third
    // stream as soon as c4 exists (c3 precedes it on the caller's stream), beside the backbone's last four layers; behind c5
    // the third stream then runs lateral5 and ONE elementwise launch for both top-down sums (fpn_merge_kernel) instead of a
    // chain of three convolutions.  Same additions, same bits.  Option fpn_early_lat = 0 / 1 pins it.
    bool early = swap67 && !X16 && id_c4 >= 0;
    { const int pin = ssd_opt(h, OPT_FPN_EARLY_LAT, -1); if (pin >= 0) early = early && pin != 0; }
    float *L4T = nullptr;
    if (early) {
        SSDCHK(falloc(&L4T, (long long)B * py.h[1] * py.w[1] * 256));
        push(make_conv_op(h, h->lat[0], C3, X3, nullptr, nullptr, B, 1, 0, SSD_ACT_NONE, {lvl(0, 256)}, true), s_lat, {id_c4});
        push(make_conv_op(h, h->lat[1], C4, L4T, nullptr, nullptr, B, 1, 0, SSD_ACT_NONE, {lvl(1, 256)}, true), s_lat);
    }
    std::vector<int> l5_deps;
    for (int c = 1; c < 4; ++c) if (id_bb_last[c] >= 0) l5_deps.push_back(id_bb_last[c]);
    if (swap67) l5_deps.push_back(id_c5);
    const int id_l5 = push(make_conv_op(h, h->lat[2], C5, X5, nullptr, nullptr, B, 1, 0, SSD_ACT_NONE, {lvl(2, 256)}, true, X16, X16, 0, FL), s_lat, l5_deps);
    std::vector<int> p6_deps = {id_c5};             // c5 of every backbone chain that is not on p6's own stream
    for (int c = 1; c < 4; ++c) if (c != 2 && id_bb_last[c] >= 0) p6_deps.push_back(id_bb_last[c]);
    int id_p7;
    {   // p6 = conv s2 (c5): BN+ReLU -> P6, ReLU(raw) -> T6 (input of p7, :60)
        LevelDesc d = dense_level(py.h[2], py.w[2], py.h[3], py.w[3], 256);
        d.out_off = py.off[3];
        push(make_conv_op(h, h->pconv[3], C5, P, T6 - py.off[3], nullptr, B, 2, 1, SSD_ACT_RELU, {d}, true, X16, X16, 0, FL), swap67 ? 0 : 2, p6_deps);
        LevelDesc d7 = dense_level(py.h[3], py.w[3], py.h[4], py.w[4], 256);
        d7.out_off = py.off[4];
        if (!p7grouped) id_p7 = push(make_conv_op(h, h->pconv[4], T6, P, nullptr, nullptr, B, 2, 1, SSD_ACT_RELU, {d7}, true, X16, X16, 0, FL), swap67 ? 0 : 2);
        else id_p7 = -1;
    }
    if (!grouped) {   // p5 = conv(x5)
        LevelDesc d = lvl(2, 256);
        d.out_off = py.off[2];
        push(make_conv_op(h, h->pconv[2], X5, P, nullptr, nullptr, B, 1, 1, SSD_ACT_RELU, {d}, true, X16, X16, 0, FL), 1, {id_l5});
    }
    // x4 = up(x5) + lateral4(c4); p4;  x3 = up(x4) + lateral3(c3); p3
    // lateral4 / lateral3 read c4 / c3, which stay fp32 rows for the depthwise layer that also consumes them: in
    // f16x3 mode the rows are split into halves while they are staged (in_fmt 2); the upsampled operand and the
    // output follow the mode
    int LF = X16 && h->lat[1].tile == IGEMM_128x128 && h->lat[0].tile == IGEMM_128x128 ? 2 : 0;
    if (!ssd_opt(h, OPT_LATERAL_SPLIT, 1)) LF = 0;       // A/B runs: 0 keeps them on the exact MFMA
    int id_l4, id_merge = -1;
    if (early) {
        Op m;
        m.cls = 5; m.flops = 0;
        m.bytes = ((double)B * py.h[0] * py.w[0] * 2 + (double)B * py.h[1] * py.w[1] * 2 + (double)B * py.h[2] * py.w[2]) * 256 * 4.0;
        const int mh = py.h[0], mw = py.w[0];
        float *l4t = L4T;
        m.run = [=](hipStream_t s) { return launch_fpn_merge(X5, l4t, X4, X3, B, mh, mw, 256, s); };
        id_l4 = id_merge = push(m, s_lat);
    } else {
        id_l4 = push(make_conv_op(h, h->lat[1], C4, X4, nullptr, X5, B, 1, 0, SSD_ACT_NONE, {lvl(1, 256)}, true, LF, X16, X16, FL), s_lat);
    }
    int id_p4, id_p3;
    if (!grouped) {
        LevelDesc d = lvl(1, 256);
        d.out_off = py.off[1];
        id_p4 = push(make_conv_op(h, h->pconv[1], X4, P, nullptr, nullptr, B, 1, 1, SSD_ACT_RELU, {d}, true, X16, X16, 0, FL), 1, {id_l4});
    }
    const int id_l3 = early ? id_merge
                            : push(make_conv_op(h, h->lat[0], C3, X3, nullptr, X4, B, 1, 0, SSD_ACT_NONE, {lvl(0, 256)}, true, LF, X16, X16, FL), s_lat);
    if (!grouped) {
        LevelDesc d = lvl(0, 256);
        d.out_off = py.off[0];
        id_p3 = push(make_conv_op(h, h->pconv[0], X3, P, nullptr, nullptr, B, 1, 1, SSD_ACT_RELU, {d}, true, X16, X16, 0, FL), 0);
    } else {
        const ConvW &g = h->pgroup;
        const float *xin[3] = {X3, X4, X5};
        std::vector<LevelDesc> lv3;
        for (int l = 0; l < 3; ++l) {
            LevelDesc d = lvl(l, 256);
            d.in_off = xin[l] - X3;                         // the three inputs are separate allocations: offsets from x3
            d.out_off = py.off[l];
            d.param_off = l * g.CoutP;
            d.wt_off = (long long)l * g.taps * g.CoutPad * g.CinP;
            lv3.push_back(d);
        }
        if (p7grouped) {
            LevelDesc d = dense_level(py.h[3], py.w[3], py.h[4], py.w[4], 256);
            d.in_off = T6 - X3;
            d.out_off = py.off[4];
            d.param_off = 3 * g.CoutP;
            d.wt_off = (long long)3 * g.taps * g.CoutPad * g.CinP;
            d.stride = 2; d.pad = 1;
            lv3.push_back(d);
        }
        std::vector<int> gdeps;
        if (swap67) gdeps.push_back(id_l3);
        id_p3 = id_p4 = push(make_conv_op(h, g, X3, P, nullptr, nullptr, B, 1, 1, SSD_ACT_RELU, lv3, true), 0, gdeps);
        if (p7grouped) id_p7 = id_p3;
    }
    for (int l = 0; l < 5; ++l) {
        char nm[8];
        snprintf(nm, sizeof nm, "p%d", l + 3);
        pl.retained[nm] = Retained{P + py.off[l], B, py.h[l], py.w[l], 256, 256, true, X16};
    }

    // ---------------- heads (box_predictor.py:36-155), all levels per launch; box tower on the
    // main stream, class tower on the second stream (independent chains)
    const int C = h->cfg.num_classes, A = 6;
    long long N = 0, aoff[5];
    for (int l = 0; l < 5; ++l) { aoff[l] = N; N += (long long)py.h[l] * py.w[l] * A; }
    pl.N = (int)N;
    float *logits, *codes;
    SSDCHK(falloc(&logits, (long long)B * N * C));
    SSDCHK(falloc(&codes, (long long)B * N * 4));
    // ---------------- anchors + post-processing
    std::vector<float> anc((size_t)N * 4);
    SSDCHK(ssd_anchors(H, W, anc.data()));
    float *anc_dev;
    SSDCHK(ap.upload(&anc_dev, anc));
    // The head towers' four ping-pong buffers as one block: inside the backbone's block when they fit (MobileNet: they do), else
    // a block of their own (ShuffleNet, whose backbone tensors are stage allocations the FPN still reads).  Either way the block is
    // dead once the two final head convolutions are done -- and the post-processing starts behind both of them (ev_join) -- so the
    // NMS candidate keys [B][C][N] u64, 46 of the workspace's 50 MB per frame and the one part of it with no state between
    // forwards, take their place (when they fit: up to 85 classes).
    float *TAB[2][2];
    unsigned char *keys_home = nullptr;
    {
        const size_t slot = scratch_slot((size_t)py.total * sizeof(float));
        unsigned char *tb = scratch;
        size_t tb_bytes = scratch_bytes;
        if (!scratch || 4 * slot > scratch_bytes) {
            SSDCHK(ap.alloc((void **)&tb, 4 * slot));
            tb_bytes = 4 * slot;
        }
        for (int t = 0; t < 2; ++t)
            for (int k = 0; k < 2; ++k) TAB[t][k] = (float *)(tb + (size_t)(2 * t + k) * slot);
        if (post_keys_bytes(B, (int)N, C) <= tb_bytes) keys_home = tb;
    }
    void *ws;
    const size_t wsb = post_workspace_bytes(B, (int)N, C, h->cfg.max_boxes_per_class, keys_home == nullptr);
    SSDCHK(ap.alloc(&ws, wsb));
    PostArgs &p = pl.post;
    memset(&p, 0, sizeof(p));
    p.logits = logits; p.codes = codes; p.anchors = anc_dev;
    p.B = B; p.N = (int)N; p.C = C;
    p.score_thr = h->cfg.score_threshold; p.iou_thr = h->cfg.iou_threshold;
    p.max_per_class = h->cfg.max_boxes_per_class;
    p.fast_max = nms_fast_max(h);
    for (int k = 0; k < 4; ++k) p.box_scaler[k] = 1.0f;               // (model.py:67-68: this call's, set by enqueue_forward)
    post_carve(p, ws, keys_home);
    HIPCHK(hipMemset(p.scan_bits, 0, post_scan_bitmap_bytes(B, (int)N, C)));
    HIPCHK(hipMemset(p.counts, 0, (size_t)B * C * sizeof(int)));      // the post-processing kernels leave these zeroed again
    p.self_clean = 1;
    std::vector<Op> tower_ops[2];
    bool all_marked = true;
    for (int t = 0; t < 2; ++t) {
        const float *in = P;
        int cur = 0;
        for (int i = 0; i < 4; ++i) {
            std::vector<LevelDesc> lv;
            for (int l = 0; l < 5; ++l) lv.push_back(dense_level(py.h[l], py.w[l], py.h[l], py.w[l], 256, py.off[l], py.off[l], l * 256));
            tower_ops[t].push_back(make_conv_op(h, h->tower[t][i], in, TAB[t][cur], nullptr, nullptr, B, 1, 1, SSD_ACT_RELU, lv, true, X16, X16, 0, FL));
            in = TAB[t][cur];
            cur ^= 1;
        }
        const int per = t == 0 ? 4 : C;     // values per anchor
        // class logits: the convolution's epilogue also marks the octets that hold a candidate (p.scan_bits) and
        // post_scan_kernel reads the bitmap instead of all logits
        const bool can_mark = t == 1 && ((long long)N * C) % 8 == 0 && (6 * C) % 8 == 0;
        std::vector<LevelDesc> lv;
        for (int l = 0; l < 5; ++l) {
            LevelDesc d = dense_level(py.h[l], py.w[l], py.h[l], py.w[l], 0, py.off[l]);
            d.out_off = aoff[l] * per;
            d.out_bstride = N * per;
            d.out_rstride = A * per;
            d.param_off = 0;
            lv.push_back(d);
        }
        bool marked = false;
        Op fop = make_conv_op(h, h->final_[t], in, t == 0 ? codes : logits, nullptr, nullptr, B, 1, 1, SSD_ACT_NONE, lv, false, X16, 0, 0, FL,
                              can_mark ? p.scan_bits : nullptr, conservative_logit_bound(h->cfg.score_threshold), &marked);
        if (t == 1) all_marked = all_marked && marked;
        tower_ops[t].push_back(fop);
    }
    p.scan_fused = all_marked ? 1 : 0;
    // enqueue order interleaved so the hardware queues stay fed.  The first box-tower layer (main) needs p4, p5 from the
    // second stream and p6, p7 from the third; the first class-tower layer (second stream) needs p3 from the main stream
    // (and p6, p7).  The box head (24 of 32 columns) runs beside the class logits.
    for (size_t i = 0; i < tower_ops[0].size(); ++i)
        for (int t = 1; t >= 0; --t) {
            std::vector<int> deps;
            if (i == 0) { deps.push_back(t == 0 ? id_p4 : id_p3); deps.push_back(id_p7); }
            push(tower_ops[t][i], t, deps);
        }
    {   // does a chain start on an internal stream without a dependency (the second backbone chain from 4 images on)?
        bool seen[4] = {true, false, false, false};
        for (const Op &op : pl.ops) {
            if (!seen[op.stream] && op.deps.empty()) pl.need_begin = true;
            seen[op.stream] = true;
        }
    }
    // events for every op another stream waits on
    for (const Op &op : pl.ops)
        for (int d : op.deps)
            if (!pl.ops[d].done) HIPCHK(hipEventCreateWithFlags(&pl.ops[d].done, ssd_sync_event_flags(h)));
    pl.retained["encoded_boxes"] = Retained{codes, B, 1, (int)N, 4, 4, false};
    pl.retained["class_predictions"] = Retained{logits, B, 1, (int)N, C, C, false};

    return SSD_OK;
}

float conservative_logit_bound(float thr)
{
    if (!(thr > 0.0f)) return -INFINITY;
    if (!(thr < 1.0f)) return INFINITY;
    const double l = log((double)thr / (1.0 - (double)thr));
    return (float)(l - 1e-3 * (1.0 + fabs(l)));
}

static hipError_t pool_event(ssd_handle *h, hipEvent_t *e)
{
    if (!h->ev_pool.empty()) { *e = h->ev_pool.back(); h->ev_pool.pop_back(); return hipSuccess; }
    return hipEventCreate(e);
}

static hipError_t run_op(ssd_handle *h, const Op &op, hipStream_t s)
{
    if (ssd_opt(h, OPT_DEBUG_SYNC, 0)) {      // fault localisation: announce every op, run it alone, wait for it
        fprintf(stderr, "[ssd] op class %d stream %d flops %.3g bytes %.3g ...", op.cls, op.stream, op.flops, op.bytes);
        fflush(stderr);
        (void)hipDeviceSynchronize();
        const auto t0 = std::chrono::steady_clock::now();
        hipError_t r = op.run(s);
        hipError_t r2 = hipDeviceSynchronize();
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        fprintf(stderr, " %s  %.1f us alone (launch + wait included): %.1f TFLOP/s, %.2f TB/s\n", r == hipSuccess && r2 == hipSuccess ? "ok" : "FAILED",
                us, op.flops / us * 1e-6, op.bytes / us * 1e-6);
        return r != hipSuccess ? r : r2;
    }
    if (!h->profiling) return op.run(s);
    EvPair e;
    e.cls = op.cls;
    e.fwd = (int)h->ref_evs.size() - 1;
    hipError_t r = pool_event(h, &e.a);
    if (r != hipSuccess) return r;
    r = pool_event(h, &e.b);
    if (r != hipSuccess) return r;
    (void)hipEventRecord(e.a, s);
    r = op.run(s);
    (void)hipEventRecord(e.b, s);
    h->evs.push_back(e);
    h->acc_flops[op.cls] += op.flops;
    h->acc_bytes[op.cls] += op.bytes;
    h->acc_n[op.cls] += 1;
    return r;
}

// The plans' internal streams (one set per device and process, build_planset below).  HIP streams share a small pool of hardware queues (4 per process and priority level by
// default), each new stream joining the least-loaded one, so WHICH queue the class tower's stream gets depends on what the
// process created before: on the caller's queue it runs behind the box tower instead of beside it (bench.py under
// torch.distributed, RCCL's streams first: 788 instead of 822 img/s; a second engine in one process: batch-1 forward +30 us).
// The streams are created once per process, with the first plan; a process that wants the clean mapping creates its first
// engine (and runs one forward) before other stream-creating libraries -- bench.py does, INTEGRATION.md section 2.
// (Measured and not adopted, profiles/r03_batch1_option_ab.log: streams of the highest priority, whose queues come from a pool
// of their own -- robust against what the framework created, but a second engine's batch-1 forward took 2.4 ms instead of 1.64;
// a CU-masked stream is a BLOCKING stream and would serialise with the legacy default stream.)
// The arena budget of a handle's cached plans: option plan_cache_mb, default a quarter of the device's memory (a batch-1 plan of
// MobileNet at 640 x 1024 holds ~0.16 GB, a 32-image one ~4.9 GB: the COCO mix of a dozen network shapes at batch 1 is ~2 GB).
size_t plan_cache_limit_bytes(const ssd_handle *h)
{
    const int mb = ssd_opt(h, OPT_PLAN_CACHE_MB, 0);
    if (mb > 0) return (size_t)mb << 20;
    static std::mutex mu;
    static std::map<int, size_t> total;                 // device -> a quarter of its memory (asked once per device)
    std::lock_guard<std::mutex> lk(mu);
    auto it = total.find(h->cfg.device);
    if (it != total.end()) return it->second;
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) != hipSuccess) { (void)hipGetLastError(); tot = (size_t)64 << 30; }
    return total[h->cfg.device] = tot / 4;
}

// Evicts least-recently-used sets until the cache fits its budget (and holds at most 64 sets); `keep` (the set about to run) stays
// whatever it weighs.  An eviction frees device memory the GPU may still be reading: the device is drained first -- the only
// device-wide wait of the forward path, and only behind a cache MISS that overflowed the budget.
int trim_plan_cache(ssd_handle *h, const PlanSet *keep)
{
    const size_t limit = plan_cache_limit_bytes(h);
    bool drained = false;
    for (;;) {
        size_t total = 0;
        for (const PlanSet *ps : h->cache) total += ps->bytes;
        if ((total <= limit && h->cache.size() <= 64) || h->cache.size() <= 1) break;
        size_t victim = h->cache.size();
        for (size_t i = 0; i < h->cache.size(); ++i)
            if (h->cache[i] != keep && (victim == h->cache.size() || h->cache[i]->last_use < h->cache[victim]->last_use)) victim = i;
        if (victim == h->cache.size()) break;
        if (!drained) { HIPCHK(hipDeviceSynchronize()); drained = true; }
        if (h->cur == h->cache[victim]) h->cur = nullptr;
        free_planset(h->cache[victim]);
        h->cache.erase(h->cache.begin() + victim);
        h->cache_evictions += 1;
    }
    return SSD_OK;
}

static int build_planset(ssd_handle *h, PlanSet &ps)
{
    const PlanKey &key = ps.key;
    int img0 = 0;
    for (int k = 0; k < key.nsub; ++k) {
        const int bk = key.B / key.nsub + (k < key.B % key.nsub ? 1 : 0);
        Plan *pl = new Plan();
        ps.plans.push_back(pl);
        {
            // ONE set of internal streams per device and process, created by the first plan that needs them and shared by every
            // handle and every plan after it (a handle's ops stay ordered per stream and across streams by their events;
            // two handles that run at the same time merely take turns on them): a second engine -- bench.py's ShuffleNet leg behind
            // the MobileNet one -- runs on the hardware queues the first one got, instead of on whatever is least loaded by then
            // (measured: 1 113 instead of 1 200 img/s when its class-tower stream landed on the caller's queue).
            static std::mutex pool_mu;
            static std::map<int, std::vector<hipStream_t>> pool;       // device -> [second, third, fourth stream]
            std::lock_guard<std::mutex> lk(pool_mu);
            std::vector<hipStream_t> &pv = pool[h->cfg.device];
            if (pv.size() < 3) pv.resize(3, nullptr);
            for (int i = 0; i < 3; ++i)
                if (!pv[i]) HIPCHK(hipStreamCreateWithFlags(&pv[i], hipStreamNonBlocking));
            pl->s_aux = pv[0];
            pl->s_bb[0] = pv[1];
            pl->s_bb[1] = pv[2];
        }
        HIPCHK(hipEventCreateWithFlags(&pl->ev_join, ssd_sync_event_flags(h)));
        HIPCHK(hipEventCreateWithFlags(&pl->ev_begin, ssd_sync_event_flags(h)));
        SSDCHK(build_plan(h, *pl, bk, key.netH, key.netW, key.ident != 0, img0));
        img0 += bk;
        ps.bytes += pl->pool.bytes;
    }
    return SSD_OK;
}

// h->cur = the plans of source frames [B, H, W, 3].  Key: what the NETWORK sees -- B, the resized + padded size, whether the source
// already has it -- so 480 x 640 and 375 x 500 (both -> 640 x 896) share one set, and a mix of image sizes through one handle
// (inference/evaluate_on_COCO.ipynb:125-150) re-plans nothing after its first pass.  A HIT touches no HIP API at all; a MISS
// allocates and fills a new arena beside the others (no device-wide wait: the running forwards use their own arenas) and may
// then evict (trim_plan_cache).
int select_plans(ssd_handle *h, int B, int H, int W)
{
    const ResizeDims rd = resize_dims(H, W, h->cfg.min_dimension, 128);
    const int netH = rd.nh + rd.ph, netW = rd.nw + rd.pw;
    const int ident = (H == netH && W == netW && rd.nh == H && rd.nw == W) ? 1 : 0;
    h->src.srcH = H; h->src.srcW = W; h->src.nh = rd.nh; h->src.nw = rd.nw;
    for (int k = 0; k < 4; ++k) h->src.box_scaler[k] = rd.box_scaler[k];
    return select_plans_net(h, B, netH, netW, ident, (long long)H * W * 3);
}

int select_plans_net(ssd_handle *h, int B, int netH, int netW, int ident, long long max_src_bytes)
{
    PlanKey key;
    key.B = B;
    key.netH = netH;
    key.netW = netW;
    key.ident = ident;
    // ONE plan per forward (round 1: 1 / 2 / 4 / 8 staggered sub-batch plans -> 730 / 696 / 647 / 587 img/s: backbone kernels
    // beside head kernels take CU slots from them and stretch far more than the overlap returns).  Consecutive sub-batch plans
    // exist for one reason: every tensor a launch addresses with 32-bit byte offsets must stay < 2 GiB.  Per image: the largest
    // backbone tensor (first conv / max-pool output [H/2, W/2, 32]; MobileNet's Conv2d_1_pointwise doubles the channels at that
    // resolution), the concatenated pyramid of a head tower (256 channels), the class logits [N, C], the box codes, and the
    // uint8 source image.  Option nsub = n forces at least n plans (the tests' way to reach this path without a 2 GiB batch).
    int nsub = 1;
    { const int v = ssd_opt(h, OPT_NSUB, 0); if (v >= 1 && v <= 8) nsub = v; }
    {
        const int nH = key.netH, nW = key.netW;
        const int cmax = h->cfg.backbone == SSD_BACKBONE_MOBILENET && !h->pw.empty() ? std::max(h->firstCp, h->pw[0].CoutP) : h->firstCp;
        long long per_img = (long long)(nH / 2) * (nW / 2) * cmax * 4;
        const Pyr py1 = make_pyr(1, nH, nW, 256);
        per_img = std::max(per_img, py1.total * 4);
        per_img = std::max(per_img, (long long)ssd_num_anchors(nH, nW) * std::max(h->cfg.num_classes, 4) * 4);
        per_img = std::max(per_img, max_src_bytes);
        if (h->cfg.backbone == SSD_BACKBONE_SHUFFLENET) {        // a ShuffleNet stage is one allocation: its producers' tensors + its two-part output
            const int un[3] = {4, 8, 4};
            int ipw = 0, hh = nH / 8, ww = nW / 8;
            for (int st = 0; st < 3 && ipw + 1 < (int)h->pw.size(); ++st) {
                per_img = std::max(per_img, (long long)hh * ww * (un[st] + 2) * h->pw[ipw + 1].CoutP * 4);
                ipw += 3 + 2 * (un[st] - 1); hh /= 2; ww /= 2;
            }
        }
        const long long bmax = ((1LL << 31) - 1) / per_img;
        if (bmax < 1) return ssd_fail(SSD_ERR_INVALID, "ssd_forward: image too large for one launch");
        const int need = (int)((B + bmax - 1) / bmax);
        if (need > nsub) nsub = need;
    }
    if (nsub > B) nsub = B;
    key.nsub = nsub;
    h->use_clock += 1;
    if (h->cur && h->cur->key == key) {                 // (the usual serving loop: same shape as the call before)
        h->cur->last_use = h->use_clock;
        h->cache_hits += 1;
        return SSD_OK;
    }
    for (PlanSet *ps : h->cache)
        if (ps->key == key) {
            ps->last_use = h->use_clock;
            h->cur = ps;
            h->cache_hits += 1;
            return SSD_OK;
        }
    h->cache_misses += 1;
    PlanSet *ps = new PlanSet();
    ps->key = key;
    ps->last_use = h->use_clock;
    int rc = build_planset(h, *ps);
    if (rc == SSD_ERR_HIP && !h->cache.empty()) {
        // (most likely out of device memory: drop every other set -- the device drained -- and try once more)
        (void)hipGetLastError();
        free_planset(ps);
        HIPCHK(hipDeviceSynchronize());
        h->cache_evictions += (long long)h->cache.size();
        free_plans(h);
        ps = new PlanSet();
        ps->key = key;
        ps->last_use = h->use_clock;
        rc = build_planset(h, *ps);
    }
    if (rc != SSD_OK) {
        const std::string msg = ssd_last_error();        // (the frees below do not touch it, but keep the first cause)
        (void)hipDeviceSynchronize();
        free_planset(ps);
        return ssd_fail(rc, msg);
    }
    h->cache.push_back(ps);
    h->cur = ps;
    return trim_plan_cache(h, ps);
}

// Enqueues one forward on stream `s` (plus the plans' internal streams): kernels only, no host synchronisation.  Sub-batch plans
// (batches past 2 GiB of activations) run one after the other: every plan starts on `s` and joins back into it.
int enqueue_forward(ssd_handle *h, const uint8_t *images_dev, float *boxes_dev, int32_t *labels_dev,
                           float *scores_dev, int32_t *num_boxes_dev, long long out_stride, hipStream_t s)
{
    if (!h->cur) return ssd_fail(SSD_ERR_STATE, "enqueue_forward: no plan selected");
    h->cur_images = images_dev;
    if (h->profiling) {
        hipEvent_t ref;
        HIPCHK(pool_event(h, &ref));
        HIPCHK(hipEventRecord(ref, s));
        h->ref_evs.push_back(ref);
    }
    const int T = h->cfg.num_classes * h->cfg.max_boxes_per_class;
    const std::vector<Plan *> &plans = h->cur->plans;
    for (size_t k = 0; k < plans.size(); ++k) {
        Plan &pl = *plans[k];
        // (k > 0: the previous plan's side streams were joined into `s` before its post-processing, and this plan's chains on
        //  them start behind ev_begin or behind an op of this plan: the shared streams need no further ordering)
        if (pl.need_begin) HIPCHK(hipEventRecord(pl.ev_begin, s));
        // option streams = 1: every op on the caller's stream, in plan order (a valid order: an op's dependencies precede it) --
        // a measurement aid that shows what the kernels cost without each other beside them
        const bool single = ssd_opt(h, OPT_STREAMS, 0) == 1;
        bool used[4] = {true, false, false, false};
        for (const Op &op : pl.ops) {
            const int os = single ? 0 : op.stream;
            hipStream_t st = os == 0 ? s : (os == 1 ? pl.s_aux : pl.s_bb[os - 2]);
            if (!used[os] && op.deps.empty())                       // a chain that starts on another stream:
                HIPCHK(hipStreamWaitEvent(st, pl.ev_begin, 0));     // behind the plan's own start
            used[os] = true;
            for (int d : op.deps)           // (same stream: already ordered)
                if (!single && pl.ops[d].stream != op.stream) HIPCHK(hipStreamWaitEvent(st, pl.ops[d].done, 0));
            HIPCHK(run_op(h, op, st));
            if (op.done && !single) HIPCHK(hipEventRecord(op.done, st));
        }
        if (used[1]) {                              // join before the post-processing reads the logits
            HIPCHK(hipEventRecord(pl.ev_join, pl.s_aux));
            HIPCHK(hipStreamWaitEvent(s, pl.ev_join, 0));
        }
        PostArgs p = pl.post;
        for (int q = 0; q < 4; ++q) p.box_scaler[q] = h->src.box_scaler[q];       // model.py:67-68, of this call's source size
        if (h->mixed) {                                                          // ... or of every frame's own
            p.per_image_scaler = 1;
            for (int b = 0; b < pl.B && pl.img0 + b < SSD_MIXED_MAX; ++b) {
                p.scaler_img[b][0] = h->mixed->scaler[pl.img0 + b][0];
                p.scaler_img[b][1] = h->mixed->scaler[pl.img0 + b][1];
            }
        }
        p.out_stride = out_stride;
        p.boxes = boxes_dev + (out_stride ? (size_t)pl.img0 * out_stride : (size_t)pl.img0 * T * 4);
        p.labels = labels_dev + (out_stride ? (size_t)pl.img0 * out_stride : (size_t)pl.img0 * T);
        p.scores = scores_dev + (out_stride ? (size_t)pl.img0 * out_stride : (size_t)pl.img0 * T);
        p.num = num_boxes_dev + (out_stride ? (size_t)pl.img0 * out_stride : (size_t)pl.img0);
        p.logit_lo = conservative_logit_bound(p.score_thr);
        Op pop;
        pop.cls = 4;
        pop.flops = 0;
        pop.bytes = (double)pl.B * p.N * (p.C + 8) * 4.0;
        pop.run = [p](hipStream_t st) { return launch_postprocess(p, st); };
        HIPCHK(run_op(h, pop, s));
    }
    return SSD_OK;
}
