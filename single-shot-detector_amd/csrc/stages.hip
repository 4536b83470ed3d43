// Pieces of the graph, each where the reference defines it: anchors and the resize arithmetic (host side), and the
// stage entry points the parity tests call (host weights in, synchronise before returning: test conveniences).
// Under -DSSD_DIAG (libssd_hip_diag.so, scripts/ only) also the timing entry points of include/ssd_hip_diag.h.
#include "host.h"

#include <cmath>
#include <cstdio>
#include <cstring>

#ifdef SSD_DIAG
int g_force_tile = -1;
long long *g_dbg_ts = nullptr;
#endif

// ----------------------------------------------------------------------------- anchors
const int A_STRIDES[5] = {8, 16, 32, 64, 128};

extern "C" int32_t ssd_num_anchors(int32_t H, int32_t W)
{
    int n = 0;
    for (int l = 0; l < 5; ++l) {
        const int h = (int)ceilf((float)H / (float)A_STRIDES[l]), w = (int)ceilf((float)W / (float)A_STRIDES[l]);
        n += h * w * 6;
    }
    return n;
}

// anchor_generator.py:13-170 in the fp32 arithmetic of the TF graph (host side, once per image size), any
// hyper-parameters: `scales = tf.constant([m * self.scales[i] ...], tf.float32)` is a Python (double) product rounded to
// fp32, the aspect ratios are the doubles rounded to fp32 (:70-75); h, w = ceil of the fp32 quotient (:59-60).
// Returns the number of anchors (written when `out` is non-null and `capacity` rows suffice) or a negative error code.
extern "C" int64_t ssd_anchors_ex(int32_t H, int32_t W, int32_t n_levels, const int32_t *strides, const double *scales,
                                  int32_t n_mult, const double *multipliers, int32_t n_ratios, const double *ratios,
                                  float *out, int64_t capacity)
{
    if (H <= 0 || W <= 0 || n_levels < 1 || n_mult < 1 || n_ratios < 1 || !strides || !scales || !multipliers || !ratios)
        return ssd_fail(SSD_ERR_INVALID, "ssd_anchors_ex: bad arguments");
    for (int l = 0; l < n_levels; ++l)
        if (strides[l] < 1) return ssd_fail(SSD_ERR_INVALID, "ssd_anchors_ex: strides must be positive");
    const int A = n_mult * n_ratios;
    const float ih = (float)H, iw = (float)W;
    long long total = 0;
    for (int l = 0; l < n_levels; ++l) {
        const float stride = (float)strides[l];
        total += (long long)ceilf(ih / stride) * (long long)ceilf(iw / stride) * A;
    }
    if (!out) return total;
    if (capacity < total) return ssd_fail(SSD_ERR_INVALID, "ssd_anchors_ex: output too small");
    std::vector<float> hh(A), hw(A);
    long long idx = 0;
    for (int l = 0; l < n_levels; ++l) {
        const float stride = (float)strides[l];
        const int h = (int)ceilf(ih / stride), w = (int)ceilf(iw / stride);
        int a = 0;
        for (int m = 0; m < n_mult; ++m)              // itertools.product(scale_multipliers, aspect_ratios)
            for (int r = 0; r < n_ratios; ++r, ++a) {
                const float scale = (float)(multipliers[m] * scales[l]);
                const float rs = sqrtf((float)ratios[r]);
                const float height = scale / rs, width = scale * rs;
                hh[a] = 0.5f * height;
                hw[a] = 0.5f * width;
            }
        float t = ((float)h - 1.0f) * stride;
        const float offy = 0.5f * (ih - t);
        t = ((float)w - 1.0f) * stride;
        const float offx = 0.5f * (iw - t);
        for (int i = 0; i < h; ++i) {
            float cy = (float)i * stride;
            cy = cy + offy;
            for (int j = 0; j < w; ++j) {
                float cx = (float)j * stride;
                cx = cx + offx;
                for (a = 0; a < A; ++a, ++idx) {
                    out[idx * 4 + 0] = (cy - hh[a]) / ih;
                    out[idx * 4 + 1] = (cx - hw[a]) / iw;
                    out[idx * 4 + 2] = (cy + hh[a]) / ih;
                    out[idx * 4 + 3] = (cx + hw[a]) / iw;
                }
            }
        }
    }
    return total;
}

// ... with the constants model.py:37-42 fixes for the exported graph (what every plan uses)
extern "C" int ssd_anchors(int32_t H, int32_t W, float *out)
{
    if (H <= 0 || W <= 0 || !out) return ssd_fail(SSD_ERR_INVALID, "ssd_anchors: bad arguments");
    static const double base[5] = {32, 64, 128, 256, 512}, mult[2] = {1.0, 1.4142}, ars[3] = {1.0, 2.0, 0.5};
    const int64_t n = ssd_anchors_ex(H, W, 5, A_STRIDES, base, 2, mult, 3, ars, out, (int64_t)ssd_num_anchors(H, W));
    return n < 0 ? (int)n : SSD_OK;
}

// resize_keeping_aspect_ratio (pipeline.py:138-194), the size arithmetic of the TF graph:
// scale_factor = to_float(min_dimension / min(h, w)); the longer side is
// to_int32(round(to_float(x) * scale_factor)) (half to even), padded up to a multiple of 128.
ResizeDims resize_dims(int height, int width, int min_dimension, int divisor)
{
    ResizeDims r;
    const int omin = height < width ? height : width;
    const float scale_factor = (float)((double)min_dimension / (double)omin);
    r.ph = r.pw = 0;
    if (height >= width) {
        const int unp = (int)nearbyintf((float)height * scale_factor);
        const int x = (int)ceil((double)unp / (double)divisor);
        r.nh = unp; r.ph = divisor * x - unp; r.nw = min_dimension;
    } else {
        const int unp = (int)nearbyintf((float)width * scale_factor);
        const int x = (int)ceil((double)unp / (double)divisor);
        r.nw = unp; r.pw = divisor * x - unp; r.nh = min_dimension;
    }
    r.box_scaler[0] = r.box_scaler[2] = (float)((double)r.nh / (double)(r.nh + r.ph));
    r.box_scaler[1] = r.box_scaler[3] = (float)((double)r.nw / (double)(r.nw + r.pw));
    return r;
}

// ----------------------------------------------------------------------------- stage entry points
static int to_dev(DevPool &pool, const float *host, size_t n, float **out)
{
    if (!host) { *out = nullptr; return SSD_OK; }
    std::vector<float> v(host, host + n);
    return pool.upload(out, v);
}

static int conv2d_impl(int x16, const float *in_dev, int32_t B, int32_t H, int32_t W, int32_t Cin, const float *w_host,
                          int32_t k, int32_t Cout, int32_t stride, int32_t pad_beg, int32_t OH, int32_t OW,
                          const float *bn_mean, const float *bn_sf, const float *bn_beta, const float *bias_host,
                          const float *up_dev, int32_t act, float *out_dev, void *stream)
{
    if (!in_dev || !w_host || !out_dev || B < 1 || H < 1 || W < 1 || Cin < 1 || Cout < 1 || (k != 1 && k != 3) ||
        stride < 1 || OH < 1 || OW < 1 || act < 0 || act > 2)
        return ssd_fail(SSD_ERR_INVALID, "ssd_conv2d: bad arguments");
    if ((bn_mean || bn_sf || bn_beta) && !(bn_mean && bn_sf && bn_beta))
        return ssd_fail(SSD_ERR_INVALID, "ssd_conv2d: batch-norm vectors must be given together");
    if ((OH - 1) * stride + k - pad_beg > H + k - 1 || (OW - 1) * stride + k - pad_beg > W + k - 1)
        return ssd_fail(SSD_ERR_INVALID, "ssd_conv2d: output size inconsistent with input size");
    if (up_dev && ((OH & 1) || (OW & 1))) return ssd_fail(SSD_ERR_INVALID, "ssd_conv2d: upsample-add needs even output size");
    // the forms the reference's graph contains: conv, conv + BN (+ act), conv + bias, conv + upsampled map
    if ((bn_mean && (bias_host || up_dev)) || (bias_host && up_dev))
        return ssd_fail(SSD_ERR_INVALID, "ssd_conv2d: batch norm, bias and upsample-add are mutually exclusive");
    hipStream_t s = (hipStream_t)stream;
    DevPool pool;
    int rc = SSD_OK;
    auto body = [&]() -> int {
        const int CinP = round_up(Cin, 32), CoutP = round_up(Cout, 8);
        ConvW cw;
        std::vector<int> inmap = phys_map(Cin, CinP), outmap = phys_map(Cout, CoutP);
        SSDCHK(pack_conv(nullptr, pool, w_host, k, Cin, Cout, inmap, outmap, cw));
        if (bn_mean) {
            BnHost b;
            for (int p : outmap) {
                b.mean.push_back(p < 0 ? 0.f : bn_mean[p]);
                b.sf.push_back(p < 0 ? 0.f : bn_sf[p]);
                b.beta.push_back(p < 0 ? 0.f : bn_beta[p]);
            }
            SSDCHK(upload_bn(pool, b, cw));
        }
        if (bias_host) {
            std::vector<float> b;
            for (int p : outmap) b.push_back(p < 0 ? 0.f : bias_host[p]);
            SSDCHK(pool.upload(&cw.bias, b));
        }
        float *tin, *tout, *tup = nullptr;
        const long long rin = (long long)B * H * W, rout = (long long)B * OH * OW;
        SSDCHK(pool.alloc((void **)&tin, (size_t)rin * CinP * 4));
        SSDCHK(pool.alloc((void **)&tout, (size_t)rout * CoutP * 4));
        // f16x3: input, upsampled operand and (unless a bias form / odd width forbids S16 rows) output in split-fp16
        const int o16 = x16 && !bias_host && CoutP % 8 == 0 ? 1 : 0;
        int *flags = nullptr;
        if (x16) { SSDCHK(pool.alloc((void **)&flags, sizeof(int))); HIPCHK(hipMemsetAsync(flags, 0, sizeof(int), s)); }
        HIPCHK(launch_permute_channels(in_dev, rin, Cin, CinP, x16 ? 3 : 1, tin, s));
        if (up_dev) {
            SSDCHK(pool.alloc((void **)&tup, (size_t)(rout / 4) * CoutP * 4));
            HIPCHK(launch_permute_channels(up_dev, rout / 4, Cout, CoutP, o16 ? 3 : 1, tup, s));
        }
        Op op = make_conv_op(nullptr, cw, tin, tout, nullptr, tup, B, stride, pad_beg, act, {dense_level(H, W, OH, OW, CoutP)}, true,
                             x16, o16, o16, flags);
        HIPCHK(op.run(s));
        HIPCHK(launch_permute_channels(tout, rout, Cout, CoutP, o16 ? 2 : 0, out_dev, s));
        HIPCHK(hipStreamSynchronize(s));
        if (x16) {
            int f = 0;
            HIPCHK(hipMemcpy(&f, flags, sizeof(int), hipMemcpyDeviceToHost));
            if (f) return ssd_fail(SSD_ERR_INVALID, "ssd_conv2d_f16x3: a value left the fp16 range (|x| > 65504)");
        }
        return SSD_OK;
    };
    rc = body();
    (void)hipStreamSynchronize(s);
    pool.free_all();
    return rc;
}

extern "C" int ssd_conv2d(const float *in_dev, int32_t B, int32_t H, int32_t W, int32_t Cin, const float *w_host,
                          int32_t k, int32_t Cout, int32_t stride, int32_t pad_beg, int32_t OH, int32_t OW,
                          const float *bn_mean, const float *bn_sf, const float *bn_beta, const float *bias_host,
                          const float *up_dev, int32_t act, float *out_dev, void *stream)
{
    return conv2d_impl(0, in_dev, B, H, W, Cin, w_host, k, Cout, stride, pad_beg, OH, OW, bn_mean, bn_sf, bn_beta, bias_host,
                       up_dev, act, out_dev, stream);
}

extern "C" int ssd_conv2d_f16x3(const float *in_dev, int32_t B, int32_t H, int32_t W, int32_t Cin, const float *w_host,
                                int32_t k, int32_t Cout, int32_t stride, int32_t pad_beg, int32_t OH, int32_t OW,
                                const float *bn_mean, const float *bn_sf, const float *bn_beta, const float *bias_host,
                                const float *up_dev, int32_t act, float *out_dev, void *stream)
{
    return conv2d_impl(1, in_dev, B, H, W, Cin, w_host, k, Cout, stride, pad_beg, OH, OW, bn_mean, bn_sf, bn_beta, bias_host,
                       up_dev, act, out_dev, stream);
}

extern "C" int ssd_depthwise3x3(const float *in_dev, int32_t B, int32_t H, int32_t W, int32_t C, const float *w_host,
                                int32_t stride, int32_t pad_beg, int32_t OH, int32_t OW, const float *bn_mean,
                                const float *bn_sf, const float *bn_beta, int32_t act, float *out_dev, void *stream)
{
    if (!in_dev || !w_host || !out_dev || B < 1 || C < 1 || stride < 1 || OH < 1 || OW < 1 || act < 0 || act > 2)
        return ssd_fail(SSD_ERR_INVALID, "ssd_depthwise3x3: bad arguments");
    if ((bn_mean || bn_sf || bn_beta) && !(bn_mean && bn_sf && bn_beta))
        return ssd_fail(SSD_ERR_INVALID, "ssd_depthwise3x3: batch-norm vectors must be given together");
    hipStream_t s = (hipStream_t)stream;
    DevPool pool;
    auto body = [&]() -> int {
        const int Cp = round_up(C, 8);
        std::vector<int> map = phys_map(C, Cp);
        std::vector<float> wt((size_t)9 * Cp, 0.f), m, sf, be;
        for (int t = 0; t < 9; ++t)
            for (int p = 0; p < Cp; ++p)
                if (map[p] >= 0) wt[(size_t)t * Cp + p] = w_host[(size_t)t * C + map[p]];
        float *dw_, *dm = nullptr, *ds = nullptr, *db = nullptr, *tin, *tout;
        SSDCHK(pool.upload(&dw_, wt));
        if (bn_mean) {
            for (int p : map) { m.push_back(p < 0 ? 0.f : bn_mean[p]); sf.push_back(p < 0 ? 0.f : bn_sf[p]); be.push_back(p < 0 ? 0.f : bn_beta[p]); }
            SSDCHK(pool.upload(&dm, m)); SSDCHK(pool.upload(&ds, sf)); SSDCHK(pool.upload(&db, be));
        }
        const long long rin = (long long)B * H * W, rout = (long long)B * OH * OW;
        SSDCHK(pool.alloc((void **)&tin, (size_t)rin * Cp * 4));
        SSDCHK(pool.alloc((void **)&tout, (size_t)rout * Cp * 4));
        HIPCHK(launch_permute_channels(in_dev, rin, C, Cp, 1, tin, s));
        HIPCHK(launch_depthwise(tin, B, H, W, Cp, dw_, stride, pad_beg, OH, OW, dm, ds, db, act, tout, s));
        HIPCHK(launch_permute_channels(tout, rout, C, Cp, 0, out_dev, s));
        HIPCHK(hipStreamSynchronize(s));
        return SSD_OK;
    };
    int rc = body();
    (void)hipStreamSynchronize(s);
    pool.free_all();
    return rc;
}

extern "C" int ssd_dw_pw(const float *in_dev, int32_t B, int32_t H, int32_t W, int32_t C, const float *dw_w_host,
                         int32_t stride, const float *dw_mean, const float *dw_sf, const float *dw_beta, int32_t dw_act,
                         const float *pw_w_host, int32_t Cout, const float *pw_mean, const float *pw_sf,
                         const float *pw_beta, int32_t pw_act, float *out_dev, void *stream)
{
    if (!in_dev || !dw_w_host || !dw_mean || !dw_sf || !dw_beta || !pw_w_host || !pw_mean || !pw_sf || !pw_beta || !out_dev ||
        B < 1 || H < 1 || W < 1 || C < 1 || Cout < 1 || (stride != 1 && stride != 2) || dw_act < 0 || dw_act > 2 || pw_act < 0 || pw_act > 2)
        return ssd_fail(SSD_ERR_INVALID, "ssd_dw_pw: bad arguments");
    if (stride == 2 && ((H & 1) || (W & 1))) return ssd_fail(SSD_ERR_INVALID, "ssd_dw_pw: stride 2 needs even H, W");
    hipStream_t s = (hipStream_t)stream;
    DevPool pool;
    auto body = [&]() -> int {
        const int Cp = round_up(C, 32), CoutP = round_up(Cout, 8);
        std::vector<int> map = phys_map(C, Cp), outmap = phys_map(Cout, CoutP);
        DwW d;
        d.Cp = Cp;
        std::vector<float> wt((size_t)9 * Cp, 0.f), m, sf, be;
        for (int t = 0; t < 9; ++t)
            for (int p = 0; p < Cp; ++p)
                if (map[p] >= 0) wt[(size_t)t * Cp + p] = dw_w_host[(size_t)t * C + map[p]];
        for (int p : map) { m.push_back(p < 0 ? 0.f : dw_mean[p]); sf.push_back(p < 0 ? 0.f : dw_sf[p]); be.push_back(p < 0 ? 0.f : dw_beta[p]); }
        SSDCHK(pool.upload(&d.w, wt)); SSDCHK(pool.upload(&d.mean, m)); SSDCHK(pool.upload(&d.sf, sf)); SSDCHK(pool.upload(&d.beta, be));
        SSDCHK(pack_dw(pool, wt, m, sf, be, d));
        ConvW cw;
        SSDCHK(pack_conv(nullptr, pool, pw_w_host, 1, C, Cout, map, outmap, cw));
        BnHost b;
        for (int p : outmap) { b.mean.push_back(p < 0 ? 0.f : pw_mean[p]); b.sf.push_back(p < 0 ? 0.f : pw_sf[p]); b.beta.push_back(p < 0 ? 0.f : pw_beta[p]); }
        SSDCHK(upload_bn(pool, b, cw));
        if (!dwpws_eligible(d, cw, B, H, W, stride))
            return ssd_fail(SSD_ERR_INVALID, "ssd_dw_pw: shape not supported by the fused kernel (every tensor below 2 GiB, stride 2 needs even H and W)");
        const int OH = H / stride, OW = W / stride;
        float *tin, *tout;
        const long long rin = (long long)B * H * W, rout = (long long)B * OH * OW;
        SSDCHK(pool.alloc((void **)&tin, (size_t)rin * Cp * 4));
        SSDCHK(pool.alloc((void **)&tout, (size_t)rout * CoutP * 4));
        HIPCHK(launch_permute_channels(in_dev, rin, C, Cp, 1, tin, s));
        Op op = make_dwpws_op(d, cw, tin, B, H, W, stride, dw_act, pw_act, tout);
        HIPCHK(op.run(s));
        HIPCHK(launch_permute_channels(tout, rout, Cout, CoutP, 0, out_dev, s));
        HIPCHK(hipStreamSynchronize(s));
        return SSD_OK;
    };
    int rc = body();
    (void)hipStreamSynchronize(s);
    pool.free_all();
    return rc;
}

extern "C" int ssd_first_conv(const uint8_t *images_dev, int32_t B, int32_t H, int32_t W, const float *w_host,
                              int32_t Cout, const float *bn_mean, const float *bn_sf, const float *bn_beta, int32_t act,
                              float *out_dev, void *stream)
{
    if (!images_dev || !w_host || !out_dev || B < 1 || H < 2 || W < 2 || (H & 1) || (W & 1) || Cout < 1 || act < 0 || act > 2)
        return ssd_fail(SSD_ERR_INVALID, "ssd_first_conv: bad arguments (H, W must be even)");
    if ((bn_mean || bn_sf || bn_beta) && !(bn_mean && bn_sf && bn_beta))
        return ssd_fail(SSD_ERR_INVALID, "ssd_first_conv: batch-norm vectors must be given together");
    hipStream_t s = (hipStream_t)stream;
    DevPool pool;
    auto body = [&]() -> int {
        const int Cp = round_up(Cout, 8);
        std::vector<int> map = phys_map(Cout, Cp);
        std::vector<float> wt((size_t)27 * Cp, 0.f), m, sf, be;
        for (int t = 0; t < 27; ++t)
            for (int p = 0; p < Cp; ++p)
                if (map[p] >= 0) wt[(size_t)t * Cp + p] = w_host[(size_t)t * Cout + map[p]];
        float *dw_, *dm = nullptr, *ds = nullptr, *db = nullptr, *tout;
        SSDCHK(pool.upload(&dw_, wt));
        if (bn_mean) {
            for (int p : map) { m.push_back(p < 0 ? 0.f : bn_mean[p]); sf.push_back(p < 0 ? 0.f : bn_sf[p]); be.push_back(p < 0 ? 0.f : bn_beta[p]); }
            SSDCHK(pool.upload(&dm, m)); SSDCHK(pool.upload(&ds, sf)); SSDCHK(pool.upload(&db, be));
        }
        const long long rout = (long long)B * (H / 2) * (W / 2);
        SSDCHK(pool.alloc((void **)&tout, (size_t)rout * Cp * 4));
        HIPCHK(launch_first_conv(images_dev, B, H, W, H, W, H, W, dw_, Cp, dm, ds, db, act, tout, s));
        HIPCHK(launch_permute_channels(tout, rout, Cout, Cp, 0, out_dev, s));
        HIPCHK(hipStreamSynchronize(s));
        return SSD_OK;
    };
    int rc = body();
    (void)hipStreamSynchronize(s);
    pool.free_all();
    return rc;
}

// MobileNet's first three layers as the one launch the layer plan uses for them (front.hip): first convolution 3x3 stride 2
// 'SAME' on the normalised uint8 frames -> depthwise 3x3 -> pointwise 1x1, each with batch norm and activation.
extern "C" int ssd_front_block(const uint8_t *images_dev, int32_t B, int32_t H, int32_t W, const float *w0_host, int32_t C0,
                               const float *bn0_mean, const float *bn0_sf, const float *bn0_beta, int32_t act0,
                               const float *dw_w_host, const float *dw_mean, const float *dw_sf, const float *dw_beta, int32_t dw_act,
                               const float *pw_w_host, int32_t Cout, const float *pw_mean, const float *pw_sf, const float *pw_beta,
                               int32_t pw_act, float *out_dev, void *stream)
{
    if (!images_dev || !w0_host || !bn0_mean || !bn0_sf || !bn0_beta || !dw_w_host || !dw_mean || !dw_sf || !dw_beta || !pw_w_host ||
        !pw_mean || !pw_sf || !pw_beta || !out_dev || B < 1 || H < 2 || W < 2 || (H & 1) || (W & 1) || act0 < 0 || act0 > 2 ||
        dw_act < 0 || dw_act > 2 || pw_act < 0 || pw_act > 2)
        return ssd_fail(SSD_ERR_INVALID, "ssd_front_block: bad arguments (H, W must be even)");
    if (!front_supports(B, H, W, C0, C0, Cout))
        return ssd_fail(SSD_ERR_INVALID, "ssd_front_block: shape not supported (32 -> 32 -> 64 channels, every tensor below 2 GiB)");
    hipStream_t s = (hipStream_t)stream;
    DevPool pool;
    auto body = [&]() -> int {
        const int Cp = 32, CoutP = 64;
        std::vector<int> map = phys_map(C0, Cp), outmap = phys_map(Cout, CoutP);
        DwW f, d;
        f.Cp = d.Cp = Cp;
        std::vector<float> w0((size_t)27 * Cp, 0.f), wd((size_t)9 * Cp, 0.f), m0, s0, b0, m, sf, be;
        for (int t = 0; t < 27; ++t)
            for (int p = 0; p < Cp; ++p) w0[(size_t)t * Cp + p] = w0_host[(size_t)t * C0 + map[p]];
        for (int t = 0; t < 9; ++t)
            for (int p = 0; p < Cp; ++p) wd[(size_t)t * Cp + p] = dw_w_host[(size_t)t * C0 + map[p]];
        for (int p : map) {
            m0.push_back(bn0_mean[p]); s0.push_back(bn0_sf[p]); b0.push_back(bn0_beta[p]);
            m.push_back(dw_mean[p]); sf.push_back(dw_sf[p]); be.push_back(dw_beta[p]);
        }
        SSDCHK(pool.upload(&f.w, w0)); SSDCHK(pool.upload(&f.mean, m0)); SSDCHK(pool.upload(&f.sf, s0)); SSDCHK(pool.upload(&f.beta, b0));
        SSDCHK(pack_dw(pool, wd, m, sf, be, d));
        ConvW cw;
        SSDCHK(pack_conv(nullptr, pool, pw_w_host, 1, C0, Cout, map, outmap, cw));
        BnHost b;
        for (int p : outmap) { b.mean.push_back(pw_mean[p]); b.sf.push_back(pw_sf[p]); b.beta.push_back(pw_beta[p]); }
        SSDCHK(upload_bn(pool, b, cw));
        float *tout;
        const long long rout = (long long)B * (H / 2) * (W / 2);
        SSDCHK(pool.alloc((void **)&tout, (size_t)rout * CoutP * 4));
        FrontArgs q;
        memset(&q, 0, sizeof(q));
        q.img = images_dev; q.w0 = f.w; q.m0 = f.mean; q.s0 = f.sf; q.b0 = f.beta; q.dwpack = d.pack;
        q.wt = cw.wt; q.mean = cw.mean; q.sf = cw.sf; q.beta = cw.beta; q.out = tout;
        q.B = B; q.H = H; q.W = W; q.act0 = act0; q.dact = dw_act; q.act = pw_act;
        q.tiles_y = (H / 2 + front_tile_y() - 1) / front_tile_y();
        q.tiles_x = (W / 2 + front_tile_x() - 1) / front_tile_x();
        HIPCHK(launch_front(q, s));
        HIPCHK(launch_permute_channels(tout, rout, Cout, CoutP, 0, out_dev, s));
        HIPCHK(hipStreamSynchronize(s));
        return SSD_OK;
    };
    int rc = body();
    (void)hipStreamSynchronize(s);
    pool.free_all();
    return rc;
}

// ShuffleNet's first two layers as the one launch the layer plan uses for them (front.hip): ssd_first_conv (3 -> 24) followed
// by ssd_maxpool3x3s2, the half-resolution tensor kept in LDS.
extern "C" int ssd_first_conv_maxpool(const uint8_t *images_dev, int32_t B, int32_t H, int32_t W, const float *w_host, int32_t Cout,
                                      const float *bn_mean, const float *bn_sf, const float *bn_beta, int32_t act, float *out_dev,
                                      void *stream)
{
    if (!images_dev || !w_host || !bn_mean || !bn_sf || !bn_beta || !out_dev || act < 0 || act > 2)
        return ssd_fail(SSD_ERR_INVALID, "ssd_first_conv_maxpool: bad arguments");
    if (Cout != 24 || !front_pool_supports(B, H, W, Cout))
        return ssd_fail(SSD_ERR_INVALID, "ssd_first_conv_maxpool: shape not supported (24 output channels, H and W multiples of 4, frames below 2 GiB)");
    hipStream_t s = (hipStream_t)stream;
    DevPool pool;
    auto body = [&]() -> int {
        const int Cp = 24;
        std::vector<int> map = phys_map(Cout, Cp);
        std::vector<float> wt((size_t)27 * Cp, 0.f), m, sf, be;
        for (int t = 0; t < 27; ++t)
            for (int p = 0; p < Cp; ++p) wt[(size_t)t * Cp + p] = w_host[(size_t)t * Cout + map[p]];
        for (int p : map) { m.push_back(bn_mean[p]); sf.push_back(bn_sf[p]); be.push_back(bn_beta[p]); }
        float *dw_, *dm, *ds, *db, *tout;
        SSDCHK(pool.upload(&dw_, wt)); SSDCHK(pool.upload(&dm, m)); SSDCHK(pool.upload(&ds, sf)); SSDCHK(pool.upload(&db, be));
        const long long rout = (long long)B * (H / 4) * (W / 4);
        SSDCHK(pool.alloc((void **)&tout, (size_t)rout * Cp * 4));
        HIPCHK(launch_front_pool(images_dev, B, H, W, dw_, Cp, dm, ds, db, act, tout, s));
        HIPCHK(launch_permute_channels(tout, rout, Cout, Cp, 0, out_dev, s));
        HIPCHK(hipStreamSynchronize(s));
        return SSD_OK;
    };
    int rc = body();
    (void)hipStreamSynchronize(s);
    pool.free_all();
    return rc;
}

extern "C" int ssd_maxpool3x3s2(const float *in_dev, int32_t B, int32_t H, int32_t W, int32_t C, float *out_dev, void *stream)
{
    if (!in_dev || !out_dev || B < 1 || C < 1 || (C & 3) || (H & 1) || (W & 1) || H < 2 || W < 2)
        return ssd_fail(SSD_ERR_INVALID, "ssd_maxpool3x3s2: bad arguments (C % 4 == 0, even H and W)");
    HIPCHK(launch_maxpool(in_dev, B, H, W, C, out_dev, (hipStream_t)stream));
    return SSD_OK;
}

extern "C" int ssd_concat_shuffle_split(const float *x_dev, const float *y_dev, int64_t rows, int32_t D, float *xo_dev,
                                        float *yo_dev, void *stream)
{
    if (!x_dev || !y_dev || !xo_dev || !yo_dev || rows < 1 || D < 1) return ssd_fail(SSD_ERR_INVALID, "ssd_concat_shuffle_split: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    DevPool pool;
    auto body = [&]() -> int {
        std::vector<int> tx(2 * D), ty(2 * D);
        for (int j = 0; j < D; ++j) {
            const int zx = j, zy = D + j;
            tx[2 * j] = zx & 1; tx[2 * j + 1] = zx >> 1;
            ty[2 * j] = zy & 1; ty[2 * j + 1] = zy >> 1;
        }
        int *dx, *dy;
        SSDCHK(pool.upload(&dx, tx)); SSDCHK(pool.upload(&dy, ty));
        HIPCHK(launch_gather_channels(x_dev, D, y_dev, D, rows, dx, D, xo_dev, s));
        HIPCHK(launch_gather_channels(x_dev, D, y_dev, D, rows, dy, D, yo_dev, s));
        HIPCHK(hipStreamSynchronize(s));
        return SSD_OK;
    };
    int rc = body();
    (void)hipStreamSynchronize(s);
    pool.free_all();
    return rc;
}

// concat_shuffle_split (shufflenet_v2.py:94-115) followed by the unit's conv1x1_before + batch norm + activation (:119) on the new x
// half -- as the ONE kernel the layer plan runs for it (sn_pw.hip): x and y stay where their producers stored them (two dense
// tensors of one allocation), the shuffle is the kernel's per-channel source table.
extern "C" int ssd_shuffle_conv1x1(const float *x_dev, const float *y_dev, int64_t rows, int32_t D, const float *w_host, int32_t Cout,
                                   const float *bn_mean_host, const float *bn_sf_host, const float *bn_beta_host, int32_t act, float *out_dev,
                                   void *stream)
{
    if (!x_dev || !y_dev || !w_host || !bn_mean_host || !bn_sf_host || !bn_beta_host || !out_dev || rows < 1 || D < 2 || (D & 1) || Cout < 1)
        return ssd_fail(SSD_ERR_INVALID, "ssd_shuffle_conv1x1: bad arguments (even D, batch norm required)");
    hipStream_t s = (hipStream_t)stream;
    DevPool pool;
    auto body = [&]() -> int {
        const int Dp = round_up(D, 32), CoutP = round_up(Cout, 32);
        std::vector<int> inmap = phys_map(D, Dp), outmap = phys_map(Cout, CoutP);
        ConvW cw;
        SSDCHK(pack_conv(nullptr, pool, w_host, 1, D, Cout, inmap, outmap, cw));
        BnHost b;
        for (int p : outmap) { b.mean.push_back(p < 0 ? 0.f : bn_mean_host[p]); b.sf.push_back(p < 0 ? 0.f : bn_sf_host[p]); b.beta.push_back(p < 0 ? 0.f : bn_beta_host[p]); }
        SSDCHK(upload_bn(pool, b, cw));
        // one allocation [x | y], both in physical channel order
        const long long tbytes = rows * Dp * 4;
        if (!pw_gather_supports(Dp, CoutP, rows, Dp * 4, 2 * tbytes, rows * CoutP * 4))
            return ssd_fail(SSD_ERR_INVALID, "ssd_shuffle_conv1x1: shape not supported by the gathering kernel (D <= 512, tensors below 2 GiB)");
        float *xy, *tout;
        SSDCHK(pool.alloc((void **)&xy, (size_t)(2 * tbytes)));
        SSDCHK(pool.alloc((void **)&tout, (size_t)rows * CoutP * 4));
        HIPCHK(launch_permute_channels(x_dev, rows, D, Dp, 1, xy, s));
        HIPCHK(launch_permute_channels(y_dev, rows, D, Dp, 1, xy + rows * Dp, s));
        // new x = z[0 : D] of z[2d] = x[d], z[2d + 1] = y[d]: input channel k comes from (k & 1 ? y : x)[k >> 1]
        std::vector<int> src(Dp, -1);
        for (int k = 0; k < D; ++k)
            src[ssd_phys_of_logical(k)] = (int)((k & 1 ? tbytes : 0) + (long long)ssd_phys_of_logical(k >> 1) * 4);
        int *src_dev;
        SSDCHK(pool.upload(&src_dev, src));
        Op op = make_pw_gather_op(cw, xy, 2 * tbytes, src_dev, Dp * 4, rows, act, tout);
        HIPCHK(op.run(s));
        HIPCHK(launch_permute_channels(tout, rows, Cout, CoutP, 0, out_dev, s));
        HIPCHK(hipStreamSynchronize(s));
        return SSD_OK;
    };
    int rc = body();
    (void)hipStreamSynchronize(s);
    pool.free_all();
    return rc;
}

extern "C" size_t ssd_postprocess_workspace_bytes(int32_t B, int32_t N, int32_t C, int32_t mp)
{
    if (B < 1 || N < 1 || C < 1 || mp < 1) return 0;
    return post_workspace_bytes(B, N, C, mp);
}

extern "C" int ssd_postprocess(const float *logits_dev, const float *codes_dev, const float *anchors_dev, int32_t B,
                               int32_t N, int32_t C, float score_threshold, float iou_threshold, int32_t mp,
                               const float *box_scaler_host, float *boxes_dev, int32_t *labels_dev, float *scores_dev,
                               int32_t *num_boxes_dev, void *workspace_dev, size_t workspace_bytes, void *stream)
{
    if (!logits_dev || !codes_dev || !anchors_dev || !boxes_dev || !labels_dev || !scores_dev || !num_boxes_dev ||
        !workspace_dev || B < 1 || N < 1 || C < 1 || mp < 1)
        return ssd_fail(SSD_ERR_INVALID, "ssd_postprocess: bad arguments");
    if (workspace_bytes < post_workspace_bytes(B, N, C, mp)) return ssd_fail(SSD_ERR_INVALID, "ssd_postprocess: workspace too small");
    PostArgs p;
    memset(&p, 0, sizeof(p));
    p.logits = logits_dev; p.codes = codes_dev; p.anchors = anchors_dev;
    p.B = B; p.N = N; p.C = C;
    p.score_thr = score_threshold; p.iou_thr = iou_threshold;
    p.logit_lo = conservative_logit_bound(score_threshold);
    p.max_per_class = mp;
    p.fast_max = nms_fast_max(nullptr);
    for (int k = 0; k < 4; ++k) p.box_scaler[k] = box_scaler_host ? box_scaler_host[k] : 1.0f;
    p.boxes = boxes_dev; p.labels = labels_dev; p.scores = scores_dev; p.num = num_boxes_dev;
    post_carve(p, workspace_dev);
    HIPCHK(launch_postprocess(p, (hipStream_t)stream));
    return SSD_OK;
}

// ----------------------------------------------------------------------------- diagnostics
#ifdef SSD_DIAG   // everything below exists only in libssd_hip_diag.so (include/ssd_hip_diag.h, scripts/)
// Times `reps` launches of one dense convolution (random data, BN + ReLU epilogue) on the
// implicit-GEMM kernel with an explicit tile variant; used by scripts/bench_conv.py to A/B
// kernel variants in one process.  nlev > 1 replicates the level `nlev` times in one launch
// (the head-tower launch shape).  Returns the average milliseconds per launch.
extern "C" int ssd_bench_conv(int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t k, int32_t stride,
                              int32_t tile, int32_t reps, int32_t pyramid, double *avg_ms, double *gflop)
{
    if (B < 1 || H < 1 || W < 1 || Cin < 1 || Cout < 1 || (k != 1 && k != 3) || reps < 1 || !avg_ms)
        return ssd_fail(SSD_ERR_INVALID, "ssd_bench_conv: bad arguments");
    DevPool pool;
    auto body = [&]() -> int {
        const int CinP = round_up(Cin, 32), CoutP = round_up(Cout, 8);
        std::vector<float> w((size_t)k * k * Cin * Cout);
        unsigned st = 12345u;
        auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xFFFF) / 65536.0f - 0.5f; };
        for (auto &v : w) v = rnd() * 0.1f;
        ConvW cw;
        g_force_tile = tile;             // (-1: the library's own choice) stays in force until the op below is made
        int rc = pack_conv(nullptr, pool, w.data(), k, Cin, Cout, phys_map(Cin, CinP), phys_map(Cout, CoutP), cw);
        if (rc != SSD_OK) g_force_tile = -1;
        SSDCHK(rc);
        BnHost b;
        const int nl = pyramid ? 5 : 1;
        for (int l = 0; l < nl; ++l)
            for (int c = 0; c < CoutP; ++c) { b.mean.push_back(0.01f * (c % 7)); b.sf.push_back(1.0f + 0.001f * (c % 5)); b.beta.push_back(0.02f); }
        SSDCHK(upload_bn(pool, b, cw));
        std::vector<LevelDesc> lv;
        long long in_total = 0, out_total = 0;
        const int pad = k == 3 ? 1 : 0;
        int h = H, wd = W;
        double fl = 0;
        for (int l = 0; l < nl; ++l) {
            const int oh = (h + 2 * pad - k) / stride + 1, ow = (wd + 2 * pad - k) / stride + 1;
            lv.push_back(dense_level(h, wd, oh, ow, CoutP, in_total, out_total, l * CoutP));
            in_total += (long long)B * h * wd * CinP;
            out_total += (long long)B * oh * ow * CoutP;
            fl += 2.0 * B * oh * ow * (double)k * k * Cin * Cout;
            h = (h + 1) / 2; wd = (wd + 1) / 2;
        }
        float *in, *out;
        SSDCHK(pool.alloc((void **)&in, (size_t)in_total * 4));
        SSDCHK(pool.alloc((void **)&out, (size_t)out_total * 4));
        {
            std::vector<float> hin((size_t)in_total);
            for (auto &v : hin) v = rnd();
            HIPCHK(hipMemcpy(in, hin.data(), hin.size() * 4, hipMemcpyHostToDevice));
        }
        long long *ts = nullptr;
        long long nblk = 0;
        if (tile == 17 || tile == 18) {   // per-block phase timestamps of the last launch -> $SSD_TS_DUMP (int64[nblk][9])
            const int tb = tile == 17 ? 128 : 64;
            for (size_t l = 0; l < lv.size(); ++l) nblk += ((long long)B * lv[l].OH * lv[l].OW + tb - 1) / tb;
            nblk *= cw.CoutPad / tb;
            SSDCHK(pool.alloc((void **)&ts, (size_t)nblk * 9 * 8));
            HIPCHK(hipMemset(ts, 0, (size_t)nblk * 9 * 8));      // kernels with fewer blocks leave zero rows
            g_dbg_ts = ts;
        }
        // SSD_BENCH_PRECISION=f16x3: the same launch on split-fp16 rows (input converted in place of the fp32 image)
        const char *bp = getenv("SSD_BENCH_PRECISION");
        const int x16 = bp && !strcmp(bp, "f16x3") ? 1 : 0;
        if (x16) {
            float *in16;
            SSDCHK(pool.alloc((void **)&in16, (size_t)in_total * 4));
            HIPCHK(launch_permute_channels(in, in_total / CinP, CinP, CinP, 3, in16, nullptr));
            in = in16;
        }
        Op op = make_conv_op(nullptr, cw, in, out, nullptr, nullptr, B, stride, pad, SSD_ACT_RELU, lv, true, x16, x16);
        g_dbg_ts = nullptr;
        g_force_tile = -1;
        hipEvent_t e0, e1;
        HIPCHK(hipEventCreate(&e0));
        HIPCHK(hipEventCreate(&e1));
        for (int i = 0; i < 2; ++i) HIPCHK(op.run(nullptr));
        HIPCHK(hipEventRecord(e0, nullptr));
        for (int i = 0; i < reps; ++i) HIPCHK(op.run(nullptr));
        HIPCHK(hipEventRecord(e1, nullptr));
        HIPCHK(hipEventSynchronize(e1));
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, e0, e1));
        *avg_ms = ms / reps;
        if (gflop) *gflop = fl / 1e9;
        if (ts) {
            if (const char *path = getenv("SSD_TS_DUMP")) {
                std::vector<long long> hts((size_t)nblk * 9);
                HIPCHK(hipMemcpy(hts.data(), ts, hts.size() * 8, hipMemcpyDeviceToHost));
                if (FILE *f = fopen(path, "wb")) { fwrite(hts.data(), 8, hts.size(), f); fclose(f); }
            }
        }
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        return SSD_OK;
    };
    int rc = body();
    g_force_tile = -1;
    g_dbg_ts = nullptr;
    (void)hipDeviceSynchronize();
    pool.free_all();
    return rc;
}

extern "C" int ssd_bench_dwpw(int32_t B, int32_t H, int32_t W, int32_t C, int32_t Cout, int32_t stride, int32_t fused,
                              int32_t reps, double *avg_ms)
{
    if (B < 1 || H < 1 || W < 1 || C < 1 || Cout < 1 || (stride != 1 && stride != 2) || reps < 1 || !avg_ms)
        return ssd_fail(SSD_ERR_INVALID, "ssd_bench_dwpw: bad arguments");
    DevPool pool;
    auto body = [&]() -> int {
        const int Cp = round_up(C, 32), CoutP = round_up(Cout, 8);
        unsigned st = 777u;
        auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xFFFF) / 65536.0f - 0.5f; };
        std::vector<int> map = phys_map(C, Cp), outmap = phys_map(Cout, CoutP);
        DwW d;
        d.Cp = Cp;
        std::vector<float> wt((size_t)9 * Cp), m(Cp, 0.01f), sf(Cp, 1.01f), be(Cp, 0.02f);
        for (auto &v : wt) v = rnd();
        SSDCHK(pool.upload(&d.w, wt)); SSDCHK(pool.upload(&d.mean, m)); SSDCHK(pool.upload(&d.sf, sf)); SSDCHK(pool.upload(&d.beta, be));
        SSDCHK(pack_dw(pool, wt, m, sf, be, d));
        std::vector<float> w((size_t)C * Cout);
        for (auto &v : w) v = rnd() * 0.1f;
        ConvW cw;
        SSDCHK(pack_conv(nullptr, pool, w.data(), 1, C, Cout, map, outmap, cw));
        BnHost b;
        for (int c = 0; c < CoutP; ++c) { b.mean.push_back(0.01f); b.sf.push_back(1.0f); b.beta.push_back(0.02f); }
        SSDCHK(upload_bn(pool, b, cw));
        const int OH = H / stride, OW = W / stride;
        float *in, *mid, *out;
        const long long nin = (long long)B * H * W * Cp;
        SSDCHK(pool.alloc((void **)&in, (size_t)nin * 4));
        SSDCHK(pool.alloc((void **)&mid, (size_t)B * OH * OW * Cp * 4));
        SSDCHK(pool.alloc((void **)&out, (size_t)B * OH * OW * CoutP * 4));
        {
            std::vector<float> hin((size_t)nin);
            for (auto &v : hin) v = rnd();
            HIPCHK(hipMemcpy(in, hin.data(), hin.size() * 4, hipMemcpyHostToDevice));
        }
        std::vector<Op> ops;
        if (fused == 1) {          // the streaming kernel (dwpw_stream.hip)
            if (!dwpws_eligible(d, cw, B, H, W, stride)) return ssd_fail(SSD_ERR_INVALID, "ssd_bench_dwpw: shape not supported by the streaming kernel");
            ops.push_back(make_dwpws_op(d, cw, in, B, H, W, stride, SSD_ACT_RELU6, SSD_ACT_RELU6, out));
        } else if (fused) {
            return ssd_fail(SSD_ERR_INVALID, "ssd_bench_dwpw: fused must be 0 (two kernels) or 1 (dwpw_stream.hip)");
        } else {
            ops.push_back(make_dw_op(d, in, B, H, W, stride, SSD_ACT_RELU6, mid, C));
            ops.push_back(make_conv_op(nullptr, cw, mid, out, nullptr, nullptr, B, 1, 0, SSD_ACT_RELU6, {dense_level(OH, OW, OH, OW, CoutP)}, true));
        }
        const char *dump = getenv("SSD_TS_DUMP");
        hipEvent_t e0, e1;
        HIPCHK(hipEventCreate(&e0));
        HIPCHK(hipEventCreate(&e1));
        for (int i = 0; i < 2; ++i) for (auto &op : ops) HIPCHK(op.run(nullptr));
        HIPCHK(hipEventRecord(e0, nullptr));
        for (int i = 0; i < reps; ++i) for (auto &op : ops) HIPCHK(op.run(nullptr));
        HIPCHK(hipEventRecord(e1, nullptr));
        HIPCHK(hipEventSynchronize(e1));
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, e0, e1));
        *avg_ms = ms / reps;
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        if (fused == 1 && dump) {   // phase cycle totals of one extra launch of the streaming kernel -> $SSD_TS_DUMP (int64[512][8])
            long long *t8 = nullptr;
            SSDCHK(pool.alloc((void **)&t8, 512 * 8 * 8));
            HIPCHK(hipMemset(t8, 0, 512 * 8 * 8));
            g_dbg_ts = t8;
            Op op = make_dwpws_op(d, cw, in, B, H, W, stride, SSD_ACT_RELU6, SSD_ACT_RELU6, out);
            g_dbg_ts = nullptr;
            HIPCHK(op.run(nullptr));
            HIPCHK(hipDeviceSynchronize());
            std::vector<long long> hts(512 * 8);
            HIPCHK(hipMemcpy(hts.data(), t8, hts.size() * 8, hipMemcpyDeviceToHost));
            if (FILE *f = fopen(dump, "wb")) { fwrite(hts.data(), 8, hts.size(), f); fclose(f); }
        }
        return SSD_OK;
    };
    int rc = body();
    (void)hipDeviceSynchronize();
    pool.free_all();
    return rc;
}
#endif  // SSD_DIAG
