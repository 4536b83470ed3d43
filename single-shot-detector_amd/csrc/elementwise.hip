// HBM-bound kernels of the backbone: first 3x3 convolution (uint8 in, preprocessing
// fused), 3x3 depthwise convolution + batch norm + activation, 3x3/2 max pool, the
// ShuffleNet concat-shuffle-split, and the logical<->physical channel permutation used
// by the stage entry points.  NHWC, 16 B (4 channels) per lane, channels innermost so a
// wave touches whole 128-B lines.  All fused multiply-adds are explicit fmaf chains in
// (ky,kx[,ci]) order; everything else is separately rounded (-ffp-contract=off).
#include "ssd_internal.h"

typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float act_apply(float v, int act)
{
    if (act >= 1) v = v > 0.0f ? v : 0.0f;
    if (act == 2) v = v < 6.0f ? v : 6.0f;
    return v;
}

__device__ __forceinline__ v4f bn_act4(v4f v, const float *mean, const float *sf, const float *beta, int c, int act)
{
    if (mean) {
        const v4f m = *(const v4f *)(mean + c), s = *(const v4f *)(sf + c), b = *(const v4f *)(beta + c);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float t = (v[i] - m[i]) * s[i];
            v[i] = t + b[i];
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = act_apply(v[i], act);
    return v;
}

// ---------------------------------------------------------------------------------------
// K1: images uint8 [B,srcH,srcW,3] -> float -> nearest-neighbour resize to [nh,nw] + zero pad to
// [H,W] (resize_keeping_aspect_ratio, pipeline.py:138-194; TF r1.12 ResizeNearestNeighbor:
// src = min(floorf(dst * in/out), in-1)) -> /255 -> 2x-1 -> conv 3x3 stride 2 'SAME' (even H,W:
// taps at rows 2oy..2oy+2, zero beyond the bottom/right edge) -> BN -> act, all fused: the
// resized / padded / normalised image never exists in memory.  Pixels of the pad band are
// 0 before normalisation, i.e. 2*0-1 = -1 after it; taps beyond [H,W] contribute 0.
// One thread = one output pixel x 4 output channels; weights [27][Cout] staged in LDS.
// IDENT: the resize is the identity (srcH == nh == H, srcW == nw == W): no index arithmetic.
template <bool IDENT>
__global__ __launch_bounds__(256, IDENT ? 4 : 2) void first_conv_kernel(const uint8_t *__restrict__ img, int B, int srcH, int srcW,
                                                          int nh, int nw, int H, int W, float hs, float ws,
                                                          const float *__restrict__ w, int Cout,
                                                          const float *mean, const float *sf, const float *beta,
                                                          int act, float *__restrict__ out)
{
    extern __shared__ float wl[];   // 27*Cout
    for (int i = threadIdx.x; i < 27 * Cout; i += blockDim.x) wl[i] = w[i];
    __syncthreads();
    const int OH = H >> 1, OW = W >> 1, C4 = Cout >> 2;
    const long long total = (long long)B * OH * OW * C4;
    const float inv255 = (float)(1.0 / 255.0);
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(idx % C4);
        long long pix = idx / C4;
        const int ox = (int)(pix % OW);
        pix /= OW;
        const int oy = (int)(pix % OH);
        const int b = (int)(pix / OH);
        v4f acc = {0.0f, 0.0f, 0.0f, 0.0f};
        if constexpr (IDENT) {
            // The 3 pixels x 3 channels under one filter row are 9 contiguous bytes at byte (..*W + 2*ox)*3, i.e.
            // 0 or 2 bytes past a dword boundary (W is even; which of the two depends on the row when W % 4 == 2): three aligned dword loads through a range-checked
            // buffer resource and a byte alignment replace nine byte loads (the kernel was bound by the rate of
            // its 27 byte-load instructions per thread, not by HBM).  Bytes past the image edge are masked below.
            const __amdgpu_buffer_rsrc_t irsrc = __builtin_amdgcn_make_buffer_rsrc((void *)img, 0, (int)((long long)B * H * W * 3), 0x00020000);
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int iy = 2 * oy + ky;
                const bool yok = iy < H;
                const int ad = ((b * H + (yok ? iy : 0)) * W + 2 * ox) * 3;
                const int a0 = ad & ~3, sh = ad & 3;           // sh: byte offset of the row's first pixel inside its dword (0 or 2)
                const unsigned w0 = __builtin_amdgcn_raw_buffer_load_b32(irsrc, a0, 0, 0);
                const unsigned w1 = __builtin_amdgcn_raw_buffer_load_b32(irsrc, a0 + 4, 0, 0);
                const unsigned w2 = __builtin_amdgcn_raw_buffer_load_b32(irsrc, a0 + 8, 0, 0);
                const unsigned d0 = __builtin_amdgcn_alignbyte(w1, w0, sh);
                const unsigned d1 = __builtin_amdgcn_alignbyte(w2, w1, sh);
                const unsigned d2 = w2 >> (8 * sh);
                const unsigned char px[9] = {(unsigned char)d0, (unsigned char)(d0 >> 8), (unsigned char)(d0 >> 16), (unsigned char)(d0 >> 24),
                                             (unsigned char)d1, (unsigned char)(d1 >> 8), (unsigned char)(d1 >> 16), (unsigned char)(d1 >> 24),
                                             (unsigned char)d2};
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const bool ok = yok && 2 * ox + kx < W;
#pragma unroll
                    for (int ci = 0; ci < 3; ++ci) {
                        float x = (float)px[kx * 3 + ci] * inv255;
                        x = 2.0f * x - 1.0f;
                        if (!ok) x = 0.0f;
                        const v4f wv = *(const v4f *)(wl + ((ky * 3 + kx) * 3 + ci) * Cout + c4 * 4);
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[i] = fmaf(x, wv[i], acc[i]);
                    }
                }
            }
        } else {
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = 2 * oy + ky;
            int sy = iy;
            if constexpr (!IDENT) {
                sy = (int)floorf((float)iy * hs);
                sy = sy < srcH - 1 ? sy : srcH - 1;
            }
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = 2 * ox + kx;
                if constexpr (IDENT) {
                    const bool ok = iy < H && ix < W;
                    const uint8_t *p = img + (((long long)b * H + (ok ? iy : 0)) * W + (ok ? ix : 0)) * 3;
#pragma unroll
                    for (int ci = 0; ci < 3; ++ci) {
                        float x = (float)p[ci] * inv255;
                        x = 2.0f * x - 1.0f;
                        if (!ok) x = 0.0f;
                        const v4f wv = *(const v4f *)(wl + ((ky * 3 + kx) * 3 + ci) * Cout + c4 * 4);
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[i] = fmaf(x, wv[i], acc[i]);
                    }
                } else {
                    int sx = (int)floorf((float)ix * ws);
                    sx = sx < srcW - 1 ? sx : srcW - 1;
                    const bool inside = iy < H && ix < W;      // else: zero padding of the convolution
                    const bool inimg = iy < nh && ix < nw;     // else (but inside): the resize's zero pad band
                    const uint8_t *p = img + (((long long)b * srcH + (inimg ? sy : 0)) * srcW + (inimg ? sx : 0)) * 3;
#pragma unroll
                    for (int ci = 0; ci < 3; ++ci) {
                        float x = inimg ? (float)p[ci] : 0.0f;
                        x = x * inv255;
                        x = 2.0f * x - 1.0f;
                        if (!inside) x = 0.0f;
                        const v4f wv = *(const v4f *)(wl + ((ky * 3 + kx) * 3 + ci) * Cout + c4 * 4);
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[i] = fmaf(x, wv[i], acc[i]);
                    }
                }
            }
        }
        }
        acc = bn_act4(acc, mean, sf, beta, c4 * 4, act);
        *(v4f *)(out + idx * 4) = acc;
    }
}

// K1b: the identity-resize case (the image is already H x W: every frame of the benchmark workload) with Cout % 8 == 0.
// K1 is bound by LDS reads -- 27 x 16 B of weights per 16 B of output, 1.5 TB/s alone -- and by its ~500 instructions
// per 16 B.  Here one LANE = one output pixel, all channels: the 27 input values are unpacked and normalised once per
// pixel, the weights are wave-uniform (scalar loads, an SGPR operand of the packed fmaf), channels go in chunks of 8
// accumulators, and the finished rows leave through a per-wave LDS transpose so that a store instruction writes 1 KB of
// consecutive bytes (the wave's 64 pixels are 64 * Cout * 4 consecutive bytes of the output).  Same (ky,kx,ci)-ordered
// fmaf chain per output.
#define FC_ROWPAD 4     // floats of padding per LDS row: 16-B aligned rows whose 16-B writes of 16 consecutive lanes miss each other's banks
// COUT: the width as a compile-time constant (scalar-load offsets become immediates), 0 = read the argument.
template <int COUT>
__global__ __launch_bounds__(256) void first_conv_px_kernel(const uint8_t *__restrict__ img, int B, int H, int W,
                                                             const float *__restrict__ w, int Cout_arg,
                                                             const float *__restrict__ mean, const float *__restrict__ sf,
                                                             const float *__restrict__ beta, int act, float *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) float fc_tr[];     // [4 waves][64 pixels][Cout + FC_ROWPAD]
    const int Cout = COUT ? COUT : Cout_arg;
    const int OH = H >> 1, OW = W >> 1;
    const long long total = (long long)B * OH * OW;
    const float inv255 = (float)(1.0 / 255.0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // 32 channels (MobileNet): rows of exactly 128 B with the 16-byte chunk c of pixel p at slot c ^ (p & 7) -- both the
    // writes (8 consecutive lanes = 8 pixels, same chunk) and the reads (a 16-lane group = chunk halves of 4 pixels) are
    // conflict-free; the padded rows of the general case cost the ds_read_b128 of the write-out two banks' worth of overlap
    // per group (SQ_LDS_BANK_CONFLICT 25 % of this kernel's LDS cycles, profiles/r02_f32_pmc_summary.txt).
    constexpr bool SWZ = COUT == 32;
    const int rowf = SWZ ? 32 : Cout + FC_ROWPAD;
    float *reg = fc_tr + (size_t)wave * 64 * rowf;
    const __amdgpu_buffer_rsrc_t irsrc = __builtin_amdgcn_make_buffer_rsrc((void *)img, 0, (int)((long long)B * H * W * 3), 0x00020000);
    const long long nwave = (long long)gridDim.x * 4;
    for (long long wbase = ((long long)blockIdx.x * 4 + wave) * 64; wbase < total; wbase += nwave * 64) {
        const long long pix = wbase + lane;
        const bool live = pix < total;
        const long long pp = live ? pix : total - 1;
        const int ox = (int)(pp % OW);
        const long long rowi = pp / OW;
        const int oy = (int)(rowi % OH);
        const int b = (int)(rowi / OH);
        // 3 pixels x 3 channels under one filter row = 9 contiguous bytes at byte (..*W + 2*ox)*3, 0 or 2 bytes past a dword
        // boundary (W even): three aligned dword loads and a byte alignment.  Only the taps of column 2ox+2 and of row
        // 2oy+2 can fall outside the image (even H, W): they are masked, the others never are.
        float x[27];
        const bool xok = 2 * ox + 2 < W;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = 2 * oy + ky;
            const bool yok = ky < 2 || iy < H;
            const int ad = ((b * H + (yok ? iy : 0)) * W + 2 * ox) * 3;
            const int a0 = ad & ~3, sh = ad & 3;
            const unsigned w0 = __builtin_amdgcn_raw_buffer_load_b32(irsrc, a0, 0, 0);
            const unsigned w1 = __builtin_amdgcn_raw_buffer_load_b32(irsrc, a0 + 4, 0, 0);
            const unsigned w2 = __builtin_amdgcn_raw_buffer_load_b32(irsrc, a0 + 8, 0, 0);
            const unsigned d0 = __builtin_amdgcn_alignbyte(w1, w0, sh);
            const unsigned d1 = __builtin_amdgcn_alignbyte(w2, w1, sh);
            const unsigned d2 = w2 >> (8 * sh);
            const unsigned char px[9] = {(unsigned char)d0, (unsigned char)(d0 >> 8), (unsigned char)(d0 >> 16), (unsigned char)(d0 >> 24),
                                         (unsigned char)d1, (unsigned char)(d1 >> 8), (unsigned char)(d1 >> 16), (unsigned char)(d1 >> 24),
                                         (unsigned char)d2};
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                float v = (float)px[k] * inv255;
                v = 2.0f * v - 1.0f;
                if (ky == 2 && !yok) v = 0.0f;
                if (k >= 6 && !xok) v = 0.0f;
                x[ky * 9 + k] = v;
            }
        }
        constexpr int CH = (COUT && COUT % 16 == 0) ? 16 : 8;       // accumulators per pass over the 27 taps
#pragma unroll 1
        for (int ch = 0; ch < Cout; ch += CH) {         // wave-uniform (not unrolled: 864 weights do not fit the SGPRs)
            float acc[CH];
#pragma unroll
            for (int i = 0; i < CH; ++i) acc[i] = 0.0f;
#pragma unroll
            for (int t = 0; t < 27; ++t) {
                const float *wr = w + t * Cout + ch;
#pragma unroll
                for (int i = 0; i < CH; ++i) acc[i] = fmaf(x[t], wr[i], acc[i]);
            }
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                if (mean) {
                    const float tq = (acc[i] - mean[ch + i]) * sf[ch + i];
                    acc[i] = tq + beta[ch + i];
                }
                acc[i] = act_apply(acc[i], act);
            }
#pragma unroll
            for (int i = 0; i < CH; i += 4) {
                const int c = SWZ ? ((((ch + i) >> 2) ^ (lane & 7)) << 2) : ch + i;
                *(v4f *)(reg + lane * rowf + c) = (v4f){acc[i], acc[i + 1], acc[i + 2], acc[i + 3]};
            }
        }
        // the wave's rows -> 64 * Cout * 4 consecutive bytes of the output, 16 B per lane and instruction (wave-private LDS
        // region: the wave's own ds_writes are ordered before its ds_reads by the counter wait the compiler places)
        const int LP = Cout >> 2;                        // 16-B pieces per pixel
        const long long nlive = (total - wbase < 64 ? total - wbase : 64) * LP;
        float *obase = out + wbase * Cout;
        for (int q = lane; q < 64 * LP; q += 64) {
            const int p = q / LP, c4 = q - p * LP;
            const v4f v = *(const v4f *)(reg + p * rowf + (SWZ ? ((c4 ^ (p & 7)) << 2) : c4 * 4));
            if (q < nlive) *(v4f *)(obase + (long long)q * 4) = v;
        }
    }
}

// K1d (round 6): K1 / K1c -- the resize fused into the first convolution, frames of ANY size, one size per batch or one per
// frame -- in K1b's form.  K1 gives 4 output channels of one pixel to a thread: the 8 (6) threads of a pixel each repeat the nine
// index computations and the 27 byte loads, and the weights come from LDS per 16 B of output -- 0.99 ms per 32 frames of 480 x 640
// against 0.17 ms for K1b on frames that need no resize, i.e. every COCO image paid 2.6 % of a 32-frame step for not being
// 640 x 896 already.  Here one LANE = one output pixel, all channels: three source rows and three source columns per lane (the
// index rule of K1, value for value: floorf((float)dst * scale), clamped to the last row / column), each of the nine source
// pixels as two aligned dwords through a range-checked buffer resource + a byte alignment (a pixel's 3 bytes start at any byte),
// then K1b's wave-uniform weights, chunks of accumulators and LDS transpose.  The same (ky,kx,ci)-ordered fmaf chain per output,
// the same separately rounded normalisation: bit-identical to K1 / K1c (tests/test_gpu_multishape.py pins both kernels on every size).
// GeomOf: frame b of the launch -> its FrameGeom (byte offset, source size, resize target, scales).
template <int COUT, class GeomOf>
__device__ __forceinline__ void first_conv_gen_body(const uint8_t *__restrict__ img, unsigned img_bytes4, GeomOf geom_of, int B, int H, int W,
                                                    const float *__restrict__ w, int Cout_arg, const float *__restrict__ mean,
                                                    const float *__restrict__ sf, const float *__restrict__ beta, int act, float *__restrict__ out,
                                                    float *fc_tr)
{
    const int Cout = COUT ? COUT : Cout_arg;
    const int OH = H >> 1, OW = W >> 1;
    const long long total = (long long)B * OH * OW;
    const float inv255 = (float)(1.0 / 255.0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr bool SWZ = COUT == 32;
    const int rowf = SWZ ? 32 : Cout + FC_ROWPAD;
    float *reg = fc_tr + (size_t)wave * 64 * rowf;
    const __amdgpu_buffer_rsrc_t irsrc = __builtin_amdgcn_make_buffer_rsrc((void *)img, 0, (int)img_bytes4, 0x00020000);
    const long long nwave = (long long)gridDim.x * 4;
    for (long long wbase = ((long long)blockIdx.x * 4 + wave) * 64; wbase < total; wbase += nwave * 64) {
        const long long pix = wbase + lane;
        const long long pp = pix < total ? pix : total - 1;
        const int ox = (int)(pp % OW);
        const long long rowi = pp / OW;
        const int oy = (int)(rowi % OH);
        const int b = (int)(rowi / OH);
        const FrameGeom g = geom_of(b);
        int roff[3], coff[3];
        bool yin[3], xin[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int iy = 2 * oy + k, ix = 2 * ox + k;
            int sy = (int)floorf((float)iy * g.hs), sx = (int)floorf((float)ix * g.ws);
            sy = sy < g.srcH - 1 ? sy : g.srcH - 1;
            sx = sx < g.srcW - 1 ? sx : g.srcW - 1;
            yin[k] = iy < g.nh;                          // else: the resize's zero pad band (or beyond the padded frame)
            xin[k] = ix < g.nw;
            roff[k] = (yin[k] ? sy : 0) * g.srcW;
            coff[k] = xin[k] ? sx : 0;
        }
        float x[27];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const bool yok = ky < 2 || 2 * oy + 2 < H;  // else: the convolution's zero padding ('SAME' on even sizes pads bottom / right only)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const bool xok = kx < 2 || 2 * ox + 2 < W;
                const unsigned p = g.off + (unsigned)(roff[ky] + coff[kx]) * 3u;
                const int a0 = (int)(p & ~3u), sh = (int)(p & 3u);
                const unsigned w0 = __builtin_amdgcn_raw_buffer_load_b32(irsrc, a0, 0, 0);
                const unsigned w1 = __builtin_amdgcn_raw_buffer_load_b32(irsrc, a0, 4, 0);
                const unsigned d = __builtin_amdgcn_alignbyte(w1, w0, sh);
                const bool inimg = yin[ky] && xin[kx];
#pragma unroll
                for (int ci = 0; ci < 3; ++ci) {
                    float v = inimg ? (float)((d >> (8 * ci)) & 0xffu) : 0.0f;
                    v = v * inv255;
                    v = 2.0f * v - 1.0f;
                    if (!(yok && xok)) v = 0.0f;
                    x[(ky * 3 + kx) * 3 + ci] = v;
                }
            }
        }
        constexpr int CH = (COUT && COUT % 16 == 0) ? 16 : 8;
#pragma unroll 1
        for (int ch = 0; ch < Cout; ch += CH) {
            float acc[CH];
#pragma unroll
            for (int i = 0; i < CH; ++i) acc[i] = 0.0f;
#pragma unroll
            for (int t = 0; t < 27; ++t) {
                const float *wr = w + t * Cout + ch;
#pragma unroll
                for (int i = 0; i < CH; ++i) acc[i] = fmaf(x[t], wr[i], acc[i]);
            }
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                if (mean) {
                    const float tq = (acc[i] - mean[ch + i]) * sf[ch + i];
                    acc[i] = tq + beta[ch + i];
                }
                acc[i] = act_apply(acc[i], act);
            }
#pragma unroll
            for (int i = 0; i < CH; i += 4) {
                const int c = SWZ ? ((((ch + i) >> 2) ^ (lane & 7)) << 2) : ch + i;
                *(v4f *)(reg + lane * rowf + c) = (v4f){acc[i], acc[i + 1], acc[i + 2], acc[i + 3]};
            }
        }
        const int LP = Cout >> 2;
        const long long nlive = (total - wbase < 64 ? total - wbase : 64) * LP;
        float *obase = out + wbase * Cout;
        for (int q = lane; q < 64 * LP; q += 64) {
            const int p = q / LP, c4 = q - p * LP;
            const v4f v = *(const v4f *)(reg + p * rowf + (SWZ ? ((c4 ^ (p & 7)) << 2) : c4 * 4));
            if (q < nlive) *(v4f *)(obase + (long long)q * 4) = v;
        }
    }
}

template <int COUT>
__global__ __launch_bounds__(256) void first_conv_gen_kernel(const uint8_t *__restrict__ img, unsigned img_bytes4, const FrameGeom g1, int B, int H, int W,
                                                              const float *__restrict__ w, int Cout_arg, const float *__restrict__ mean,
                                                              const float *__restrict__ sf, const float *__restrict__ beta, int act, float *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) float fc_tr[];
    const unsigned frame = (unsigned)g1.srcH * (unsigned)g1.srcW * 3u;
    first_conv_gen_body<COUT>(img, img_bytes4, [&](int b) { FrameGeom g = g1; g.off = g1.off + (unsigned)b * frame; return g; }, B, H, W, w, Cout_arg,
                              mean, sf, beta, act, out, fc_tr);
}

template <int COUT>
__global__ __launch_bounds__(256) void first_conv_gen_mixed_kernel(const uint8_t *__restrict__ img, unsigned img_bytes4, const MixedGeom mg, int first, int B,
                                                                    int H, int W, const float *__restrict__ w, int Cout_arg,
                                                                    const float *__restrict__ mean, const float *__restrict__ sf,
                                                                    const float *__restrict__ beta, int act, float *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) float fc_tr[];
    first_conv_gen_body<COUT>(img, img_bytes4, [&](int b) { return mg.f[first + b]; }, B, H, W, w, Cout_arg, mean, sf, beta, act, out, fc_tr);
}

// K1d's launch: false when this width has no instantiation (then K1 / K1c run).  The first layer's physical width is the logical
// one rounded up to 32 (weights.hip load_first): 32 for every MobileNet-v1 up to depth multiplier 1.0 and every ShuffleNet-v2, 64 at
// MobileNet 2.0 -- whose 68 KB of transpose rows would need the opt-in LDS size; that one stays on K1.
static bool launch_first_conv_gen(const uint8_t *img, unsigned long long img_bytes, const FrameGeom *one, const MixedGeom *mg, int first, int B, int H, int W,
                                  const float *w, int Cout, const float *mean, const float *sf, const float *beta, int act, float *out, hipStream_t s)
{
    if (Cout != 32 || img_bytes + 6 >= (1ull << 31)) return false;
    // The kernel's dword loads are aligned RELATIVE TO ITS BASE: a base that is not a multiple of 4 itself (the second chain's or
    // sub-batch's first frame when a frame's byte size is odd, a caller's slice) is rounded down and the remainder added to every
    // frame's byte offset, so that no load depends on the device's unaligned-access mode.
    const unsigned adj = (unsigned)((uintptr_t)img & 3u);
    img -= adj;
    const size_t lds = (size_t)256 * 32 * sizeof(float);
    const unsigned bytes4 = (unsigned)((img_bytes + adj + 3) & ~3ull);      // (a pixel's second dword may reach <= 3 bytes past a size that is no multiple of 4: the same word)
    const long long px = (long long)B * (H / 2) * (W / 2);
    long long blocks = (px + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    const dim3 grid((unsigned)blocks), blk(256);
    if (one) {
        FrameGeom g = *one;
        g.off += adj;
        hipLaunchKernelGGL(first_conv_gen_kernel<32>, grid, blk, lds, s, img, bytes4, g, B, H, W, w, Cout, mean, sf, beta, act, out);
    } else {
        MixedGeom t = *mg;
        for (int b = first; b < first + B; ++b) t.f[b].off += adj;
        hipLaunchKernelGGL(first_conv_gen_mixed_kernel<32>, grid, blk, lds, s, img, bytes4, t, first, B, H, W, w, Cout, mean, sf, beta, act, out);
    }
    return true;
}

// K1c: K1 (the general, non-identity form) for a batch whose frames have DIFFERENT source sizes and resize to the same [H,W]:
// frame b reads its geometry from the argument table.  The arithmetic per output is K1's, value for value: the same index rule
// (floorf((float)dst * scale), clamped), the same (ky,kx,ci)-ordered fmaf chain.
__global__ __launch_bounds__(256, 2) void first_conv_mixed_kernel(const uint8_t *__restrict__ img, const MixedGeom mg, int first, int B, int H, int W,
                                                                  const float *__restrict__ w, int Cout, const float *mean, const float *sf,
                                                                  const float *beta, int act, float *__restrict__ out)
{
    extern __shared__ float wl[];   // 27*Cout
    for (int i = threadIdx.x; i < 27 * Cout; i += blockDim.x) wl[i] = w[i];
    __syncthreads();
    const int OH = H >> 1, OW = W >> 1, C4 = Cout >> 2;
    const long long total = (long long)B * OH * OW * C4;
    const float inv255 = (float)(1.0 / 255.0);
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(idx % C4);
        long long pix = idx / C4;
        const int ox = (int)(pix % OW);
        pix /= OW;
        const int oy = (int)(pix % OH);
        const int b = (int)(pix / OH);
        const FrameGeom g = mg.f[first + b];
        const uint8_t *src = img + g.off;
        v4f acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = 2 * oy + ky;
            int sy = (int)floorf((float)iy * g.hs);
            sy = sy < g.srcH - 1 ? sy : g.srcH - 1;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = 2 * ox + kx;
                int sx = (int)floorf((float)ix * g.ws);
                sx = sx < g.srcW - 1 ? sx : g.srcW - 1;
                const bool inside = iy < H && ix < W;          // else: zero padding of the convolution
                const bool inimg = iy < g.nh && ix < g.nw;     // else (but inside): the resize's zero pad band
                const uint8_t *p = src + ((long long)(inimg ? sy : 0) * g.srcW + (inimg ? sx : 0)) * 3;
#pragma unroll
                for (int ci = 0; ci < 3; ++ci) {
                    float x = inimg ? (float)p[ci] : 0.0f;
                    x = x * inv255;
                    x = 2.0f * x - 1.0f;
                    if (!inside) x = 0.0f;
                    const v4f wv = *(const v4f *)(wl + ((ky * 3 + kx) * 3 + ci) * Cout + c4 * 4);
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i] = fmaf(x, wv[i], acc[i]);
                }
            }
        }
        acc = bn_act4(acc, mean, sf, beta, c4 * 4, act);
        *(v4f *)(out + idx * 4) = acc;
    }
}

hipError_t launch_first_conv_mixed(const uint8_t *img, const MixedGeom &mg, int first, int B, int H, int W, const float *w, int Cout,
                                   const float *mean, const float *sf, const float *beta, int act, float *out, hipStream_t s, int variant)
{
    if (B < 1 || first < 0 || first + B > SSD_MIXED_MAX || Cout % 4 || (H & 1) || (W & 1) || 27 * Cout * 4 > 65536) return hipErrorInvalidValue;
    unsigned long long end = 0;
    for (int b = first; b < first + B; ++b) {
        const FrameGeom &g = mg.f[b];
        if (g.srcH < 1 || g.srcW < 1 || g.nh < 1 || g.nw < 1 || g.nh > H || g.nw > W) return hipErrorInvalidValue;
        if ((unsigned long long)g.off + (unsigned long long)g.srcH * g.srcW * 3 >= (1ull << 31)) return hipErrorInvalidValue;
        end = std::max(end, (unsigned long long)g.off + (unsigned long long)g.srcH * g.srcW * 3);
    }
    if (variant != 1 && launch_first_conv_gen(img, end, nullptr, &mg, first, B, H, W, w, Cout, mean, sf, beta, act, out, s)) return hipGetLastError();
    const long long total = (long long)B * (H / 2) * (W / 2) * (Cout / 4);
    long long blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(first_conv_mixed_kernel, dim3((unsigned)blocks), dim3(256), 27 * Cout * sizeof(float), s, img, mg, first, B, H, W, w, Cout,
                       mean, sf, beta, act, out);
    return hipGetLastError();
}

hipError_t launch_first_conv(const uint8_t *img, int B, int srcH, int srcW, int nh, int nw, int H, int W,
                             const float *w, int Cout, const float *mean, const float *sf, const float *beta, int act,
                             float *out, hipStream_t s, int variant)
{
    if ((long long)B * srcH * srcW * 3 >= (1LL << 31)) return hipErrorInvalidValue;      // 32-bit byte offsets into the image
    if (Cout % 4 || (H & 1) || (W & 1) || 27 * Cout * 4 > 65536 || nh < 1 || nw < 1 || nh > H || nw > W || srcH < 1 || srcW < 1)
        return hipErrorInvalidValue;
    const float hs = (float)srcH / (float)nh, ws = (float)srcW / (float)nw;
    const long long total = (long long)B * (H / 2) * (W / 2) * (Cout / 4);
    long long blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (srcH == nh && nh == H && srcW == nw && nw == W && Cout % 8 == 0 && Cout <= 128) {
        const long long px = (long long)B * (H / 2) * (W / 2);
        blocks = (px + 255) / 256;
        if (blocks > 256 * 32) blocks = 256 * 32;
        const size_t lds = (size_t)256 * (Cout == 32 ? 32 : Cout + FC_ROWPAD) * sizeof(float);
        if (Cout == 32)        // MobileNet-v1 at depth multiplier 1
            hipLaunchKernelGGL(first_conv_px_kernel<32>, dim3((unsigned)blocks), dim3(256), lds, s, img, B, H, W, w, Cout, mean, sf, beta, act, out);
        else if (Cout == 24)   // ShuffleNet-v2
            hipLaunchKernelGGL(first_conv_px_kernel<24>, dim3((unsigned)blocks), dim3(256), lds, s, img, B, H, W, w, Cout, mean, sf, beta, act, out);
        else {
            // wider first layers (depth multiplier 2: 64 channels = 68 KB, up to 128 = 135 KB): above the 64 KB a kernel gets
            // without asking, so ask -- per device
            static std::atomic<unsigned> attr_done{0};
            if (lds > 65536) {
                hipError_t e = ssd_allow_lds((const void *)first_conv_px_kernel<0>, (int)lds, attr_done);
                if (e != hipSuccess) return e;
            }
            hipLaunchKernelGGL(first_conv_px_kernel<0>, dim3((unsigned)blocks), dim3(256), lds, s, img, B, H, W, w, Cout, mean, sf, beta, act, out);
        }
    } else if (srcH == nh && nh == H && srcW == nw && nw == W)
        hipLaunchKernelGGL(first_conv_kernel<true>, dim3((unsigned)blocks), dim3(256), 27 * Cout * sizeof(float), s, img, B,
                           srcH, srcW, nh, nw, H, W, hs, ws, w, Cout, mean, sf, beta, act, out);
    else {
        const FrameGeom g1 = {0u, srcH, srcW, nh, nw, hs, ws};
        // K1d (one lane per output pixel) where the width has an instantiation; variant 1 pins K1 (tests, A/B)
        if (variant == 1 || !launch_first_conv_gen(img, (unsigned long long)B * srcH * srcW * 3, &g1, nullptr, 0, B, H, W, w, Cout, mean, sf, beta, act, out, s))
            hipLaunchKernelGGL(first_conv_kernel<false>, dim3((unsigned)blocks), dim3(256), 27 * Cout * sizeof(float), s, img, B,
                               srcH, srcW, nh, nw, H, W, hs, ws, w, Cout, mean, sf, beta, act, out);
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// K2: depthwise 3x3, stride 1/2, zero padding `pad` on top/left (TF 'SAME': 1 for stride 1,
// 0 for stride 2 on even sizes), + BN + act.  One thread = R output rows x PX consecutive output
// pixels x 4 channels.  The input rows it needs are streamed top to bottom, each loaded once
// ((PX-1)*STRIDE+3 float4) and applied to every output row it belongs to, so every output keeps
// its own (ky,kx)-ordered fmaf chain.  An out-of-image tap contributes fmaf(0, w, acc) == acc, so
// zero filling is exact.  Stride 1 runs 2 rows x 4 pixels (24 loads for 8 outputs instead of 36:
// the layer is bound by L1/L2 reads, 3.2 -> 3.7 TB/s alone), stride 2 1 row x 2 pixels (4.3 -> 4.7 TB/s):
// profiles/r02_depthwise_probe.log, scripts/experiments/dw_probe.hip.
// OUT16: the result rows are written in split-fp16 form (h | l per octet, igemm.hip) for a pointwise
// convolution that runs in precision mode f16x3; the values are exact fp32 results, rounded to h + l.
template <int STRIDE, int OUT16 = 0, int R = (STRIDE == 1 ? 2 : 1), int PX = (STRIDE == 1 ? 4 : 2)>
__global__ __launch_bounds__(256) void depthwise_kernel(const float *__restrict__ in, int B, int H, int W, int C,
                                                         const float *__restrict__ w, int pad, int OH, int OW,
                                                         const float *mean, const float *sf, const float *beta,
                                                         int act, float *__restrict__ out, int *flags)
{
    constexpr int NCOL = (PX - 1) * STRIDE + 3, NROW = (R - 1) * STRIDE + 3;
    bool ovf = false;
    const int C4 = C >> 2, XG = (OW + PX - 1) / PX, YG = (OH + R - 1) / R;
    const long long total = (long long)B * YG * XG * C4;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C4) * 4;
        long long q = idx / C4;
        const int ox0 = (int)(q % XG) * PX;
        q /= XG;
        const int oy0 = (int)(q % YG) * R;
        const int b = (int)(q / YG);
        v4f wv[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) wv[t] = *(const v4f *)(w + t * C + c);
        v4f acc[R][PX];
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int p = 0; p < PX; ++p) acc[r][p] = (v4f){0.0f, 0.0f, 0.0f, 0.0f};
        const int ix0 = ox0 * STRIDE - pad, iy0 = oy0 * STRIDE - pad;
#pragma unroll
        for (int j = 0; j < NROW; ++j) {
            const int iy = iy0 + j;
            const bool rowok = (unsigned)iy < (unsigned)H;
            const float *rowp = in + (((long long)b * H + (rowok ? iy : 0)) * W) * C + c;
            v4f x[NCOL];
#pragma unroll
            for (int k = 0; k < NCOL; ++k) {
                const int ix = ix0 + k;
                const bool ok = rowok && (unsigned)ix < (unsigned)W;
                x[k] = (v4f){0.0f, 0.0f, 0.0f, 0.0f};
                if (ok) x[k] = *(const v4f *)(rowp + (long long)ix * C);
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int ky = j - r * STRIDE;            // this input row is tap row ky of output row r
                if (ky >= 0 && ky < 3) {
#pragma unroll
                    for (int p = 0; p < PX; ++p)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                acc[r][p][i] = fmaf(x[p * STRIDE + kx][i], wv[ky * 3 + kx][i], acc[r][p][i]);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (oy0 + r >= OH) continue;
            if constexpr (OUT16) {
                typedef _Float16 v4h __attribute__((ext_vector_type(4)));
                typedef unsigned v2u __attribute__((ext_vector_type(2)));
                // 4 channels = half an octet: h at (octet * 8 + half * 2) floats, l 4 floats further
                float *o = out + (((long long)b * OH + oy0 + r) * OW + ox0) * C + (c >> 3) * 8 + ((c >> 2) & 1) * 2;
#pragma unroll
                for (int p = 0; p < PX; ++p)
                    if (ox0 + p < OW) {
                        const v4f v = bn_act4(acc[r][p], mean, sf, beta, c, act);
                        v4h h, l;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            ovf |= !(fabsf(v[e]) <= 65504.0f);       // also true for NaN
                            const float x = fminf(fmaxf(v[e], -65504.0f), 65504.0f);
                            h[e] = (_Float16)x;
                            l[e] = (_Float16)(x - (float)h[e]);
                        }
                        *(v2u *)(o + (long long)p * C) = __builtin_bit_cast(v2u, h);
                        *(v2u *)(o + (long long)p * C + 4) = __builtin_bit_cast(v2u, l);
                    }
            } else {
                float *o = out + (((long long)b * OH + oy0 + r) * OW + ox0) * C + c;
#pragma unroll
                for (int p = 0; p < PX; ++p)
                    if (ox0 + p < OW) *(v4f *)(o + (long long)p * C) = bn_act4(acc[r][p], mean, sf, beta, c, act);
            }
        }
    }
    if constexpr (OUT16) { if (ovf && flags) atomicOr(flags, 1); }
}

hipError_t launch_depthwise(const float *in, int B, int H, int W, int C, const float *w, int stride, int pad, int OH,
                            int OW, const float *mean, const float *sf, const float *beta, int act, float *out,
                            hipStream_t s, int out16, int *flags)
{
    if (C % 4 || (stride != 1 && stride != 2) || (out16 && C % 8)) return hipErrorInvalidValue;
    const int R = stride == 1 ? 2 : 1, PX = stride == 1 ? 4 : 2;      // as in the kernel
    const long long total = (long long)B * ((OH + R - 1) / R) * ((OW + PX - 1) / PX) * (C / 4);
    long long blocks = (total + 255) / 256;
    if (blocks > 256 * 64) blocks = 256 * 64;
    if (blocks < 1) blocks = 1;
    if (stride == 1 && !out16)
        hipLaunchKernelGGL(depthwise_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, s, in, B, H, W, C, w, pad, OH, OW,
                           mean, sf, beta, act, out, flags);
    else if (!out16)
        hipLaunchKernelGGL(depthwise_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, s, in, B, H, W, C, w, pad, OH, OW,
                           mean, sf, beta, act, out, flags);
    else if (stride == 1)
        hipLaunchKernelGGL((depthwise_kernel<1, 1>), dim3((unsigned)blocks), dim3(256), 0, s, in, B, H, W, C, w, pad, OH, OW,
                           mean, sf, beta, act, out, flags);
    else
        hipLaunchKernelGGL((depthwise_kernel<2, 1>), dim3((unsigned)blocks), dim3(256), 0, s, in, B, H, W, C, w, pad, OH, OW,
                           mean, sf, beta, act, out, flags);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// K7: max pool 3x3 stride 2 'SAME' on even sizes: window rows 2oy..2oy+2, cells beyond
// the edge do not take part.
__global__ __launch_bounds__(256) void maxpool_kernel(const float *__restrict__ in, int B, int H, int W, int C,
                                                       float *__restrict__ out)
{
    const int OH = H >> 1, OW = W >> 1, C4 = C >> 2;
    const long long total = (long long)B * OH * OW * C4;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C4) * 4;
        long long pix = idx / C4;
        const int ox = (int)(pix % OW);
        pix /= OW;
        const int oy = (int)(pix % OH);
        const int b = (int)(pix / OH);
        v4f m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = 2 * oy + ky;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = 2 * ox + kx;
                if (iy < H && ix < W) {
                    const v4f x = *(const v4f *)(in + (((long long)b * H + iy) * W + ix) * C + c);
#pragma unroll
                    for (int i = 0; i < 4; ++i) m[i] = x[i] > m[i] ? x[i] : m[i];
                }
            }
        }
        *(v4f *)(out + idx * 4) = m;
    }
}

hipError_t launch_maxpool(const float *in, int B, int H, int W, int C, float *out, hipStream_t s)
{
    if (C % 4 || (H & 1) || (W & 1)) return hipErrorInvalidValue;
    const long long total = (long long)B * (H / 2) * (W / 2) * (C / 4);
    long long blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(maxpool_kernel, dim3((unsigned)blocks), dim3(256), 0, s, in, B, H, W, C, out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// The FPN's top-down merges as one elementwise launch (feature_extractor.py:64-69, nearest_neighbor_upsample :79-100):
//     x4 = up2(x5) + lateral4(c4),   x3 = up2(x4) + lateral3(c3)          out[2i+a, 2j+b] = in[i, j]
// on lateral convolutions that ran EARLY, without their upsampled operand (batch 1: they finish beside the backbone's last
// layers, and what is left behind c5 is lateral5 and this kernel instead of a chain of three launches, plan.hip).  Each sum
// is one fp32 addition, exactly what the lateral convolution's epilogue performs when it adds the operand itself.
// One thread = 4 channels of one x3 position; the thread of the even (y, x) also writes its x4 value.
__global__ __launch_bounds__(256) void fpn_merge_kernel(const float *__restrict__ x5, const float *__restrict__ l4, float *__restrict__ x4,
                                                         float *__restrict__ x3, int B, int H3, int W3, int C)
{
    const int C4 = C >> 2, H4 = H3 >> 1, W4 = W3 >> 1, H5 = H4 >> 1, W5 = W4 >> 1;
    const long long total = (long long)B * H3 * W3 * C4;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C4) * 4;
        long long q = idx / C4;
        const int x = (int)(q % W3);
        q /= W3;
        const int y = (int)(q % H3), b = (int)(q / H3);
        const long long o4 = (((long long)b * H4 + (y >> 1)) * W4 + (x >> 1)) * C + c;
        const v4f a5 = *(const v4f *)(x5 + (((long long)b * H5 + (y >> 2)) * W5 + (x >> 2)) * C + c);
        const v4f a4 = *(const v4f *)(l4 + o4);
        v4f v4, v3 = *(const v4f *)(x3 + idx * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v4[e] = a5[e] + a4[e];
            v3[e] = v4[e] + v3[e];
        }
        *(v4f *)(x3 + idx * 4) = v3;
        if (!((y | x) & 1)) *(v4f *)(x4 + o4) = v4;
    }
}

hipError_t launch_fpn_merge(const float *x5, const float *l4, float *x4, float *x3, int B, int H3, int W3, int C, hipStream_t s)
{
    if (C % 4 || (H3 & 3) || (W3 & 3) || B < 1) return hipErrorInvalidValue;
    const long long total = (long long)B * H3 * W3 * (C / 4);
    long long blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(fpn_merge_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x5, l4, x4, x3, B, H3, W3, C);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
__device__ __forceinline__ int phys_of_logical(int l) { const int r = l & 7; return (l & ~7) + ((r & 1) ? 4 + (r >> 1) : (r >> 1)); }
__device__ __forceinline__ int logical_of_phys(int p) { const int r = p & 7; return (p & ~7) + (r < 4 ? 2 * r : 2 * (r - 4) + 1); }

// K8: table-driven channel gather: out[r][j] = (tab[2j]==0 ? x : y)[r][tab[2j+1]], or 0 when
// tab[2j] < 0.  Serves concat_shuffle_split (shufflenet_v2.py:94-115: z[2d+g] = (g?y:x)[d],
// new x = z[:D], new y = z[D:]) and the stage-output concat (:89) in any channel order.
__global__ __launch_bounds__(256) void gather_kernel(const float *__restrict__ x, int xs, const float *__restrict__ y,
                                                      int ys, long long rows, const int *__restrict__ tab, int Cout,
                                                      float *__restrict__ out)
{
    const long long total = rows * Cout;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(idx % Cout);
        const long long r = idx / Cout;
        const int src = tab[2 * j], k = tab[2 * j + 1];
        float v = 0.0f;
        if (src == 0) v = x[r * xs + k];
        else if (src == 1) v = y[r * ys + k];
        out[idx] = v;
    }
}

hipError_t launch_gather_channels(const float *x, int xs, const float *y, int ys, long long rows, const int *tab,
                                  int Cout, float *out, hipStream_t s)
{
    const long long total = rows * Cout;
    long long blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(gather_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, xs, y, ys, rows, tab, Cout, out);
    return hipGetLastError();
}

// Table-driven row gather from several dense tensors of ONE allocation with one row stride (the y half of a ShuffleNet stage
// output, shufflenet_v2.py:89: channels that earlier units left untouched): out[r][c] = base[r * rs + src[c]] (byte
// offsets; src[c] < 0: 0) for c < Cw, four channels per thread, 16-byte stores.
__global__ __launch_bounds__(256) void gather_rows_kernel(const unsigned char *__restrict__ base, const int *__restrict__ src, int rs,
                                                           long long rows, int Cw, float *__restrict__ out, int ors)
{
    const int C4 = Cw >> 2;
    const long long total = rows * C4;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C4) * 4;
        const long long r = idx / C4;
        const int4 sv = *(const int4 *)(src + c);
        const unsigned char *row = base + r * rs;
        float4 v;
        v.x = sv.x >= 0 ? *(const float *)(row + sv.x) : 0.0f;
        v.y = sv.y >= 0 ? *(const float *)(row + sv.y) : 0.0f;
        v.z = sv.z >= 0 ? *(const float *)(row + sv.z) : 0.0f;
        v.w = sv.w >= 0 ? *(const float *)(row + sv.w) : 0.0f;
        *(float4 *)(out + r * ors + c) = v;
    }
}

hipError_t launch_gather_rows(const float *base, const int *src, int rs, long long rows, int Cw, float *out, int ors, hipStream_t s)
{
    if (!base || !src || !out || Cw < 4 || (Cw & 3) || (ors & 3) || (rs & 3) || rows < 1 || (reinterpret_cast<uintptr_t>(out) & 15)) return hipErrorInvalidValue;
    const long long total = rows * (Cw / 4);
    long long blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (const unsigned char *)base, src, rs, rows, Cw, out, ors);
    return hipGetLastError();
}

// to_phys = 1: in [rows][C] logical -> out [rows][Cpad] physical (pad channels zero)
// to_phys = 0: in [rows][Cpad] physical -> out [rows][C] logical
// to_phys = 2 / 3: the same with the physical side in split-fp16 (S16) rows: per octet of 8 physical
//   channels 8 halves h then 8 halves l, value = h + l (igemm.hip)
// split > 0 (to_phys = 0 only): a two-part physical row [first `split` logical channels in their own standard layout over
// Cpad / 2 | the rest likewise] -- a ShuffleNet stage output (plan.hip)
__global__ __launch_bounds__(256) void permute_kernel(const float *__restrict__ in, long long rows, int C, int Cpad,
                                                       int to_phys, float *__restrict__ out, int split)
{
    const int Cw = (to_phys & 1) ? Cpad : C;
    const long long total = rows * Cw;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(idx % Cw);
        const long long r = idx / Cw;
        if (to_phys == 1) {
            const int l = logical_of_phys(j);
            out[idx] = l < C ? in[r * C + l] : 0.0f;
        } else if (to_phys == 0) {
            out[idx] = in[r * Cpad + ((split > 0 && j >= split) ? (Cpad >> 1) + phys_of_logical(j - split) : phys_of_logical(j))];
        } else if (to_phys == 3) {
            const int l = logical_of_phys(j);
            float x = l < C ? in[r * C + l] : 0.0f;
            x = fminf(fmaxf(x, -65504.0f), 65504.0f);
            const _Float16 h = (_Float16)x;
            const _Float16 lo = (_Float16)(x - (float)h);
            _Float16 *row = (_Float16 *)(out + r * Cpad);
            row[(j >> 3) * 16 + (j & 7)] = h;
            row[(j >> 3) * 16 + 8 + (j & 7)] = lo;
        } else {
            const int p = phys_of_logical(j);
            const _Float16 *row = (const _Float16 *)(in + r * Cpad);
            out[idx] = (float)row[(p >> 3) * 16 + (p & 7)] + (float)row[(p >> 3) * 16 + 8 + (p & 7)];
        }
    }
}

hipError_t launch_permute_channels(const float *in, long long rows, int C, int Cpad, int to_phys, float *out,
                                   hipStream_t s, int split)
{
    if (split && to_phys != 0) return hipErrorInvalidValue;
    const long long total = rows * ((to_phys & 1) ? Cpad : C);
    long long blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(permute_kernel, dim3((unsigned)blocks), dim3(256), 0, s, in, rows, C, Cpad, to_phys, out, split);
    return hipGetLastError();
}
