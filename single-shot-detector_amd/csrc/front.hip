// MobileNet's first three layers in ONE kernel: Conv2d_0 (3x3 stride 2 on the uint8 frame, 3 -> 32, batch norm, ReLU6;
// mobilenet_v1.py:37-45 behind the normalisation of model.py / detector.py) -> Conv2d_1_depthwise (3x3, batch norm, ReLU6)
// -> Conv2d_1_pointwise (32 -> 64, batch norm, ReLU6; mobilenet_v1.py:59-67, depthwise_conv.py:5-26).  The 32-channel
// tensor at half resolution -- the second-largest tensor of a forward, 18.3 MB per 640x896 frame, written by one launch
// and read back by the next -- never exists: a block computes the first convolution for the 14 x 18 patch its 12 x 16
// output tile's depthwise taps read (1.31x the arithmetic of the plain layer, 27 fused multiply-adds per value) and keeps
// it in LDS.  One launch less per forward, 36.6 MB less traffic per frame.
//
//   tile        12 x 16 output positions of one image x all 64 output channels; a block walks tiles b, b + G, ...
//   phase 1     first convolution: one LANE = one patch position, all 32 channels (the arithmetic of first_conv_px_kernel,
//               elementwise.hip: 27 inputs unpacked and normalised once, wave-uniform weights from scalar loads, the
//               (ky,kx,ci)-ordered fmaf chain, batch norm in three separately rounded steps, activation); positions outside
//               the layer's output are the depthwise layer's zero padding.  The frame bytes of the NEXT tile are fetched
//               here and used one tile later.
//   phase 2     depthwise 3x3: a thread = 4 channels x one column x 6 rows (8 patch rows x 3 columns read once); the
//               (ky,kx)-ordered fmaf chain, batch norm, activation of dwpw_stream.hip -> the A image of the 1x1
//   phase 3     1x1 on v_mfma_f32_32x32x2_f32, weights (the wave's fragments stay in registers for the whole kernel) as the
//               A operand and positions as the B operand, k in channel order = the chain of igemm.hip; 12 accumulator
//               tiles of 32 channels x 32 positions, three per wave
//   phase 4     batch norm + activation, 16-byte stores straight from the accumulators (a lane holds 4 consecutive channels
//               of one position), as dwpw_stream.hip's dense epilogue
// Bit-identical to first_conv_px_kernel -> dwpw_stream_kernel (and so to the three-kernel chain and the oracle).
#include "ssd_internal.h"

typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

namespace {
constexpr int TY = 12, TX = 16, PH = TY + 2, PW = TX + 2, NSLOT = PH * PW;     // 252 patch positions: lanes 0 .. 251 of the block
constexpr int OFF_A = NSLOT * 128, OFF_P = OFF_A + TY * TX * 128, LDS_BYTES = OFF_P + 3 * 64 * 4;
static_assert(NSLOT <= 256 && LDS_BYTES <= 64 * 1024, "one lane per patch position, two blocks per CU");
}

// The frame bytes under the 3x3 stride-2 window of first-convolution output (cy, cx) of image b (coordinates inside the layer's
// output): three filter rows of 9 contiguous bytes at byte ((b H + 2 cy + ky) W + 2 cx) 3, each fetched as three aligned
// dwords -- the addressing of first_conv_px_kernel (elementwise.hip).  Row 2 cy + 2 may lie below the frame (even H): it is
// fetched from row 0 and masked in frame_unpack.
static __device__ __forceinline__ void frame_fetch(const __amdgpu_buffer_rsrc_t irsrc, bool live, int b, int H, int W, int cy, int cx,
                                                   unsigned (&raw)[9])
{
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = 2 * cy + ky;
        const int ad = ((b * H + (iy < H ? iy : 0)) * W + 2 * cx) * 3;
        const int a0 = live ? (ad & ~3) : (int)0x80000000u;
        raw[ky * 3 + 0] = __builtin_amdgcn_raw_buffer_load_b32(irsrc, a0, 0, 0);
        raw[ky * 3 + 1] = __builtin_amdgcn_raw_buffer_load_b32(irsrc, a0, 4, 0);
        raw[ky * 3 + 2] = __builtin_amdgcn_raw_buffer_load_b32(irsrc, a0, 8, 0);
    }
}

// ... -> the 27 normalised inputs x / 255 -> 2 x - 1 in (ky, kx, ci) order; taps below / right of the frame ('SAME' on even
// sizes pads there only) are 0.  The arithmetic of first_conv_px_kernel, value for value.
static __device__ __forceinline__ void frame_unpack(const unsigned (&raw)[9], int b, int H, int W, int cy, int cx, float (&x)[27])
{
    const float inv255 = (float)(1.0 / 255.0);
    const bool xok = 2 * cx + 2 < W;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = 2 * cy + ky;
        const bool yok = ky < 2 || iy < H;
        const int ad = ((b * H + (iy < H ? iy : 0)) * W + 2 * cx) * 3;
        const int sh = ad & 3;
        const unsigned w0 = raw[ky * 3], w1 = raw[ky * 3 + 1], w2 = raw[ky * 3 + 2];
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
}

// Round 6: the same for a RESIZED frame (resize_keeping_aspect_ratio fused, pipeline.py:138-194 -- the index rule of
// elementwise.hip K1 / K1d, value for value) whose width is not reduced (srcW <= nw, i.e. every COCO image at min_dimension 640):
// the three taps of a filter row then read source columns sx(2 cx), sx(2 cx + 1), sx(2 cx + 2) that lie at most two pixels apart,
// so a filter row is still 9 contiguous source bytes from the first tap's pixel -- three aligned dwords, like the frame of the
// network's own size -- and tap kx takes the pixel sx(2 cx + kx) - sx(2 cx) in {0, 1, 2} of them.  Rows are three independent
// source rows (any vertical scale).  Beyond the resize's target (the zero pad band) a tap is 0 before the normalisation, beyond
// the padded frame 0 after it.
struct FrontGeom { int srcH, srcW, nh, nw; float hs, ws; unsigned off0; };      // off0: byte offset of frame 0 from the (4-byte aligned) base
// MODE of the two kernels below: 0 = frames of the network's size, 1 = resized frames of ONE size (FrontGeom; frame b starts at byte
// b * srcH * srcW * 3), 2 = a batch of frames of DIFFERENT sizes (ssd_forward_mixed): frame b's geometry and byte offset are entry
// first + b of the table in the kernel's arguments -- a tile belongs to one frame, so the entry is read with scalar loads
struct FrontMixed { MixedGeom mg; int first; unsigned bytes4; };
template <int MODE> struct FrontGeomArg { typedef FrontGeom type; };
template <> struct FrontGeomArg<2> { typedef FrontMixed type; };
template <int MODE>
static __device__ __forceinline__ void front_geom_of(const typename FrontGeomArg<MODE>::type &ga, int b, FrontGeom &g, unsigned &base)
{
    if constexpr (MODE == 2) {
        const FrameGeom f = ga.mg.f[ga.first + b];
        g.srcH = f.srcH; g.srcW = f.srcW; g.nh = f.nh; g.nw = f.nw; g.hs = f.hs; g.ws = f.ws;
        base = f.off;
    } else {
        g = ga;
        base = ga.off0 + (unsigned)b * (unsigned)ga.srcH * (unsigned)ga.srcW * 3u;
    }
}

static __device__ __forceinline__ int front_src(int dst, float scale, int n)
{
    const int v = (int)floorf((float)dst * scale);
    return v < n - 1 ? v : n - 1;
}

static __device__ __forceinline__ void frame_fetch_gen(const __amdgpu_buffer_rsrc_t irsrc, bool live, unsigned base, const FrontGeom &g, int cy, int cx,
                                                       unsigned (&raw)[9])
{
    const int sx0 = front_src(2 * cx, g.ws, g.srcW);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int sy = front_src(2 * cy + ky, g.hs, g.srcH);
        const int ad = (int)(base + (unsigned)(sy * g.srcW + sx0) * 3u);
        const int a0 = live ? (ad & ~3) : (int)0x80000000u;
        raw[ky * 3 + 0] = __builtin_amdgcn_raw_buffer_load_b32(irsrc, a0, 0, 0);
        raw[ky * 3 + 1] = __builtin_amdgcn_raw_buffer_load_b32(irsrc, a0, 4, 0);
        raw[ky * 3 + 2] = __builtin_amdgcn_raw_buffer_load_b32(irsrc, a0, 8, 0);
    }
}

static __device__ __forceinline__ void frame_unpack_gen(const unsigned (&raw)[9], unsigned base, int H, int W, const FrontGeom &g, int cy, int cx, float (&x)[27])
{
    const float inv255 = (float)(1.0 / 255.0);
    const int sx0 = front_src(2 * cx, g.ws, g.srcW);
    int off[3];
    bool xin[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        off[kx] = 3 * (front_src(2 * cx + kx, g.ws, g.srcW) - sx0);      // 0, 3 or 6
        xin[kx] = 2 * cx + kx < g.nw;
    }
    const bool xok = 2 * cx + 2 < W;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = 2 * cy + ky;
        const bool yok = ky < 2 || iy < H;
        const bool yin = iy < g.nh;
        const int sy = front_src(iy, g.hs, g.srcH);
        const int sh = (int)((base + (unsigned)(sy * g.srcW + sx0) * 3u) & 3u);
        const unsigned w0 = raw[ky * 3], w1 = raw[ky * 3 + 1], w2 = raw[ky * 3 + 2];
        const unsigned d0 = __builtin_amdgcn_alignbyte(w1, w0, sh);      // source bytes 0..3 of the row's span
        const unsigned d1 = __builtin_amdgcn_alignbyte(w2, w1, sh);      // 4..7
        const unsigned d2 = w2 >> (8 * sh);                              // 8..
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const unsigned lo = off[kx] < 4 ? d0 : d1, hi = off[kx] < 4 ? d1 : d2;
            const unsigned px = __builtin_amdgcn_alignbyte(hi, lo, off[kx] & 3);
            const bool inimg = yin && xin[kx];
#pragma unroll
            for (int ci = 0; ci < 3; ++ci) {
                float v = inimg ? (float)((px >> (8 * ci)) & 0xffu) : 0.0f;
                v = v * inv255;
                v = 2.0f * v - 1.0f;
                if (!yok || (kx == 2 && !xok)) v = 0.0f;
                x[ky * 9 + kx * 3 + ci] = v;
            }
        }
    }
}

// (the read-only operands as `const __restrict__` kernel parameters: with them inside the by-value struct the compiler cannot
//  prove the first convolution's weights invariant and loads them per lane into vector registers instead of with scalar loads)
struct FrontDims { int B, H, W, act0, dact, act, tiles_y, tiles_x; };
template <int MODE>
__global__ __launch_bounds__(256, 2) void front_kernel(const uint8_t *__restrict__ a_img, const float *__restrict__ a_w0,
                                                        const float *__restrict__ a_m0, const float *__restrict__ a_s0,
                                                        const float *__restrict__ a_b0, const float *__restrict__ a_dwpack,
                                                        const float *__restrict__ a_wt, const float *__restrict__ a_mean,
                                                        const float *__restrict__ a_sf, const float *__restrict__ a_beta,
                                                        float *__restrict__ a_out, const FrontDims a, const typename FrontGeomArg<MODE>::type gg)
{
    constexpr bool GEN = MODE != 0;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = a.H, W = a.W, OH = H >> 1, OW = W >> 1;
    const int tiles_img = a.tiles_y * a.tiles_x, total = a.B * tiles_img;
    // (GEN: the size rounded up to whole dwords -- a row's third dword may reach <= 3 bytes past a size that is no multiple of 4)
    int ibytes;
    if constexpr (MODE == 2) ibytes = (int)gg.bytes4;
    else if constexpr (MODE == 1) ibytes = (int)((((long long)a.B * gg.srcH * gg.srcW * 3) + gg.off0 + 3) & ~3LL);
    else ibytes = (int)((long long)a.B * H * W * 3);
    const __amdgpu_buffer_rsrc_t irsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a_img, 0, ibytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a_out, 0, (int)((long long)a.B * OH * OW * 64 * 4), 0x00020000);
    constexpr unsigned OOB = 0x80000000u;

    // ---- per-thread constants
    // phase 1: patch position of this lane
    const int ppy = tid / PW, ppx = tid - ppy * PW;
    // phase 2: 4 channels c4, column dcol, rows 6 dhalf .. 6 dhalf + 5
    const int c4 = tid & 7;
    v4f dwv[9], dmean, dsf, dbeta;
#pragma unroll
    for (int t = 0; t < 9; ++t) dwv[t] = *(const v4f *)(a_dwpack + t * 32 + c4 * 4);
    dmean = *(const v4f *)(a_dwpack + 9 * 32 + c4 * 4);
    dsf = *(const v4f *)(a_dwpack + 10 * 32 + c4 * 4);
    dbeta = *(const v4f *)(a_dwpack + 11 * 32 + c4 * 4);
    // phase 3: accumulator tile j of this wave = pair 3 wave + j of (position tile m = pair >> 1, channel tile n = pair & 1)
    v4f wf[3][4];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int n = (3 * wave + j) & 1;
#pragma unroll
        for (int g = 0; g < 4; ++g) wf[j][g] = *(const v4f *)(a_wt + (n * 32 + (lane & 31)) * 32 + (2 * g + (lane >> 5)) * 4);
    }
    int roff[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) roff[g] = (lane & 31) * 128 + (((2 * g + (lane >> 5)) ^ (((lane & 31) >> 1) & 7)) << 4);
    // phase 4: batch norm of the 64 output channels -> LDS
    {
        float *pp = (float *)(lds + OFF_P);
        if (tid < 64) { pp[tid] = a_mean[tid]; pp[64 + tid] = a_sf[tid]; pp[128 + tid] = a_beta[tid]; }
    }
    const bool dact_on = a.dact >= 1, pact_on = a.act >= 1;
    const float dact_hi = a.dact == 2 ? 6.0f : __builtin_inff(), pact_hi = a.act == 2 ? 6.0f : __builtin_inff();

    // the frame bytes under this lane's patch position of tile `t`: three filter rows of 9 contiguous bytes, each as three
    // aligned dwords (the arithmetic of first_conv_px_kernel); positions outside the layer's output are fetched clamped
    // and zeroed after the activation
    unsigned raw[9];
    auto fetch = [&](int t) {
        const int b = t / tiles_img, r = t - b * tiles_img, ty = r / a.tiles_x, tx = r - ty * a.tiles_x;
        int fy = ty * TY - 1 + ppy, fx = tx * TX - 1 + ppx;
        fy = fy < 0 ? 0 : (fy >= OH ? OH - 1 : fy);
        fx = fx < 0 ? 0 : (fx >= OW ? OW - 1 : fx);
        if constexpr (GEN) {
            FrontGeom g;
            unsigned base;
            front_geom_of<MODE>(gg, b, g, base);
            frame_fetch_gen(irsrc, tid < NSLOT, base, g, fy, fx, raw);
        } else frame_fetch(irsrc, tid < NSLOT, b, H, W, fy, fx, raw);
    };
    int t = blockIdx.x;
    if (t < total) fetch(t);
    __syncthreads();

    for (; t < total; t += (int)gridDim.x) {
        const int b = t / tiles_img, rt = t - b * tiles_img, ty = rt / a.tiles_x, tx = rt - ty * a.tiles_x;
        // ---- phase 1: first convolution of this lane's patch position -> patch row tid (chunk c at slot c ^ (tid & 7))
        {
            const int fy = ty * TY - 1 + ppy, fx = tx * TX - 1 + ppx;
            const bool inside = tid < NSLOT && (unsigned)fy < (unsigned)OH && (unsigned)fx < (unsigned)OW;
            const int cy = fy < 0 ? 0 : (fy >= OH ? OH - 1 : fy), cx = fx < 0 ? 0 : (fx >= OW ? OW - 1 : fx);
            float x[27];
            if constexpr (GEN) {
                FrontGeom g;
                unsigned base;
                front_geom_of<MODE>(gg, b, g, base);
                frame_unpack_gen(raw, base, H, W, g, cy, cx, x);
            } else frame_unpack(raw, b, H, W, cy, cx, x);
            // the next tile's bytes: in flight until the next iteration's phase 1
            if (t + (int)gridDim.x < total) fetch(t + (int)gridDim.x);
            unsigned char *prow = lds + tid * 128;
            {                                                // all 32 accumulators in one pass over the 27 taps
                float acc[32];
#pragma unroll
                for (int i = 0; i < 32; ++i) acc[i] = 0.0f;
#pragma unroll
                for (int k = 0; k < 27; ++k) {
                    const float *wr = a_w0 + k * 32;
#pragma unroll
                    for (int i = 0; i < 32; ++i) acc[i] = fmaf(x[k], wr[i], acc[i]);
                }
#pragma unroll
                for (int i = 0; i < 32; ++i) {
                    const float tq = (acc[i] - a_m0[i]) * a_s0[i];
                    float v = tq + a_b0[i];
                    if (a.act0 >= 1) v = v > 0.0f ? v : 0.0f;
                    if (a.act0 == 2) v = v < 6.0f ? v : 6.0f;
                    acc[i] = inside ? v : 0.0f;
                }
                if (tid < NSLOT) {
#pragma unroll
                    for (int i = 0; i < 32; i += 4)
                        *(v4f *)(prow + (((i >> 2) ^ (tid & 7)) << 4)) = (v4f){acc[i], acc[i + 1], acc[i + 2], acc[i + 3]};
                }
            }
        }
        __syncthreads();                 // (1) the patch is complete; the previous tile's MFMAs have read the A image
        // ---- phase 2: depthwise 3x3 + batch norm + activation -> A image rows (dhalf * 6 + o) * 16 + dcol
        {
            // (the thread index through an opaque copy: the 24 + 6 swizzled LDS addresses below are loop invariants, and
            //  hoisted out of the tile loop they cost ~60 registers -- with them the kernel spilled)
            int tz = tid;
            asm volatile("" : "+v"(tz));
            const int c4 = tz & 7, dcol = (tz >> 3) & 15, dhalf = tz >> 7;
            // rows in a rolling window: patch row r is tap row 0 of output r, row 1 of output r - 1, row 2 of output r - 2 --
            // every output still receives its taps in (ky,kx) order
            v4f part[3];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                v4f xr[3];
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int slot = (dhalf * 6 + r) * PW + dcol + kx;
                    xr[kx] = *(const v4f *)(lds + slot * 128 + ((c4 ^ (slot & 7)) << 4));
                }
#pragma unroll
                for (int ky = 2; ky >= 0; --ky) {
                    const int o = r - ky;
                    if (o < 0 || o >= 6) continue;
                    v4f &v = part[o % 3];
                    if (ky == 0) v = (v4f){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaf(xr[kx][e], dwv[ky * 3 + kx][e], v[e]);
                    if (ky == 2) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float tq = (v[e] - dmean[e]) * dsf[e];
                            v[e] = tq + dbeta[e];
                            if (dact_on) v[e] = __builtin_amdgcn_fmed3f(v[e], 0.0f, dact_hi);
                        }
                        const int m = (dhalf * 6 + o) * TX + dcol;
                        *(v4f *)(lds + OFF_A + m * 128 + ((c4 ^ ((m >> 1) & 7)) << 4)) = v;
                    }
                }
            }
        }
        __syncthreads();                 // (2) the A image is complete; the patch is free for the next tile
        // ---- phases 3 + 4: 1x1 and epilogue of this wave's three accumulator tiles
        {
            const float *pp = (const float *)(lds + OFF_P);
            const int eh = lane >> 5;
            v16f acc[3];
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                v4f af[3];
#pragma unroll
                for (int j = 0; j < 3; ++j) af[j] = *(const v4f *)(lds + OFF_A + ((3 * wave + j) >> 1) * 4096 + roff[g]);
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int j = 0; j < 3; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[j][g][k], af[j][k], acc[j], 0, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                // acc[j][r]: channel nt * 32 + (r & 3) + 8 (r >> 2) + 4 (lane >> 5), position mt * 32 + (lane & 31) of the tile
                const int pair = 3 * wave + j, mt = pair >> 1, nt = pair & 1;
                const int p = mt * 32 + (lane & 31);
                const int oy = ty * TY + (p >> 4), ox = tx * TX + (p & 15);
                const bool ok = oy < OH && ox < OW;
                const int pos = (b * OH + oy) * OW + ox;
#pragma unroll
                for (int m4 = 0; m4 < 4; ++m4) {
                    const int cl = nt * 32 + 8 * m4 + 4 * eh;
                    const v4f mean = *(const v4f *)(pp + cl), sf = *(const v4f *)(pp + 64 + cl), beta = *(const v4f *)(pp + 128 + cl);
                    v4f v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float xv = acc[j][4 * m4 + e];
                        const float tq = (xv - mean[e]) * sf[e];
                        xv = tq + beta[e];
                        if (pact_on) xv = __builtin_amdgcn_fmed3f(xv, 0.0f, pact_hi);
                        v[e] = xv;
                    }
                    const unsigned o = ok ? (unsigned)((pos * 64 + cl) * 4) : OOB;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, v), orsrc, (int)o, 0, 0);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// ShuffleNet's first two layers in one kernel: Conv1 (3x3 stride 2 on the uint8 frame, 3 -> 24, batch norm, ReLU) -> MaxPool
// (3x3 stride 2 'SAME'; shufflenet_v2.py:50-54).  The 24-channel tensor at half resolution (9.8 MB per 640x640 frame) stays
// in LDS: a block computes the convolution for the 15 x 17 patch under its 7 x 8 pooled positions (1.14x the plain layer's
// arithmetic), one lane per patch position as above, then takes the maxima.  Cells beyond the convolution's output do not
// take part in a window (maxpool_kernel, elementwise.hip): they are -inf in the patch.  max is exact: bit-identical to
// first_conv_px_kernel -> maxpool_kernel.
namespace {
constexpr int QY = 7, QX = 8, QPH = 2 * QY + 1, QPW = 2 * QX + 1, QSLOT = QPH * QPW;      // 255 patch positions
static_assert(QSLOT <= 256, "one lane per patch position");
}
struct FrontPoolDims { int B, H, W, act0, tiles_y, tiles_x; };
// COUT: channels the convolution computes (24: the layer's width is 3 whole octets, so its physical channels are 0 .. 23);
// CS: physical channels of a stored row -- 24, or 32 inside the network, whose tensors carry 8 zero pad channels (stored as zeros
// here: round 5 -- computing them cost a quarter of the kernel's multiply-adds)
// MODE: as front_kernel
template <int COUT, int CS, int MODE>
__global__ __launch_bounds__(256, 2) void front_pool_kernel(const uint8_t *__restrict__ a_img, const float *__restrict__ a_w0,
                                                             const float *__restrict__ a_m0, const float *__restrict__ a_s0,
                                                             const float *__restrict__ a_b0, float *__restrict__ a_out,
                                                             const FrontPoolDims a, const typename FrontGeomArg<MODE>::type gg)
{
    constexpr bool GEN = MODE != 0;
    __shared__ __attribute__((aligned(16))) unsigned char lds[QSLOT * 128];      // patch rows of 128 B: 6 chunks used, chunk c at slot c ^ (row & 7)
    const int tid = threadIdx.x;
    const int H = a.H, W = a.W, OH = H >> 1, OW = W >> 1, PH2 = OH >> 1, PW2 = OW >> 1;
    const int tiles_img = a.tiles_y * a.tiles_x, total = a.B * tiles_img;
    int ibytes;
    if constexpr (MODE == 2) ibytes = (int)gg.bytes4;
    else if constexpr (MODE == 1) ibytes = (int)((((long long)a.B * gg.srcH * gg.srcW * 3) + gg.off0 + 3) & ~3LL);
    else ibytes = (int)((long long)a.B * H * W * 3);
    const __amdgpu_buffer_rsrc_t irsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a_img, 0, ibytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a_out, 0, (int)((long long)a.B * PH2 * PW2 * CS * 4), 0x00020000);
    const int ppy = tid / QPW, ppx = tid - ppy * QPW;
    unsigned raw[9];
    auto fetch = [&](int t) {
        const int b = t / tiles_img, r = t - b * tiles_img, ty = r / a.tiles_x, tx = r - ty * a.tiles_x;
        int fy = 2 * QY * ty + ppy, fx = 2 * QX * tx + ppx;
        fy = fy >= OH ? OH - 1 : fy;
        fx = fx >= OW ? OW - 1 : fx;
        if constexpr (GEN) {
            FrontGeom g;
            unsigned base;
            front_geom_of<MODE>(gg, b, g, base);
            frame_fetch_gen(irsrc, tid < QSLOT, base, g, fy, fx, raw);
        } else frame_fetch(irsrc, tid < QSLOT, b, H, W, fy, fx, raw);
    };
    int t = blockIdx.x;
    if (t < total) fetch(t);
    for (; t < total; t += (int)gridDim.x) {
        const int b = t / tiles_img, rt = t - b * tiles_img, ty = rt / a.tiles_x, tx = rt - ty * a.tiles_x;
        {   // ---- phase 1: the convolution at this lane's patch position
            const int fy = 2 * QY * ty + ppy, fx = 2 * QX * tx + ppx;
            const bool inside = tid < QSLOT && fy < OH && fx < OW;
            const int cy = fy >= OH ? OH - 1 : fy, cx = fx >= OW ? OW - 1 : fx;
            float x[27];
            if constexpr (GEN) {
                FrontGeom g;
                unsigned base;
                front_geom_of<MODE>(gg, b, g, base);
                frame_unpack_gen(raw, base, H, W, g, cy, cx, x);
            } else frame_unpack(raw, b, H, W, cy, cx, x);
            if (t + (int)gridDim.x < total) fetch(t + (int)gridDim.x);
            unsigned char *prow = lds + tid * 128;
            {                                                // all COUT accumulators in one pass over the 27 taps
                float acc[COUT];
#pragma unroll
                for (int i = 0; i < COUT; ++i) acc[i] = 0.0f;
#pragma unroll
                for (int k = 0; k < 27; ++k) {
                    const float *wr = a_w0 + k * CS;             // (weights [27][CS])
#pragma unroll
                    for (int i = 0; i < COUT; ++i) acc[i] = fmaf(x[k], wr[i], acc[i]);
                }
#pragma unroll
                for (int i = 0; i < COUT; ++i) {
                    const float tq = (acc[i] - a_m0[i]) * a_s0[i];
                    float v = tq + a_b0[i];
                    if (a.act0 >= 1) v = v > 0.0f ? v : 0.0f;
                    if (a.act0 == 2) v = v < 6.0f ? v : 6.0f;
                    acc[i] = inside ? v : -__builtin_inff();
                }
                if (tid < QSLOT) {
#pragma unroll
                    for (int i = 0; i < COUT; i += 4)
                        *(v4f *)(prow + (((i >> 2) ^ (tid & 7)) << 4)) = (v4f){acc[i], acc[i + 1], acc[i + 2], acc[i + 3]};
                }
            }
        }
        __syncthreads();
        // ---- phase 2: 56 pooled positions x CS / 4 channel quads: a thread = (quad tid & 7, position tid >> 3 [+ 32]); quads past the
        // computed channels are the tensor's zero pad channels
#pragma unroll
        for (int rnd = 0; rnd < 2; ++rnd) {
            const int c4 = tid & 7, p = (tid >> 3) + 32 * rnd;
            if (c4 < CS / 4 && p < QY * QX) {
                const int py = p / QX, qx = p - py * QX;
                const int oy = QY * ty + py, ox = QX * tx + qx;
                v4f m = {0.0f, 0.0f, 0.0f, 0.0f};
                if (c4 < COUT / 4) {
                    m = (v4f){-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) {
                            const int slot = (2 * py + ky) * QPW + 2 * qx + kx;
                            const v4f xv = *(const v4f *)(lds + slot * 128 + ((c4 ^ (slot & 7)) << 4));
#pragma unroll
                            for (int e = 0; e < 4; ++e) m[e] = xv[e] > m[e] ? xv[e] : m[e];
                        }
                }
                const unsigned o = (oy < PH2 && ox < PW2) ? (unsigned)((((b * PH2 + oy) * PW2 + ox) * CS + c4 * 4) * 4) : 0x80000000u;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, m), orsrc, (int)o, 0, 0);
            }
        }
        __syncthreads();                 // the patch is free for the next tile
    }
}

bool front_pool_supports(int B, int H, int W, int C0)
{
    if ((C0 != 24 && C0 != 32) || B < 1 || H < 4 || W < 4 || (H & 3) || (W & 3)) return false;
    return (long long)B * H * W * 3 < (1LL << 31);
}

// resized frames: the width must not shrink (a filter row = 9 contiguous source bytes), 32-bit byte offsets into the frames
bool front_gen_supports(int B, int srcH, int srcW, int nh, int nw)
{
    return B >= 1 && srcH >= 1 && srcW >= 1 && nh >= 1 && nw >= srcW && (long long)B * srcH * srcW * 3 + 6 < (1LL << 31);
}

// ... frames first .. first + B - 1 of a mixed-size batch: each of them, and the table's byte range (*end_bytes: its exact end)
bool front_mixed_supports(const MixedGeom &mg, int first, int B, int H, int W, unsigned *end_bytes)
{
    if (B < 1 || first < 0 || first + B > SSD_MIXED_MAX) return false;
    unsigned long long end = 0;
    for (int b = first; b < first + B; ++b) {
        const FrameGeom &g = mg.f[b];
        if (g.srcH < 1 || g.srcW < 1 || g.nh < 1 || g.nw < g.srcW || g.nh > H || g.nw > W) return false;
        end = std::max(end, (unsigned long long)g.off + (unsigned long long)g.srcH * g.srcW * 3);
    }
    if (end + 6 >= (1ull << 31)) return false;
    if (end_bytes) *end_bytes = (unsigned)end;                // (the exact end; the launch rounds it up behind its base adjustment)
    return true;
}

// src: null = frames of the network's size [B,H,W,3]; else {srcH, srcW, nh, nw}: frames [B,srcH,srcW,3] resized to [nh,nw] and padded
// to [H,W] on the fly (front_gen_supports: the width does not shrink); mixed: frames of different sizes, entries first .. of the table
// (then img is the base the entries' byte offsets count from)
hipError_t launch_front_pool(const uint8_t *img, int B, int H, int W, const float *w0, int C0, const float *m0, const float *s0, const float *b0,
                             int act0, float *out, hipStream_t s, const int *src, const MixedGeom *mixed, int first)
{
    if (!img || !w0 || !m0 || !s0 || !b0 || !out || !front_pool_supports(B, H, W, C0)) return hipErrorInvalidValue;
    if (src && (!front_gen_supports(B, src[0], src[1], src[2], src[3]) || src[2] > H || src[3] > W)) return hipErrorInvalidValue;
    FrontPoolDims d = {B, H, W, act0, (H / 4 + QY - 1) / QY, (W / 4 + QX - 1) / QX};
    const long long total = (long long)B * d.tiles_y * d.tiles_x;
    if (total > 0x7fffffffLL) return hipErrorInvalidValue;
    const int grid = (int)(total < 4096 ? total : 4096);
    // (resized / mixed: the kernel's dword loads are aligned relative to its base -- a base that is no multiple of 4 is rounded down
    //  and the remainder added to the frames' byte offsets, elementwise.hip launch_first_conv_gen)
    const unsigned adj = (src || mixed) ? (unsigned)((uintptr_t)img & 3u) : 0u;
    img -= adj;
    if (mixed) {
        FrontMixed fm;
        if (!front_mixed_supports(*mixed, first, B, H, W, &fm.bytes4)) return hipErrorInvalidValue;
        fm.mg = *mixed;
        for (int b = first; b < first + B; ++b) fm.mg.f[b].off += adj;
        fm.bytes4 = (fm.bytes4 + adj + 3u) & ~3u;
        fm.first = first;
        if (C0 == 24) hipLaunchKernelGGL((front_pool_kernel<24, 24, 2>), dim3((unsigned)grid), dim3(256), 0, s, img, w0, m0, s0, b0, out, d, fm);
        else hipLaunchKernelGGL((front_pool_kernel<24, 32, 2>), dim3((unsigned)grid), dim3(256), 0, s, img, w0, m0, s0, b0, out, d, fm);
        return hipGetLastError();
    }
    const FrontGeom g = src ? FrontGeom{src[0], src[1], src[2], src[3], (float)src[0] / (float)src[2], (float)src[1] / (float)src[3], adj}
                            : FrontGeom{H, W, H, W, 1.0f, 1.0f, 0u};
    if (src) {
        if (C0 == 24) hipLaunchKernelGGL((front_pool_kernel<24, 24, 1>), dim3((unsigned)grid), dim3(256), 0, s, img, w0, m0, s0, b0, out, d, g);
        else hipLaunchKernelGGL((front_pool_kernel<24, 32, 1>), dim3((unsigned)grid), dim3(256), 0, s, img, w0, m0, s0, b0, out, d, g);
    } else {
        if (C0 == 24) hipLaunchKernelGGL((front_pool_kernel<24, 24, 0>), dim3((unsigned)grid), dim3(256), 0, s, img, w0, m0, s0, b0, out, d, g);
        else hipLaunchKernelGGL((front_pool_kernel<24, 32, 0>), dim3((unsigned)grid), dim3(256), 0, s, img, w0, m0, s0, b0, out, d, g);
    }
    return hipGetLastError();
}

bool front_supports(int B, int H, int W, int C0, int K, int Cout)
{
    if (C0 != 32 || K != 32 || Cout != 64 || B < 1 || H < 2 || W < 2 || (H & 1) || (W & 1)) return false;
    return (long long)B * H * W * 3 < (1LL << 31) && (long long)B * (H / 2) * (W / 2) * 64 * 4 < (1LL << 31);
}

hipError_t launch_front(const FrontArgs &q, hipStream_t s)
{
    if (!q.img || !q.w0 || !q.m0 || !q.s0 || !q.b0 || !q.dwpack || !q.wt || !q.mean || !q.sf || !q.beta || !q.out) return hipErrorInvalidValue;
    if (!front_supports(q.B, q.H, q.W, 32, 32, 64)) return hipErrorInvalidValue;
    if (!q.mixed && q.resized && (!front_gen_supports(q.B, q.srcH, q.srcW, q.nh, q.nw) || q.nh > q.H || q.nw > q.W)) return hipErrorInvalidValue;
    const int OH = q.H / 2, OW = q.W / 2;
    if (q.tiles_y != (OH + TY - 1) / TY || q.tiles_x != (OW + TX - 1) / TX) return hipErrorInvalidValue;
    const long long total = (long long)q.B * q.tiles_y * q.tiles_x;
    if (total > 0x7fffffffLL) return hipErrorInvalidValue;
    // two resident blocks per CU, a block walks tiles b, b + grid, ...  (one frame = 756 tiles, measured: grids of 256 / 384 /
    // 512 / 756 blocks -> 34.2 / 32.8 / 28.8 / 32.1 us; 32 frames: 512 / 1024 / 2048 the same step time)
    const int grid = (int)(total < 512 ? total : 512);
    const FrontDims d = {q.B, q.H, q.W, q.act0, q.dact, q.act, q.tiles_y, q.tiles_x};
    // (resized / mixed: loads aligned relative to a base that is itself a multiple of 4, as in launch_front_pool)
    const unsigned adj = (q.mixed || q.resized) ? (unsigned)((uintptr_t)q.img & 3u) : 0u;
    const uint8_t *img = q.img - adj;
    if (q.mixed) {
        FrontMixed fm;
        if (!front_mixed_supports(*q.mixed, q.mixed_first, q.B, q.H, q.W, &fm.bytes4)) return hipErrorInvalidValue;
        fm.mg = *q.mixed;
        for (int b = q.mixed_first; b < q.mixed_first + q.B; ++b) fm.mg.f[b].off += adj;
        fm.bytes4 = (fm.bytes4 + adj + 3u) & ~3u;
        fm.first = q.mixed_first;
        hipLaunchKernelGGL(front_kernel<2>, dim3((unsigned)grid), dim3(256), LDS_BYTES, s, img, q.w0, q.m0, q.s0, q.b0, q.dwpack, q.wt, q.mean,
                           q.sf, q.beta, q.out, d, fm);
    } else if (q.resized) {
        // (the scale factors as launch_first_conv forms them: the same float expressions as elementwise.hip K1 / K1d)
        const FrontGeom g = {q.srcH, q.srcW, q.nh, q.nw, (float)q.srcH / (float)q.nh, (float)q.srcW / (float)q.nw, adj};
        hipLaunchKernelGGL(front_kernel<1>, dim3((unsigned)grid), dim3(256), LDS_BYTES, s, img, q.w0, q.m0, q.s0, q.b0, q.dwpack, q.wt, q.mean,
                           q.sf, q.beta, q.out, d, g);
    } else {
        const FrontGeom g = {q.H, q.W, q.H, q.W, 1.0f, 1.0f, 0u};
        hipLaunchKernelGGL(front_kernel<0>, dim3((unsigned)grid), dim3(256), LDS_BYTES, s, q.img, q.w0, q.m0, q.s0, q.b0, q.dwpack, q.wt, q.mean,
                           q.sf, q.beta, q.out, d, g);
    }
    return hipGetLastError();
}

int front_tile_y() { return TY; }
int front_tile_x() { return TX; }
