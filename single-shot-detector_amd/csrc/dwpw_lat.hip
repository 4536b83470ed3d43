// Depthwise 3x3 + BN + act -> pointwise 1x1 + BN + act as ONE launch in the LATENCY form (batch 1-2: MobileNet Conv2d_5 .. 13,
// mobilenet_v1.py:59-67, depthwise_conv.py:5-26): the four-wave block of igemm_lat.hip (16 positions x 4 x CT x 16 channels on
// v_mfma_f32_16x16x4_f32) whose position operand is not loaded but PRODUCED -- every thread computes the depthwise output of one
// position x 4 channels and writes it into the block's LDS image in MFMA role order.
//
// Why a second fused kernel.  dwpw_stream.hip is the serving-batch design (persistent blocks streaming 8x8 position tiles, one
// resident round of blocks); at batch 1 a 40x56 layer is 35 of its tiles and the pair of launches it replaces costs
// 7.5 + 19 us for 7.5 us of matrix work: two kernel boundaries, two rounds of "every block waits for its first operands", and
// the depthwise result's trip through L2 (profiles/r04_batch1_timeline_before.txt).  Here the depthwise arithmetic rides in front
// of each 64-channel slice of the K loop: per slice 36 fmaf + batch norm per thread against 2 x CT x 8 MFMAs per wave.
//
//   GEMM view   rows m = (image, oy, ox) of the depthwise OUTPUT, cols n = pointwise output channel (physical order),
//               k = depthwise channel (physical order in memory, logical order in the MFMA chain: the LDS image of
//               igemm_lat.hip does the regrouping)
//   iteration   64 channels = two K-steps of 32: thread (r = tid >> 4, c = tid & 15) owns position r, channels 4c .. 4c+3
//               of the slice: nine 16-byte taps (zero padding = buffer range check, per-tap masks), nine 16-byte depthwise
//               weights, mean / sf / beta -- one (ky, kx)-ordered fmaf chain from +0 per channel, batch norm as three
//               separately rounded operations, the activation: the arithmetic of depthwise_kernel (elementwise.hip), bit for bit
//   redundancy  the 4 x CT x 16-channel block recomputes the depthwise values of its 16 positions for every channel tile of
//               the layer (CoutPad / (64 CT) of them); the depthwise part of a layer is 3 % of its arithmetic
//   weights     ConvW::wlat, straight into MFMA registers (igemm_lat.hip), one iteration ahead
//   epilogue    igemm_mfma16.h (transposed product: 16-byte stores from the accumulators)
#include "igemm_mfma16.h"
#include <type_traits>

typedef float v2f __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ int dl_swz(int r) { return (int)((0x32765410u >> (4 * ((r >> 1) & 7))) & 7u) ^ ((r & 1) << 2); }   // = lat_swz (igemm_lat.hip)

template <int CT, int STRIDE>
__global__ __launch_bounds__(256) void dwpw_lat_kernel(const DwPwLArgs q)
{
    const IgemmArgs &a = q.g;
    constexpr int BNW = CT * 16, BN = 4 * BNW;
    constexpr unsigned OOB = 0x80000000u;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * 4096];      // [stage][K-step of the slice][16 rows x 128 B]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    int swz;
    {
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int qq = nblk >> 3, rr = nblk & 7, xcd = bid & 7;
        swz = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
    }
    const int tile_m = (int)udivl((unsigned)swz, a.dN);
    const int tile_n = swz - tile_m * a.n_tiles_n;
    const IgemmLevel L = a.lv[0];
    const int H = q.H, W = q.W, OW = L.OW, M = L.M, P = L.OH * L.OW, K = a.Cin;
    const int m0 = tile_m * 16, n0 = tile_n * BN + wave * BNW;
    const int KC = K >> 5, NI = K >> 6;

    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.in, 0, (int)((long long)a.B * H * W * K * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t drsrc = __builtin_amdgcn_make_buffer_rsrc((void *)q.dw_w, 0, 9 * K * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t mrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)q.dw_mean, 0, K * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t srsrc = __builtin_amdgcn_make_buffer_rsrc((void *)q.dw_sf, 0, K * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t brsrc = __builtin_amdgcn_make_buffer_rsrc((void *)q.dw_beta, 0, K * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.wt_lat, 0, (int)((long long)a.CoutPad * K * 4), 0x00020000);

    // ---- producer item: position r of the tile, channels 4c .. 4c + 3 of the slice
    const int r = tid >> 4, c = tid & 15;
    unsigned xoff[9];
    {
        const int m = m0 + r;
        const bool rowok = m < M;
        const int mm = rowok ? m : 0;
        const int b = (int)udivl((unsigned)mm, L.dP), pp = mm - b * P;
        const int oy = (int)udivl((unsigned)pp, L.dOW), ox = pp - oy * OW;
        const int iy0 = oy * STRIDE - q.dpad, ix0 = ox * STRIDE - q.dpad;
        const int xbase = ((b * H * W + iy0 * W + ix0) * K + c * 4) * 4;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const bool ok = rowok && (unsigned)(iy0 + ky) < (unsigned)H && (unsigned)(ix0 + kx) < (unsigned)W;
                xoff[ky * 3 + kx] = ok ? (unsigned)(xbase + (ky * W + kx) * K * 4) : OOB;
            }
    }
    int wlo, whi;                   // LDS image (igemm_lat.hip): K-step c >> 3, 16-byte chunk c & 7 = (octet o, half hh) of the row
    {
        const int hk = c >> 3, cc = c & 7, o = cc >> 1, hh = cc & 1, f = dl_swz(r);
        wlo = hk * 2048 + r * 128 + ((((hh) * 2 + (o >> 1)) ^ f) << 4) + (o & 1) * 8;
        whi = hk * 2048 + r * 128 + ((((hh + 2) * 2 + (o >> 1)) ^ f) << 4) + (o & 1) * 8;
    }
    int roff[2];
    {
        const int i = lane & 15, kk = lane >> 4, f = dl_swz(i);
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) roff[hf] = i * 128 + (((2 * kk + hf) ^ f) << 4);
    }
    int wbase[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) wbase[ct] = ((n0 >> 4) + ct) * KC * 2048 + lane * 16;

    v4f xr[9], wd[9], bm, bs, bb;                       // the producer's operands of ONE slice
    auto issue_prod = [&](int it) __attribute__((always_inline)) {
        const int so = it * 256;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            xr[t] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (int)xoff[t], so, 0));
            wd[t] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(drsrc, c * 16, t * K * 4 + so, 0));
        }
        bm = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(mrsrc, c * 16, so, 0));
        bs = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(srsrc, c * 16, so, 0));
        bb = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(brsrc, c * 16, so, 0));
    };
    const int dact = q.dact;
    auto produce = [&](int stage) __attribute__((always_inline)) {
        v4f acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = fmaf(xr[t][i], wd[t][i], acc[i]);
        v4f v;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float t = (acc[i] - bm[i]) * bs[i];
            float y = t + bb[i];
            if (dact >= 1) y = y > 0.0f ? y : 0.0f;
            if (dact == 2) y = y < 6.0f ? y : 6.0f;
            v[i] = y;
        }
        unsigned char *base = lds + stage * 4096;
        *(v2f *)(base + wlo) = v2f{v[0], v[2]};
        *(v2f *)(base + whi) = v2f{v[1], v[3]};
    };
    v4f wr[2][2][CT][2];            // [set][K-step of the slice][channel tile][half]
    auto issue_w = [&](auto set_tag, int it) __attribute__((always_inline)) {
        constexpr int S = decltype(set_tag)::value;
#pragma unroll
        for (int hk = 0; hk < 2; ++hk)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int hf = 0; hf < 2; ++hf)
                    wr[S][hk][ct][hf] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wbase[ct] + hf * 1024, (2 * it + hk) * 2048, 0));
    };
    v4f xf[2][2];                   // [K-step of the slice][half]
    auto lread = [&](int stage) __attribute__((always_inline)) {
        const unsigned char *base = lds + stage * 4096;
#pragma unroll
        for (int hk = 0; hk < 2; ++hk)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) xf[hk][hf] = *(const v4f *)(base + hk * 2048 + roff[hf]);
    };
    v4f acc[CT][1];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) acc[ct][0] = v4f{0.f, 0.f, 0.f, 0.f};
    auto mfmas = [&](auto set_tag) __attribute__((always_inline)) {
        constexpr int S = decltype(set_tag)::value;
#pragma unroll
        for (int hk = 0; hk < 2; ++hk)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct)
                        acc[ct][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[S][hk][ct][hf][e], xf[hk][hf][e], acc[ct][0], 0, 0, 0);
    };
    // ---- pipeline.  Iteration `it` multiplies slice it (fragments in registers, weights in set it & 1); in front of its
    // MFMAs the block produces slice it + 1 into the other LDS stage (operands issued an iteration ago) and issues the
    // producer operands of slice it + 2 and the weights of slice it + 1; behind them one barrier, then the fragments of
    // slice it + 1.  Stage (it + 1) & 1 was last read behind the barrier of iteration it - 2 ... it - 1: every wave has
    // passed that barrier, with its reads complete, before any wave writes here.
    issue_prod(0);
    issue_w(std::integral_constant<int, 0>{}, 0);
    produce(0);
    issue_prod(NI > 1 ? 1 : 0);
    __syncthreads();
    lread(0);
    auto iter = [&](auto set_tag, int it) __attribute__((always_inline)) {
        constexpr int S = decltype(set_tag)::value;
        if (it + 1 < NI) {                                   // block-uniform
            produce((it + 1) & 1);
            issue_prod(it + 2 < NI ? it + 2 : it + 1);
            issue_w(std::integral_constant<int, S ^ 1>{}, it + 1);
        }
        mfmas(set_tag);
        __syncthreads();
        if (it + 1 < NI) lread((it + 1) & 1);
    };
    int it = 0;
    for (; it + 2 <= NI; it += 2) {
        iter(std::integral_constant<int, 0>{}, it);
        iter(std::integral_constant<int, 1>{}, it + 1);
    }
    if (it < NI) iter(std::integral_constant<int, 0>{}, it);

    epilogue_16x16<1, CT>(a, L, acc, m0, n0, lane);
}

bool dwpw_lat_supports(const DwPwLArgs &q, int ct)
{
    const IgemmArgs &a = q.g;
    if (ct != 1 && ct != 2 && ct != 4) return false;
    if (a.Cin < 64 || a.Cin % 64 != 0 || a.CoutPad % (64 * ct) != 0 || a.Cout % 4 != 0 || !a.wt_lat || !a.mean || a.bias || a.res || a.out2) return false;
    if (a.in_fmt || a.out_fmt || a.res_fmt || a.nlevels != 1 || a.taps != 1) return false;
    if ((q.dstride != 1 && q.dstride != 2) || q.dpad < 0 || q.dpad > 1) return false;
    const IgemmLevel &L = a.lv[0];
    if ((L.out_rstride | (int)L.out_bstride | (int)L.out_off) & 3) return false;
    if ((L.OH - 1) * q.dstride + 2 - q.dpad > q.H + 1 || (L.OW - 1) * q.dstride + 2 - q.dpad > q.W + 1) return false;
    if ((long long)a.B * q.H * q.W * a.Cin * 4 >= (1LL << 31) || (long long)a.CoutPad * a.Cin * 4 >= (1LL << 31)) return false;
    if ((long long)a.B * L.out_bstride * 4 >= (1LL << 31) || L.out_bstride < 0) return false;
    return true;
}

template <int CT>
static hipError_t launch_c(const DwPwLArgs &q, long long nblk, hipStream_t s)
{
    if (q.dstride == 1) hipLaunchKernelGGL((dwpw_lat_kernel<CT, 1>), dim3((unsigned)nblk), dim3(256), 0, s, q);
    else hipLaunchKernelGGL((dwpw_lat_kernel<CT, 2>), dim3((unsigned)nblk), dim3(256), 0, s, q);
    return hipGetLastError();
}

hipError_t launch_dwpw_lat(int ct, const DwPwLArgs &q, hipStream_t s)
{
    if (!dwpw_lat_supports(q, ct)) return hipErrorInvalidValue;
    const IgemmArgs &a = q.g;
    if (a.n_tiles_n * 64 * ct != a.CoutPad) return hipErrorInvalidValue;
    const long long nblk = (long long)((a.lv[0].M + 15) / 16) * a.n_tiles_n;
    if (nblk <= 0 || nblk > 0x7fffffffLL) return hipErrorInvalidValue;
    switch (ct) {
    case 1: return launch_c<1>(q, nblk, s);
    case 2: return launch_c<2>(q, nblk, s);
    case 4: return launch_c<4>(q, nblk, s);
    }
    return hipErrorInvalidValue;
}
