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
#define DWL_SLICE 1664           // bytes between the 32-channel slices of the depthwise table in LDS (1 536 of data + 128)
static __device__ __forceinline__ int dl_swz(int r) { return (int)((0x32765410u >> (4 * ((r >> 1) & 7))) & 7u) ^ ((r & 1) << 2); }   // = lat_swz (igemm_lat.hip)

template <int CT, int STRIDE>
__global__ __launch_bounds__(256, (CT <= 2 ? 3 : 1)) void dwpw_lat_kernel(const DwPwLArgs q)
{
    const IgemmArgs &a = q.g;
    constexpr int BNW = CT * 16, BN = 4 * BNW;
    constexpr unsigned OOB = 0x80000000u;
    // LDS: [stage][K-step of the slice][16 rows x 128 B] A images, then the layer's depthwise table -- per 32-channel slice
    // 9 taps + mean + sf + beta x 32 channels (DwW::pack), slices DWL_SLICE bytes apart: the two slices an iteration reads
    // must not share banks (1 536 B = 6 x 256 B would put them on the same 32)
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char *const tab = lds + 2 * 4096;
    typedef __attribute__((address_space(3))) void *lds_as_t;
    const int lds0 = (int)(size_t)(lds_as_t)lds;
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
    {   // the depthwise table of the whole layer into LDS, once per block (K / 32 slices of 96 sixteen-byte pieces)
        const int npiece = KC * 96;
        for (int i = tid; i < npiece; i += 256) {
            const int sl = i / 96, o = i - sl * 96;
            *(v4f *)(tab + sl * DWL_SLICE + o * 16) = *(const v4f *)(q.dw_pack + (long long)sl * 384 + o * 4);
        }
    }
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

    v4f xr[9];                                          // the nine taps of ONE slice (the weights come from the LDS table)
    auto issue_prod = [&](int it) __attribute__((always_inline)) {
        const int so = it * 256;
#pragma unroll
        for (int t = 0; t < 9; ++t) xr[t] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (int)xoff[t], so, 0));
    };
    const int dact = q.dact;
    const int toff = (c >> 3) * DWL_SLICE + (c & 7) * 16;        // this thread's column of the table within an iteration's two slices
    auto produce = [&](int stage, int it) __attribute__((always_inline)) {
        const unsigned char *tb = tab + it * (2 * DWL_SLICE) + toff;
        v4f acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const v4f wd = *(const v4f *)(tb + t * 128);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = fmaf(xr[t][i], wd[i], acc[i]);
        }
        const v4f bm = *(const v4f *)(tb + 9 * 128), bs = *(const v4f *)(tb + 10 * 128), bb = *(const v4f *)(tb + 11 * 128);
        v4f v;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float t = (acc[i] - bm[i]) * bs[i];
            float y = t + bb[i];
            if (dact >= 1) y = y > 0.0f ? y : 0.0f;
            if (dact == 2) y = y < 6.0f ? y : 6.0f;
            v[i] = y;
        }
        // (ds_write2_b32 takes the two dwords from the registers where they are: igemm_lat.hip lstore; the explicit wait because the
        //  barrier behind the iteration must find these writes complete and the compiler does not count them)
        asm volatile("ds_write2_b32 %0, %1, %2 offset1:1" ::"v"(lds0 + stage * 4096 + wlo), "v"(v[0]), "v"(v[2]) : "memory");
        asm volatile("ds_write2_b32 %0, %1, %2 offset1:1\n\ts_waitcnt lgkmcnt(0)" ::"v"(lds0 + stage * 4096 + whi), "v"(v[1]), "v"(v[3]) : "memory");
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
    __syncthreads();                                         // the table is in LDS
    produce(0, 0);
    issue_prod(NI > 1 ? 1 : 0);
    __syncthreads();
    lread(0);
    // One iteration that has a successor, as ONE straight-line region with the instruction mix pinned (sched_group_barrier):
    // behind every MFMA of slice `it` its share of the NEXT slice's production (table reads first, the fmaf chain, the two
    // LDS writes last) and of the loads two slices / one slice ahead.  Issued as blocks -- production, then 17 loads, then 32
    // MFMAs -- the four waves of a block queued their loads together, sat in the texture-address path's queue and reached
    // their MFMAs late, and every block of a CU did so at the same time: the phases added (35 us per 512 -> 512 layer with
    // per-thread depthwise weights from global memory, 26.5 with the LDS table, profiles/r04_batch1_option_ab.log).
    auto iter_mid = [&](auto set_tag, int it) __attribute__((always_inline)) {
        constexpr int S = decltype(set_tag)::value;
        constexpr int NM = 16 * CT, NV = 9 + 4 * CT, VAL = (72 + NM - 1) / NM;
        __builtin_amdgcn_sched_barrier(0);
        produce((it + 1) & 1, it + 1);
        issue_w(std::integral_constant<int, S ^ 1>{}, it + 1);
        issue_prod(it + 2 < NI ? it + 2 : it + 1);
        mfmas(set_tag);
        __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);             // the slice's twelve table reads up front: they land under
#pragma unroll                                                           // the first MFMAs (read one by one in front of its fmaf
        for (int i = 0; i < NM; ++i) {                                   // group, each cost the wave an LDS round trip between two MFMAs)
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                  // one MFMA
            if (i > 0) __builtin_amdgcn_sched_group_barrier(0x002, VAL, 0);                     // producer arithmetic
            if ((i * NV) / NM != ((i + 1) * NV) / NM) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // a load
            if (i >= NM - 2) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                 // the A image's two writes
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        lread((it + 1) & 1);
    };
    int it = 0;
    for (; it + 2 < NI; it += 2) {
        iter_mid(std::integral_constant<int, 0>{}, it);
        iter_mid(std::integral_constant<int, 1>{}, it + 1);
    }
    if (it + 1 < NI) {                                       // NI - it == 2
        iter_mid(std::integral_constant<int, 0>{}, it);
        mfmas(std::integral_constant<int, 1>{});
    } else {                                                 // NI - it == 1
        mfmas(std::integral_constant<int, 0>{});
    }

    epilogue_16x16<1, CT>(a, L, acc, m0, n0, lane);
}

bool dwpw_lat_supports(const DwPwLArgs &q, int ct)
{
    const IgemmArgs &a = q.g;
    if (ct != 1 && ct != 2 && ct != 4) return false;
    if (2 * 4096 + (a.Cin >> 5) * DWL_SLICE > 64 * 1024 || !q.dw_pack) return false;      // the layer's depthwise table lives in LDS
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
    const int lds = 2 * 4096 + (q.g.Cin >> 5) * DWL_SLICE;
    if (q.dstride == 1) hipLaunchKernelGGL((dwpw_lat_kernel<CT, 1>), dim3((unsigned)nblk), dim3(256), lds, s, q);
    else hipLaunchKernelGGL((dwpw_lat_kernel<CT, 2>), dim3((unsigned)nblk), dim3(256), lds, s, q);
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
