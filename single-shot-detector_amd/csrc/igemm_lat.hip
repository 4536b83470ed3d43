// Latency form of the implicit-GEMM convolution for SMALL launches (batch 1-2: fpn p4..p7 and laterals, the late
// MobileNet pointwise layers, coarse pyramid levels): v_mfma_f32_16x16x4_f32, one wave per block, operands straight
// from global memory into MFMA registers -- no LDS, no barrier.
//
// Why a second kernel.  The time of a small launch on the 32x32x2 kernel (igemm.hip) is the length of ONE accumulator's
// K chain: an accumulator of v_mfma_f32_32x32x2_f32 advances k by 2 per 64 pipe cycles, so fpn p6 at batch 1
// (k = 9 x 1024) is 4 608 dependent MFMAs = 123 us at 2.4 GHz on 12 tiles, whatever else is done well (measured 199 us,
// profiles/r03_batch1_timeline_before.txt), and every K-step pays a barrier + LDS round trip on top.
// v_mfma_f32_16x16x4_f32 advances k by 4 per 32 cycles -- a quarter of the chain -- on 16x16 tiles, i.e. four times as
// many independent chains for the 256 CUs, and it is bit for bit the same k-ascending fmaf chain
// (scripts/experiments/mfma_16x16x4_probe.hip, profiles/r03_mfma_16x16x4_probe.log: 256 of 256 outputs equal to the
// chain at K = 4 .. 2304; one dependent accumulator issues every 32.0 cycles).  Results are therefore bit-identical
// to igemm.hip and to the oracle; tests compare all three.
//
//   GEMM view   as igemm.hip: rows m = (image, oy, ox), cols n = output channel (physical order), k = (ky, kx, ci)
//   wave tile   PT x 16 positions  x  CT x 16 channels, PT*CT accumulators of 4 registers
//   product     TRANSPOSED: the weights are the MFMA's A operand (rows of D = channels), the activations its B operand
//               (columns of D = positions); fma(w, x, acc) == fma(x, w, acc).  A lane then holds 4 CONSECUTIVE channels
//               of one position: batch norm with vector parameter loads and 16-byte stores straight from the
//               accumulators, no transpose.
//   operands    lane l = (i = l & 15, kk = l >> 4) supplies k = 4t + kk of MFMA t.  With the physical channel order of
//               ssd_internal.h (octet = logical [0,2,4,6,1,3,5,7]) the lane's two values of an octet, logical kk and
//               4 + kk, sit 8 bytes apart: ONE 16-byte load at  row + octet*32 + (kk & 1)*16 + (kk >> 1)*4  brings both
//               (elements 0 and 2; lanes kk >= 2 load at a 4-byte-shifted address, which a dword-aligned
//               buffer_load_dwordx4 allows, so every lane finds its operands in the same registers: no select).
//               The shifted load of a tensor's last chunk touches 4 bytes behind it: every allocation of the library
//               carries that slack (DevPool) and the descriptors here are 4 bytes longer; the value is never used.
//   pipeline    the loads of K-step s + D are issued behind the MFMAs of K-step s (D = 2 .. 4 register sets); waits are
//               the compiler's counted vmcnt.  Zero padding and rows past M are the buffer range check, tap changes a
//               per-row mask (as igemm.hip).
#include "igemm_mfma16.h"
#include <type_traits>

template <int PT, int CT, int TAPS, int D>
__global__ __launch_bounds__(64) void igemm_lat_kernel(const IgemmArgs a)
{
    constexpr int BM = PT * 16, BN = CT * 16;
    const int lane = threadIdx.x, i = lane & 15, kk = lane >> 4;
    const int sub = (kk & 1) * 16 + (kk >> 1) * 4;            // byte offset of the lane's operands inside an octet

    // blocks b, b+8, ... share an XCD: consecutive tiles (the channel tiles of one position tile first) per XCD
    int swz;
    {
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
        swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tile_m = (int)udivl((unsigned)swz, a.dN);
    const int tile_n = swz - tile_m * a.n_tiles_n;
    int lvl = 0;
#pragma unroll
    for (int l = 1; l < SSD_MAX_LEVELS; ++l)
        if (l < a.nlevels && tile_m >= a.lv[l].tile_begin) lvl = l;
    const IgemmLevel L = a.lv[lvl];
    const int H = L.H, W = L.W, OW = L.OW, M = L.M, P = L.OH * L.OW, Cin = a.Cin;
    const int m0 = (tile_m - L.tile_begin) * BM, n0 = tile_n * BN;

    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(a.in + L.in_off), 0, (int)((long long)a.B * H * W * Cin * 4 + 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(a.wt + L.wt_off), 0, (int)((long long)TAPS * a.CoutPad * Cin * 4 + 4), 0x00020000);
    constexpr unsigned OOB = 0x80000000u;

    // ---- positions: byte offset of tap (0,0) of row m0 + p*16 + i, and "tap t reads inside the image" bits
    int xbase[PT];
    unsigned xmask[PT];
    const bool dense1x1 = TAPS == 1 && a.stride == 1 && a.pad == 0 && L.OH == H && OW == W;
#pragma unroll
    for (int p = 0; p < PT; ++p) {
        const int m = m0 + p * 16 + i;
        const bool rowok = m < M;
        const int mm = rowok ? m : 0;
        if (dense1x1) {
            xbase[p] = mm * Cin * 4 + sub;
            xmask[p] = rowok ? 1u : 0u;
        } else {
            const int b = (int)udivl((unsigned)mm, L.dP), pp = mm - b * P;
            const int oy = (int)udivl((unsigned)pp, L.dOW), ox = pp - oy * OW;
            const int iy0 = oy * a.stride - a.pad, ix0 = ox * a.stride - a.pad;
            xbase[p] = (b * H * W + iy0 * W + ix0) * Cin * 4 + sub;
            unsigned vx = 0, mk = 0;
#pragma unroll
            for (int k = 0; k < (TAPS == 9 ? 3 : 1); ++k)
                if ((unsigned)(ix0 + k) < (unsigned)W) vx |= 1u << k;
#pragma unroll
            for (int k = 0; k < (TAPS == 9 ? 3 : 1); ++k)
                if ((unsigned)(iy0 + k) < (unsigned)H) mk |= vx << (3 * k);
            xmask[p] = rowok ? mk : 0u;
        }
    }
    auto tap_offsets = [&](int t, unsigned (&off)[PT]) {
        const int tky = TAPS == 9 ? t / 3 : 0, tkx = TAPS == 9 ? t - 3 * tky : 0;
        const int d = (tky * W + tkx) * Cin * 4;
#pragma unroll
        for (int p = 0; p < PT; ++p) off[p] = ((xmask[p] >> t) & 1u) ? (unsigned)(xbase[p] + d) : OOB;
    };
    // ---- weights: row n0 + c*16 + i of wt [tap][CoutPad][Cin]
    int wbase[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) wbase[c] = (n0 + c * 16 + i) * Cin * 4 + sub;
    const int w_tapstride = a.CoutPad * Cin * 4;

    const int KC = Cin >> 5, KS = TAPS * KC;
    unsigned offc[PT], offn[PT];
    int ltap = 0, lkc = 0, kload = 0;
    tap_offsets(0, offc);
    tap_offsets(1, offn);
    v4f xr[D][PT][4], wr[D][CT][4];
    auto load = [&](auto slot_tag) __attribute__((always_inline)) {
        constexpr int S = decltype(slot_tag)::value;
        const int so = lkc * 128, wso = ltap * w_tapstride + so;
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int o = 0; o < 4; ++o)
                wr[S][c][o] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wbase[c] + o * 32, wso, 0));
#pragma unroll
        for (int p = 0; p < PT; ++p)
#pragma unroll
            for (int o = 0; o < 4; ++o)
                xr[S][p][o] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (int)offc[p] + o * 32, so, 0));
        // advance the load stream; past the last step the counters stay put (the surplus prefetch re-reads valid memory)
        if (++kload < KS) {
            if (++lkc == KC) {
                lkc = 0;
                ++ltap;
#pragma unroll
                for (int p = 0; p < PT; ++p) offc[p] = offn[p];
                tap_offsets(ltap + 1, offn);
            }
        }
    };
    v4f acc[CT][PT];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int p = 0; p < PT; ++p) acc[c][p] = v4f{0.f, 0.f, 0.f, 0.f};
    // one K-step = 32 channels = 8 MFMAs per accumulator: octet o, then its two halves (elements 0 and 2 of the lane's load)
    auto compute = [&](auto slot_tag) __attribute__((always_inline)) {
        constexpr int S = decltype(slot_tag)::value;
#pragma unroll
        for (int o = 0; o < 4; ++o)
#pragma unroll
            for (int e = 0; e < 4; e += 2)
#pragma unroll
                for (int c = 0; c < CT; ++c)
#pragma unroll
                    for (int p = 0; p < PT; ++p)
                        acc[c][p] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[S][c][o][e], xr[S][p][o][e], acc[c][p], 0, 0, 0);
    };
    auto for_slots = [&](auto fn) __attribute__((always_inline)) {
        fn(std::integral_constant<int, 0>{});
        if constexpr (D > 1) fn(std::integral_constant<int, 1>{});
        if constexpr (D > 2) fn(std::integral_constant<int, 2>{});
        if constexpr (D > 3) fn(std::integral_constant<int, 3>{});
    };
    for_slots([&](auto s) __attribute__((always_inline)) { load(s); });
    int ks = 0;
    for (; ks + D <= KS; ks += D)
        for_slots([&](auto s) __attribute__((always_inline)) { compute(s); load(s); });
    {
        const int rem = KS - ks;                              // 0 .. D-1 steps left, already loaded
        for_slots([&](auto s) __attribute__((always_inline)) { if (decltype(s)::value < rem) compute(s); });
    }

    epilogue_16x16<PT, CT>(a, L, acc, m0, n0, lane);
}

int igemm_lat_bm(int tile) { return (tile == IGEMM_LAT_2x1 || tile == IGEMM_LAT_2x2) ? 32 : 16; }
int igemm_lat_bn(int tile) { return (tile == IGEMM_LAT_1x2 || tile == IGEMM_LAT_2x2) ? 32 : 16; }

template <int PT, int CT, int D>
static hipError_t launch_l(const IgemmArgs &a, int total_tiles_m, hipStream_t s)
{
    const long long nblk = (long long)total_tiles_m * a.n_tiles_n;
    if (nblk <= 0 || nblk > 0x7fffffffLL) return hipErrorInvalidValue;
    if (a.taps == 9) hipLaunchKernelGGL((igemm_lat_kernel<PT, CT, 9, D>), dim3((unsigned)nblk), dim3(64), 0, s, a);
    else hipLaunchKernelGGL((igemm_lat_kernel<PT, CT, 1, D>), dim3((unsigned)nblk), dim3(64), 0, s, a);
    return hipGetLastError();
}

// the forms this kernel implements (make_conv_op asks before choosing it)
bool igemm_lat_supports(const IgemmArgs &a)
{
    if (a.in_fmt || a.out_fmt || a.res_fmt || a.ts) return false;               // exact fp32 rows only
    if (a.Cout % 4 != 0 || a.Cin % 32 != 0 || a.CoutPad % 32 != 0) return false;
    for (int l = 0; l < a.nlevels; ++l)
        if ((a.lv[l].out_rstride | (int)a.lv[l].out_bstride | (int)a.lv[l].out_off) & 3) return false;    // 16-byte stores
    if (a.res && (a.Cout & 3)) return false;
    return true;
}

hipError_t launch_igemm_lat(int tile, const IgemmArgs &a, int total_tiles_m, hipStream_t s)
{
    if (!igemm_lat_supports(a) || (a.taps != 1 && a.taps != 9)) return hipErrorInvalidValue;
    if ((long long)a.taps * a.CoutPad * a.Cin * 4 + 4 >= (1LL << 31)) return hipErrorInvalidValue;
    for (int l = 0; l < a.nlevels; ++l) {
        if ((long long)a.B * a.lv[l].H * a.lv[l].W * a.Cin * 4 + 4 >= (1LL << 31)) return hipErrorInvalidValue;
        if ((long long)a.B * a.lv[l].out_bstride * 4 >= (1LL << 31) || a.lv[l].out_bstride < 0) return hipErrorInvalidValue;
    }
    if ((a.mean != nullptr) != (a.sf != nullptr) || (a.mean != nullptr) != (a.beta != nullptr)) return hipErrorInvalidValue;
    if (a.mean && (a.bias || a.res)) return hipErrorInvalidValue;
    if (a.bias && a.res) return hipErrorInvalidValue;
    if (a.out2 && !a.mean) return hipErrorInvalidValue;
    if (a.n_tiles_n * igemm_lat_bn(tile) != a.CoutPad || a.nlevels < 1 || a.nlevels > SSD_MAX_LEVELS) return hipErrorInvalidValue;
    switch (tile) {
    case IGEMM_LAT_1x1: return launch_l<1, 1, 4>(a, total_tiles_m, s);
    case IGEMM_LAT_1x2: return launch_l<1, 2, 3>(a, total_tiles_m, s);
    case IGEMM_LAT_2x1: return launch_l<2, 1, 3>(a, total_tiles_m, s);
    case IGEMM_LAT_2x2: return launch_l<2, 2, 2>(a, total_tiles_m, s);
    }
    return hipErrorInvalidValue;
}
