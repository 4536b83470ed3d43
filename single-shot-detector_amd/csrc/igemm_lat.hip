// Latency form of the implicit-GEMM convolution for TINY launches (batch 1-2: fpn p6 and p7 -- 140 and 35 positions per
// image; coarse pyramid levels): v_mfma_f32_16x16x4_f32, ONE wave per block, no barrier.
//
// Why a second kernel.  The time of a tiny launch on the 32x32x2 kernel (igemm.hip) is the length of ONE accumulator's
// K chain: an accumulator of v_mfma_f32_32x32x2_f32 advances k by 2 per 64 pipe cycles, so fpn p6 at batch 1
// (k = 9 x 1024) is 4 608 dependent MFMAs = 123 us at 2.4 GHz on 12 tiles, whatever else is done well (measured 199 us in
// the network, profiles/r03_batch1_timeline_before.txt), and every K-step pays a barrier + LDS round trip on top.
// v_mfma_f32_16x16x4_f32 advances k by 4 per 32 cycles -- a quarter of the chain -- on 16x16 tiles, i.e. sixteen times as
// many independent chains as 64x64 tiles give the 256 CUs, and it is bit for bit the same k-ascending fmaf chain
// (scripts/experiments/mfma_16x16x4_probe.hip, profiles/r03_mfma_16x16x4_probe.log: 256 of 256 outputs equal to the
// chain at K = 4 .. 2304; one dependent accumulator issues every 32.0 cycles).  Results are therefore bit-identical
// to igemm.hip and to the oracle; tests compare all three.
//
//   GEMM view   as igemm.hip: rows m = (image, oy, ox), cols n = output channel (physical order), k = (ky, kx, ci)
//   wave tile   PT x 16 positions  x  CT x 16 channels, PT*CT accumulators of 4 registers
//   product     transposed (weights = the MFMA's A operand), epilogue straight from the accumulators: igemm_mfma16.h
//   weights     a second packing made at ssd_finalize (ConvW::wlat): per (tap, 16-channel tile, K-step of 32 channels) two
//               1-KB pieces in LANE order -- lane (i, kk) finds its operands t = 0..3 / 4..7 (k = 4 t + kk) of channel i at
//               lane * 16: a fragment is TWO perfectly coalesced loads straight into MFMA registers
//   positions   coalesced too: 8 rows x 128 B per load instruction (zero padding = buffer range check, per-row tap masks, as
//               igemm.hip), regrouped by MFMA role through a wave-private LDS image: a row's 32 channels stored as
//               [kk = channel & 3][t = channel >> 2].  In the physical channel order a lane's 16-byte chunk (octet o, half h)
//               holds logical 8o + {0,2,4,6} + h: elements (0,2) go to kk = h at t = 2o, 2o+1, elements (1,3) to kk = h + 2
//               -- two 8-byte writes per load; lane (i, kk) reads its eight operands with two ds_read_b128.  The 16-byte
//               slot (2 kk + half) of row r sits at slot ^ lat_swz(r), lat_swz(r) = P[(r >> 1) & 7] ^ 4 (r & 1) with
//               P = 0 1 4 5 6 7 2 3, found by exhaustive search over the lane groups and bank rules of MI355X_MICROARCH.md's
//               LDS table: ds_read_b128 is served in four NON-contiguous groups of 16 lanes over 64 banks, ds_write_b64 in
//               four contiguous groups of 16 lanes over 32 banks -- rows r and r + 1, which one write group covers, must
//               differ in the slot half they write.  SQ_LDS_BANK_CONFLICT 0 (the first swizzle, a function of r >> 1
//               alone, was conflict-free for the reads and 2-way on every write group: 25-40 % of the kernel's LDS
//               cycles, profiles/r03_batch1_pmc_summary.txt).  No barrier: LDS operations of one wave
//               execute in order
//   pipeline    D register sets: the loads of K-step s + D are issued behind the MFMAs of K-step s; the positions of step
//               s + 1 go to the other LDS stage before the MFMAs of step s and are read back behind them; waits are the
//               compiler's counted vmcnt / lgkmcnt.
// (The first form of this kernel, round 3, loaded every fragment in MFMA lane order straight from the activation / weight
//  rows -- 16 rows x 4 lanes per instruction, half of each 16-byte load unused: such a load costs ~64 cycles of the CU's
//  address path, a K-step 8 of them against 256 cycles of MFMA; p6 65 us alone, and the laterals running beside it lost
//  25 us to its loads.  This form issues 4 coalesced loads per K-step.)
#include "igemm_mfma16.h"
#include <type_traits>

typedef float v2f __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ int lat_swz(int r) { return (int)((0x32765410u >> (4 * ((r >> 1) & 7))) & 7u) ^ ((r & 1) << 2); }

template <int PT, int CT, int TAPS, int D, int WB, int IL>
__global__ __launch_bounds__(64 * WB) void igemm_lat_kernel(const IgemmArgs a)
{
    static_assert(D == 2 || D == 4 || D == 8 || D == 16, "the loop is unrolled over D register sets and two LDS stages");
    static_assert(WB == 1 || WB == 2 || WB == 4, "waves per block");
    // (Beside a launch that saturates the matrix pipe -- fpn p6 beside the grouped p3+p4+p5 launch -- this kernel's dependent
    //  32-cycle MFMAs queue behind the other waves' 64-cycle ones: p6 59 us alone, 172 us there, with or without s_setprio 3,
    //  4 or 8 K-steps of loads in flight, a high-priority stream: DESIGN section 8.)
    // WB > 1 (the form for launches of a few blocks per CU): WB waves share the block's PT x 16 positions -- each loads its
    // share of the rows, the LDS image is the block's, one barrier per K-step between its write and its reads -- and each
    // wave owns CT x 16 of the block's WB x CT x 16 channels (weights stay per wave, straight into registers).  Per MFMA a
    // quarter to a half of the one-wave form's operand traffic, the same 16x16 granularity.
    constexpr int BM = PT * 16, BNW = CT * 16, BN = WB * BNW;
    constexpr int NXB = 2 * PT;                   // position loads per K-step and block: 8 rows x 128 B each
    constexpr int NX = (NXB + WB - 1) / WB;       // ... per wave: load j of wave w is the block's load w + WB * j
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * PT * 2048];
    const int lane = threadIdx.x & 63;
    const int wave = WB == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);

    // blocks b, b+8, ... share an XCD: consecutive tiles (the channel tiles of one position tile first) per XCD
    int swz;
    {
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
        swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    int tile_m, tile_n;
    if (a.n_major) { tile_n = (int)udivl((unsigned)swz, a.dM); tile_m = swz - tile_n * a.tiles_m; }
    else { tile_m = (int)udivl((unsigned)swz, a.dN); tile_n = swz - tile_m * a.n_tiles_n; }
    int lvl = 0;
#pragma unroll
    for (int l = 1; l < SSD_MAX_LEVELS; ++l)
        if (l < a.nlevels && tile_m >= a.lv[l].tile_begin) lvl = l;
    const IgemmLevel L = a.lv[lvl];
    const int H = L.H, W = L.W, OW = L.OW, M = L.M, P = L.OH * L.OW, Cin = a.Cin;
    const int m0 = (tile_m - L.tile_begin) * BM, n0 = tile_n * BN + wave * BNW;
    const int KC = Cin >> 5, KS = TAPS * KC;

    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(a.in + L.in_off), 0, (int)((long long)a.B * H * W * Cin * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(a.wt_lat + L.wt_off), 0, (int)((long long)TAPS * a.CoutPad * Cin * 4), 0x00020000);
    constexpr unsigned OOB = 0x80000000u;

    // ---- positions: the block's load ub covers rows 8 ub + (lane >> 3), chunk lane & 7
    int xbase[NX];
    unsigned xmask[NX];
    int woff_lo[NX], woff_hi[NX];
    bool xact[NX];                                // (wave-uniform) this wave has a j-th load
    const bool dense1x1 = TAPS == 1 && L.stride == 1 && L.pad == 0 && L.OH == H && OW == W;
#pragma unroll
    for (int u = 0; u < NX; ++u) {
        const int ub = wave + WB * u;
        xact[u] = WB == 1 || ub < NXB;
        const int r = (8 * ub + (lane >> 3)) & (BM - 1), c = lane & 7;
        const int m = m0 + r;
        const bool rowok = m < M;
        const int mm = rowok ? m : 0;
        if (dense1x1) {
            xbase[u] = (mm * Cin + c * 4) * 4;
            xmask[u] = rowok ? 1u : 0u;
        } else {
            const int b = (int)udivl((unsigned)mm, L.dP), pp = mm - b * P;
            const int oy = (int)udivl((unsigned)pp, L.dOW), ox = pp - oy * OW;
            const int iy0 = oy * L.stride - L.pad, ix0 = ox * L.stride - L.pad;
            xbase[u] = ((b * H * W + iy0 * W + ix0) * Cin + c * 4) * 4;
            unsigned vx = 0, mk = 0;
#pragma unroll
            for (int k = 0; k < (TAPS == 9 ? 3 : 1); ++k)
                if ((unsigned)(ix0 + k) < (unsigned)W) vx |= 1u << k;
#pragma unroll
            for (int k = 0; k < (TAPS == 9 ? 3 : 1); ++k)
                if ((unsigned)(iy0 + k) < (unsigned)H) mk |= vx << (3 * k);
            xmask[u] = rowok ? mk : 0u;
        }
        // LDS image: octet o, half h of the row -> elements (0,2) at role kk = h, (1,3) at kk = h + 2
        const int o = c >> 1, hh = c & 1;
        const int f = lat_swz(r);
        woff_lo[u] = r * 128 + ((((hh) * 2 + (o >> 1)) ^ f) << 4) + (o & 1) * 8;
        woff_hi[u] = r * 128 + ((((hh + 2) * 2 + (o >> 1)) ^ f) << 4) + (o & 1) * 8;
    }
    auto tap_offsets = [&](int t, unsigned (&off)[NX]) {
        const int tky = TAPS == 9 ? t / 3 : 0, tkx = TAPS == 9 ? t - 3 * tky : 0;
        const int d = (tky * W + tkx) * Cin * 4;
#pragma unroll
        for (int u = 0; u < NX; ++u) off[u] = ((xmask[u] >> t) & 1u) ? (unsigned)(xbase[u] + d) : OOB;
    };
    int roff[2];
    {
        const int i = lane & 15, kk = lane >> 4;
        const int f = lat_swz(i);
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) roff[hf] = i * 128 + (((2 * kk + hf) ^ f) << 4);
    }
    // ---- weights: piece (tap, channel tile, K-step, half) of wlat at lane * 16
    int wbase[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) wbase[c] = ((n0 >> 4) + c) * KC * 2048 + lane * 16;
    const int w_tapstride = a.CoutPad * Cin * 4;

    unsigned offc[NX], offn[NX];
    int ltap = 0, lkc = 0, kload = 0;
    tap_offsets(0, offc);
    tap_offsets(1, offn);
    v4f xr[D][NX], wr[D][CT][2];
    auto gload = [&](auto set_tag) __attribute__((always_inline)) {
        constexpr int S = decltype(set_tag)::value;
        const int so = lkc * 128, wso = ltap * w_tapstride + lkc * 2048;
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
                wr[S][c][hf] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wbase[c] + hf * 1024, wso, 0));
#pragma unroll
        for (int u = 0; u < NX; ++u)
            if (xact[u]) xr[S][u] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (int)offc[u], so, 0));
        if (++kload < KS) {          // advance the load stream (past the end the counters stay put: valid memory is re-read)
            if (++lkc == KC) {
                lkc = 0;
                ++ltap;
#pragma unroll
                for (int u = 0; u < NX; ++u) offc[u] = offn[u];
                tap_offsets(ltap + 1, offn);
            }
        }
    };
    // ds_write2_b32 takes its two dwords from two registers WHERE THEY ARE (elements 0, 2 / 1, 3 of the loaded chunk); the
    // 64-bit store the compiler makes of any C form needs them copied side by side first -- two v_mov per store, and every
    // vector instruction is added to the exact-fp32 MFMA's time on its SIMD (alone on the chip fpn p6 53 -> 41 us,
    // profiles/r04_lat_one_wave.log).  LDS operations of one wave execute in order; an operation the compiler does not
    // track only makes its counted waits a little longer.
    typedef __attribute__((address_space(3))) void *lds_as_t;
    const int lds0 = (int)(size_t)(lds_as_t)lds;                 // the image's byte address in LDS
    auto lstore = [&](int stage, auto set_tag) __attribute__((always_inline)) {
        constexpr int S = decltype(set_tag)::value;
        const int base = lds0 + stage * (PT * 2048);
#pragma unroll
        for (int u = 0; u < NX; ++u)
            if (xact[u]) {
                asm volatile("ds_write2_b32 %0, %1, %2 offset1:1" ::"v"(base + woff_lo[u]), "v"(xr[S][u][0]), "v"(xr[S][u][2]) : "memory");
                asm volatile("ds_write2_b32 %0, %1, %2 offset1:1" ::"v"(base + woff_hi[u]), "v"(xr[S][u][1]), "v"(xr[S][u][3]) : "memory");
            }
        // several waves share the image: the barrier behind this must find the writes complete, and the compiler's wait in front
        // of it does not count operations it did not emit
        if constexpr (WB > 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    v4f xf[2][PT][2];                // fragments by LDS stage
    auto lread = [&](auto stage_tag) __attribute__((always_inline)) {
        constexpr int ST = decltype(stage_tag)::value;
        const unsigned char *base = lds + ST * (PT * 2048);
#pragma unroll
        for (int p = 0; p < PT; ++p)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) xf[ST][p][hf] = *(const v4f *)(base + p * 2048 + roff[hf]);
    };
    v4f acc[CT][PT];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int p = 0; p < PT; ++p) acc[c][p] = v4f{0.f, 0.f, 0.f, 0.f};
    auto mfmas = [&](auto set_tag) __attribute__((always_inline)) {
        constexpr int S = decltype(set_tag)::value;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int c = 0; c < CT; ++c)
#pragma unroll
                    for (int p = 0; p < PT; ++p)
                        acc[c][p] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[S][c][hf][e], xf[S & 1][p][hf][e], acc[c][p], 0, 0, 0);
    };
    // one K-step whose operands are register set S and the fragments of LDS stage S & 1 (D is even): the positions of the
    // next step go through the other stage into the other fragment registers FIRST (write and read of one wave execute in
    // order; both complete under the MFMAs), then the multiply, then the set is refilled with step + D
    auto kstep = [&](auto set_tag) __attribute__((always_inline)) {
        constexpr int S = decltype(set_tag)::value, N = (S + 1) % D;
        lstore((S + 1) & 1, std::integral_constant<int, N>{});
        if constexpr (WB > 1) __syncthreads();    // every wave's rows of the stage are written; its previous readers passed the
                                                  // barrier of the step before with their reads complete
        lread(std::integral_constant<int, (S + 1) & 1>{});
        __builtin_amdgcn_sched_barrier(0);
        mfmas(set_tag);
        gload(set_tag);
    };
    auto for_sets = [&](auto fn) __attribute__((always_inline)) {
        fn(std::integral_constant<int, 0>{});
        fn(std::integral_constant<int, 1>{});
        if constexpr (D > 2) { fn(std::integral_constant<int, 2>{}); fn(std::integral_constant<int, 3>{}); }
        if constexpr (D > 4) {
            fn(std::integral_constant<int, 4>{}); fn(std::integral_constant<int, 5>{});
            fn(std::integral_constant<int, 6>{}); fn(std::integral_constant<int, 7>{});
        }
        if constexpr (D > 8) {
            fn(std::integral_constant<int, 8>{}); fn(std::integral_constant<int, 9>{});
            fn(std::integral_constant<int, 10>{}); fn(std::integral_constant<int, 11>{});
            fn(std::integral_constant<int, 12>{}); fn(std::integral_constant<int, 13>{});
            fn(std::integral_constant<int, 14>{}); fn(std::integral_constant<int, 15>{});
        }
    };
    if constexpr (IL) {
        static_assert(WB == 1 && PT == 1 && CT == 1, "the interleaved K-step is written for the one-accumulator wave");
        // ---- the ONE-accumulator wave (fpn p6 / p7 / lateral5 at batch 1-2): its 8 MFMAs per K-step are one dependent chain, a
        // wave issues in order, and while MFMA i + 1 waits for MFMA i nothing behind it in the program can issue -- with the
        // K-step written as [stage next step | 8 MFMAs | loads] only the LAST MFMA covered any of the ~35 other instructions:
        // 488 cycles per K-step for 256 of matrix pipe, whatever the prefetch depth or the tile order (p6 59 us alone at every
        // D and either order, profiles/r04_lat_one_wave.log).  Here every MFMA is followed by its share of the other work
        // (sched_barrier fences pin the order): the 32 cycles of its execution carry them.
        // Set k % D holds K-step k; step k multiplies set S, stages the positions of step k + 1 (set N) through LDS stage
        // (k + 1) & 1, and refills the set step k - 1 freed (T) with step k + D - 1.
        auto issue_w = [&](auto set_tag) __attribute__((always_inline)) {
            constexpr int S = decltype(set_tag)::value;
            const int wso = ltap * w_tapstride + lkc * 2048;
            wr[S][0][0] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wbase[0], wso, 0));
            wr[S][0][1] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wbase[0] + 1024, wso, 0));
        };
        auto issue_x = [&](auto set_tag) __attribute__((always_inline)) {
            constexpr int S = decltype(set_tag)::value;
            const int so = lkc * 128;
#pragma unroll
            for (int u = 0; u < NX; ++u) xr[S][u] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (int)offc[u], so, 0));
        };
        auto advance = [&]() __attribute__((always_inline)) {
            if (++kload < KS) {
                if (++lkc == KC) {
                    lkc = 0;
                    ++ltap;
#pragma unroll
                    for (int u = 0; u < NX; ++u) offc[u] = offn[u];
                    tap_offsets(ltap + 1, offn);
                }
            }
        };
#define SSD_SB __builtin_amdgcn_sched_barrier(0)
        auto kstep_il = [&](auto set_tag) __attribute__((always_inline)) {
            constexpr int S = decltype(set_tag)::value, N = (S + 1) % D, T = (S + D - 1) % D, STN = (S + 1) & 1;
            unsigned char *base = lds + STN * (PT * 2048);
            auto M = [&](int hf, int e) __attribute__((always_inline)) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[S][0][hf][e], xf[S & 1][0][hf][e], acc[0][0], 0, 0, 0);
            };
            // (IL >= 2: timing ablations of the diagnostics build, results wrong -- 2 without the LDS round trip, 3 also without
            //  the position loads, 4 also without the weight loads, 5 also without the load stream's bookkeeping)
            SSD_SB; M(0, 0); SSD_SB;
            if constexpr (IL < 2) {
                // (ds_write2_b32: see lstore)
                asm volatile("ds_write2_b32 %0, %1, %2 offset1:1" ::"v"(lds0 + STN * (PT * 2048) + woff_lo[0]), "v"(xr[N][0][0]), "v"(xr[N][0][2]) : "memory");
                asm volatile("ds_write2_b32 %0, %1, %2 offset1:1" ::"v"(lds0 + STN * (PT * 2048) + woff_hi[0]), "v"(xr[N][0][1]), "v"(xr[N][0][3]) : "memory");
            }
            SSD_SB; M(0, 1); SSD_SB;
            if constexpr (IL < 2) {
                asm volatile("ds_write2_b32 %0, %1, %2 offset1:1" ::"v"(lds0 + STN * (PT * 2048) + woff_lo[1]), "v"(xr[N][1][0]), "v"(xr[N][1][2]) : "memory");
                asm volatile("ds_write2_b32 %0, %1, %2 offset1:1" ::"v"(lds0 + STN * (PT * 2048) + woff_hi[1]), "v"(xr[N][1][1]), "v"(xr[N][1][3]) : "memory");
            }
            SSD_SB; M(0, 2); SSD_SB;
            if constexpr (IL < 2) {
                xf[STN][0][0] = *(const v4f *)(base + roff[0]);
                xf[STN][0][1] = *(const v4f *)(base + roff[1]);
            }
            SSD_SB; M(0, 3); SSD_SB;
            if constexpr (IL < 4) issue_w(std::integral_constant<int, T>{});
            SSD_SB; M(1, 0); SSD_SB;
            if constexpr (IL < 3) issue_x(std::integral_constant<int, T>{});
            SSD_SB; M(1, 1); SSD_SB;
            if constexpr (IL < 5) advance();
            SSD_SB; M(1, 2); SSD_SB; M(1, 3); SSD_SB;
        };
#undef SSD_SB
        static_assert(NX == 2, "one wave, 16 positions: two row loads per K-step");
        for_sets([&](auto s) __attribute__((always_inline)) {
            if constexpr (decltype(s)::value < D - 1) { issue_w(s); issue_x(s); advance(); }
        });
        lstore(0, std::integral_constant<int, 0>{});
        lread(std::integral_constant<int, 0>{});
        int ks = 0;
        for (; ks + D <= KS; ks += D) for_sets([&](auto s) __attribute__((always_inline)) { kstep_il(s); });
        const int rem = KS - ks;
        for_sets([&](auto s) __attribute__((always_inline)) { if (decltype(s)::value < rem) kstep_il(s); });
    } else {
    for_sets([&](auto s) __attribute__((always_inline)) { gload(s); });
    lstore(0, std::integral_constant<int, 0>{});
    if constexpr (WB > 1) __syncthreads();
    lread(std::integral_constant<int, 0>{});
    int ks = 0;
    for (; ks + D <= KS; ks += D) for_sets([&](auto s) __attribute__((always_inline)) { kstep(s); });
    {
        const int rem = KS - ks;     // 0 .. D-1 steps left: their operands are loaded (set d = step's index in the group)
        for_sets([&](auto s) __attribute__((always_inline)) { if (decltype(s)::value < rem) kstep(s); });
    }
    }

    epilogue_16x16<PT, CT>(a, L, acc, m0, n0, lane);
}

int igemm_lat_bm(int tile) { return (tile == IGEMM_LAT_2x1 || tile == IGEMM_LAT_2x2 || tile == IGEMM_LAT_W4_2x1) ? 32 : 16; }
bool igemm_lat_n_major(int tile) { return tile == IGEMM_LAT_1x1_NM || tile == IGEMM_LAT_1x1_D8_NM; }
int igemm_lat_bn(int tile)
{
    switch (tile) {
    case IGEMM_LAT_1x2: case IGEMM_LAT_2x2: case IGEMM_LAT_W2_1x1: return 32;
    case IGEMM_LAT_W4_1x1: case IGEMM_LAT_W4_2x1: return 64;
    case IGEMM_LAT_W4_1x2: return 128;
    default: return 16;
    }
}

template <int PT, int CT, int D, int WB = 1, int IL = 0>
static hipError_t launch_l(const IgemmArgs &a, int total_tiles_m, hipStream_t s)
{
    const long long nblk = (long long)total_tiles_m * a.n_tiles_n;
    if (nblk <= 0 || nblk > 0x7fffffffLL) return hipErrorInvalidValue;
    if (a.taps == 9) hipLaunchKernelGGL((igemm_lat_kernel<PT, CT, 9, D, WB, IL>), dim3((unsigned)nblk), dim3(64 * WB), 0, s, a);
    else hipLaunchKernelGGL((igemm_lat_kernel<PT, CT, 1, D, WB, IL>), dim3((unsigned)nblk), dim3(64 * WB), 0, s, a);
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
    if (!igemm_lat_supports(a) || !a.wt_lat || (a.taps != 1 && a.taps != 9)) return hipErrorInvalidValue;
    if ((long long)a.taps * a.CoutPad * a.Cin * 4 >= (1LL << 31)) return hipErrorInvalidValue;
    for (int l = 0; l < a.nlevels; ++l) {
        if ((long long)a.B * a.lv[l].H * a.lv[l].W * a.Cin * 4 >= (1LL << 31)) return hipErrorInvalidValue;
        if ((long long)a.B * a.lv[l].out_bstride * 4 >= (1LL << 31) || a.lv[l].out_bstride < 0) return hipErrorInvalidValue;
    }
    if ((a.mean != nullptr) != (a.sf != nullptr) || (a.mean != nullptr) != (a.beta != nullptr)) return hipErrorInvalidValue;
    if (a.mean && (a.bias || a.res)) return hipErrorInvalidValue;
    if (a.bias && a.res) return hipErrorInvalidValue;
    if (a.out2 && !a.mean) return hipErrorInvalidValue;
    if (a.n_tiles_n * igemm_lat_bn(tile) != a.CoutPad || a.nlevels < 1 || a.nlevels > SSD_MAX_LEVELS) return hipErrorInvalidValue;
    if (a.n_major && (a.tiles_m != total_tiles_m || total_tiles_m < 1)) return hipErrorInvalidValue;
    switch (tile) {
    case IGEMM_LAT_1x1: return launch_l<1, 1, 4>(a, total_tiles_m, s);
    case IGEMM_LAT_1x1_D16: return launch_l<1, 1, 16, 1, 1>(a, total_tiles_m, s);
#ifdef SSD_DIAG   // libssd_hip_diag.so only: the other prefetch depths / tile orders of the interleaved K-step (measured, round 4), and
                  // its timing ablations (results wrong)
    case IGEMM_LAT_1x1_IL: case IGEMM_LAT_1x1_NM: return launch_l<1, 1, 4, 1, 1>(a, total_tiles_m, s);
    case IGEMM_LAT_1x1_D8: case IGEMM_LAT_1x1_D8_NM: return launch_l<1, 1, 8, 1, 1>(a, total_tiles_m, s);
    case 33: return launch_l<1, 1, 16, 1, 2>(a, total_tiles_m, s);
    case 34: return launch_l<1, 1, 16, 1, 3>(a, total_tiles_m, s);
    case 35: return launch_l<1, 1, 16, 1, 4>(a, total_tiles_m, s);
    case 36: return launch_l<1, 1, 16, 1, 5>(a, total_tiles_m, s);
#endif
    case IGEMM_LAT_1x2: return launch_l<1, 2, 4>(a, total_tiles_m, s);
    case IGEMM_LAT_2x1: return launch_l<2, 1, 4>(a, total_tiles_m, s);
    case IGEMM_LAT_2x2: return launch_l<2, 2, 2>(a, total_tiles_m, s);
    case IGEMM_LAT_W2_1x1: return launch_l<1, 1, 4, 2>(a, total_tiles_m, s);
    case IGEMM_LAT_W4_1x1: return launch_l<1, 1, 4, 4>(a, total_tiles_m, s);
    case IGEMM_LAT_W4_2x1: return launch_l<2, 1, 4, 4>(a, total_tiles_m, s);
    case IGEMM_LAT_W4_1x2: return launch_l<1, 2, 4, 4>(a, total_tiles_m, s);
    }
    return hipErrorInvalidValue;
}
