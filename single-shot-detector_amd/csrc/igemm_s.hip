// Small-tile form of the implicit-GEMM convolution: v_mfma_f32_16x16x4_f32 on LDS-staged tiles of 32 or 64 positions by
// 32 or 64 channels, for the launches of a batch-1 / batch-2 forward that have only a few hundred 64x64 tiles (the late
// MobileNet pointwise layers, the FPN laterals): on v_mfma_f32_32x32x2_f32 a wave's unit of work is a 32x32 accumulator with
// a 64-cycle step, so a layer of 1 120 such units leaves 96 of the 1 024 SIMDs with two units and everyone waits for them
// (2 240 x 512 outputs, K = 512: 13.6 us of chain for 7.5 us of work), and a 64x64 tile's K loop cannot be shorter than
// 16 x 1 024 cycles.  The 16x16x4 instruction has the same throughput (64 FLOP per cycle and SIMD), a 32-cycle step on a
// 16x16 accumulator, and is bit for bit the same k-ascending fmaf chain (scripts/experiments/mfma_16x16x4_probe.hip), so the
// same work splits four times finer.  Results are bit-identical to igemm.hip, igemm_lat.hip and the oracle.
//
//   block       256 threads = 2 x 2 waves; wave tile TM x 16 positions by TN x 16 channels; tile BM = 32 TM, BN = 32 TN
//   staging     as igemm.hip: buffer_load_dwordx4 (zero padding = range check, per-row tap masks) into registers, two
//               register sets (loads three K-steps ahead), registers -> LDS stage, one barrier per K-step of 32 channels
//   LDS image   a row's 32 channels are stored by MFMA role: [kk = channel & 3][t = channel >> 2] -- lane (i, kk) of a
//               fragment read takes its eight operands t = 0..7 with two ds_read_b128.  In the physical channel order of
//               ssd_internal.h a thread's 16-byte chunk (octet o, half h) holds logical 8o + {0,2,4,6} + h: elements (0,2)
//               go to kk = h at t = 2o, 2o+1 and elements (1,3) to kk = h + 2 -- two 8-byte LDS writes.  The 16-byte slot
//               (2 kk + half) of row r sits at slot ^ ((r >> 1) & 7) ^ (2 * ((r >> 2) & 1)): conflict-free for the four
//               16-lane groups of ds_read_b128 (computed for the group table of MI355X_MICROARCH.md; checked with
//               SQ_LDS_BANK_CONFLICT).
//   product     transposed (weights = the MFMA's A operand), epilogue straight from the accumulators: igemm_mfma16.h
#include "igemm_mfma16.h"
#include <type_traits>

typedef float v2f __attribute__((ext_vector_type(2)));

template <int TM, int TN, int TAPS>
__global__ __launch_bounds__(256) void igemm_s_kernel(const IgemmArgs a)
{
    constexpr int BM = 32 * TM, BN = 32 * TN;     // positions x channels per block
    constexpr int NA = TM, NB = TN;               // 16-byte loads per thread and K-step (rows (tid >> 3) + 32 u)
    constexpr int X_BYTES = BM * 128, STAGE = (BM + BN) * 128;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_m = wave >> 1, wave_n = wave & 1;
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
        (void *)(a.in + L.in_off), 0, (int)((long long)a.B * H * W * Cin * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(a.wt + L.wt_off), 0, (int)((long long)TAPS * a.CoutPad * Cin * 4), 0x00020000);
    constexpr unsigned OOB = 0x80000000u;

    // ---- loader bookkeeping (igemm.hip): thread owns chunk (tid & 7) of rows (tid >> 3) + 32 u
    int xbase[NA];
    unsigned xmask[NA];
    const bool dense1x1 = TAPS == 1 && a.stride == 1 && a.pad == 0 && L.OH == H && OW == W;
#pragma unroll
    for (int u = 0; u < NA; ++u) {
        const int m = m0 + (tid >> 3) + 32 * u;
        const bool rowok = m < M;
        const int mm = rowok ? m : 0;
        if (dense1x1) {
            xbase[u] = (mm * Cin + (tid & 7) * 4) * 4;
            xmask[u] = rowok ? 1u : 0u;
        } else {
            const int b = (int)udivl((unsigned)mm, L.dP), pp = mm - b * P;
            const int oy = (int)udivl((unsigned)pp, L.dOW), ox = pp - oy * OW;
            const int iy0 = oy * a.stride - a.pad, ix0 = ox * a.stride - a.pad;
            xbase[u] = ((b * H * W + iy0 * W + ix0) * Cin + (tid & 7) * 4) * 4;
            unsigned vx = 0, mk = 0;
#pragma unroll
            for (int k = 0; k < (TAPS == 9 ? 3 : 1); ++k)
                if ((unsigned)(ix0 + k) < (unsigned)W) vx |= 1u << k;
#pragma unroll
            for (int k = 0; k < (TAPS == 9 ? 3 : 1); ++k)
                if ((unsigned)(iy0 + k) < (unsigned)H) mk |= vx << (3 * k);
            xmask[u] = rowok ? mk : 0u;
        }
    }
    auto tap_offsets = [&](int t, unsigned (&off)[NA]) {
        const int tky = TAPS == 9 ? t / 3 : 0, tkx = TAPS == 9 ? t - 3 * tky : 0;
        const int d = (tky * W + tkx) * Cin * 4;
#pragma unroll
        for (int u = 0; u < NA; ++u) off[u] = ((xmask[u] >> t) & 1u) ? (unsigned)(xbase[u] + d) : OOB;
    };
    const int wvoff = ((n0 + (tid >> 3)) * Cin + (tid & 7) * 4) * 4;
    const int w_ustride = 32 * Cin * 4, w_tapstride = a.CoutPad * Cin * 4;
    // LDS write offsets of this thread's chunk: octet o, half h -> elements (0,2) at role kk = h, (1,3) at kk = h + 2
    int woff_lo, woff_hi;
    {
        const int r = tid >> 3, c = tid & 7, o = c >> 1, hh = c & 1;
        const int f = ((r >> 1) & 7) ^ (2 * ((r >> 2) & 1));
        woff_lo = r * 128 + ((((hh) * 2 + (o >> 1)) ^ f) << 4) + (o & 1) * 8;
        woff_hi = r * 128 + ((((hh + 2) * 2 + (o >> 1)) ^ f) << 4) + (o & 1) * 8;
    }
    // fragment read offsets of this lane: row i of a 16-row group, role kk
    int roff[2];
    {
        const int i = lane & 15, kk = lane >> 4;
        const int f = ((i >> 1) & 7) ^ (2 * ((i >> 2) & 1));
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) roff[hf] = i * 128 + (((2 * kk + hf) ^ f) << 4);
    }

    const int KC = Cin >> 5, KS = TAPS * KC;
    unsigned offc[NA], offn[NA];
    int ltap = 0, lkc = 0, kload = 0;
    tap_offsets(0, offc);
    tap_offsets(1, offn);
    typedef v4f (&RegX)[NA];
    typedef v4f (&RegW)[NB];
    v4f rx0[NA], rw0[NB], rx1[NA], rw1[NB];
    auto gload = [&](RegX qx, RegW qw) __attribute__((always_inline)) {
        const int so = lkc * 128;
#pragma unroll
        for (int u = 0; u < NA; ++u)
            qx[u] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (int)offc[u], so, 0));
#pragma unroll
        for (int u = 0; u < NB; ++u)
            qw[u] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wvoff + u * w_ustride, ltap * w_tapstride + so, 0));
        if (++kload < KS) {          // advance the load stream (past the end the counters stay put: valid memory is re-read)
            if (++lkc == KC) {
                lkc = 0;
                ++ltap;
#pragma unroll
                for (int u = 0; u < NA; ++u) offc[u] = offn[u];
                tap_offsets(ltap + 1, offn);
            }
        }
    };
    auto lstore = [&](int stage, RegX qx, RegW qw) __attribute__((always_inline)) {
        unsigned char *base = lds + stage * STAGE;
#pragma unroll
        for (int u = 0; u < NA; ++u) {
            *(v2f *)(base + woff_lo + u * 4096) = v2f{qx[u][0], qx[u][2]};
            *(v2f *)(base + woff_hi + u * 4096) = v2f{qx[u][1], qx[u][3]};
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            *(v2f *)(base + X_BYTES + woff_lo + u * 4096) = v2f{qw[u][0], qw[u][2]};
            *(v2f *)(base + X_BYTES + woff_hi + u * 4096) = v2f{qw[u][1], qw[u][3]};
        }
    };
    v4f acc[TN][TM];
#pragma unroll
    for (int c = 0; c < TN; ++c)
#pragma unroll
        for (int p = 0; p < TM; ++p) acc[c][p] = v4f{0.f, 0.f, 0.f, 0.f};
    // one K-step from LDS stage `stage`: fragments (two 16-byte reads each), then 8 MFMAs per accumulator in t order
    auto compute = [&](int stage) __attribute__((always_inline)) {
        const unsigned char *xb = lds + stage * STAGE + wave_m * TM * 2048;
        const unsigned char *wb = lds + stage * STAGE + X_BYTES + wave_n * TN * 2048;
        v4f xf[TM][2], wf[TN][2];
#pragma unroll
        for (int p = 0; p < TM; ++p)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) xf[p][hf] = *(const v4f *)(xb + p * 2048 + roff[hf]);
#pragma unroll
        for (int c = 0; c < TN; ++c)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) wf[c][hf] = *(const v4f *)(wb + c * 2048 + roff[hf]);
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int c = 0; c < TN; ++c)
#pragma unroll
                    for (int p = 0; p < TM; ++p)
                        acc[c][p] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[c][hf][e], xf[p][hf][e], acc[c][p], 0, 0, 0);
    };
    // ---- pipeline: set j & 1 holds K-step j from its load (issued during step j - 3) until it is written to LDS (step j - 1)
    gload(rx0, rw0);                 // step 0
    gload(rx1, rw1);                 // step 1
    lstore(0, rx0, rw0);
    gload(rx0, rw0);                 // step 2
    __syncthreads();
    auto kstep = [&](auto cur_tag) __attribute__((always_inline)) {
        constexpr int cur = decltype(cur_tag)::value, nxt = cur ^ 1;
        // registers of step ks + 1 (set nxt) -> LDS stage nxt; reload that set with step ks + 3
        if constexpr (nxt) { lstore(nxt, rx1, rw1); gload(rx1, rw1); }
        else { lstore(nxt, rx0, rw0); gload(rx0, rw0); }
        compute(cur);
        __syncthreads();             // everyone has read stage cur and written stage nxt
    };
    {
        int ks = 0;
        for (; ks + 2 <= KS - 1; ks += 2) {
            kstep(std::integral_constant<int, 0>{});
            kstep(std::integral_constant<int, 1>{});
        }
        if (ks < KS - 1) kstep(std::integral_constant<int, 0>{});
    }
    compute((KS - 1) & 1);           // last K-step: nothing left to stage

    epilogue_16x16<TM, TN>(a, L, acc, m0 + wave_m * TM * 16, n0 + wave_n * TN * 16, lane);
}

int igemm_s_bm(int tile) { return (tile == IGEMM_S_64x32 || tile == IGEMM_S_64x64) ? 64 : 32; }
int igemm_s_bn(int tile) { return (tile == IGEMM_S_32x64 || tile == IGEMM_S_64x64) ? 64 : 32; }

template <int TM, int TN>
static hipError_t launch_s(const IgemmArgs &a, int total_tiles_m, hipStream_t s)
{
    const long long nblk = (long long)total_tiles_m * a.n_tiles_n;
    if (nblk <= 0 || nblk > 0x7fffffffLL) return hipErrorInvalidValue;
    if (a.taps == 9) hipLaunchKernelGGL((igemm_s_kernel<TM, TN, 9>), dim3((unsigned)nblk), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((igemm_s_kernel<TM, TN, 1>), dim3((unsigned)nblk), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_igemm_s(int tile, const IgemmArgs &a, int total_tiles_m, hipStream_t s)
{
    // the forms of the 16x16 epilogue (igemm_lat_supports), and everything the kernel assumes
    if (!igemm_lat_supports(a) || (a.taps != 1 && a.taps != 9)) return hipErrorInvalidValue;
    if ((long long)a.taps * a.CoutPad * a.Cin * 4 >= (1LL << 31)) return hipErrorInvalidValue;
    for (int l = 0; l < a.nlevels; ++l) {
        if ((long long)a.B * a.lv[l].H * a.lv[l].W * a.Cin * 4 >= (1LL << 31)) return hipErrorInvalidValue;
        if ((long long)a.B * a.lv[l].out_bstride * 4 >= (1LL << 31) || a.lv[l].out_bstride < 0) return hipErrorInvalidValue;
    }
    if ((a.mean != nullptr) != (a.sf != nullptr) || (a.mean != nullptr) != (a.beta != nullptr)) return hipErrorInvalidValue;
    if (a.mean && (a.bias || a.res)) return hipErrorInvalidValue;
    if (a.bias && a.res) return hipErrorInvalidValue;
    if (a.out2 && !a.mean) return hipErrorInvalidValue;
    if (a.n_tiles_n * igemm_s_bn(tile) != a.CoutPad || a.nlevels < 1 || a.nlevels > SSD_MAX_LEVELS) return hipErrorInvalidValue;
    switch (tile) {
    case IGEMM_S_32x32: return launch_s<1, 1>(a, total_tiles_m, s);
    case IGEMM_S_32x64: return launch_s<1, 2>(a, total_tiles_m, s);
    case IGEMM_S_64x32: return launch_s<2, 1>(a, total_tiles_m, s);
    case IGEMM_S_64x64: return launch_s<2, 2>(a, total_tiles_m, s);
    }
    return hipErrorInvalidValue;
}
