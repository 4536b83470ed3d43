// Implicit-GEMM convolution on the exact-fp32 matrix cores of gfx950
// (v_mfma_f32_32x32x2_f32): every dense 1x1 / 3x3 convolution of the detector --
// MobileNet/ShuffleNet pointwise layers, FPN laterals (with the nearest-upsample merge
// fused), FPN 3x3 outputs (stride 1 and the explicit-pad stride-2 p6/p7), the shared
// head towers (all five pyramid levels in one launch, per-level batch-norm) and the
// final class/box convolutions.
//
//   GEMM view   rows  m = (image, oy, ox) output positions of one pyramid level
//               cols  n = output channel (physical order)
//               k     = (ky, kx, ci), ci fastest, ci in PHYSICAL order (ssd_internal.h)
//   block       256 threads = 4 waves, tile BM x BN, K-step 32 channels of one tap
//   LDS         2 stages x (BM + BN) rows x 128 B; row r keeps its eight 16-B chunks
//               at chunk ^ ((r >> 1) & 7): ds_read_b128 of 16 rows x 1 chunk and
//               ds_write_b128 of 1 row x 8 chunks are both bank-conflict free.
//   pipeline    global_load_dwordx4 of a later K-step -> registers while the MFMAs of
//               K-step s run from LDS; registers -> LDS (other stage); one barrier.
//               fp32 operands: the loop runs its K-steps in pairs (LDS stage = compile-time
//               constant: no vector address arithmetic per step), the wide tiles keep two
//               register sets (loads three steps ahead); tap changes and row decompositions
//               use per-row masks and host-made multiply-shift divisors -- the exact-fp32
//               MFMA shares the SIMD's issue with every other vector instruction, so the
//               kernel is tuned by instruction count.
//   numerics    each output element is ONE accumulator that receives its k terms in
//               increasing logical (ky,kx,ci) order: bit-identical to the fmaf chain
//               of the oracle.  Epilogue ops are separately rounded (-ffp-contract=off).
#include "ssd_internal.h"
#include <type_traits>

typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef _Float16 v4h __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------
// S16 ("split fp16", precision mode f16x3).  An fp32 value x is kept as two halves
// h = f16(x), l = f16(x - h) (x - h is exact in fp32; h + l carries ~22 significand bits and
// fp16 subnormals are honoured by the matrix cores, measured: scripts/experiments/
// mfma_f16_probe.hip).  A tensor row of Cp channels is Cp*4 bytes as in fp32: per octet of 8
// physical channels 16 B of h followed by 16 B of l.  A product x*w is evaluated as
// xh*wh + xh*wl + xl*wh on v_mfma_f32_32x32x16_f16 (every term exact in fp32, the dropped
// xl*wl term is 2^-22 relative), accumulated in fp32: 3 MFMAs of 32 pipe cycles per 16
// channels instead of 8 exact-fp32 MFMAs of 64 cycles -- 5.3x the matrix throughput at the
// accuracy of an fp32 accumulation chain (K = 2304: max error vs double 7.2e-6 for both).
// The result is NOT bit-identical to the oracle's fmaf chain; tests bound it by the
// north-star tolerance instead.  The K-step stays 32 channels = 128 B per row, so the
// global -> LDS path and the LDS image are the ones of the fp32 kernel.
// ---------------------------------------------------------------------------------------
static __device__ __forceinline__ void split_f16(const v4f v, v2u &hi, v2u &lo, bool &ovf)
{
    v4h h, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float x = v[e];
        ovf |= !(fabsf(x) <= 65504.0f);
        x = __builtin_fminf(__builtin_fmaxf(x, -65504.0f), 65504.0f);
        h[e] = (_Float16)x;
        l[e] = (_Float16)(x - (float)h[e]);
    }
    hi = __builtin_bit_cast(v2u, h);
    lo = __builtin_bit_cast(v2u, l);
}
static __device__ __forceinline__ v4f join_f16(const v2u hi, const v2u lo)
{
    const v4h h = __builtin_bit_cast(v4h, hi), l = __builtin_bit_cast(v4h, lo);
    v4f v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (float)h[e] + (float)l[e];
    return v;
}

__device__ __forceinline__ unsigned udiv(unsigned n, UDiv u)      // ssd_internal.h: n < 2^31
{
    return u.sh < 0 ? n : __umulhi(n, u.mag) >> u.sh;
}

// DBG (timing experiments only, results are wrong): 1 = no global loads / LDS writes in the
// K loop, 2 = additionally no LDS fragment reads, 3 = additionally no barrier.
// DBG 7 (results are right): thread 0 of every block records 100 MHz timestamps of its phases.
template <int WAVES_M, int WAVES_N, int WM, int WN, int TAPS, int DBGT = 0, int S16 = 0>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, 2) void igemm_kernel(const IgemmArgs a)
{
    constexpr int DBG = (DBGT == 7 || DBGT == 8) ? 0 : DBGT;     // 8: no diagnostic, the deep-prefetch form of a 64-wide tile
    long long stamp[8];
    auto mark = [&](int i) {
        if constexpr (DBGT == 7) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); stamp[i] = wall_clock64(); }
    };
    if constexpr (DBGT == 7) stamp[0] = wall_clock64();
    constexpr int NT = 64 * WAVES_M * WAVES_N;   // threads per block (256 or 512)
    constexpr int RPP = NT / 8;                  // tile rows covered by one pass of the block
    constexpr int BM = WAVES_M * WM * 32;
    constexpr int BN = WAVES_N * WN * 32;
    constexpr int NA = BM / RPP;         // 16-B loads per thread per K-step, A tile
    constexpr int NB = BN / RPP;         // same, B tile
    static_assert(BM % RPP == 0 && BN % RPP == 0 && RPP % 16 == 0, "tile/thread mismatch");
    constexpr int A_BYTES = BM * 128;
    constexpr int STAGE = (BM + BN) * 128;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_m = wave / WAVES_N, wave_n = wave % WAVES_N;

    // blocks b, b+8, b+16 ... share an XCD (its L2): give them consecutive tiles so the
    // N-tiles of one row panel and neighbouring row panels hit the same L2.
    int swz;
    {
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
        swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tile_m = (int)udiv((unsigned)swz, a.dN);
    const int tile_n = swz - tile_m * a.n_tiles_n;
    int lvl = 0;
#pragma unroll
    for (int i = 1; i < SSD_MAX_LEVELS; ++i)
        if (i < a.nlevels && tile_m >= a.lv[i].tile_begin) lvl = i;
    const IgemmLevel L = a.lv[lvl];       // by value: one wide scalar load up front (a reference re-reads its fields, each behind its own wait)
    const int H = L.H, W = L.W, OW = L.OW, M = L.M;
    const int P = L.OH * L.OW;
    const int Cin = a.Cin;
    const int m0 = (tile_m - L.tile_begin) * BM;

    // ---- per-thread load bookkeeping: thread owns chunk (tid&7) of rows (tid>>3)+RPP*u.
    // Loads go through buffer resources (SGPR base + 32-bit byte offset per lane): an
    // offset >= num_records is answered with zeros by the range check, which IS the zero
    // padding of the convolution (and of rows beyond M) -- no select, no branch.
    const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(a.in + L.in_off), 0, (int)((long long)a.B * H * W * Cin * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t brsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(a.wt + L.wt_off), 0, (int)((long long)TAPS * a.CoutPad * Cin * 4), 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    // Per row: byte offset of its tap (0,0) (used only where that tap exists) and 9 bits "tap t reads inside the image"; the
    // offset of tap t is then one wave-uniform displacement away.  (A tap change used to redo the coordinate arithmetic
    // per row: ~100 vector instructions every KC K-steps, 0.2 per MFMA in the 3x3 launches -- every one of them takes an
    // issue slot from the matrix pipe, profiles/r02_pointwise_phases.log.)
    int abase0[NA];
    unsigned amask[NA];
    const bool dense1x1 = TAPS == 1 && L.stride == 1 && L.pad == 0 && L.OH == H && OW == W;   // rows of A = rows of the input
#pragma unroll
    for (int u = 0; u < NA; ++u) {
        const int m = m0 + (tid >> 3) + RPP * u;
        const bool rowok = m < M;
        const int mm = rowok ? m : 0;
        if (dense1x1) {
            abase0[u] = (mm * Cin + (tid & 7) * 4) * 4;
            amask[u] = rowok ? 1u : 0u;
        } else {
            const int b = (int)udiv((unsigned)mm, L.dP), p = mm - b * P;
            const int oy = (int)udiv((unsigned)p, L.dOW), ox = p - oy * OW;
            const int iy0 = oy * L.stride - L.pad, ix0 = ox * L.stride - L.pad;
            abase0[u] = (b * H * W * Cin + (tid & 7) * 4) * 4 + (iy0 * W + ix0) * Cin * 4;
            unsigned vx = 0, mk = 0;                              // bit 3*ky + kx
#pragma unroll
            for (int k = 0; k < (TAPS == 9 ? 3 : 1); ++k)
                if ((unsigned)(ix0 + k) < (unsigned)W) vx |= 1u << k;
#pragma unroll
            for (int k = 0; k < (TAPS == 9 ? 3 : 1); ++k)
                if ((unsigned)(iy0 + k) < (unsigned)H) mk |= vx << (3 * k);
            amask[u] = rowok ? mk : 0u;
        }
    }
    // byte offsets of this thread's A chunks for filter tap `t` (OOB when the tap is padding, or t == TAPS)
    auto tap_offsets = [&](int t, unsigned (&off)[NA]) {
        const int tky = TAPS == 9 ? t / 3 : 0, tkx = TAPS == 9 ? t - 3 * tky : 0;
        const int d = (tky * W + tkx) * Cin * 4;                  // wave-uniform
#pragma unroll
        for (int u = 0; u < NA; ++u) off[u] = ((amask[u] >> t) & 1u) ? (unsigned)(abase0[u] + d) : OOB;
    };
    const int bvoff = ((tile_n * BN + (tid >> 3)) * Cin + (tid & 7) * 4) * 4;
    const int b_ustride = RPP * Cin * 4;
    const int b_tapstride = a.CoutPad * Cin * 4;
    const int woff = (tid >> 3) * 128 + (((tid & 7) ^ ((tid >> 4) & 7)) << 4);

    // ---- per-lane fragment read offsets (4 octets of the 32-channel K-step) and the accumulators: filled by late_init(),
    // which the pipelines call right BEHIND their first global loads -- nothing here is needed to issue those, and every
    // instruction in front of them is latency the block's first MFMA waits for (the blocks of a 1x1 launch sit in their
    // prologues together).
    int roff[4];
    v16f acc[WM][WN];
    auto late_init = [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            // fp32: octet g, lanes 0-31 physical channels 0-3, lanes 32-63 physical 4-7.
            // S16: g = 2*s + hl: 16-channel step s, lane group (lane >> 5) takes octet 2s + group,
            //      hl = 0 the h chunk, 1 the l chunk of that octet.
            const int chunk = S16 ? ((g >> 1) * 4 + 2 * (lane >> 5) + (g & 1)) : (2 * g + (lane >> 5));
            roff[g] = (lane & 31) * 128 + ((chunk ^ (((lane & 31) >> 1) & 7)) << 4);
        }
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    };

    const int KC = Cin >> 5;
    const int KS = TAPS * KC;
    v4f ra[NA], rb[NB];
    typedef v4f (&RegA)[NA];
    typedef v4f (&RegB)[NB];
    unsigned offc[NA], offn[NA];       // A offsets of the load stream's tap and of the next tap
    int ltap = 0, lkc = 0, kload = 0;  // coordinates of the NEXT K-step to load
    tap_offsets(0, offc);
    tap_offsets(1, offn);
    auto gload_into = [&](RegA qa, RegB qb) __attribute__((always_inline)) {
        const int so = lkc * 128;      // scalar offset: 32 channels per K-step
#pragma unroll
        for (int u = 0; u < NA; ++u)
            qa[u] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(arsrc, (int)offc[u], so, 0));
#pragma unroll
        for (int u = 0; u < NB; ++u)
            qb[u] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(brsrc, bvoff + u * b_ustride, ltap * b_tapstride + so, 0));
    };
    auto gload = [&]() __attribute__((always_inline)) { gload_into(ra, rb); };
    // advance the load stream by one K-step (uniform branch once per tap; kept out of the
    // MFMA phases).  Past the last step the counters stay put: the software pipeline's
    // surplus prefetch re-reads valid memory.
    auto gadvance = [&]() {
        if (++kload < KS) {
            if (++lkc == KC) {
                lkc = 0;
                ++ltap;
#pragma unroll
                for (int u = 0; u < NA; ++u) offc[u] = offn[u];
                tap_offsets(ltap + 1, offn);
            }
        }
    };
    // S16 == 2: the A rows in memory are fp32 (a tensor that an fp32 consumer also reads: c3 / c4 under the FPN
    // laterals); the split into halves happens here, on the way into the S16 LDS image.  A thread's 16 bytes
    // are 4 channels = half an octet: h into chunk 2o, l into chunk 2o+1, 8 bytes each.
    const int woff_h = (tid >> 3) * 128 + ((((tid & 6)) ^ ((tid >> 4) & 7)) << 4) + (tid & 1) * 8;
    const int woff_l = (tid >> 3) * 128 + ((((tid & 6) + 1) ^ ((tid >> 4) & 7)) << 4) + (tid & 1) * 8;
    bool in_ovf = false;               // S16 == 2: a staged fp32 value was outside the fp16 range (or NaN) -> status word
    auto lstore_from = [&](int stage, RegA ra, RegB rb) __attribute__((always_inline)) {
        unsigned char *base = lds + stage * STAGE;
#pragma unroll
        for (int u = 0; u < NA; ++u) {
            if constexpr (S16 == 2) {
                v4h hh, ll;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    in_ovf |= !(fabsf(ra[u][e]) <= 65504.0f);
                    const float x = __builtin_fminf(__builtin_fmaxf(ra[u][e], -65504.0f), 65504.0f);
                    hh[e] = (_Float16)x;
                    ll[e] = (_Float16)(x - (float)hh[e]);
                }
                *(v2u *)(base + woff_h + u * (RPP * 128)) = __builtin_bit_cast(v2u, hh);
                *(v2u *)(base + woff_l + u * (RPP * 128)) = __builtin_bit_cast(v2u, ll);
            } else {
                *(v4f *)(base + woff + u * (RPP * 128)) = ra[u];
            }
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) *(v4f *)(base + A_BYTES + woff + u * (RPP * 128)) = rb[u];
    };
    auto lstore = [&](int stage) __attribute__((always_inline)) { lstore_from(stage, ra, rb); };
    // ---- fragment reads / MFMA phases -------------------------------------------------
    // A K-step is four phases (one 8-channel octet each, 4*WM*WN MFMAs = 4*WM*WN*64 pipe
    // cycles).  Phase g issues the LDS reads of octet g+1 FIRST and then its MFMAs, so a
    // fragment has a whole phase to arrive (v_mfma issue is in order: a wave that waits
    // for an LDS read right before an MFMA leaves the matrix pipe idle, and the partner
    // wave on the SIMD runs the same program in lockstep and does not cover it).  The
    // barrier sits between phases 2 and 3; right after it the wave reads octet 0 of the
    // NEXT stage, which then has phase 3's MFMAs to arrive.
    auto rdfrag = [&](int stage, int g, v4f (&af)[WM], v4f (&bf)[WN]) {
        const unsigned char *abase = lds + stage * STAGE + wave_m * WM * 4096;
        const unsigned char *bbase = lds + stage * STAGE + A_BYTES + wave_n * WN * 4096;
#pragma unroll
        for (int i = 0; i < WM; ++i) af[i] = *(const v4f *)(abase + i * 4096 + roff[g]);
#pragma unroll
        for (int j = 0; j < WN; ++j) bf[j] = *(const v4f *)(bbase + j * 4096 + roff[g]);
    };
    auto mfma16 = [&](const v4f (&af)[WM], const v4f (&bf)[WN]) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][t], bf[j][t], acc[i][j], 0, 0, 0);
    };
    constexpr int PM = 4 * WM * WN;          // MFMAs per phase
    constexpr int NFR = WM + WN;             // LDS reads per fragment set
    // scheduling request for one phase: `nrd` LDS reads first, then every MFMA followed by
    // a few scalar/vector ALU ops and, while they last, one LDS write / one global load.
    auto phase_sched = [&](int nrd, int nwr, int nld) {
        if (nrd) __builtin_amdgcn_sched_group_barrier(0x100, NFR, 0);
#pragma unroll
        for (int i = 0; i < PM; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (i < nwr) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            if (i < nld) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x006, 4, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    if constexpr (!S16) {
    v4f fa0[WM], fb0[WN], fa1[WM], fb1[WN];
    // Two register sets: the loads of K-step j are issued during step j-3 and written to LDS during step j-1 (set j & 1
    // is written in phase 0 of a step and reloaded in its phase 1), so a load has two K-steps to arrive instead of
    // three quarters of one -- the 16-step 1x1 launches, whose blocks all fetch at once, were waiting for them.
    // (The 64-wide tiles keep one set: five blocks per CU need <= 96 VGPRs -- except in the instance for launches that do not
    //  fill the chip anyway (DBGT 8, tile IGEMM_64x64D: batch-1 FPN / pointwise launches, one or two blocks per CU, whose
    //  K-steps wait for operands that come from HBM or the Infinity Cache, not from a warm L2).)
    constexpr bool DEEP = BM * BN >= 128 * 64 || DBGT == 8;
    v4f ra1[DEEP ? NA : 1], rb1[DEEP ? NB : 1];
    if constexpr (DEEP) {
        gload_into(ra, rb);        // step 0
        gadvance();
        gload_into((RegA)ra1, (RegB)rb1);      // step 1
        gadvance();
        late_init();
        lstore_from(0, ra, rb);
        gload_into(ra, rb);        // step 2
        gadvance();
    } else {
        gload();
        gadvance();
        late_init();
        lstore(0);
        gload();
        gadvance();
    }
    __syncthreads();
    mark(1);
    rdfrag(0, 0, fa0, fb0);
    // One K-step with the LDS stage as a compile-time constant: every LDS address of the loop is then a loop-invariant
    // register plus an immediate (with `ks & 1` as the stage the loop spent 8 vector adds per K-step on them -- 0.125 per
    // MFMA, each taking an issue slot from the matrix pipe).  The loop runs the steps in pairs, stage 0 then stage 1.
    auto kstep = [&](auto cur_tag) __attribute__((always_inline)) {
        constexpr int cur = decltype(cur_tag)::value, nxt = cur ^ 1;
        // phase 0: registers (step ks+1, set nxt) -> LDS stage nxt
        if (DBG < 2) rdfrag(cur, 1, fa1, fb1);
        if (DBG < 1) { if constexpr (DEEP && nxt) lstore_from(nxt, (RegA)ra1, (RegB)rb1); else lstore_from(nxt, ra, rb); }
        mfma16(fa0, fb0);
        phase_sched(1, NA + NB, 0);
        // phase 1: global loads of step ks+3 (one set: ks+2) -> the set just written
        if (DBG < 2) rdfrag(cur, 2, fa0, fb0);
        if (DBG < 1) { if constexpr (DEEP && nxt) gload_into((RegA)ra1, (RegB)rb1); else gload_into(ra, rb); }
        mfma16(fa1, fb1);
        phase_sched(1, 0, NA + NB);
        // phase 2
        if (DBG < 2) rdfrag(cur, 3, fa1, fb1);
        mfma16(fa0, fb0);
        phase_sched(1, 0, 0);
        // every wave has read stage cur and written stage nxt
        if (DBG < 3) __syncthreads();
        // phase 3: first fragment of the next K-step, last MFMAs of this one
        if (DBG < 2) rdfrag(nxt, 0, fa0, fb0);
        mfma16(fa1, fb1);
        phase_sched(1, 0, 0);
        gadvance();
    };
    {
        int ks = 0;                                   // K-steps 0 .. KS-2 stage their successor; step ks reads stage ks & 1
        // (Measured and not kept, batch 1: the waves that share a SIMD -- one per co-resident block, all started together -- do
        //  not advance alike (K-loop times of the blocks of a tower launch 75 .. 115 us, profiles/r03_tower_phases_b1.log) and the
        //  last ones finish alone on their SIMDs.  s_setprio 3 -> 0 as a wave's K loop advances, so that the arbiter favours
        //  whichever wave is behind: 1.745 -> 1.770 ms per forward, one stream 1.782 -> 1.786.)
        for (; ks + 2 <= KS - 1; ks += 2) {
            kstep(std::integral_constant<int, 0>{});
            kstep(std::integral_constant<int, 1>{});
        }
        if (ks < KS - 1) kstep(std::integral_constant<int, 0>{});
    }
    {   // last K-step: nothing left to stage
        const int cur = (KS - 1) & 1;
        rdfrag(cur, 1, fa1, fb1);
        mfma16(fa0, fb0);
        rdfrag(cur, 2, fa0, fb0);
        mfma16(fa1, fb1);
        rdfrag(cur, 3, fa1, fb1);
        mfma16(fa0, fb0);
        mfma16(fa1, fb1);
    }

    } else {
    // ---- S16 operands: a K-step is two 16-channel steps of 3*WM*WN MFMAs (32 pipe cycles each).
    // Phase 0: fragments of step 1, registers (K-step ks+1) -> LDS stage nxt, MFMAs of step 0, then
    // the global loads of K-step ks+2; barrier; phase 1: first fragments of the next stage, MFMAs of
    // step 1.
    auto rd16 = [&](int stage, int st, v4f (&ah)[WM], v4f (&al)[WM], v4f (&bh)[WN], v4f (&bl)[WN]) {
        const unsigned char *abase = lds + stage * STAGE + wave_m * WM * 4096;
        const unsigned char *bbase = lds + stage * STAGE + A_BYTES + wave_n * WN * 4096;
#pragma unroll
        for (int i = 0; i < WM; ++i) {
            ah[i] = *(const v4f *)(abase + i * 4096 + roff[2 * st]);
            al[i] = *(const v4f *)(abase + i * 4096 + roff[2 * st + 1]);
        }
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            bh[j] = *(const v4f *)(bbase + j * 4096 + roff[2 * st]);
            bl[j] = *(const v4f *)(bbase + j * 4096 + roff[2 * st + 1]);
        }
    };
    // small terms first; consecutive MFMAs go to different accumulators
    auto mf16 = [&](const v4f (&ah)[WM], const v4f (&al)[WM], const v4f (&bh)[WN], const v4f (&bl)[WN]) {
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8h, al[i]), __builtin_bit_cast(v8h, bh[j]), acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8h, ah[i]), __builtin_bit_cast(v8h, bl[j]), acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8h, ah[i]), __builtin_bit_cast(v8h, bh[j]), acc[i][j], 0, 0, 0);
    };
    constexpr int PM16 = 3 * WM * WN;
    constexpr int NFR16 = 2 * (WM + WN);
    // phase 0 carries the NA+NB LDS writes and the NA+NB global loads, phase 1 only fragment reads
    auto sched16 = [&](auto staging_tag) {
        constexpr bool STG = decltype(staging_tag)::value;
        constexpr int NST = NA + NB;
        constexpr int PER = (NST + PM16 - 1) / PM16;      // staging instructions of each kind per MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, NFR16, 0);
#pragma unroll
        for (int i = 0; i < PM16; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if constexpr (STG) {
                if (i * PER < NST) {
                    __builtin_amdgcn_sched_group_barrier(0x200, PER, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, PER, 0);
                }
            }
            __builtin_amdgcn_sched_group_barrier(0x006, 4, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    v4f ah0[WM], al0[WM], bh0[WN], bl0[WN], ah1[WM], al1[WM], bh1[WN], bl1[WN];
    gload();
    gadvance();
    late_init();
    lstore(0);
    gload();
    gadvance();
    __syncthreads();
    mark(1);
    rd16(0, 0, ah0, al0, bh0, bl0);
    if (KS > 1) {
      int ks = 0;
      do {
        const int cur = ks & 1, nxt = cur ^ 1;
        rd16(cur, 1, ah1, al1, bh1, bl1);
        lstore(nxt);
        mf16(ah0, al0, bh0, bl0);
        gload();
        sched16(std::true_type{});
        __syncthreads();
        rd16(nxt, 0, ah0, al0, bh0, bl0);
        mf16(ah1, al1, bh1, bl1);
        sched16(std::false_type{});
        gadvance();
      } while (++ks < KS - 1);
    }
    {
        const int cur = (KS - 1) & 1;
        rd16(cur, 1, ah1, al1, bh1, bl1);
        mf16(ah0, al0, bh0, bl0);
        mf16(ah1, al1, bh1, bl1);
    }
    }

    // ---- epilogue: batch norm / bias / upsample-add / activation.
    // The accumulators (lane = column, 16 registers = 16 rows of a 32x32 tile) go through a
    // per-wave LDS transpose so that every lane ends up with 4 consecutive channels of one
    // row: 16-B global stores (256 contiguous bytes per row and instruction) and vector loads
    // of the BN parameters, a quarter of the store instructions of the direct form.  The LDS
    // image is plain row-major: a 32-lane ds_write_b32 covers 32 consecutive dwords of one row
    // and a 16-lane ds_read_b128 group covers 16 distinct 16-B chunks -- conflict-free as is.
    mark(2);
    if constexpr (S16 == 2) { if (in_ovf && a.flags) atomicOr(a.flags, 1); }
    const bool has_bn = a.mean != nullptr;
    constexpr int RW = WN * 32;                  // floats per row of the wave's sub-tile
    constexpr int C4N = RW / 4;                  // 16-B chunks per row
    constexpr int WAVE_REGION = WM * 32 * RW * 4;
    static_assert(WAVE_REGION * WAVES_M * WAVES_N <= 2 * STAGE, "epilogue transpose does not fit the staging LDS");
    __syncthreads();                             // every wave is done with the staging buffers
    {
        float *reg = (float *)(lds + wave * WAVE_REGION);
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    const int col = j * 32 + (lane & 31);
                    reg[row * RW + col] = acc[i][j][r];
                }
        __syncthreads();
        mark(3);
        const int c4 = lane % C4N;
        const int col = tile_n * BN + wave_n * RW + c4 * 4;
        // S16 rows: the lane's 4 channels are half an octet -- h at (octet*8 + half*2) floats, l 4 floats on
        const int col16 = tile_n * BN + wave_n * RW + (c4 >> 1) * 8 + (c4 & 1) * 2;
        constexpr int ROWS_PER_IT = 64 / C4N;    // rows per wave-instruction (C4N = 24: 2 rows, lanes 48-63 idle)
        constexpr int ITS = WM * 32 / ROWS_PER_IT;
        const bool colok = col < a.Cout && lane < ROWS_PER_IT * C4N;   // parameter vectors are padded to CoutPad
        v4f mean = {0.f, 0.f, 0.f, 0.f}, sf = {1.f, 1.f, 1.f, 1.f}, beta = {0.f, 0.f, 0.f, 0.f}, bias = {0.f, 0.f, 0.f, 0.f};
        if (colok && has_bn) {
            mean = *(const v4f *)(a.mean + L.param_off + col);
            sf = *(const v4f *)(a.sf + L.param_off + col);
            beta = *(const v4f *)(a.beta + L.param_off + col);
        }
        if (colok && a.bias) bias = *(const v4f *)(a.bias + L.param_off + col);
        // Row geometry without per-row divisions: row m = (image b, position p); the rows of one
        // lane advance by ROWS_PER_IT, i.e. by (qR images, rR positions) with one possible wrap.
        const int row0 = (lane / C4N) % ROWS_PER_IT;
        const int mfirst = m0 + wave_m * WM * 32 + row0;
        const int b0 = (int)udiv((unsigned)mfirst, L.dP), p0 = mfirst - b0 * P;
        const int qR = (int)udiv((unsigned)ROWS_PER_IT, L.dP), rR = ROWS_PER_IT - qR * P;
        const int bstride = (int)L.out_bstride, rstride = L.out_rstride;
        const int step = (qR * bstride + rR * rstride) * 4, wrapstep = (bstride - P * rstride) * 4;   // bytes
        // upsample-add operand (FPN laterals): all rows' loads first, then ONE wait (see below)
        v4f rv[ITS];
        if (a.res) {
#pragma unroll
            for (int it = 0; it < ITS; ++it) {
                const int m = mfirst + it * ROWS_PER_IT;
                rv[it] = v4f{0.f, 0.f, 0.f, 0.f};
                if (m < M && colok) {
                    const int b = (int)udiv((unsigned)m, L.dP), p = m - b * P;
                    const int oy = (int)udiv((unsigned)p, L.dOW), ox = p - oy * OW;
                    const int ch = L.OH >> 1, cw = OW >> 1;
                    const float *rrow = a.res + L.res_off + (((long long)b * ch + (oy >> 1)) * cw + (ox >> 1)) * a.Cout;
                    if (a.res_fmt) rv[it] = join_f16(*(const v2u *)(rrow + col16), *(const v2u *)(rrow + col16 + 4));
                    else rv[it] = *(const v4f *)(rrow + col);
                }
            }
#pragma unroll
            for (int it = 0; it < ITS; ++it) asm volatile("" : "+v"(rv[it]));
        }
        // The parameters must have ARRIVED before the store loop: gfx9 counts loads and stores in
        // one in-order counter (vmcnt), and a parameter still pending on any path makes the
        // compiler wait for vmcnt(0) in every iteration -- i.e. for the previous iteration's stores
        // to retire.  Using the values here puts the one wait in front of the loop.
        asm volatile("" : "+v"(mean), "+v"(sf), "+v"(beta), "+v"(bias));
        if constexpr (DBGT == 7) stamp[5] = wall_clock64();
        // Stores go through buffer resources like the loads: rows beyond M and columns beyond Cout
        // get an out-of-range offset and are dropped by the range check, so the loop is branch-free
        // straight-line code the scheduler can interleave across rows (the wave shares its SIMD with
        // a block that is issuing MFMAs; dependent address arithmetic per row cost 1.3 us a row).
        constexpr unsigned OOBS = 0x80000000u;
        const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(a.out + L.out_off), 0, (int)OOBS, 0x00020000);
        const __amdgpu_buffer_rsrc_t o2rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)((a.out2 ? a.out2 : a.out) + L.out_off), 0, (int)OOBS, 0x00020000);
        const unsigned off0 = (unsigned)(b0 * bstride + p0 * rstride + (a.out_fmt ? col16 : col)) * 4u;
        const bool vec_rows = ((a.Cout | rstride | bstride | (int)L.out_off) & 3) == 0;
        bool ovf = false;
        // One straight-line loop per epilogue form (uniform switch below): all LDS reads first,
        // then arithmetic and stores of independent rows for the scheduler to interleave.
        const float act_hi = a.act == 2 ? 6.0f : __builtin_inff();
        auto store_rows_act = [&](auto mode_tag, auto o16_tag, auto act_tag) __attribute__((always_inline)) {
            constexpr int MODE = decltype(mode_tag)::value;   // 0 plain, 1 +upsampled, 2 BN, 3 BN + relu(raw) copy, 4 bias
            constexpr bool O16 = decltype(o16_tag)::value;    // output rows in S16 form
            constexpr int ACT = decltype(act_tag)::value;     // 1: ReLU, then min with act_hi (6 or +inf); -1: a.act read per value
            v4f raw[ITS];
#pragma unroll
            for (int it = 0; it < ITS; ++it) raw[it] = *(const v4f *)(reg + (it * ROWS_PER_IT + row0) * RW + (c4 << 2));
            if constexpr (S16) {                              // weights were scaled by a power of two (exact)
#pragma unroll
                for (int it = 0; it < ITS; ++it)
#pragma unroll
                    for (int e = 0; e < 4; ++e) raw[it][e] = raw[it][e] * a.acc_scale;
            }
            unsigned off = off0;
            int p = p0;
#pragma unroll
            for (int it = 0; it < ITS; ++it) {
                const int m = mfirst + it * ROWS_PER_IT;
                if constexpr (DBGT == 7) { if (it == 1) stamp[6] = wall_clock64(); if (it == ITS / 2) stamp[7] = wall_clock64(); }
                v4f v = raw[it];
                if constexpr (MODE == 2 || MODE == 3) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float t = (v[e] - mean[e]) * sf[e];
                        v[e] = t + beta[e];
                    }
                }
                if constexpr (MODE == 4) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] + bias[e];
                }
                if constexpr (MODE == 1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = rv[it][e] + v[e];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if constexpr (ACT < 0) {
                        if (a.act >= 1) v[e] = v[e] > 0.0f ? v[e] : 0.0f;
                        if (a.act == 2) v[e] = v[e] < 6.0f ? v[e] : 6.0f;
                    } else {
                        v[e] = v[e] > 0.0f ? v[e] : 0.0f;
                        v[e] = v[e] < act_hi ? v[e] : act_hi;
                    }
                }
                const unsigned o = (m < M && colok) ? off : OOBS;
                if constexpr (MODE == 4 && !O16) {
                    // bias form = the class logits: first half of the post-processing's score filter, here where the logits
                    // are in registers (as igemm16.hip): mark the octet of 8 consecutive logits that holds a value at or above
                    // the conservative logit bound; post_scan_kernel then reads the bitmap and the marked octets only
                    if (a.scan_bits && o != OOBS) {
                        const float mx = __builtin_fmaxf(__builtin_fmaxf(v[0], v[1]), __builtin_fmaxf(v[2], v[3]));
                        if (mx >= a.scan_lo) {
                            const unsigned oct = ((unsigned)L.out_off + (o >> 2)) >> 3;      // octet index in [B][N][C]
                            atomicOr(a.scan_bits + (oct >> 5), 1u << (oct & 31));
                        }
                    }
                }
                if constexpr (O16) {
                    v2u hi, lo;
                    split_f16(v, hi, lo, ovf);
                    __builtin_amdgcn_raw_buffer_store_b64(hi, orsrc, (int)o, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b64(lo, orsrc, (int)(o == OOBS ? OOBS : o + 16u), 0, 0);
                } else if (vec_rows) {
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, v), orsrc, (int)o, 0, 0);
                } else {   // output rows that are not 16-B aligned (head widths 6*C with odd C): per-element stores
                    // (hipcc 7.2: bit-casting v[e] element by element inside the unrolled loop stored element 0
                    //  four times; cast the vector once and index the integer vector)
                    const v4u vu = __builtin_bit_cast(v4u, v);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        __builtin_amdgcn_raw_buffer_store_b32(vu[e], orsrc, (int)((col + e < a.Cout) ? o + 4u * e : OOBS), 0, 0);
                }
                if constexpr (MODE == 3) {
                    v4f q;
#pragma unroll
                    for (int e = 0; e < 4; ++e) q[e] = raw[it][e] > 0.0f ? raw[it][e] : 0.0f;
                    if constexpr (O16) {
                        v2u hi, lo;
                        split_f16(q, hi, lo, ovf);
                        __builtin_amdgcn_raw_buffer_store_b64(hi, o2rsrc, (int)o, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b64(lo, o2rsrc, (int)(o == OOBS ? OOBS : o + 16u), 0, 0);
                    } else {
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, q), o2rsrc, (int)o, 0, 0);
                    }
                }
                p += rR;
                const bool wrap = p >= P;
                p -= wrap ? P : 0;
                off += (unsigned)(step + (wrap ? wrapstep : 0));
            }
        };
        // a.act is uniform.  Batch-norm forms with an activation on the wide fp32 tiles (backbone, FPN outputs and towers at
        // serving batch sizes): ReLU, then min with 6 or +inf -- two vector instructions per value; reading a.act per value
        // costs two selects on top of each (5 of the 8 vector instructions per value).  The other forms, the S16 instances
        // (at their register limit) and the 64-wide tiles (five blocks per CU need <= 96 VGPRs) keep reading it.
        auto store_rows = [&](auto mode_tag, auto o16_tag) __attribute__((always_inline)) {
            constexpr int MODE = decltype(mode_tag)::value;
            if constexpr (!S16 && BM * BN >= 128 * 64 && (MODE == 2 || MODE == 3)) {
                if (a.act >= 1) store_rows_act(mode_tag, o16_tag, std::integral_constant<int, 1>{});
                else store_rows_act(mode_tag, o16_tag, std::integral_constant<int, -1>{});
            } else {
                store_rows_act(mode_tag, o16_tag, std::integral_constant<int, -1>{});
            }
        };
        // (host checks: bias and batch norm are exclusive, the upsampled operand comes without
        // either, the second output only with batch norm)
        using F = std::false_type;
        using T = std::true_type;
        if (a.out_fmt) {          // S16 rows (host: never together with a bias or unaligned widths)
            if (has_bn) {
                if (a.out2) store_rows(std::integral_constant<int, 3>{}, T{});
                else store_rows(std::integral_constant<int, 2>{}, T{});
            } else if (a.res) {
                store_rows(std::integral_constant<int, 1>{}, T{});
            } else {
                store_rows(std::integral_constant<int, 0>{}, T{});
            }
            // a value outside the fp16 range was clamped: tell the host (ssd_status)
            if (ovf && a.flags) atomicOr(a.flags, 1);
        } else if (has_bn) {
            bool fast = false;
            if constexpr (!S16 && BM * BN >= 128 * 64) fast = !a.out2 && a.act >= 1 && a.dense_out && vec_rows;
            if (fast) {
                if constexpr (!S16 && BM * BN >= 128 * 64) {
                    // The form every backbone / FPN-output / tower launch takes: batch norm + ReLU|ReLU6, fp32 rows, dense
                    // output ((image, position) -> row m of the level: no wrap arithmetic), 16-B aligned: 14 instructions per
                    // row group instead of ~30.  v_med3_f32(x, 0, hi) is the clamp in one instruction; a NaN comes out as 0,
                    // as it does from `x > 0 ? x : 0` (oracle act_apply; tests/test_gpu_stages.py::test_conv2d_nan_inf).
                    v4f raw[ITS];
#pragma unroll
                    for (int it = 0; it < ITS; ++it) raw[it] = *(const v4f *)(reg + (it * ROWS_PER_IT + row0) * RW + (c4 << 2));
                    const int mrem = colok ? M - mfirst : 0;                 // rows of this lane's column group that exist
                    unsigned off = (unsigned)(mfirst * rstride + col) * 4u;
                    const unsigned rstep = (unsigned)(ROWS_PER_IT * rstride) * 4u;
#pragma unroll
                    for (int it = 0; it < ITS; ++it) {
                        v4f v = raw[it];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float t = (v[e] - mean[e]) * sf[e];
                            v[e] = __builtin_amdgcn_fmed3f(t + beta[e], 0.0f, act_hi);
                        }
                        const unsigned o = it * ROWS_PER_IT < mrem ? off : OOBS;
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, v), orsrc, (int)o, 0, 0);
                        off += rstep;
                    }
                }
            } else if (a.out2) store_rows(std::integral_constant<int, 3>{}, F{});
            else store_rows(std::integral_constant<int, 2>{}, F{});
        } else if (a.bias) {
            store_rows(std::integral_constant<int, 4>{}, F{});
        } else if (a.res) {
            store_rows(std::integral_constant<int, 1>{}, F{});
        } else {
            store_rows(std::integral_constant<int, 0>{}, F{});
        }
    }
    if constexpr (DBGT == 7) {
        // stamp 4 is taken with the stores still in flight (a wave does not wait for them to retire)
        stamp[4] = wall_clock64();
        if (tid == 0 && a.ts) {
            unsigned hw;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            unsigned xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            long long *t = a.ts + (long long)blockIdx.x * 9;
            for (int i = 0; i < 8; ++i) t[i] = stamp[i];
            t[8] = ((long long)xcc << 32) | hw;
        }
    }
}

int igemm_tile_bm(int tile) { return tile == IGEMM_256x128 ? 256 : ((tile == IGEMM_64x64 || tile == IGEMM_64x64D || tile == 18) ? 64 : 128); }
int igemm_tile_bn(int tile)
{
    switch (tile) {
    case IGEMM_128x256: return 256;
    case IGEMM_128x128: case IGEMM_256x128: return 128;
    case IGEMM_128x64: case IGEMM_64x64: case IGEMM_64x64D: case 18: return 64;
    case IGEMM_128x96: return 96;
    case IGEMM_128x32: case 8: return 32;
    default: return 128;   // diagnostic variants of 128x128
    }
}

template <int WAVES_M, int WAVES_N, int WM, int WN, int TAPS, int DBG = 0, int S16 = 0>
static hipError_t launch_tt(const IgemmArgs &a, int total_tiles_m, hipStream_t s)
{
    constexpr int BM = WAVES_M * WM * 32, BN = WAVES_N * WN * 32;
    constexpr int lds_bytes = 2 * (BM + BN) * 128;
    static std::atomic<unsigned> attr_done{0};
    auto k = igemm_kernel<WAVES_M, WAVES_N, WM, WN, TAPS, DBG, S16>;
    {
        hipError_t e = ssd_allow_lds((const void *)k, lds_bytes, attr_done);
        if (e != hipSuccess) return e;
    }
    const long long nblk = (long long)total_tiles_m * a.n_tiles_n;
    if (nblk <= 0 || nblk > 0x7fffffffLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k, dim3((unsigned)nblk), dim3(64 * WAVES_M * WAVES_N), lds_bytes, s, a);
    return hipGetLastError();
}

template <int WAVES_M, int WAVES_N, int WM, int WN, int DBG = 0>
static hipError_t launch_t(const IgemmArgs &a, int total_tiles_m, hipStream_t s)
{
    if constexpr (DBG == 0) {
        if (a.in_fmt == 2) {     // fp32 rows split in the loader: the 1x1 convolutions under the FPN laterals
            if (a.taps != 1) return hipErrorInvalidValue;
            if constexpr ((WAVES_M == 2 && WAVES_N == 2 && WM == WN) ) return launch_tt<WAVES_M, WAVES_N, WM, WN, 1, 0, 2>(a, total_tiles_m, s);
            else return hipErrorInvalidValue;
        }
        if (a.in_fmt)
            return a.taps == 9 ? launch_tt<WAVES_M, WAVES_N, WM, WN, 9, 0, 1>(a, total_tiles_m, s)
                               : launch_tt<WAVES_M, WAVES_N, WM, WN, 1, 0, 1>(a, total_tiles_m, s);
    }
    return a.taps == 9 ? launch_tt<WAVES_M, WAVES_N, WM, WN, 9, DBG>(a, total_tiles_m, s)
                       : launch_tt<WAVES_M, WAVES_N, WM, WN, 1, DBG>(a, total_tiles_m, s);
}

hipError_t launch_igemm(int tile, const IgemmArgs &a, int total_tiles_m, hipStream_t s)
{
    // host-side shape checks: the kernel assumes them
    // widths that are not a multiple of 4 take per-element stores; the vector forms need aligned rows
    if (a.Cout % 4 != 0 && (a.mean || a.res || a.out2)) return hipErrorInvalidValue;
    if (a.Cin % 32 != 0 || a.CoutPad % igemm_tile_bn(tile) != 0 || (a.taps != 1 && a.taps != 9)) return hipErrorInvalidValue;
    // 32-bit byte offsets inside buffer resources: every tensor of a launch stays < 2 GiB
    if ((long long)a.taps * a.CoutPad * a.Cin * 4 >= (1LL << 31)) return hipErrorInvalidValue;
    for (int i = 0; i < a.nlevels; ++i)
        if ((long long)a.B * a.lv[i].H * a.lv[i].W * a.Cin * 4 >= (1LL << 31)) return hipErrorInvalidValue;
    // epilogue forms the kernel implements
    if ((a.mean != nullptr) != (a.sf != nullptr) || (a.mean != nullptr) != (a.beta != nullptr)) return hipErrorInvalidValue;
    if (a.mean && (a.bias || a.res)) return hipErrorInvalidValue;
    if (a.bias && a.res) return hipErrorInvalidValue;
    if (a.out2 && !a.mean) return hipErrorInvalidValue;
    // S16 rows are whole octets: no bias form, widths and strides multiples of 8 floats; diagnostics are fp32 only
    if (a.out_fmt && (a.bias || (a.Cout & 7) || !a.dense_out)) return hipErrorInvalidValue;
    if (a.res_fmt && !a.res) return hipErrorInvalidValue;
    if (a.in_fmt && tile >= 10) return hipErrorInvalidValue;
    for (int i = 0; i < a.nlevels; ++i)     // 32-bit byte offsets in the epilogue's buffer stores, relative to the level's base
        if ((long long)a.B * a.lv[i].out_bstride * 4 >= (1LL << 31) || a.lv[i].out_bstride < 0) return hipErrorInvalidValue;
    if (a.n_tiles_n * igemm_tile_bn(tile) != a.CoutPad || a.nlevels < 1 || a.nlevels > SSD_MAX_LEVELS) return hipErrorInvalidValue;
    switch (tile) {
    case IGEMM_128x128: return launch_t<2, 2, 2, 2>(a, total_tiles_m, s);
    case IGEMM_128x64: return launch_t<4, 1, 1, 2>(a, total_tiles_m, s);
    case IGEMM_128x32: return launch_t<4, 1, 1, 1>(a, total_tiles_m, s);
    case IGEMM_64x64: return launch_t<2, 2, 1, 1>(a, total_tiles_m, s);
    case IGEMM_64x64D: return a.in_fmt ? launch_t<2, 2, 1, 1>(a, total_tiles_m, s) : launch_t<2, 2, 1, 1, 8>(a, total_tiles_m, s);
    case IGEMM_128x96: return launch_t<4, 1, 1, 3>(a, total_tiles_m, s);
#ifdef SSD_DIAG   // ablation / phase-stamp instances exist only in libssd_hip_diag.so
    case 10: return launch_t<2, 2, 2, 2, 1>(a, total_tiles_m, s);
    case 11: return launch_t<2, 2, 2, 2, 2>(a, total_tiles_m, s);
    case 12: return launch_t<2, 2, 2, 2, 3>(a, total_tiles_m, s);
    case 17: return launch_t<2, 2, 2, 2, 7>(a, total_tiles_m, s);
    case 18: return launch_t<2, 2, 1, 1, 7>(a, total_tiles_m, s);      // phase stamps of the 64x64 tile
    case 8: return launch_t<4, 1, 1, 1, 8>(a, total_tiles_m, s);       // 128x32 with the two register sets (loads three K-steps ahead)
#endif
    }
    return hipErrorInvalidValue;
}
