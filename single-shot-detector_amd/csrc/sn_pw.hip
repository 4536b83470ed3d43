// ShuffleNet v2's conv1x1_before (shufflenet_v2.py:118-121) with concat_shuffle_split (:94-115) folded into its LOADS:
// a 1x1 convolution + batch norm + activation whose input row is GATHERED -- input channel k of a position comes from
// column src[k] of one of several dense producer tensors (unit_1's two branches, the conv1x1_after outputs of earlier
// units), all rows of one allocation with one row stride.  Producers store their channels dense (16-byte stores, every
// line written whole by one launch); the interleave-and-split of the reference is a static per-channel source table of
// the consumer.  The k order of the accumulation stays the logical one: bit-identical to gather -> igemm.
//
//   work item   64 consecutive rows (positions) x BN output channels; a block walks tiles q, q + Q, ...; K is streamed in
//               32-channel slices: iteration = (tile, slice)
//   input       per slice the A image of dwpw_stream.hip (64 rows x 128 B, 16-byte chunk c of row r at slot
//               c ^ ((r >> 1) & 7)), filled by LDS-DMA with FOUR-byte elements (buffer_load_dword ... lds): one wave
//               instruction = two rows of the image; lane -> (row of the pair, dword slot), and the lane's source address
//               is the row offset (scalar) + src[the channel its slot holds under the pair's swizzle key].  Wave w issues the
//               pairs i = w, w + 4, ...: their keys are w and w + 4, so a lane needs two table entries per slice (read from
//               the LDS copy of the table while the previous slice's loads land).  Zero channels (the pad channels of K,
//               src = -1) and rows past the allocation are the buffer range check: the lane delivers 0.
//   weights     the slice of the [rows][K] kernel as dwpw_stream.hip's B image (LDS-DMA, 16 bytes per lane)
//   pipeline    A(it+1), B(it+1) are issued behind the barrier of iteration `it` and land under its MFMAs; one
//               s_barrier per iteration, counted vmcnt (the previous tile's stores stay in flight)
//   product     v_mfma_f32_32x32x2_f32, weights as the A operand, positions as the B operand (the transposed product of
//               dwpw_stream.hip): a lane holds 4 consecutive channels of one position -> batch norm + activation ->
//               16-byte stores into a dense [M][out_rs] tensor
#include "ssd_internal.h"

typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void *lds_ptr_t;

#define WAIT_VM_CASE(n) case n: asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); break;
static __device__ __forceinline__ void wait_vmcnt_pw(int n)
{
    switch (n) {
        WAIT_VM_CASE(0) WAIT_VM_CASE(1) WAIT_VM_CASE(2) WAIT_VM_CASE(3) WAIT_VM_CASE(4) WAIT_VM_CASE(5) WAIT_VM_CASE(6) WAIT_VM_CASE(7)
        WAIT_VM_CASE(8)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

template <int WN>
__global__ __launch_bounds__(256, WN == 1 ? 4 : 3) void pw_gather_kernel(const PwGArgs a)
{
    constexpr int BM = 64, WAVES_N = 2, BN = WAVES_N * WN * 32;
    constexpr int NBW = BN / 32;                      // B-slice DMA instructions per wave
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, P_BYTES = BN * 12;
    constexpr int OFF_B = 2 * A_BYTES, OFF_P = OFF_B + 2 * B_BYTES, OFF_S = OFF_P + P_BYTES;
    constexpr int NSTORE = WN * 4;                    // epilogue stores per wave and tile
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];      // the ONE shared array of this kernel

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave / WAVES_N, wave_n = wave % WAVES_N;
    // blocks b, b + 8, ... share an XCD: there, consecutive blocks take the n-tiles of one m-tile sequence
    const int xcd = blockIdx.x & 7, kk = blockIdx.x >> 3;
    const int tile_n = kk % a.n_tiles;
    const int Q = (int)gridDim.x / a.n_tiles;
    const int q = (kk / a.n_tiles) * 8 + xcd;
    if (q >= a.m_tiles) return;                       // whole block, before any barrier
    const int my_tiles = (a.m_tiles - q + Q - 1) / Q;
    const int K = a.K, KC = K >> 5;
    const int T = my_tiles * KC;

    const __amdgpu_buffer_rsrc_t irsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.base, 0, a.base_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t brsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.wt, 0, (int)((long long)a.wt_rows * K * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.out, 0, a.out_bytes, 0x00020000);

    // ---- source table and epilogue parameters -> LDS (ordinary loads, retired before the DMA pipeline starts)
    {
        int *sl = (int *)(lds + OFF_S);
        for (int k = tid; k < K; k += 256) sl[k] = a.src[k];
        float *pp = (float *)(lds + OFF_P);
        for (int c = tid; c < BN; c += 256) {
            const int n = tile_n * BN + c;
            const bool ok = n < a.Cout;
            pp[c] = ok ? a.mean[n] : 0.0f;
            pp[BN + c] = ok ? a.sf[n] : 0.0f;
            pp[2 * BN + c] = ok ? a.beta[n] : 0.0f;
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // ---- gather bookkeeping.  Pair instruction i = wave + 4 m (m = 0 .. 7) fills rows 2i, 2i + 1 of the image; its
    // swizzle key is i & 7 = wave + 4 (m & 1).  Lane -> row 2i + (lane >> 5), dword slot d = lane & 31 = chunk slot d >> 2,
    // element d & 3; the slot holds chunk (d >> 2) ^ key of the slice.
    const int d = lane & 31;
    const int swz0 = ((((d >> 2) ^ wave) << 2) | (d & 3)) * 4, swz1 = ((((d >> 2) ^ (wave + 4)) << 2) | (d & 3)) * 4;   // byte index into a slice of the table
    const int vrow = (lane >> 5) * a.rs;
    const unsigned char *sl = lds + OFF_S;
    // B slice: rows n = (wave * NBW + k) * 8 + (lane >> 3) of the block's BN, LDS slot lane & 7 <- source chunk slot ^ ((n >> 1) & 7)
    int boff[NBW];
#pragma unroll
    for (int k = 0; k < NBW; ++k) {
        const int n = (wave * NBW + k) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((n >> 1) & 7);
        boff[k] = ((tile_n * BN + n) * K + chunk * 4) * 4;
    }
    auto dma_a = [&](int iter, int tile, int g0, int g1) {
        unsigned char *dst = lds + (iter & 1) * A_BYTES + wave * 256;
        const int r0 = tile * BM + 2 * wave;                     // scalar: first row of pair `wave`; pairs i and i + 4 are 8 rows apart
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            // (the scalar offset takes no part in the buffer's range check: rows past the tensor's last one -- the ragged last
            //  tile, whose rows are never stored -- re-read row M - 1 instead of leaving the tensor; a pair whose FIRST row is
            //  the last one (odd M) reads it for both of its rows)
            const int row = r0 + 8 * m;
            const int so = (row < a.M - 1 ? row : a.M - 1) * a.rs;
            const int vr = row + 1 < a.M ? vrow : 0;
            const int g = (m & 1) ? g1 : g0;
            const int off = g < 0 ? (int)OOB : g + vr;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(irsrc, (lds_ptr_t)(dst + m * 1024), 4, off, so, 0, 0);
        }
    };
    auto dma_b = [&](int iter, int s) {
        unsigned char *dst = lds + OFF_B + (iter & 1) * B_BYTES + wave * NBW * 1024;
#pragma unroll
        for (int k = 0; k < NBW; ++k) {
            const int o = boff[k];
            __builtin_amdgcn_raw_ptr_buffer_load_lds(brsrc, (lds_ptr_t)(dst + k * 1024), 16, o, s * 128, 0, 0);
        }
    };

    int roff[4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
        roff[g] = (lane & 31) * 128 + (((2 * g + (lane >> 5)) ^ (((lane & 31) >> 1) & 7)) << 4);
    v16f acc[WN];
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
    const int eh = lane >> 5;
    const bool pact_on = a.act >= 1;
    const float pact_hi = a.act == 2 ? 6.0f : __builtin_inff();

    // cursors: `tp`, `sp` run one iteration ahead (the loads), `te`, `se` with the iteration being computed
    int tp = q, sp = 0, te = q, se = 0;
    auto advance = [&](int &t, int &s) { if (++s == KC) { s = 0; t += Q; } };
    {
        const int g0 = *(const int *)(sl + swz0), g1 = *(const int *)(sl + swz1);
        dma_a(0, tp, g0, g1);
        dma_b(0, 0);
        advance(tp, sp);
    }
    bool prev_last = false;
    for (int it = 0; it < T; ++it) {
        // the next slice's source offsets: read while this iteration's loads land
        int g0 = -1, g1 = -1;
        if (it + 1 < T) { g0 = *(const int *)(sl + sp * 128 + swz0); g1 = *(const int *)(sl + sp * 128 + swz1); }
        wait_vmcnt_pw(prev_last ? NSTORE : 0);                          // A(it), B(it): everything older than the previous tile's stores
        __builtin_amdgcn_s_barrier();                                   // every wave's share has landed; MFMA(it-1) is over
        __builtin_amdgcn_sched_barrier(0);
        if (it + 1 < T) {
            dma_a(it + 1, tp, g0, g1);
            dma_b(it + 1, sp);
            advance(tp, sp);
        }
        __builtin_amdgcn_sched_barrier(0);
        {   // ---- acc[channel][position] += W slice (BN x 32) * A^T (32 x BM)
            const unsigned char *ab = lds + (it & 1) * A_BYTES + wave_m * 4096;
            const unsigned char *bb = lds + OFF_B + (it & 1) * B_BYTES + wave_n * WN * 4096;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const v4f af = *(const v4f *)(ab + roff[g]);
                v4f bf[WN];
#pragma unroll
                for (int j = 0; j < WN; ++j) bf[j] = *(const v4f *)(bb + j * 4096 + roff[g]);
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int j = 0; j < WN; ++j)
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[j][t], af[t], acc[j], 0, 0, 0);
            }
        }
        prev_last = se == KC - 1;
        if (prev_last) {
            // ---- epilogue.  acc[j][r]: channel (wave_n * WN + j) * 32 + (r & 3) + 8 (r >> 2) + 4 (lane >> 5) of the block's
            // BN, position = tile row wave_m * 32 + (lane & 31)
            const int pos = te * BM + wave_m * 32 + (lane & 31);
            const bool ok = pos < a.M;
            const float *pp = (const float *)(lds + OFF_P);
#pragma unroll
            for (int j = 0; j < WN; ++j) {
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const int cl = (wave_n * WN + j) * 32 + 8 * m + 4 * eh;
                    const v4f mean = *(const v4f *)(pp + cl), sf = *(const v4f *)(pp + BN + cl), beta = *(const v4f *)(pp + 2 * BN + cl);
                    v4f v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float x = acc[j][4 * m + e];
                        const float t = (x - mean[e]) * sf[e];
                        x = t + beta[e];
                        if (pact_on) x = __builtin_amdgcn_fmed3f(x, 0.0f, pact_hi);
                        v[e] = x;
                        acc[j][4 * m + e] = 0.0f;
                    }
                    const int n = tile_n * BN + cl;
                    const unsigned o = (ok && n < a.Cout) ? (unsigned)((pos * a.out_rs + n) * 4) : OOB;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, v), orsrc, (int)o, 0, 0);
                }
            }
        }
        advance(te, se);
    }
}

int pw_gather_tile_n(int CoutP) { return CoutP <= 64 ? 64 : 128; }

template <int WN>
static hipError_t launch_g(const PwGArgs &a, hipStream_t s)
{
    constexpr int BN = 2 * WN * 32;
    const int lds_bytes = 2 * 64 * 128 + 2 * BN * 128 + BN * 12 + a.K * 4;
    constexpr int PER_CU = WN == 1 ? 4 : 3;
    static std::atomic<unsigned> attr_done{0};
    auto k = pw_gather_kernel<WN>;
    {
        hipError_t e = ssd_allow_lds((const void *)k, 2 * 64 * 128 + 2 * BN * 128 + BN * 12 + 512 * 4, attr_done);
        if (e != hipSuccess) return e;
    }
    int Q = 256 * PER_CU / a.n_tiles / 8 * 8;
    const int need = (a.m_tiles + 7) / 8 * 8;
    if (Q > need) Q = need;
    if (Q < 8) Q = 8;
    hipLaunchKernelGGL(k, dim3((unsigned)(Q * a.n_tiles)), dim3(256), lds_bytes, s, a);
    return hipGetLastError();
}

bool pw_gather_supports(int K, int CoutP, long long M, int rs, long long base_bytes, long long out_bytes)
{
    if (K < 32 || K % 32 || K > 512 || CoutP < 4 || CoutP % 4 || M < 1 || rs < 4 || (rs & 3)) return false;
    if (base_bytes <= 0 || base_bytes >= (1LL << 31) || out_bytes <= 0 || out_bytes >= (1LL << 31)) return false;
    const int BN = pw_gather_tile_n(CoutP);
    if ((CoutP + BN - 1) / BN > 32) return false;
    return M * (long long)rs < (1LL << 31);          // scalar row offsets
}

hipError_t launch_pw_gather(const PwGArgs &a, hipStream_t s)
{
    // host-side checks of everything the kernel assumes
    if (!a.base || !a.src || !a.wt || !a.mean || !a.sf || !a.beta || !a.out) return hipErrorInvalidValue;
    if (!pw_gather_supports(a.K, a.Cout, a.M, a.rs, a.base_bytes, a.out_bytes)) return hipErrorInvalidValue;
    const int BN = pw_gather_tile_n(a.Cout);
    if (a.wt_rows < a.Cout || a.n_tiles != (a.Cout + BN - 1) / BN || a.m_tiles != (a.M + 63) / 64) return hipErrorInvalidValue;
    if ((long long)a.wt_rows * a.K * 4 >= (1LL << 31) || a.out_rs < a.Cout || (long long)a.M * a.out_rs * 4 > (long long)a.out_bytes) return hipErrorInvalidValue;
    return BN == 64 ? launch_g<1>(a, s) : launch_g<2>(a, s);
}
