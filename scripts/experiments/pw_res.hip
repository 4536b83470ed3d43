// EXPERIMENT, round 5 -- NOT part of the library (measured slower, profiles/r05_pw_res_experiment.log; log entry in README.md).
// To re-run: copy into single-shot-detector_amd/csrc/, add the file to _lib.py's _SOURCES, declare PwRArgs / pw_res_supports /
// launch_pw_res in ssd_internal.h (struct below) and route make_conv_op's dense single-level 1x1 case to launch_pw_res.
//   struct PwRArgs { const float *in, *wt, *mean, *sf, *beta; float *out; int out_rs, out_bytes, M, K, Cout, wt_rows, act, m_tiles, n_tiles; };
//
// 1x1 convolution (+ batch norm + activation) with the block's weight tile RESIDENT in LDS: MobileNet's pointwise layers and
// the FPN laterals at a serving batch (mobilenet_v1.py:59-67, feature_extractor.py:57,66), K <= 512.
//
//   block       64 output channels of the layer, all K of them in LDS for the whole launch (K x 64 x 4 B: 128 KB at K = 512;
//               one block per CU), as K / 32 slice images of dwpw_stream.hip's B layout
//   work        the block walks 64-row tiles of the [M][K] input q, q + Q, ...; only the positions are streamed: per
//               K-step one 8 KB A image (LDS-DMA, 16 bytes per lane, two wave instructions per wave) into a ring of three,
//               two K-steps ahead; one s_barrier per K-step, counted vmcnt, the stream continues across tiles (no
//               per-tile prologue)
//   product     v_mfma_f32_32x32x2_f32, weights as the A operand, positions as the B operand (dwpw_stream.hip's transposed
//               product): k ascends in physical order = the chain of igemm.hip, bit for bit; a lane holds 4 consecutive
//               channels of one position -> batch norm + activation -> 16-byte stores
// Per K-step and wave: 2 LDS-DMA instructions, 8 ds_read_b128, 16 MFMAs -- against 8 global loads, 8 ds_write_b128,
// 16 ds_read_b128 and 64 MFMAs (x 1/4) of the 128x128 tile of igemm.hip, whose staging instructions take issue slots from the
// exact-fp32 matrix pipe.
#include "ssd_internal.h"

typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void *lds_ptr_t;

#define WAIT_VM_CASE(n) case n: asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); break;
static __device__ __forceinline__ void wait_vmcnt_pr(int n)
{
    switch (n) {
        WAIT_VM_CASE(0) WAIT_VM_CASE(1) WAIT_VM_CASE(2) WAIT_VM_CASE(3) WAIT_VM_CASE(4) WAIT_VM_CASE(5) WAIT_VM_CASE(6) WAIT_VM_CASE(7)
        WAIT_VM_CASE(8) WAIT_VM_CASE(9) WAIT_VM_CASE(10)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

__global__ __launch_bounds__(256, 1) void pw_res_kernel(const PwRArgs a)
{
    constexpr int BM = 64, BN = 64, NST = 3;
    constexpr int A_BYTES = BM * 128, B_SLICE = BN * 128, P_BYTES = BN * 12;
    constexpr int OFF_P = NST * A_BYTES, OFF_B = OFF_P + 1024;
    constexpr int NSTORE = 4;
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];      // the ONE shared array of this kernel

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave >> 1, wave_n = wave & 1;
    // blocks b, b + 8, ... share an XCD: there, consecutive blocks take the n-tiles of one m-tile sequence (they stream the
    // same rows through that L2)
    const int xcd = blockIdx.x & 7, kk = blockIdx.x >> 3;
    const int tile_n = kk % a.n_tiles;
    const int Q = (int)gridDim.x / a.n_tiles;
    const int q = (kk / a.n_tiles) * 8 + xcd;
    if (q >= a.m_tiles) return;                       // whole block, before any barrier
    const int my_tiles = (a.m_tiles - q + Q - 1) / Q;
    const int K = a.K, KC = K >> 5;
    const int T = my_tiles * KC;

    const __amdgpu_buffer_rsrc_t irsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.in, 0, (int)((long long)a.M * K * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t brsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.wt, 0, (int)((long long)a.wt_rows * K * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.out, 0, a.out_bytes, 0x00020000);

    // ---- epilogue parameters -> LDS
    if (a.mean) {
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

    // ---- the weight tile: slice s -> image s, rows n = (wave * 2 + k) * 8 + (lane >> 3), LDS slot lane & 7 <- source chunk slot ^ ((n >> 1) & 7)
    {
        int boff[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int n = (wave * 2 + k) * 8 + (lane >> 3);
            const int chunk = (lane & 7) ^ ((n >> 1) & 7);
            boff[k] = ((tile_n * BN + n) * K + chunk * 4) * 4;
        }
        for (int s = 0; s < KC; ++s) {
            unsigned char *dst = lds + OFF_B + s * B_SLICE + wave * 2048;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int o = boff[k];
                __builtin_amdgcn_raw_ptr_buffer_load_lds(brsrc, (lds_ptr_t)(dst + k * 1024), 16, o, s * 128, 0, 0);
            }
        }
    }
    // ---- position stream: instruction i = wave, wave + 4 covers rows 8i .. 8i + 7 of a tile; lane -> row 8i + (lane >> 3),
    // LDS slot lane & 7 <- source chunk slot ^ key, key = ((row >> 1) & 7) = (4 wave + (lane >> 4)) & 7 for both of the wave's instructions
    const int akey = (4 * wave + (lane >> 4)) & 7;
    const int avoff = (((lane >> 3) + 8 * wave) * K + (((lane & 7) ^ akey) << 2)) * 4;      // row 8 wave + (lane >> 3) of tile 0, slice 0
    const int arow32 = 32 * K * 4;                                                          // instruction wave + 4: 32 rows further
    auto dma_a = [&](int slot, int tile, int s) {
        unsigned char *dst = lds + slot * A_BYTES + wave * 1024;
        const int base = tile * (BM * K * 4) + s * 128;          // scalar; added to the lane offset so that the range check sees rows past M
        const int o0 = avoff + base, o1 = avoff + arow32 + base;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(irsrc, (lds_ptr_t)dst, 16, o0, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(irsrc, (lds_ptr_t)(dst + 4096), 16, o1, 0, 0, 0);
    };

    int roff[4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
        roff[g] = (lane & 31) * 128 + (((2 * g + (lane >> 5)) ^ (((lane & 31) >> 1) & 7)) << 4);
    v16f acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    const int eh = lane >> 5;
    const bool has_bn = a.mean != nullptr, pact_on = a.act >= 1;
    const float pact_hi = a.act == 2 ? 6.0f : __builtin_inff();

    // cursors: (tp, sp) two iterations ahead (the loads), (te, se) the iteration being computed
    int tp = q, sp = 0, te = q, se = 0;
    auto advance = [&](int &t, int &s) { if (++s == KC) { s = 0; t += Q; } };
    dma_a(0, tp, sp); advance(tp, sp);
    if (T > 1) { dma_a(1, tp, sp); advance(tp, sp); }
    int slot = 0, slot_n = 2;                        // ring slot of iteration `it`, slot of the loads issued in it (it + 2)
    bool last1 = false, last2 = false;               // iteration it - 1 / it - 2 ended a tile (its stores are in flight)
    for (int it = 0; it < T; ++it) {
        // A(it) (and, at it = 0, the weight tile) has landed once everything older than A(it+1) and the stores behind it is done
        wait_vmcnt_pr((it + 1 < T ? 2 : 0) + (last1 ? NSTORE : 0) + ((last2 && it + 1 < T) ? NSTORE : 0));
        __builtin_amdgcn_s_barrier();                                   // every wave's share has landed; MFMA(it-1) is over
        __builtin_amdgcn_sched_barrier(0);
        if (it + 2 < T) { dma_a(slot_n, tp, sp); advance(tp, sp); }
        __builtin_amdgcn_sched_barrier(0);
        {
            const unsigned char *ab = lds + slot * A_BYTES + wave_m * 4096;
            const unsigned char *bb = lds + OFF_B + se * B_SLICE + wave_n * 4096;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const v4f af = *(const v4f *)(ab + roff[g]);
                const v4f bf = *(const v4f *)(bb + roff[g]);
#pragma unroll
                for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[t], af[t], acc, 0, 0, 0);
            }
        }
        last2 = last1;
        last1 = se == KC - 1;
        if (last1) {
            // ---- epilogue.  acc[r]: channel wave_n * 32 + (r & 3) + 8 (r >> 2) + 4 (lane >> 5) of the block's 64, position = tile row
            // wave_m * 32 + (lane & 31)
            const int pos = te * BM + wave_m * 32 + (lane & 31);
            const bool ok = pos < a.M;
            const float *pp = (const float *)(lds + OFF_P);
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int cl = wave_n * 32 + 8 * m + 4 * eh;
                v4f v;
                if (has_bn) {
                    const v4f mean = *(const v4f *)(pp + cl), sf = *(const v4f *)(pp + BN + cl), beta = *(const v4f *)(pp + 2 * BN + cl);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float t = (acc[4 * m + e] - mean[e]) * sf[e];
                        v[e] = t + beta[e];
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[4 * m + e];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (pact_on) v[e] = __builtin_amdgcn_fmed3f(v[e], 0.0f, pact_hi);
                    acc[4 * m + e] = 0.0f;
                }
                const int n = tile_n * BN + cl;
                const unsigned o = (ok && n < a.Cout) ? (unsigned)((pos * a.out_rs + n) * 4) : OOB;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, v), orsrc, (int)o, 0, 0);
            }
        }
        advance(te, se);
        slot = slot == NST - 1 ? 0 : slot + 1;
        slot_n = slot_n == NST - 1 ? 0 : slot_n + 1;
    }
}

bool pw_res_supports(int K, int CoutP, long long M)
{
    if (K < 32 || K % 32 || K > 512 || CoutP < 4 || CoutP % 4 || M < 1) return false;
    return M * (long long)K * 4 < (1LL << 31) - (64LL * 512 * 4 + 4096);      // lane offsets of the ragged last tile stay positive
}

hipError_t launch_pw_res(const PwRArgs &a, hipStream_t s)
{
    if (!a.in || !a.wt || !a.out || (a.mean && (!a.sf || !a.beta))) return hipErrorInvalidValue;
    if (!pw_res_supports(a.K, a.Cout, a.M)) return hipErrorInvalidValue;
    if (a.wt_rows < a.Cout || a.n_tiles != (a.Cout + 63) / 64 || a.m_tiles != (a.M + 63) / 64 || a.n_tiles > 32) return hipErrorInvalidValue;
    if ((long long)a.wt_rows * a.K * 4 >= (1LL << 31) || a.out_rs < a.Cout || (long long)a.M * a.out_rs * 4 >= (1LL << 31) ||
        ((long long)a.M - 1) * a.out_rs * 4 + (long long)a.Cout * 4 > (long long)a.out_bytes) return hipErrorInvalidValue;
    const int lds_bytes = 3 * 64 * 128 + 1024 + (a.K / 32) * 64 * 128;
    static std::atomic<unsigned> attr_done{0};
    {
        hipError_t e = ssd_allow_lds((const void *)pw_res_kernel, 3 * 64 * 128 + 1024 + 16 * 64 * 128, attr_done);
        if (e != hipSuccess) return e;
    }
    // one resident block per CU; m-tile sequences in multiples of 8 (one per XCD)
    int Q = 256 / a.n_tiles / 8 * 8;
    const int need = (a.m_tiles + 7) / 8 * 8;
    if (Q > need) Q = need;
    if (Q < 8) Q = 8;
    hipLaunchKernelGGL(pw_res_kernel, dim3((unsigned)(Q * a.n_tiles)), dim3(256), lds_bytes, s, a);
    return hipGetLastError();
}
