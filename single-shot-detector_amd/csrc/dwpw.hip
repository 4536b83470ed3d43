// Depthwise 3x3 (+ batch norm + activation) and the pointwise 1x1 convolution that consumes it
// (+ batch norm + activation) in ONE kernel: the MobileNet block of mobilenet_v1.py:59-67 and the
// depthwise -> conv1x1_after pairs of shufflenet_v2.py:118-137.  The depthwise result never goes
// to memory: a block computes it for its BM output positions x all K channels straight into the
// LDS image the MFMA loop reads as its A operand, then streams the 1x1 weights (B operand)
// through two LDS stages exactly like igemm.hip.
//
//   numerics  the depthwise part is the code of depthwise_kernel (elementwise.hip): per output
//             one (ky,kx)-ordered fmaf chain, batch norm as (x-mean)*sf+beta separately rounded;
//             the 1x1 part is the k-ordered accumulation of igemm.hip.  Bit-identical to running
//             the two kernels one after the other, and to the oracle.
//   traffic   reads the depthwise input once (3-row halo through L2), writes the 1x1 output:
//             the depthwise output (write + read of B*OH*OW*K floats) is gone.
//   limits    K <= 256 (A operand resident in LDS: BM*K*4 bytes), OW and OH*OW multiples of 4
//             (a thread produces 4 adjacent positions of one image row); host-checked.
#include "ssd_internal.h"
#include <cstdlib>

typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

template <int STRIDE, int WAVES_M, int WAVES_N, int WM, int WN, bool BATCH>
__global__ __launch_bounds__(256, 2) void dwpw_kernel(const DwPwArgs a)
{
    static_assert(WAVES_M * WAVES_N == 4, "256 threads");
    constexpr int BM = WAVES_M * WM * 32;
    constexpr int BN = WAVES_N * WN * 32;
    constexpr int NB = BN / 32;                  // 16-B loads per thread per K-step, B tile
    constexpr int NCOL = 3 * STRIDE + 3;         // input columns under 4 adjacent outputs
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_m = wave / WAVES_N, wave_n = wave % WAVES_N;
    long long stamp[5];
    auto mark = [&](int i) { if (a.ts) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); stamp[i] = wall_clock64(); } };
    if (a.ts) stamp[0] = wall_clock64();
    int swz;
    {   // blocks b, b+8, ... share an XCD: consecutive tiles per XCD (igemm.hip)
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
        swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tile_n = swz % a.n_tiles_n, tile_m = swz / a.n_tiles_n;
    const int m0 = tile_m * BM;
    const int K = a.K, KC = K >> 5, H = a.H, W = a.W, OW = a.OW, M = a.M;
    const int P = a.OH * a.OW;
    const int A_BYTES = KC * BM * 128;
    unsigned char *ldsB = lds + A_BYTES;

    const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void *)a.in, 0, (int)((long long)a.B * H * W * K * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t brsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void *)a.wt, 0, (int)((long long)a.CoutPad * K * 4), 0x00020000);

    // ---- 1x1 weights of K-step 0 on their way while the depthwise part runs
    const int bvoff = ((tile_n * BN + (tid >> 3)) * K + (tid & 7) * 4) * 4;
    const int b_ustride = 32 * K * 4;
    const int woff = (tid >> 3) * 128 + (((tid & 7) ^ ((tid >> 4) & 7)) << 4);
    v4f rb[NB];
    auto loadB = [&](int kc) {
#pragma unroll
        for (int u = 0; u < NB; ++u)
            rb[u] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(brsrc, bvoff + u * b_ustride, kc * 128, 0));
    };
    auto storeB = [&](int stage) {
#pragma unroll
        for (int u = 0; u < NB; ++u) *(v4f *)(ldsB + stage * (BN * 128) + woff + u * 4096) = rb[u];
    };
    loadB(0);

    // ---- depthwise 3x3 + BN + act for BM positions x K channels -> LDS A image.
    // item = (strip of 4 adjacent output positions, 4 channels); lanes run along channels.
    {
        // a thread keeps ONE channel chunk (weights and batch norm loaded once) and walks the strips
        const int K4 = K >> 2, T = 256 / K4;
        const int slot = tid / K4, cc = tid - slot * K4;
        const int c = cc * 4;
        v4f wv[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) wv[t] = *(const v4f *)(a.wdw + t * K + c);
        const v4f dm = *(const v4f *)(a.dmean + c), ds = *(const v4f *)(a.dsf + c), db = *(const v4f *)(a.dbeta + c);
        for (int strip = slot < T ? slot : BM; strip < BM / 4; strip += T) {
            const int m = m0 + strip * 4;
            const bool rowok = m < M;
            const int mm = rowok ? m : 0;
            const int b = mm / P, p = mm - b * P;
            const int oy = p / OW, ox0 = p - oy * OW;
            // every load of the item is issued before the first use (one memory round trip per item,
            // not one per tap row): the scheduling barrier keeps the compiler from re-serialising them
            const int ix0 = ox0 * STRIDE - a.pad;
            v4f x[3][NCOL];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int iy = oy * STRIDE + ky - a.pad;
                const bool yok = rowok && (unsigned)iy < (unsigned)H;
                const int rowbase = (((b * H + iy) * W) * K + c) * 4;
#pragma unroll
                for (int j = 0; j < NCOL; ++j) {
                    const int ix = ix0 + j;
                    const bool ok = yok && (unsigned)ix < (unsigned)W;
                    // out-of-image taps: range-checked buffer load returns 0, fmaf(0, w, acc) == acc
                    x[ky][j] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(arsrc, ok ? rowbase + ix * K * 4 : (int)OOB, 0, 0));
                }
            }
            // BATCH: one memory round trip per item at the price of registers (2 waves per SIMD);
            // otherwise the compiler interleaves loads and FMAs row by row (4 waves per SIMD).
            if constexpr (BATCH) __builtin_amdgcn_sched_barrier(0);
            v4f acc[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = (v4f){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[q][i] = fmaf(x[ky][q * STRIDE + kx][i], wv[ky * 3 + kx][i], acc[q][i]);
            const int kc = cc >> 3, ch8 = cc & 7;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                v4f v = acc[q];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float t = (v[i] - dm[i]) * ds[i];
                    v[i] = t + db[i];
                    if (a.dact >= 1) v[i] = v[i] > 0.0f ? v[i] : 0.0f;
                    if (a.dact == 2) v[i] = v[i] < 6.0f ? v[i] : 6.0f;
                }
                const int r = strip * 4 + q;
                *(v4f *)(lds + kc * (BM * 128) + r * 128 + ((ch8 ^ ((r >> 1) & 7)) << 4)) = v;
            }
        }
    }

    // ---- 1x1 convolution: A resident, B double-buffered
    int roff[4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
        roff[g] = (lane & 31) * 128 + (((2 * g + (lane >> 5)) ^ (((lane & 31) >> 1) & 7)) << 4);
    v16f acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    mark(1);
    storeB(0);
    if (KC > 1) loadB(1);
    __syncthreads();
    mark(2);
    for (int s = 0; s < KC; ++s) {
        const int cur = s & 1;
        if (s + 1 < KC) storeB(cur ^ 1);          // rb holds K-step s+1; stage cur^1 was released by the last barrier
        if (s + 2 < KC) loadB(s + 2);
        const unsigned char *ab = lds + s * (BM * 128) + wave_m * WM * 4096;
        const unsigned char *bb = ldsB + cur * (BN * 128) + wave_n * WN * 4096;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            v4f af[WM], bf[WN];
#pragma unroll
            for (int i = 0; i < WM; ++i) af[i] = *(const v4f *)(ab + i * 4096 + roff[g]);
#pragma unroll
            for (int j = 0; j < WN; ++j) bf[j] = *(const v4f *)(bb + j * 4096 + roff[g]);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][t], bf[j][t], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    mark(3);
    // ---- epilogue (igemm.hip): per-wave LDS transpose, BN + act, 16-B buffer stores
    constexpr int RW = WN * 32, C4N = RW / 4;
    constexpr int WAVE_REGION = WM * 32 * RW * 4;
    constexpr int ROWS_PER_IT = 64 / C4N;
    constexpr int ITS = WM * 32 / ROWS_PER_IT;
    static_assert(64 % C4N == 0, "wave sub-tile width");
    float *reg = (float *)(lds + wave * WAVE_REGION);
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                reg[row * RW + j * 32 + (lane & 31)] = acc[i][j][r];
            }
    __syncthreads();
    const int c4 = lane % C4N;
    const int col = tile_n * BN + wave_n * RW + c4 * 4;
    const bool colok = col < a.Cout;
    v4f mean = {0.f, 0.f, 0.f, 0.f}, sf = {1.f, 1.f, 1.f, 1.f}, beta = {0.f, 0.f, 0.f, 0.f};
    if (colok) {
        mean = *(const v4f *)(a.mean + col);
        sf = *(const v4f *)(a.sf + col);
        beta = *(const v4f *)(a.beta + col);
    }
    asm volatile("" : "+v"(mean), "+v"(sf), "+v"(beta));    // one wait in front of the store loop (igemm.hip)
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.out, 0, (int)OOB, 0x00020000);
    const int row0 = lane / C4N;
    const int mfirst = m0 + wave_m * WM * 32 + row0;
    v4f raw[ITS];
#pragma unroll
    for (int it = 0; it < ITS; ++it) raw[it] = *(const v4f *)(reg + (it * ROWS_PER_IT + row0) * RW + (c4 << 2));
#pragma unroll
    for (int it = 0; it < ITS; ++it) {
        const int m = mfirst + it * ROWS_PER_IT;
        v4f v = raw[it];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float t = (v[e] - mean[e]) * sf[e];
            v[e] = t + beta[e];
            if (a.act >= 1) v[e] = v[e] > 0.0f ? v[e] : 0.0f;
            if (a.act == 2) v[e] = v[e] < 6.0f ? v[e] : 6.0f;
        }
        const unsigned o = (m < M && colok) ? (unsigned)(m * a.Cout + col) * 4u : OOB;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, v), orsrc, (int)o, 0, 0);
    }
    if (a.ts) {
        stamp[4] = wall_clock64();
        if (tid == 0) for (int i = 0; i < 5; ++i) a.ts[(long long)blockIdx.x * 5 + i] = stamp[i];
    }
}

int dwpw_tile_bm(int shape) { return shape == DWPW_128x64 ? 128 : 64; }
int dwpw_tile_bn(int shape) { return shape == DWPW_128x64 ? 64 : 128; }

template <int STRIDE, int WAVES_M, int WAVES_N, int WM, int WN, bool BATCH>
static hipError_t launch_shape(const DwPwArgs &a, hipStream_t s)
{
    constexpr int BM = WAVES_M * WM * 32, BN = WAVES_N * WN * 32;
    const int lds_main = (a.K / 32) * BM * 128 + 2 * BN * 128;
    const int lds_epi = 4 * WM * 32 * WN * 32 * 4;
    const int lds_bytes = lds_main > lds_epi ? lds_main : lds_epi;
    static int attr_max = 0;
    auto k = dwpw_kernel<STRIDE, WAVES_M, WAVES_N, WM, WN, BATCH>;
    if (lds_bytes > attr_max) {
        hipError_t e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e != hipSuccess) return e;
        attr_max = lds_bytes;
    }
    const long long nblk = (long long)((a.M + BM - 1) / BM) * a.n_tiles_n;
    if (nblk <= 0 || nblk > 0x7fffffffLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k, dim3((unsigned)nblk), dim3(256), lds_bytes, s, a);
    return hipGetLastError();
}

hipError_t launch_dwpw(int shape, int stride, const DwPwArgs &a, hipStream_t s)
{
    // host-side shape checks: the kernel assumes them
    if (!a.in || !a.wdw || !a.dmean || !a.dsf || !a.dbeta || !a.wt || !a.mean || !a.sf || !a.beta || !a.out) return hipErrorInvalidValue;
    if (a.K < 32 || a.K % 32 != 0 || a.K > 256 || a.Cout % 4 != 0 || a.OW % 4 != 0 || (a.OH * a.OW) % 4 != 0) return hipErrorInvalidValue;
    if (a.M != a.B * a.OH * a.OW || a.M < 1 || (stride != 1 && stride != 2)) return hipErrorInvalidValue;
    if (a.CoutPad % dwpw_tile_bn(shape) != 0 || a.n_tiles_n * dwpw_tile_bn(shape) != a.CoutPad || a.Cout > a.CoutPad) return hipErrorInvalidValue;
    // 32-bit byte offsets in the buffer resources
    if ((long long)a.B * a.H * a.W * a.K * 4 >= (1LL << 31) || (long long)a.M * a.Cout * 4 >= (1LL << 31) ||
        (long long)a.CoutPad * a.K * 4 >= (1LL << 31)) return hipErrorInvalidValue;
    // every tap of every output must come from rows/columns the index arithmetic covers
    if ((a.OH - 1) * stride + 2 - a.pad > a.H + 1 || (a.OW - 1) * stride + 2 - a.pad > a.W + 1 || a.pad < 0 || a.pad > 1) return hipErrorInvalidValue;
    static int batch_env = -2;
    if (batch_env == -2) { const char *e = getenv("SSD_DWPW_BATCH"); batch_env = e ? atoi(e) : -1; }
    // measured (scripts/bench_dwpw.py): with the weights hoisted the batched form no longer pays
    // (Conv2d_1..4: 0.659/0.522/0.624/0.393 ms batched vs 0.612/0.489/0.640/0.366 ms); kept for A/B runs
    const bool batch = batch_env > 0;
    if (shape == DWPW_128x64) {
        if (batch) return stride == 1 ? launch_shape<1, 4, 1, 1, 2, true>(a, s) : launch_shape<2, 4, 1, 1, 2, true>(a, s);
        return stride == 1 ? launch_shape<1, 4, 1, 1, 2, false>(a, s) : launch_shape<2, 4, 1, 1, 2, false>(a, s);
    }
    if (shape == DWPW_64x128) {
        if (batch) return stride == 1 ? launch_shape<1, 2, 2, 1, 2, true>(a, s) : launch_shape<2, 2, 2, 1, 2, true>(a, s);
        return stride == 1 ? launch_shape<1, 2, 2, 1, 2, false>(a, s) : launch_shape<2, 2, 2, 1, 2, false>(a, s);
    }
    return hipErrorInvalidValue;
}
