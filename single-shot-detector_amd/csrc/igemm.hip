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
//   pipeline    global_load_dwordx4 of K-step s+1 -> registers while the MFMAs of
//               K-step s run from LDS; registers -> LDS (other stage); one barrier.
//   numerics    each output element is ONE accumulator that receives its k terms in
//               increasing logical (ky,kx,ci) order: bit-identical to the fmaf chain
//               of the oracle.  Epilogue ops are separately rounded (-ffp-contract=off).
#include "ssd_internal.h"

typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int WAVES_M, int WAVES_N, int WM, int WN>
__global__ __launch_bounds__(256, 2) void igemm_kernel(const IgemmArgs a)
{
    constexpr int BM = WAVES_M * WM * 32;
    constexpr int BN = WAVES_N * WN * 32;
    constexpr int NA = BM / 32;          // 16-B loads per thread per K-step, A tile
    constexpr int NB = BN / 32;          // same, B tile
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
    const int tile_n = swz % a.n_tiles_n;
    const int tile_m = swz / a.n_tiles_n;
    int lvl = 0;
#pragma unroll
    for (int i = 1; i < SSD_MAX_LEVELS; ++i)
        if (i < a.nlevels && tile_m >= a.lv[i].tile_begin) lvl = i;
    const IgemmLevel &L = a.lv[lvl];
    const int H = L.H, W = L.W, OW = L.OW, M = L.M;
    const int P = L.OH * L.OW;
    const int Cin = a.Cin;
    const int m0 = (tile_m - L.tile_begin) * BM;

    // ---- per-thread load bookkeeping: thread owns chunk (tid&7) of rows (tid>>3)+32u
    const float *arow[NA];
    int aiy0[NA], aix0[NA];
#pragma unroll
    for (int u = 0; u < NA; ++u) {
        const int m = m0 + (tid >> 3) + 32 * u;
        if (m < M) {
            const int b = m / P, p = m - b * P;
            const int oy = p / OW, ox = p - oy * OW;
            arow[u] = a.in + L.in_off + (long long)b * H * W * Cin + (tid & 7) * 4;
            aiy0[u] = oy * a.stride - a.pad;
            aix0[u] = ox * a.stride - a.pad;
        } else {
            arow[u] = a.in;
            aiy0[u] = -(1 << 20);
            aix0[u] = 0;
        }
    }
    const float *brow = a.wt + (long long)(tile_n * BN + (tid >> 3)) * Cin + (tid & 7) * 4;
    const long long b_ustride = 32LL * Cin;
    const long long b_tapstride = (long long)a.CoutPad * Cin;
    const int woff = (tid >> 3) * 128 + (((tid & 7) ^ ((tid >> 4) & 7)) << 4);

    // ---- per-lane fragment read offsets (4 octets of the 32-channel K-step)
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

    const int KC = Cin >> 5;
    const int KS = a.taps * KC;
    v4f ra[NA], rb[NB];
    int ky = 0, kx = 0, kc = 0, tap = 0;   // coordinates of the NEXT K-step to load

    auto gload = [&]() {
#pragma unroll
        for (int u = 0; u < NA; ++u) {
            const int iy = aiy0[u] + ky, ix = aix0[u] + kx;
            const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
            v4f v = {0.0f, 0.0f, 0.0f, 0.0f};
            if (ok) v = *(const v4f *)(arow[u] + (long long)(iy * W + ix) * Cin + kc * 32);
            ra[u] = v;
        }
#pragma unroll
        for (int u = 0; u < NB; ++u)
            rb[u] = *(const v4f *)(brow + tap * b_tapstride + u * b_ustride + kc * 32);
        if (++kc == KC) {
            kc = 0;
            ++tap;
            if (++kx == 3) { kx = 0; ++ky; }
        }
    };
    auto lstore = [&](int stage) {
        unsigned char *base = lds + stage * STAGE;
#pragma unroll
        for (int u = 0; u < NA; ++u) *(v4f *)(base + woff + u * 4096) = ra[u];
#pragma unroll
        for (int u = 0; u < NB; ++u) *(v4f *)(base + A_BYTES + woff + u * 4096) = rb[u];
    };
    auto compute = [&](int stage) {
        const unsigned char *abase = lds + stage * STAGE + wave_m * WM * 4096;
        const unsigned char *bbase = lds + stage * STAGE + A_BYTES + wave_n * WN * 4096;
        // all 16 fragment reads of the K-step are issued up front; the MFMAs of octet g
        // only wait for their own operands (counted lgkmcnt), later reads stay in flight.
        v4f af[4][WM], bf[4][WN];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int i = 0; i < WM; ++i) af[g][i] = *(const v4f *)(abase + i * 4096 + roff[g]);
#pragma unroll
            for (int j = 0; j < WN; ++j) bf[g][j] = *(const v4f *)(bbase + j * 4096 + roff[g]);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[g][i][t], bf[g][j][t], acc[i][j], 0, 0, 0);
    };

    gload();
    lstore(0);
    __syncthreads();
    for (int ks = 0; ks < KS; ++ks) {
        const bool more = ks + 1 < KS;
        if (more) gload();
        compute(ks & 1);
        if (more) lstore((ks + 1) & 1);
        __syncthreads();
    }

    // ---- epilogue: batch norm / bias / upsample-add / activation, straight from the
    // accumulators (lane = column, 16 registers = 16 rows of the 32x32 tile).
    const bool has_bn = a.mean != nullptr;
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        const int col = tile_n * BN + (wave_n * WN + j) * 32 + (lane & 31);
        const bool colok = col < a.Cout;
        float mean = 0.0f, sf = 1.0f, beta = 0.0f, bias = 0.0f;
        if (colok && has_bn) {
            mean = a.mean[L.param_off + col];
            sf = a.sf[L.param_off + col];
            beta = a.beta[L.param_off + col];
        }
        if (colok && a.bias) bias = a.bias[L.param_off + col];
#pragma unroll
        for (int i = 0; i < WM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (wave_m * WM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int m = m0 + row;
                if (m < M && colok) {
                    const float raw = acc[i][j][r];
                    float v = raw;
                    if (has_bn) {
                        v = (v - mean) * sf;
                        v = v + beta;
                    }
                    if (a.bias) v = v + bias;
                    long long off;
                    if (a.dense_out && !a.res) {
                        off = L.out_off + (long long)m * L.out_rstride + col;
                    } else {
                        const int b = m / P, p = m - b * P;
                        off = L.out_off + b * L.out_bstride + (long long)p * L.out_rstride + col;
                        if (a.res) {
                            const int oy = p / OW, ox = p - oy * OW;
                            const int ch = L.OH >> 1, cw = OW >> 1;
                            v = a.res[L.res_off + (((long long)b * ch + (oy >> 1)) * cw + (ox >> 1)) * a.Cout + col] + v;
                        }
                    }
                    if (a.act >= 1) v = v > 0.0f ? v : 0.0f;
                    if (a.act == 2) v = v < 6.0f ? v : 6.0f;
                    a.out[off] = v;
                    if (a.out2) a.out2[off] = raw > 0.0f ? raw : 0.0f;
                }
            }
        }
    }
}

int igemm_tile_bm(int) { return 128; }
int igemm_tile_bn(int tile) { return tile == IGEMM_128x128 ? 128 : (tile == IGEMM_128x64 ? 64 : 32); }

template <int WAVES_M, int WAVES_N, int WM, int WN>
static hipError_t launch_t(const IgemmArgs &a, int total_tiles_m, hipStream_t s)
{
    constexpr int BM = WAVES_M * WM * 32, BN = WAVES_N * WN * 32;
    constexpr int lds_bytes = 2 * (BM + BN) * 128;
    static bool attr_set = false;
    auto k = igemm_kernel<WAVES_M, WAVES_N, WM, WN>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const long long nblk = (long long)total_tiles_m * a.n_tiles_n;
    if (nblk <= 0 || nblk > 0x7fffffffLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k, dim3((unsigned)nblk), dim3(256), lds_bytes, s, a);
    return hipGetLastError();
}

hipError_t launch_igemm(int tile, const IgemmArgs &a, int total_tiles_m, hipStream_t s)
{
    // host-side shape checks: the kernel assumes them
    if (a.Cin % 32 != 0 || a.CoutPad % igemm_tile_bn(tile) != 0 || (a.taps != 1 && a.taps != 9)) return hipErrorInvalidValue;
    if (a.n_tiles_n * igemm_tile_bn(tile) != a.CoutPad || a.nlevels < 1 || a.nlevels > SSD_MAX_LEVELS) return hipErrorInvalidValue;
    switch (tile) {
    case IGEMM_128x128: return launch_t<2, 2, 2, 2>(a, total_tiles_m, s);
    case IGEMM_128x64: return launch_t<4, 1, 1, 2>(a, total_tiles_m, s);
    case IGEMM_128x32: return launch_t<4, 1, 1, 1>(a, total_tiles_m, s);
    }
    return hipErrorInvalidValue;
}
