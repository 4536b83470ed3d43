// Is v_mfma_f32_16x16x4_f32 -- like v_mfma_f32_32x32x2_f32 -- bit for bit a k-ORDERED fmaf chain, and what does a
// dependent chain of either cost?  (Round 3, batch-1 latency: a 32x32x2 accumulator advances k by 2 per 64 pipe cycles,
// a 16x16x4 accumulator by 4 per 32 -- a quarter of the chain latency for the small launches whose time is the length
// of ONE accumulator's K chain: fpn p6 at batch 1 is 4 608 dependent 32x32x2 MFMAs = 123 us at 2.4 GHz whatever the tile.)
//   part 1  D = A(16xK) * B(Kx16) with the 16x16x4 instruction, K = 4 .. 2304, against host chains in several k orders
//   part 2  the same for 32x32x2 (control: the shipped kernels rely on it)
//   part 3  cycles per MFMA: one dependent accumulator / four independent accumulators, both shapes, 1 wave per SIMD
// build: hipcc --offload-arch=gfx950 -O2 -ffp-contract=off -o bin/mfma_16x16x4_probe mfma_16x16x4_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

// A [16][K] row-major, B [K][16] row-major, D [16][16]
__global__ void k16(const float *A, const float *B, float *D, int K)
{
    const int lane = threadIdx.x, i = lane & 15, g = lane >> 4;
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < K; s += 4)
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[i * K + s + g], B[(s + g) * 16 + i], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * g + r) * 16 + i] = acc[r];     // D[row = 4*(lane/16) + r][col = lane%16]
}

// A [32][K], B [K][32], D [32][32]
__global__ void k32(const float *A, const float *B, float *D, int K)
{
    const int lane = threadIdx.x, i = lane & 31, g = lane >> 5;
    v16f acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int s = 0; s < K; s += 2)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[i * K + s + g], B[(s + g) * 32 + i], acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * g) * 32 + i] = acc[r];
}

template <int SHAPE, int NACC>
__global__ __launch_bounds__(256) void chain(float *out, long long *cyc, int iters)
{
    const float a = threadIdx.x * 1e-3f, b = 1.0001f;
    v16f acc32[NACC];
    v4f acc16[NACC];
    for (int i = 0; i < NACC; ++i) { for (int r = 0; r < 16; ++r) acc32[i][r] = 0.f; for (int r = 0; r < 4; ++r) acc16[i][r] = 0.f; }
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            if constexpr (SHAPE == 32) acc32[m % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc32[m % NACC], 0, 0, 0);
            else acc16[m % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc16[m % NACC], 0, 0, 0);
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) { for (int r = 0; r < 16; ++r) s += acc32[i][r]; for (int r = 0; r < 4; ++r) s += acc16[i][r]; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int SHAPE, int NACC>
static void time_chain(const char *name)
{
    const int blocks = 256, iters = 2000;
    float *out; long long *cyc;
    hipMalloc(&out, blocks * 256 * 4);
    hipMalloc(&cyc, blocks * 4 * 8);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((chain<SHAPE, NACC>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    std::vector<long long> h(blocks * 4);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (auto v : h) s += v;
    printf("%-64s %.1f cycles per MFMA\n", name, s / h.size() / iters / 16);
    hipFree(out); hipFree(cyc);
}

static unsigned rng = 2463534242u;
static float rnd() { rng ^= rng << 13; rng ^= rng >> 17; rng ^= rng << 5; return ((rng >> 8) / 8388608.0f - 1.0f) * 1.7f; }

int main()
{
    const int Ks[] = {4, 8, 32, 256, 2304};
    for (int K : Ks) {
        std::vector<float> A(16 * K), B(K * 16), D(256);
        for (auto &v : A) v = rnd();
        for (auto &v : B) v = rnd();
        float *dA, *dB, *dD;
        hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, 256 * 4);
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k16, dim3(1), dim3(64), 0, 0, dA, dB, dD, K);
        hipMemcpy(D.data(), dD, 256 * 4, hipMemcpyDeviceToHost);
        // hypotheses for one instruction's four products p0..p3 added to acc
        int eq_seq = 0, eq_rev = 0, eq_pair = 0, eq_tree = 0, eq_nofma = 0;
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f;
                for (int s = 0; s < K; s += 4) {
                    const float *a = &A[i * K + s];
                    float b[4];
                    for (int q = 0; q < 4; ++q) b[q] = B[(s + q) * 16 + j];
                    for (int q = 0; q < 4; ++q) s0 = fmaf(a[q], b[q], s0);                               // k ascending fmaf chain
                    for (int q = 3; q >= 0; --q) s1 = fmaf(a[q], b[q], s1);                              // k descending
                    s2 = fmaf(a[1], b[1], fmaf(a[0], b[0], s2)); s2 = fmaf(a[3], b[3], fmaf(a[2], b[2], s2));   // same as s0 (sanity)
                    s3 = s3 + (fmaf(a[1], b[1], a[0] * b[0]) + fmaf(a[3], b[3], a[2] * b[2]));          // dot-product tree
                    for (int q = 0; q < 4; ++q) { volatile float p = a[q] * b[q]; s4 = s4 + p; }         // separate multiply and add
                }
                const float d = D[i * 16 + j];
                eq_seq += !memcmp(&d, &s0, 4); eq_rev += !memcmp(&d, &s1, 4); eq_pair += !memcmp(&d, &s2, 4);
                eq_tree += !memcmp(&d, &s3, 4); eq_nofma += !memcmp(&d, &s4, 4);
            }
        printf("16x16x4  K=%4d: of 256 outputs bit-equal to  k-ascending fmaf chain %3d | k-descending %3d | (sanity) %3d | dot tree %3d | mul+add %3d\n",
               K, eq_seq, eq_rev, eq_pair, eq_tree, eq_nofma);
        hipFree(dA); hipFree(dB); hipFree(dD);
    }
    for (int K : Ks) {
        if (K & 1) continue;
        std::vector<float> A(32 * K), B(K * 32), D(1024);
        for (auto &v : A) v = rnd();
        for (auto &v : B) v = rnd();
        float *dA, *dB, *dD;
        hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, 1024 * 4);
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k32, dim3(1), dim3(64), 0, 0, dA, dB, dD, K);
        hipMemcpy(D.data(), dD, 1024 * 4, hipMemcpyDeviceToHost);
        int eq = 0;
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) {
                float s0 = 0.f;
                for (int s = 0; s < K; ++s) s0 = fmaf(A[i * K + s], B[s * 32 + j], s0);
                eq += !memcmp(&D[i * 32 + j], &s0, 4);
            }
        printf("32x32x2  K=%4d: of 1024 outputs bit-equal to the k-ascending fmaf chain %4d\n", K, eq);
        hipFree(dA); hipFree(dB); hipFree(dD);
    }
    time_chain<32, 1>("32x32x2, ONE dependent accumulator, 1 wave per SIMD");
    time_chain<32, 2>("32x32x2, two accumulators");
    time_chain<32, 4>("32x32x2, four accumulators");
    time_chain<16, 1>("16x16x4, ONE dependent accumulator, 1 wave per SIMD");
    time_chain<16, 2>("16x16x4, two accumulators");
    time_chain<16, 4>("16x16x4, four accumulators");
    return 0;
}
