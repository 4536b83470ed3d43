// Does the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32) run BESIDE f32 VALU work on the same SIMD, or instead of it?
// One block of 256 threads per CU (4 waves = one per SIMD) or 512 threads (two per SIMD).  Variants:
//   0  MFMA only            (NM dependent-free MFMAs per loop iteration, 4 accumulators)
//   1  VALU only            (NV v_fma_f32 per iteration on 8 independent registers)
//   2  both in ONE wave, interleaved by the compiler's order (MFMA, NV/NM FMAs, MFMA, ...)
//   3  two waves per SIMD: even waves MFMA only, odd waves VALU only
// Prints cycles per iteration (s_memtime) for each; if (2) ~ max(0, 1) the pipes overlap inside a wave, if ~ sum they do not.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v16f __attribute__((ext_vector_type(16)));

template <int VAR, int NM, int NV>
__global__ __launch_bounds__(512) void probe(float *out, long long *cyc, int iters)
{
    const int wave = threadIdx.x >> 6;
    v16f acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.001f + i;
    const float a = threadIdx.x * 1e-3f, b = 1.0001f;
    const bool do_m = VAR == 0 || VAR == 2 || (VAR == 3 && (wave & 1) == 0);
    const bool do_v = VAR == 1 || VAR == 2 || (VAR == 3 && (wave & 1) == 1);
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            if (do_m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m & 3], 0, 0, 0);
            if (do_v) {
#pragma unroll
                for (int k = 0; k < NV / NM; ++k) v[k & 7] = __builtin_fmaf(v[k & 7], b, a);
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + wave] = t1 - t0;
}

template <int VAR, int NM, int NV>
static void run(const char *name, int threads)
{
    const int blocks = 256, iters = 2000;
    float *out; long long *cyc;
    hipMalloc(&out, blocks * threads * 4);
    hipMalloc(&cyc, blocks * 8 * 8);
    hipLaunchKernelGGL((probe<VAR, NM, NV>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL((probe<VAR, NM, NV>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    std::vector<long long> h(blocks * (threads / 64));
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double even = 0, odd = 0; int ne = 0, no = 0;
    for (size_t i = 0; i < h.size(); ++i) { if ((i % (threads / 64)) & 1) { odd += h[i]; ++no; } else { even += h[i]; ++ne; } }
    printf("%-52s NM=%2d NV=%3d threads=%d: cycles/iteration even waves %.0f, odd waves %.0f\n", name, NM, NV, threads, even / ne / iters, no ? odd / no / iters : 0.0);
    hipFree(out); hipFree(cyc);
}

int main()
{
    run<0, 16, 0>("MFMA only, one wave per SIMD", 256);
    run<1, 16, 64>("VALU only (64 FMA), one wave per SIMD", 256);
    run<1, 16, 128>("VALU only (128 FMA), one wave per SIMD", 256);
    run<2, 16, 64>("one wave: 16 MFMA + 64 FMA interleaved", 256);
    run<2, 16, 128>("one wave: 16 MFMA + 128 FMA interleaved", 256);
    run<2, 16, 256>("one wave: 16 MFMA + 256 FMA interleaved", 256);
    run<0, 16, 0>("MFMA only, two waves per SIMD", 512);
    run<3, 16, 64>("two waves per SIMD: even MFMA, odd 64 FMA", 512);
    run<3, 16, 128>("two waves per SIMD: even MFMA, odd 128 FMA", 512);
    run<3, 16, 256>("two waves per SIMD: even MFMA, odd 256 FMA", 512);
    return 0;
}
