// Probe for the split-fp16 ("f16x3") convolution mode: v_mfma_f32_32x32x16_f16 on gfx950.
//   1. operand / result lane layout (checked against a CPU product of small integers)
//   2. are fp16 denormal inputs honoured?
//   3. accuracy of  x*w ~= xh*wh + xh*wl + xl*wh  (x = xh + xl, halves) over K = 2304 terms,
//      against a double-precision sum and against the sequential fp32 fmaf chain.
// Build: hipcc --offload-arch=gfx950 -O2 -o mfma_f16_probe mfma_f16_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

// A [32][K] halves, B [32][K] halves (both k-contiguous), D [32][32] floats: D = A * B^T
// nsplit = 1: plain; 3: A = Ah|Al, B = Bh|Bl given as separate arrays
__global__ void probe(const _Float16 *Ah, const _Float16 *Al, const _Float16 *Bh, const _Float16 *Bl, int K, int nsplit, float *D)
{
    const int lane = threadIdx.x, i = lane & 31, g = lane >> 5;
    v16f acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int k0 = 0; k0 < K; k0 += 16) {
        const v8h ah = *(const v8h *)(Ah + (size_t)i * K + k0 + 8 * g);
        const v8h bh = *(const v8h *)(Bh + (size_t)i * K + k0 + 8 * g);
        if (nsplit == 3) {
            const v8h al = *(const v8h *)(Al + (size_t)i * K + k0 + 8 * g);
            const v8h bl = *(const v8h *)(Bl + (size_t)i * K + k0 + 8 * g);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
    }
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * g;
        D[row * 32 + i] = acc[r];
    }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

static int run(const std::vector<_Float16> &ah, const std::vector<_Float16> &al, const std::vector<_Float16> &bh,
               const std::vector<_Float16> &bl, int K, int nsplit, std::vector<float> &D)
{
    _Float16 *d[4];
    const std::vector<_Float16> *src[4] = {&ah, &al, &bh, &bl};
    for (int t = 0; t < 4; ++t) {
        CK(hipMalloc(&d[t], 32 * K * 2));
        CK(hipMemcpy(d[t], src[t]->data(), 32 * K * 2, hipMemcpyHostToDevice));
    }
    float *dd;
    CK(hipMalloc(&dd, 32 * 32 * 4));
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d[0], d[1], d[2], d[3], K, nsplit, dd);
    CK(hipDeviceSynchronize());
    D.resize(1024);
    CK(hipMemcpy(D.data(), dd, 4096, hipMemcpyDeviceToHost));
    for (int t = 0; t < 4; ++t) (void)hipFree(d[t]);
    (void)hipFree(dd);
    return 0;
}

int main()
{
    srand(1);
    {   // 1. layout
        const int K = 32;
        std::vector<_Float16> a(32 * K), b(32 * K), z(32 * K, (_Float16)0.f);
        for (auto &v : a) v = (_Float16)(float)(rand() % 7 - 3);
        for (auto &v : b) v = (_Float16)(float)(rand() % 7 - 3);
        std::vector<float> D;
        if (run(a, z, b, z, K, 1, D)) return 1;
        int bad = 0;
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) {
                float s = 0;
                for (int k = 0; k < K; ++k) s += (float)a[i * K + k] * (float)b[j * K + k];
                if (s != D[i * 32 + j]) ++bad;
            }
        printf("layout: %d mismatches of 1024\n", bad);
    }
    {   // 2. denormals: a = 2^-20 (fp16 subnormal), b = 2^10 -> each product 2^-10, K = 16 -> 2^-6
        const int K = 16;
        std::vector<_Float16> a(32 * K, (_Float16)ldexpf(1.f, -20)), b(32 * K, (_Float16)1024.f), z(32 * K, (_Float16)0.f);
        std::vector<float> D;
        if (run(a, z, b, z, K, 1, D)) return 1;
        printf("denormal inputs: D[0] = %g (expected %g if honoured, 0 if flushed)\n", D[0], ldexp(1.0, -6));
        // both subnormal: 2^-20 * 2^-20 = 2^-40 * 16 = 2^-36
        std::vector<_Float16> c(32 * K, (_Float16)ldexpf(1.f, -20));
        if (run(a, z, c, z, K, 1, D)) return 1;
        printf("denormal x denormal: D[0] = %g (expected %g)\n", D[0], ldexp(1.0, -36));
    }
    for (int variant = 0; variant < 3; ++variant) {   // 3. accuracy, K = 2304
        const int K = 2304;
        const int ashift = variant == 2 ? 8 : 0, wshift = variant == 0 ? 0 : 12;
        std::vector<float> x(32 * K), w(32 * K);
        for (auto &v : x) { float u = (rand() / (float)RAND_MAX); v = u < 0.4f ? 0.f : 6.f * (u - 0.4f); }   // relu-like, [0, 3.6]
        for (auto &v : w) { float u = 0; for (int t = 0; t < 12; ++t) u += rand() / (float)RAND_MAX; v = (u - 6.f) * 0.0295f; }  // ~N(0, 2/2304)
        std::vector<_Float16> xh(32 * K), xl(32 * K), wh(32 * K), wl(32 * K);
        for (int t = 0; t < 32 * K; ++t) {
            const float xs = ldexpf(x[t], ashift), ws = ldexpf(w[t], wshift);
            xh[t] = (_Float16)xs; xl[t] = (_Float16)(xs - (float)xh[t]);
            wh[t] = (_Float16)ws; wl[t] = (_Float16)(ws - (float)wh[t]);
        }
        std::vector<float> D3, D1;
        if (run(xh, xl, wh, wl, K, 3, D3)) return 1;
        if (run(xh, xl, wh, wl, K, 1, D1)) return 1;
        double e3 = 0, e1 = 0, ec = 0, rms = 0;
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) {
                double s = 0;
                float c = 0.f;
                for (int k = 0; k < K; ++k) {
                    s += (double)x[i * K + k] * (double)w[j * K + k];
                    c = fmaf(x[i * K + k], w[j * K + k], c);
                }
                const double sc = ldexp(1.0, -(ashift + wshift));
                e3 = fmax(e3, fabs(D3[i * 32 + j] * sc - s));
                e1 = fmax(e1, fabs(D1[i * 32 + j] * sc - s));
                ec = fmax(ec, fabs((double)c - s));
                rms += s * s;
            }
        rms = sqrt(rms / 1024);
        printf("K=2304 ashift=%d wshift=%d: output rms %.3g; max abs err vs double: f16x3 %.3g, f16x1 %.3g, fp32 fmaf chain %.3g\n",
               ashift, wshift, rms, e3, e1, ec);
    }
    return 0;
}
