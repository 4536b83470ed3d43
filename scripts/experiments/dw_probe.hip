// Depthwise 3x3 (+BN+ReLU6) at the MobileNet backbone shapes that run as their own launch: what does the shipped
// kernel reach alone, what does a plain copy of the same bytes reach, and do more outputs per thread (R rows x PX
// pixels, fewer L1/L2 reads per output) change it?  Every variant is checked bit-for-bit against the shipped kernel.
//   build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I single-shot-detector_amd/csrc scripts/experiments/dw_probe.hip -o scripts/experiments/bin/dw_probe
#include "../../single-shot-detector_amd/csrc/elementwise.hip"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

template <int STRIDE, int R, int PX>
__global__ __launch_bounds__(256) void dw_var(const float *__restrict__ in, int B, int H, int W, int C,
                                               const float *__restrict__ w, int pad, int OH, int OW, const float *mean,
                                               const float *sf, const float *beta, int act, float *__restrict__ out)
{
    constexpr int NCOL = (PX - 1) * STRIDE + 3, NROW = (R - 1) * STRIDE + 3;
    const int C4 = C >> 2, XG = (OW + PX - 1) / PX, YG = (OH + R - 1) / R;
    const long long total = (long long)B * YG * XG * C4;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C4) * 4;
        long long q = idx / C4;
        const int ox0 = (int)(q % XG) * PX;
        q /= XG;
        const int oy0 = (int)(q % YG) * R;
        const int b = (int)(q / YG);
        v4f wv[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) wv[t] = *(const v4f *)(w + t * C + c);
        v4f acc[R][PX];
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int p = 0; p < PX; ++p) acc[r][p] = (v4f){0.0f, 0.0f, 0.0f, 0.0f};
        const int ix0 = ox0 * STRIDE - pad, iy0 = oy0 * STRIDE - pad;
#pragma unroll
        for (int j = 0; j < NROW; ++j) {
            const int iy = iy0 + j;
            const bool rowok = (unsigned)iy < (unsigned)H;
            const float *rowp = in + (((long long)b * H + (rowok ? iy : 0)) * W) * C + c;
            v4f x[NCOL];
#pragma unroll
            for (int k = 0; k < NCOL; ++k) {
                const int ix = ix0 + k;
                const bool ok = rowok && (unsigned)ix < (unsigned)W;
                x[k] = (v4f){0.0f, 0.0f, 0.0f, 0.0f};
                if (ok) x[k] = *(const v4f *)(rowp + (long long)ix * C);
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int ky = j - r * STRIDE;
                if (ky >= 0 && ky < 3) {
#pragma unroll
                    for (int p = 0; p < PX; ++p)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                acc[r][p][i] = fmaf(x[p * STRIDE + kx][i], wv[ky * 3 + kx][i], acc[r][p][i]);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (oy0 + r >= OH) continue;
            float *o = out + (((long long)b * OH + oy0 + r) * OW + ox0) * C + c;
#pragma unroll
            for (int p = 0; p < PX; ++p)
                if (ox0 + p < OW) *(v4f *)(o + (long long)p * C) = bn_act4(acc[r][p], mean, sf, beta, c, act);
        }
    }
}

__global__ __launch_bounds__(256) void copy_like(const v4f *__restrict__ in, long long nin, v4f *__restrict__ out, long long nout)
{
    // reads nin float4 and writes nout float4 (nout <= nin): the bytes a depthwise layer moves, nothing else
    const long long step = (long long)gridDim.x * blockDim.x, ratio = nin / nout;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nout; i += step) {
        v4f a = in[i];
        for (long long r = 1; r < ratio; ++r) { const v4f t = in[i + r * nout]; a[0] += t[0]; a[1] += t[1]; a[2] += t[2]; a[3] += t[3]; }
        out[i] = a;
    }
}

__global__ void fill(float *p, long long n, unsigned seed)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed;
        x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = (x & 0xFFFF) / 65536.0f * 6.0f;
    }
}

struct Shape { const char *name; int H, W, C, stride; };

template <int STRIDE, int R, int PX>
static void launch_var(const float *in, int B, int H, int W, int C, const float *w, int OH, int OW, const float *mean,
                       const float *sf, const float *beta, float *out, int cap)
{
    const long long total = (long long)B * ((OH + R - 1) / R) * ((OW + PX - 1) / PX) * (C / 4);
    long long blocks = (total + 255) / 256;
    if (blocks > 256LL * cap) blocks = 256LL * cap;
    hipLaunchKernelGGL((dw_var<STRIDE, R, PX>), dim3((unsigned)blocks), dim3(256), 0, 0, in, B, H, W, C, w, STRIDE == 1 ? 1 : 0, OH,
                       OW, mean, sf, beta, 2, out);
}

int main(int argc, char **argv)
{
    const int B = argc > 1 ? atoi(argv[1]) : 32;
    const Shape shapes[] = {{"Conv2d_5  s1 80x112x256", 80, 112, 256, 1}, {"Conv2d_6  s2 80x112x256", 80, 112, 256, 2},
                            {"Conv2d_7  s1 40x56x512", 40, 56, 512, 1},   {"Conv2d_12 s2 40x56x512", 40, 56, 512, 2},
                            {"Conv2d_13 s1 20x28x1024", 20, 28, 1024, 1}, {"Conv2d_1  s1 320x448x32", 320, 448, 32, 1}};
    const int NBUF = 4, reps = 20;
    {   // first convolution (uint8 frames 640x896x3 -> 320x448x32, stride 2) of the MobileNet backbone, alone
        const int H = 640, W = 896, Cout = 32;
        const long long nimg = (long long)B * H * W * 3, nout = (long long)B * (H / 2) * (W / 2) * Cout;
        uint8_t *img; float *out[2], *w, *mean, *sf, *beta;
        hipMalloc(&img, nimg); hipMalloc(&out[0], nout * 4); hipMalloc(&out[1], nout * 4);
        hipMalloc(&w, 27 * Cout * 4); hipMalloc(&mean, Cout * 4); hipMalloc(&sf, Cout * 4); hipMalloc(&beta, Cout * 4);
        hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, (float *)img, nimg / 4, 3u);
        hipLaunchKernelGGL(fill, dim3(16), dim3(256), 0, 0, w, 27LL * Cout, 5u);
        hipLaunchKernelGGL(fill, dim3(16), dim3(256), 0, 0, mean, (long long)Cout, 6u);
        hipLaunchKernelGGL(fill, dim3(16), dim3(256), 0, 0, sf, (long long)Cout, 7u);
        hipLaunchKernelGGL(fill, dim3(16), dim3(256), 0, 0, beta, (long long)Cout, 8u);
        hipDeviceSynchronize();
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        for (int i = 0; i < 3; ++i) launch_first_conv(img, B, H, W, H, W, H, W, w, Cout, mean, sf, beta, 2, out[i & 1], 0);
        hipEventRecord(e0, 0);
        for (int i = 0; i < reps; ++i) launch_first_conv(img, B, H, W, H, W, H, W, w, Cout, mean, sf, beta, 2, out[i & 1], 0);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        ms /= reps;
        printf("first conv 640x896x3 u8 -> 320x448x32, %d images: %.1f MB in + %.1f MB out: %.1f us, %.2f TB/s\n", B, nimg / 1e6, nout * 4 / 1e6,
               ms * 1e3, (nimg + nout * 4.0) / (ms * 1e-3) / 1e12);
        hipFree(img); hipFree(out[0]); hipFree(out[1]); hipFree(w); hipFree(mean); hipFree(sf); hipFree(beta);
    }
    for (const Shape &s : shapes) {
        const int OH = s.H / s.stride, OW = s.W / s.stride;
        const long long nin = (long long)B * s.H * s.W * s.C, nout = (long long)B * OH * OW * s.C;
        float *in[NBUF], *out[NBUF], *ref, *w, *mean, *sf, *beta;
        for (int i = 0; i < NBUF; ++i) {
            if (hipMalloc(&in[i], nin * 4) != hipSuccess || hipMalloc(&out[i], nout * 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
            hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, in[i], nin, 17u * i + 1u);
        }
        hipMalloc(&ref, nout * 4);
        hipMalloc(&w, 9 * s.C * 4); hipMalloc(&mean, s.C * 4); hipMalloc(&sf, s.C * 4); hipMalloc(&beta, s.C * 4);
        hipLaunchKernelGGL(fill, dim3(16), dim3(256), 0, 0, w, 9LL * s.C, 5u);
        hipLaunchKernelGGL(fill, dim3(16), dim3(256), 0, 0, mean, (long long)s.C, 6u);
        hipLaunchKernelGGL(fill, dim3(16), dim3(256), 0, 0, sf, (long long)s.C, 7u);
        hipLaunchKernelGGL(fill, dim3(16), dim3(256), 0, 0, beta, (long long)s.C, 8u);
        hipDeviceSynchronize();
        const double bytes = (double)(nin + nout) * 4;
        printf("%s, %d images: %.1f MB in + %.1f MB out\n", s.name, B, nin * 4 / 1e6, nout * 4 / 1e6);
        auto timeit = [&](const char *name, auto &&go, bool check) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            for (int i = 0; i < 3; ++i) go(in[i % NBUF], out[i % NBUF]);
            hipEventRecord(e0, 0);
            for (int i = 0; i < reps; ++i) go(in[i % NBUF], out[i % NBUF]);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            ms /= reps;
            const char *verdict = "";
            if (check) {
                go(in[0], out[0]);
                hipDeviceSynchronize();
                std::vector<float> a((size_t)nout), b((size_t)nout);
                hipMemcpy(a.data(), out[0], nout * 4, hipMemcpyDeviceToHost);
                hipMemcpy(b.data(), ref, nout * 4, hipMemcpyDeviceToHost);
                verdict = memcmp(a.data(), b.data(), nout * 4) == 0 ? "  bit-equal" : "  DIFFERS";
            }
            printf("    %-34s %7.1f us  %5.2f TB/s%s\n", name, ms * 1e3, bytes / (ms * 1e-3) / 1e12, verdict);
            hipEventDestroy(e0); hipEventDestroy(e1);
        };
        auto shipped = [&](const float *i, float *o) {
            launch_depthwise(i, B, s.H, s.W, s.C, w, s.stride, s.stride == 1 ? 1 : 0, OH, OW, mean, sf, beta, 2, o, 0, 0, nullptr);
        };
        shipped(in[0], ref);
        hipDeviceSynchronize();
        timeit("copy of the same bytes", [&](const float *i, float *o) {
            hipLaunchKernelGGL(copy_like, dim3(256 * 16), dim3(256), 0, 0, (const v4f *)i, nin / 4, (v4f *)o, nout / 4); }, false);
        timeit("shipped kernel", shipped, true);
#define VAR(ST, R, PX, CAP)                                                                                                        \
    if (s.stride == ST)                                                                                                            \
        timeit("R=" #R " PX=" #PX " cap " #CAP, [&](const float *i, float *o) {                                                     \
            launch_var<ST, R, PX>(i, B, s.H, s.W, s.C, w, OH, OW, mean, sf, beta, o, CAP); }, true);
        VAR(1, 1, 4, 64) VAR(1, 2, 4, 64) VAR(1, 4, 4, 64) VAR(1, 2, 2, 64) VAR(1, 1, 8, 64) VAR(1, 2, 4, 16) VAR(1, 1, 2, 64) VAR(1, 1, 1, 64)
        VAR(2, 1, 4, 64) VAR(2, 2, 4, 64) VAR(2, 2, 2, 64) VAR(2, 1, 2, 64) VAR(2, 4, 2, 64) VAR(2, 2, 4, 16)
        for (int i = 0; i < NBUF; ++i) { hipFree(in[i]); hipFree(out[i]); }
        hipFree(ref); hipFree(w); hipFree(mean); hipFree(sf); hipFree(beta);
    }
    return 0;
}
