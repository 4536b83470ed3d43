// Post-processing of the detector on the GPU: ssd.py:60 (sigmoid) + nms.py:48-102
// (confidence filter, decode, clip, per-class tf.image.non_max_suppression, class-major
// concat, zero pad) + model.py:67-68 (divide by box_scaler).
//
//   K9a scan     one pass over logits [B,N,C] (HBM-bound, 16 B per lane): a logit below a
//                conservative bound cannot reach score_threshold and is skipped; the
//                rare survivors get the correctly rounded fp32 sigmoid (evaluated in
//                double), and every (anchor, class) with score > threshold is appended
//                to the candidate list of its (image, class) as a 64-bit key
//                (score bits << 32 | ~anchor) -- key order == (score desc, anchor asc).
//                The anchor's box is decoded + clipped once into dec[B,N,4].
//                (The reference's `max_c score >= thr` row filter, nms.py:71, only
//                 decides which rows reach the NMS op; rows it keeps whose scores are all
//                 <= thr can never be selected, so the candidate set is `score > thr`.)
//   K9c nms      one wavefront per (image, class): <= max_per_class rounds of
//                {wave arg-max over live keys -> keep -> every lane kills its live
//                candidates whose IoU with the kept box is > iou_threshold}.  The arg-max
//                of the survivors is exactly the next box greedy NMS keeps, so no sort is
//                needed and any candidate count works (<= 512 stay in registers).
//   K9d pack     per image: prefix over class counts, class-major copy, / box_scaler,
//                zero padding, num_boxes.
#include "ssd_internal.h"
#include <cstdlib>

typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;

__device__ __forceinline__ float sigmoid_cr(float x) { return (float)(1.0 / (1.0 + exp(-(double)x))); }
__device__ __forceinline__ float exp_cr(float x) { return (float)exp((double)x); }

// box_utils.py:114-142 + nms.py:77
__device__ __forceinline__ v4f decode_clip(const v4f c, const v4f a)
{
    const float ha = a[2] - a[0], wa = a[3] - a[1];
    float t = 0.5f * ha;
    const float cya = a[0] + t;
    t = 0.5f * wa;
    const float cxa = a[1] + t;
    const float ty = c[0] / 10.0f, tx = c[1] / 10.0f, th = c[2] / 5.0f, tw = c[3] / 5.0f;
    const float h = exp_cr(th) * ha, w = exp_cr(tw) * wa;
    t = ty * ha;
    const float cy = t + cya;
    t = tx * wa;
    const float cx = t + cxa;
    const float hh = 0.5f * h, hw = 0.5f * w;
    v4f b = {cy - hh, cx - hw, cy + hh, cx + hw};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float v = b[k];
        v = v < 0.0f ? 0.0f : v;
        v = v > 1.0f ? 1.0f : v;
        b[k] = v;
    }
    return b;
}

// IOUGreaterThanThreshold of TF r1.12 non_max_suppression_op.cc
__device__ __forceinline__ bool iou_greater(const v4f bi, const v4f bj, float thr)
{
    const float ymin_i = fminf(bi[0], bi[2]), xmin_i = fminf(bi[1], bi[3]);
    const float ymax_i = fmaxf(bi[0], bi[2]), xmax_i = fmaxf(bi[1], bi[3]);
    const float ymin_j = fminf(bj[0], bj[2]), xmin_j = fminf(bj[1], bj[3]);
    const float ymax_j = fmaxf(bj[0], bj[2]), xmax_j = fmaxf(bj[1], bj[3]);
    const float area_i = (ymax_i - ymin_i) * (xmax_i - xmin_i);
    const float area_j = (ymax_j - ymin_j) * (xmax_j - xmin_j);
    // branch-free form of `if (area_i <= 0 || area_j <= 0) return false;` -- with both areas
    // positive the union is positive, otherwise the (possibly NaN) quotient is masked.
    const bool valid = (area_i > 0.0f) & (area_j > 0.0f);
    const float iy0 = fmaxf(ymin_i, ymin_j), ix0 = fmaxf(xmin_i, xmin_j);
    const float iy1 = fminf(ymax_i, ymax_j), ix1 = fminf(xmax_i, xmax_j);
    const float ih = fmaxf(iy1 - iy0, 0.0f), iw = fmaxf(ix1 - ix0, 0.0f);
    const float inter = ih * iw;
    float uni = area_i + area_j;
    uni = uni - inter;
    const float iou = inter / uni;
    return valid & (iou > thr);
}

__device__ __forceinline__ void emit_candidate(const PostArgs &p, int b, int i, int c, float logit)
{
    const float s = sigmoid_cr(logit);
    if (!(s > p.score_thr)) return;
    const int slot = atomicAdd(&p.counts[b * p.C + c], 1);
    p.keys[((long long)b * p.C + c) * p.N + slot] = ((u64)__float_as_uint(s) << 32) | (u64)(0xFFFFFFFFu - (unsigned)i);
    const v4f code = *(const v4f *)(p.codes + ((long long)b * p.N + i) * 4);
    const v4f anc = *(const v4f *)(p.anchors + (long long)i * 4);
    *(v4f *)(p.dec + ((long long)b * p.N + i) * 4) = decode_clip(code, anc);
}

__global__ __launch_bounds__(256) void post_scan_kernel(const PostArgs p)
{
    const int C = p.C;
    if ((C & 3) == 0) {
        const int C4 = C >> 2;
        const long long total = (long long)p.B * p.N * C4;
        for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
             idx += (long long)gridDim.x * blockDim.x) {
            const v4f x = *(const v4f *)(p.logits + idx * 4);
            const float mx = fmaxf(fmaxf(x[0], x[1]), fmaxf(x[2], x[3]));
            if (mx >= p.logit_lo) {
                const int q = (int)(idx % C4);
                const long long row = idx / C4;
                const int i = (int)(row % p.N), b = (int)(row / p.N);
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (x[k] >= p.logit_lo) emit_candidate(p, b, i, q * 4 + k, x[k]);
            }
        }
    } else {
        const long long total = (long long)p.B * p.N * C;
        for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
             idx += (long long)gridDim.x * blockDim.x) {
            const float x = p.logits[idx];
            if (x >= p.logit_lo) {
                const int c = (int)(idx % C);
                const long long row = idx / C;
                emit_candidate(p, (int)(row / p.N), (int)(row % p.N), c, x);
            }
        }
    }
}

__device__ __forceinline__ u64 wave_max_u64(u64 v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const unsigned lo = __shfl_xor((unsigned)(v & 0xFFFFFFFFu), off, 64);
        const unsigned hi = __shfl_xor((unsigned)(v >> 32), off, 64);
        const u64 o = ((u64)hi << 32) | lo;
        v = o > v ? o : v;
    }
    return v;
}

#define NMS_R 8   // candidates per lane kept in registers (fast path: n <= 64*NMS_R)

__global__ __launch_bounds__(64) void post_nms_kernel(const PostArgs p)
{
    const int bc = blockIdx.x;              // b*C + c
    const int b = bc / p.C;
    const int lane = threadIdx.x;
    int n = p.counts[bc];
    if (n > p.N) n = p.N;
    u64 *keys = p.keys + (long long)bc * p.N;
    const float *dec = p.dec + (long long)b * p.N * 4;
    float *ob = p.cls_boxes + (long long)bc * p.max_per_class * 4;
    float *os = p.cls_scores + (long long)bc * p.max_per_class;
    int kept = 0;
    if (n > 0 && n <= p.fast_max) {
        // keys and boxes are immutable; liveness is one bit per register slot.  (A version
        // that zeroed key[r] under `key == best || iou > thr` was miscompiled by hipcc
        // 7.2: the kill of the IoU branch was dropped -- keep this form branch-free.)
        u64 key[NMS_R];
        v4f box[NMS_R];
        unsigned alive = 0;
#pragma unroll
        for (int r = 0; r < NMS_R; ++r) {
            const int i = lane + 64 * r;
            const bool ok = i < n;
            key[r] = ok ? keys[ok ? i : 0] : 0ull;
            const unsigned anchor = 0xFFFFFFFFu - (unsigned)(key[r] & 0xFFFFFFFFu);
            box[r] = *(const v4f *)(dec + (long long)(ok ? anchor : 0u) * 4);
            alive |= ok ? (1u << r) : 0u;
        }
        while (kept < p.max_per_class) {
            u64 best = 0;
#pragma unroll
            for (int r = 0; r < NMS_R; ++r) {
                const u64 k = ((alive >> r) & 1u) ? key[r] : 0ull;
                best = k > best ? k : best;
            }
            best = wave_max_u64(best);
            if (best == 0) break;
            const v4f wb = *(const v4f *)(dec + (long long)(0xFFFFFFFFu - (unsigned)(best & 0xFFFFFFFFu)) * 4);
            if (lane == 0) {
                *(v4f *)(ob + kept * 4) = wb;
                os[kept] = __uint_as_float((unsigned)(best >> 32));
            }
            ++kept;
            unsigned kill = 0;
#pragma unroll
            for (int r = 0; r < NMS_R; ++r) {
                const bool k = (key[r] == best) | iou_greater(box[r], wb, p.iou_thr);
                kill |= k ? (1u << r) : 0u;
            }
            alive &= ~kill;
        }
    } else if (n > 0) {
        // any candidate count: keys stay in global memory, dead candidates are zeroed
        while (kept < p.max_per_class) {
            u64 best = 0;
            for (int i = lane; i < n; i += 64) {
                const u64 k = keys[i];
                best = k > best ? k : best;
            }
            best = wave_max_u64(best);
            if (best == 0) break;
            const v4f wb = *(const v4f *)(dec + (long long)(0xFFFFFFFFu - (unsigned)(best & 0xFFFFFFFFu)) * 4);
            if (lane == 0) {
                *(v4f *)(ob + kept * 4) = wb;
                os[kept] = __uint_as_float((unsigned)(best >> 32));
            }
            ++kept;
            for (int i = lane; i < n; i += 64) {
                const u64 k = keys[i];
                if (k == 0) continue;
                const v4f bx = *(const v4f *)(dec + (long long)(0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFu)) * 4);
                if (k == best || iou_greater(bx, wb, p.iou_thr)) keys[i] = 0;
            }
        }
    }
    if (lane == 0) p.cls_counts[bc] = kept;
}

__global__ __launch_bounds__(256) void post_pack_kernel(const PostArgs p)
{
    extern __shared__ int pre[];            // C+1 exclusive prefix
    const int b = blockIdx.x, C = p.C, mp = p.max_per_class, T = C * mp;
    if (threadIdx.x == 0) {
        int s = 0;
        for (int c = 0; c < C; ++c) { pre[c] = s; s += p.cls_counts[b * C + c]; }
        pre[C] = s;
        p.num[b] = s;
    }
    __syncthreads();
    const int total = pre[C];
    float *boxes = p.boxes + (long long)b * T * 4;
    float *scores = p.scores + (long long)b * T;
    int32_t *labels = p.labels + (long long)b * T;
    for (int t = threadIdx.x; t < T; t += blockDim.x) {
        const int c = t / mp, j = t - c * mp;
        if (j < pre[c + 1] - pre[c]) {
            const int d = pre[c] + j;
            const long long src = (long long)(b * C + c) * mp + j;
#pragma unroll
            for (int k = 0; k < 4; ++k) boxes[d * 4 + k] = p.cls_boxes[src * 4 + k] / p.box_scaler[k];
            scores[d] = p.cls_scores[src];
            labels[d] = c;
        }
    }
    for (int d = total + threadIdx.x; d < T; d += blockDim.x) {
#pragma unroll
        for (int k = 0; k < 4; ++k) boxes[d * 4 + k] = 0.0f;
        scores[d] = 0.0f;
        labels[d] = 0;
    }
}

static size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

size_t post_workspace_bytes(int B, int N, int C, int mp)
{
    size_t s = 0;
    s += align_up((size_t)B * C * N * sizeof(u64));
    s += align_up((size_t)B * C * sizeof(int));
    s += align_up((size_t)B * N * 4 * sizeof(float));
    s += align_up((size_t)B * C * mp * 4 * sizeof(float));
    s += align_up((size_t)B * C * mp * sizeof(float));
    s += align_up((size_t)B * C * sizeof(int));
    return s;
}

void post_carve(PostArgs &p, void *ws)
{
    unsigned char *q = (unsigned char *)ws;
    const size_t B = p.B, N = p.N, C = p.C, mp = p.max_per_class;
    p.keys = (u64 *)q;          q += align_up(B * C * N * sizeof(u64));
    p.counts = (int *)q;        q += align_up(B * C * sizeof(int));
    p.dec = (float *)q;         q += align_up(B * N * 4 * sizeof(float));
    p.cls_boxes = (float *)q;   q += align_up(B * C * mp * 4 * sizeof(float));
    p.cls_scores = (float *)q;  q += align_up(B * C * mp * sizeof(float));
    p.cls_counts = (int *)q;
}

hipError_t launch_postprocess(const PostArgs &pin, hipStream_t s)
{
    PostArgs p = pin;
    p.fast_max = 64 * NMS_R;
    if (const char *e = getenv("SSD_NMS_FAST_MAX")) { int v = atoi(e); if (v >= 0 && v < p.fast_max) p.fast_max = v; }
    if (p.B < 1 || p.N < 1 || p.C < 1 || p.max_per_class < 1) return hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(p.counts, 0, (size_t)p.B * p.C * sizeof(int), s);
    if (e != hipSuccess) return e;
    const long long units = (long long)p.B * p.N * ((p.C & 3) ? p.C : p.C / 4);
    long long blocks = (units + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(post_scan_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p);
    hipLaunchKernelGGL(post_nms_kernel, dim3((unsigned)(p.B * p.C)), dim3(64), 0, s, p);
    hipLaunchKernelGGL(post_pack_kernel, dim3((unsigned)p.B), dim3(256), (p.C + 1) * sizeof(int), s, p);
    return hipGetLastError();
}
