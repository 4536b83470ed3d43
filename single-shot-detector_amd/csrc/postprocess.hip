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

// One candidate (logit above the conservative bound): exact score, strict threshold,
// append to its (image, class) list, decode the anchor's box.
__device__ __forceinline__ void emit_candidate(const PostArgs &p, unsigned elem, float logit)
{
    const float s = sigmoid_cr(logit);
    if (!(s > p.score_thr)) return;
    const int c = (int)(elem % (unsigned)p.C);
    const unsigned row = elem / (unsigned)p.C;
    const int i = (int)(row % (unsigned)p.N), b = (int)(row / (unsigned)p.N);
    const int slot = atomicAdd(&p.counts[b * p.C + c], 1);
    p.keys[((long long)b * p.C + c) * p.N + slot] = ((u64)__float_as_uint(s) << 32) | (u64)(0xFFFFFFFFu - (unsigned)i);
    const v4f code = *(const v4f *)(p.codes + ((long long)b * p.N + i) * 4);
    const v4f anc = *(const v4f *)(p.anchors + (long long)i * 4);
    *(v4f *)(p.dec + ((long long)b * p.N + i) * 4) = decode_clip(code, anc);
}

// K9a.  The scan itself is a pure HBM stream; the rare logits above the bound are pushed
// into a per-block LDS queue and the queue is drained DENSELY by all 256 threads (the
// double-precision sigmoid / exp and the atomics would otherwise run with one active lane
// per wave).  The element index fits 32 bits (host check: B*N*C < 2^32).
#define SCAN_U 4                       // 16-B loads in flight per thread
#define SCAN_Q (2048 + 256 * SCAN_U * 4)
__global__ __launch_bounds__(256) void post_scan_kernel(const PostArgs p)
{
    __shared__ unsigned q_elem[SCAN_Q];
    __shared__ float q_val[SCAN_Q];
    __shared__ int q_n;
    const int C = p.C;
    const bool vec = (C & 3) == 0;
    const long long total = vec ? (long long)p.B * p.N * (C >> 2) : (long long)p.B * p.N * C;
    if (threadIdx.x == 0) q_n = 0;
    __syncthreads();
    auto drain = [&]() {
        const int n = q_n;
        for (int t = threadIdx.x; t < n; t += 256) emit_candidate(p, q_elem[t], q_val[t]);
        __syncthreads();
        if (threadIdx.x == 0) q_n = 0;
        __syncthreads();
    };
    auto push = [&](unsigned elem, float v) {
        const int slot = atomicAdd(&q_n, 1);
        q_elem[slot] = elem;
        q_val[slot] = v;
    };
    if (p.scan_fused) {
        // The logits convolution marked the octets that hold a candidate (igemm16.hip): read the bitmap (B*N*C/64
        // bytes) and the marked octets only.  Same filter, same emit_candidate: the per-class lists hold the same set.
        const long long nwords = ((long long)p.B * p.N * C / 8 + 31) / 32;
        for (long long w0 = (long long)blockIdx.x * 256; w0 < nwords; w0 += (long long)gridDim.x * 256) {
            const long long w = w0 + threadIdx.x;
            unsigned bits = w < nwords ? p.scan_bits[w] : 0u;
            // one marked octet per thread and round: at most 256 * 8 pushes between two looks at the queue
            // (capacity 2048 + 4096), however dense the marks are (saturated logits mark every octet)
            while (__syncthreads_or(bits != 0)) {
                if (bits) {
                    const int k = __builtin_ctz(bits);
                    bits &= bits - 1;
                    const long long e0 = (w * 32 + k) * 8;
                    const v4f x0 = *(const v4f *)(p.logits + e0), x1 = *(const v4f *)(p.logits + e0 + 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (x0[j] >= p.logit_lo) push((unsigned)(e0 + j), x0[j]);
                        if (x1[j] >= p.logit_lo) push((unsigned)(e0 + 4 + j), x1[j]);
                    }
                }
                if (__syncthreads_or(q_n > 2048)) drain();
            }
        }
        drain();
        return;
    }
    const long long stride = (long long)gridDim.x * 256 * SCAN_U;
    for (long long base = (long long)blockIdx.x * 256 * SCAN_U; base < total; base += stride) {
        if (vec) {
            v4f x[SCAN_U];
#pragma unroll
            for (int u = 0; u < SCAN_U; ++u) {
                const long long idx = base + u * 256 + threadIdx.x;
                x[u] = idx < total ? *(const v4f *)(p.logits + idx * 4) : (v4f){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            }
#pragma unroll
            for (int u = 0; u < SCAN_U; ++u) {
                const long long idx = base + u * 256 + threadIdx.x;
                const float mx = fmaxf(fmaxf(x[u][0], x[u][1]), fmaxf(x[u][2], x[u][3]));
                if (mx >= p.logit_lo) {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (x[u][k] >= p.logit_lo) push((unsigned)(idx * 4 + k), x[u][k]);
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < SCAN_U; ++u) {
                const long long idx = base + u * 256 + threadIdx.x;
                if (idx < total) {
                    const float x = p.logits[idx];
                    if (x >= p.logit_lo) push((unsigned)idx, x);
                }
            }
        }
        // the decision must be block-uniform AND separated by a barrier from the next iteration's pushes (a fast
        // wave's atomicAdd on q_n could otherwise flip it for a slow wave: divergent __syncthreads in drain)
        if (__syncthreads_or(q_n > 2048)) drain();
    }
    drain();
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

#define NMS_R 8      // candidates per thread kept in registers
#define NMS_BIG 1024 // threads of the large-list kernel

// Greedy NMS of one (image, class) list by ONE wavefront with R candidates per lane in registers.  keys and boxes are
// immutable; liveness is one bit per register slot.  (A version that zeroed key[r] under `key == best || iou > thr` was
// miscompiled by hipcc 7.2: the kill of the IoU branch was dropped -- keep this form branch-free.)
template <int R>
__device__ __forceinline__ int nms_one_wave(const PostArgs &p, const u64 *keys, const float *dec, int n, int lane, float *ob, float *os)
{
    u64 key[R];
    v4f box[R];
    unsigned alive = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = lane + 64 * r;
        const bool ok = i < n;
        key[r] = ok ? keys[ok ? i : 0] : 0ull;
        const unsigned anchor = 0xFFFFFFFFu - (unsigned)(key[r] & 0xFFFFFFFFu);
        box[r] = *(const v4f *)(dec + (long long)(ok ? anchor : 0u) * 4);
        alive |= ok ? (1u << r) : 0u;
    }
    int kept = 0;
    while (kept < p.max_per_class) {
        u64 best = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
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
        for (int r = 0; r < R; ++r) {
            const bool k = (key[r] == best) | iou_greater(box[r], wb, p.iou_thr);
            kill |= k ? (1u << r) : 0u;
        }
        alive &= ~kill;
    }
    return kept;
}

// K9c, lists of up to 64*NMS_R candidates: one wavefront, everything in registers.  (Measured and not adopted: the same wave
// with 32 candidates per lane for lists of 513 .. 2 048 instead of the 1 024-thread kernel -- batch-1 post-processing 0.22 ->
// 0.28 ms: 32 IoU tests per lane and round cost more than the block kernel's barrier per round.)
__global__ __launch_bounds__(64) void post_nms_small_kernel(const PostArgs p)
{
    const int bc = blockIdx.x;              // b*C + c
    const int b = bc / p.C;
    const int lane = threadIdx.x;
    int n = p.counts[bc];
    if (n > p.N) n = p.N;
    if (n > p.fast_max) {                   // handled by post_nms_big_kernel: put the pair on its work list
        if (lane == 0) p.big_list[atomicAdd(p.big_n, 1)] = bc;
        return;
    }
    const u64 *keys = p.keys + (long long)bc * p.N;
    const float *dec = p.dec + (long long)b * p.N * 4;
    float *ob = p.cls_boxes + (long long)bc * p.max_per_class * 4;
    float *os = p.cls_scores + (long long)bc * p.max_per_class;
    int kept = 0;
    if (n > 0) kept = nms_one_wave<NMS_R>(p, keys, dec, n, lane, ob, os);
    if (lane == 0) p.cls_counts[bc] = kept;
}

// K9c, longer lists: 1024 threads per (image, class).  Up to 1024*NMS_R candidates live in
// registers (same scheme, block-wide arg-max through LDS); beyond that the keys stay in
// global memory and dead candidates are zeroed there.
// The blocks take their (image, class) pairs from the work list the small kernel filled: most lists are short
// and never come here, and a block of 1024 threads with 80 KB of LDS that only looks at its count and leaves
// still costs its launch -- one block per pair made this kernel 0.31 ms of a 32-image step for 64 long lists.
__global__ __launch_bounds__(NMS_BIG) void post_nms_big_kernel(const PostArgs p)
{
    __shared__ u64 wbest[2][NMS_BIG / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nbig = *p.big_n;
    for (int item = blockIdx.x; item < nbig; item += gridDim.x) {
    const int bc = p.big_list[item];
    const int b = bc / p.C;
    int n = p.counts[bc];
    if (n > p.N) n = p.N;
    u64 *keys = p.keys + (long long)bc * p.N;
    const float *dec = p.dec + (long long)b * p.N * 4;
    float *ob = p.cls_boxes + (long long)bc * p.max_per_class * 4;
    float *os = p.cls_scores + (long long)bc * p.max_per_class;
    // Lists longer than the register capacity: greedy NMS only consumes candidates in
    // descending score order until max_per_class boxes are kept, so first try the top
    // scores alone -- a score histogram picks the largest score cut that leaves at most
    // CAP candidates, those are compacted into LDS and processed in registers.  If that
    // keeps max_per_class boxes the result is exact; otherwise (rare: massive ties or
    // massive suppression) everything is redone by the global-memory path below.
    constexpr int CAP = NMS_BIG * NMS_R;
    constexpr int NBIN = 4096;
    __shared__ unsigned hist[NBIN];
    __shared__ u64 chunk[CAP];
    __shared__ int chunk_n, cut_bin;
    const unsigned lo_bits = __float_as_uint(p.score_thr > 0.0f ? p.score_thr : 0.0f);
    int shift = 0;
    while (((0x3F800000u - lo_bits) >> shift) >= (unsigned)NBIN) ++shift;
    bool in_regs = n <= CAP, trial = false;
    int nn = n;                              // candidates resident in registers
    if (!in_regs) {
        for (int i = tid; i < NBIN; i += NMS_BIG) hist[i] = 0;
        if (tid == 0) chunk_n = 0;
        __syncthreads();
        for (int i = tid; i < n; i += NMS_BIG) {
            const unsigned sb = (unsigned)(keys[i] >> 32);
            atomicAdd(&hist[(sb - lo_bits) >> shift], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            unsigned cum = 0;
            int bin = NBIN;                  // keep bins >= cut_bin
            while (bin > 0 && cum + hist[bin - 1] <= (unsigned)CAP) { cum += hist[bin - 1]; --bin; }
            cut_bin = bin;
        }
        __syncthreads();
        const int cb = cut_bin;
        for (int i = tid; i < n; i += NMS_BIG) {
            const u64 k = keys[i];
            if ((int)(((unsigned)(k >> 32) - lo_bits) >> shift) >= cb) chunk[atomicAdd(&chunk_n, 1)] = k;
        }
        __syncthreads();
        nn = chunk_n;
        trial = nn >= p.max_per_class;       // fewer than the cap can never fill it: skip the trial
        in_regs = trial;
    }
    int kept = 0;
    u64 key[NMS_R];
    v4f box[NMS_R];
    unsigned alive = 0;
restart:
    if (in_regs) {
#pragma unroll
        for (int r = 0; r < NMS_R; ++r) {
            const int i = tid + NMS_BIG * r;
            const bool ok = i < nn;
            key[r] = ok ? (trial ? chunk[ok ? i : 0] : keys[ok ? i : 0]) : 0ull;
            const unsigned anchor = 0xFFFFFFFFu - (unsigned)(key[r] & 0xFFFFFFFFu);
            box[r] = *(const v4f *)(dec + (long long)(ok ? anchor : 0u) * 4);
            alive |= ok ? (1u << r) : 0u;
        }
    }
    while (kept < p.max_per_class) {        // uniform trip count: `best` is block-uniform
        u64 best = 0;
        if (in_regs) {
#pragma unroll
            for (int r = 0; r < NMS_R; ++r) {
                const u64 k = ((alive >> r) & 1u) ? key[r] : 0ull;
                best = k > best ? k : best;
            }
        } else {
            for (int i = tid; i < n; i += NMS_BIG) {
                const u64 k = keys[i];
                best = k > best ? k : best;
            }
        }
        best = wave_max_u64(best);
        const int buf = kept & 1;
        if (lane == 0) wbest[buf][wave] = best;
        __syncthreads();
#pragma unroll
        for (int w = 0; w < NMS_BIG / 64; ++w) {
            const u64 o = wbest[buf][w];
            best = o > best ? o : best;
        }
        if (best == 0) break;
        const v4f wb = *(const v4f *)(dec + (long long)(0xFFFFFFFFu - (unsigned)(best & 0xFFFFFFFFu)) * 4);
        if (tid == 0) {
            *(v4f *)(ob + kept * 4) = wb;
            os[kept] = __uint_as_float((unsigned)(best >> 32));
        }
        ++kept;
        if (in_regs) {
            unsigned kill = 0;
#pragma unroll
            for (int r = 0; r < NMS_R; ++r) {
                const bool k = (key[r] == best) | iou_greater(box[r], wb, p.iou_thr);
                kill |= k ? (1u << r) : 0u;
            }
            alive &= ~kill;
        } else {
            for (int i = tid; i < n; i += NMS_BIG) {
                const u64 k = keys[i];
                const unsigned anchor = k ? 0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFu) : 0u;   // dead: any mapped box
                const v4f bx = *(const v4f *)(dec + (long long)anchor * 4);
                const bool kill = (k == best) | iou_greater(bx, wb, p.iou_thr);
                if (k != 0 && kill) keys[i] = 0;
            }
        }
    }
    if (trial && kept < p.max_per_class) {   // block-uniform: the cut did not suffice -> exact redo
        __syncthreads();
        trial = false;
        in_regs = false;
        kept = 0;
        alive = 0;
        goto restart;
    }
    if (tid == 0) p.cls_counts[bc] = kept;
    __syncthreads();                         // the shared arrays are reused by the next pair
    }
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
size_t post_scan_bitmap_bytes(int B, int N, int C) { return (((size_t)B * N * C / 8 + 31) / 32) * 4 + 4; }

size_t post_workspace_bytes(int B, int N, int C, int mp)
{
    size_t s = 0;
    s += align_up((size_t)B * C * N * sizeof(u64));
    s += align_up((size_t)B * C * sizeof(int));
    s += align_up((size_t)B * N * 4 * sizeof(float));
    s += align_up((size_t)B * C * mp * 4 * sizeof(float));
    s += align_up((size_t)B * C * mp * sizeof(float));
    s += align_up((size_t)B * C * sizeof(int));
    s += align_up(((size_t)B * C + 1) * sizeof(int));      // work list of the long-list NMS kernel, its length first
    s += align_up(post_scan_bitmap_bytes(B, N, C));        // candidate-octet bitmap of the fused scan
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
    p.cls_counts = (int *)q;    q += align_up(B * C * sizeof(int));
    p.big_n = (int *)q;
    p.big_list = (int *)q + 1;  q += align_up((B * C + 1) * sizeof(int));
    p.scan_bits = (unsigned *)q;
}

hipError_t launch_postprocess(const PostArgs &pin, hipStream_t s)
{
    PostArgs p = pin;
    // lists up to fast_max candidates stay in one wave's registers; the caller may lower it (tests route every
    // list through the 1024-thread kernel), 0 / out of range = the default
    if (p.fast_max < 1 || p.fast_max > 64 * NMS_R) p.fast_max = pin.fast_max == -1 ? 0 : 64 * NMS_R;
    if (p.B < 1 || p.N < 1 || p.C < 1 || p.max_per_class < 1) return hipErrorInvalidValue;
    if ((long long)p.B * p.N * p.C >= (1LL << 32)) return hipErrorInvalidValue;   // 32-bit element index in the scan queue
    hipError_t e = hipMemsetAsync(p.counts, 0, (size_t)p.B * p.C * sizeof(int), s);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(p.big_n, 0, sizeof(int), s);
    if (e != hipSuccess) return e;
    const long long units = (long long)p.B * p.N * ((p.C & 3) ? p.C : p.C / 4);
    long long blocks = (units + 256 * SCAN_U - 1) / (256 * SCAN_U);
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (p.scan_fused) {          // bitmap walk: one resident round of blocks (48 KB of LDS each: three per CU)
        const long long nwords = ((long long)p.B * p.N * p.C / 8 + 31) / 32;
        blocks = (nwords + 255) / 256;
        if (blocks > 768) blocks = 768;
    }
    hipLaunchKernelGGL(post_scan_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p);
    hipLaunchKernelGGL(post_nms_small_kernel, dim3((unsigned)(p.B * p.C)), dim3(64), 0, s, p);
    const int big_blocks = p.B * p.C < 512 ? p.B * p.C : 512;     // two resident blocks per CU
    hipLaunchKernelGGL(post_nms_big_kernel, dim3((unsigned)big_blocks), dim3(NMS_BIG), 0, s, p);
    hipLaunchKernelGGL(post_pack_kernel, dim3((unsigned)p.B), dim3(256), (p.C + 1) * sizeof(int), s, p);
    if (p.scan_fused) {          // clean bitmap for the next forward's logits convolution (ordered behind this stream)
        e = hipMemsetAsync(p.scan_bits, 0, post_scan_bitmap_bytes(p.B, p.N, p.C), s);
        if (e != hipSuccess) return e;
    }
    return hipGetLastError();
}
