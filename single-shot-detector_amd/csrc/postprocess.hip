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

// K9a.  Candidates = (anchor, class) pairs with score > threshold.  A logit below a conservative bound is skipped without
// evaluating the sigmoid; the survivors are queued in LDS and drained DENSELY by all 256 threads (the double-precision
// sigmoid / exp would otherwise run with one active lane per wave).  The element index fits 32 bits (host check).
//
// Drain with block-aggregated list appends: untrained or real, the scores of a frame concentrate on a few classes (bench
// frames: 4 085 and 1 246 of 6 503 candidates in two of 80 classes), and one returning global atomic per candidate on the
// class counter serialised there -- the scan took 52 us at batch 1 whatever its loads did.  Now every candidate takes its
// slot from an LDS counter of its (image, class), one thread per used counter reserves the block's range with ONE global
// atomic, and the candidates write at base + slot.  The list ORDER was and stays unspecified (the NMS is an arg-max).
#define SCAN_U 4                       // 16-B loads in flight per thread (plain path)
#define SCAN_Q (2048 + 256 * SCAN_U * 4)
#define SCAN_TAB 1024                  // LDS counters: (images spanned by a pass) x C must fit, else direct global atomics
#define SCAN_OQ 4096                   // fused path: marked octets of a pass of 128 bitmap words
__global__ __launch_bounds__(256) void post_scan_kernel(const PostArgs p)
{
    __shared__ unsigned q_elem[SCAN_Q];
    __shared__ float q_val[SCAN_Q];
    __shared__ unsigned q_slot[SCAN_Q];
    __shared__ unsigned oq[SCAN_OQ];
    __shared__ int tab_cnt[SCAN_TAB], tab_base[SCAN_TAB];
    __shared__ int q_n, oq_n;
    const int C = p.C;
    const unsigned NC = (unsigned)p.N * (unsigned)C;
    const bool vec = (C & 3) == 0;
    const long long total = vec ? (long long)p.B * p.N * (C >> 2) : (long long)p.B * p.N * C;
    if (threadIdx.x == 0) q_n = 0;
    __syncthreads();
    auto push = [&](unsigned elem, float v) {
        const int slot = atomicAdd(&q_n, 1);
        q_elem[slot] = elem;
        q_val[slot] = v;
    };
    // e_lo / e_hi: first and last element index the queued candidates can have
    auto drain = [&](unsigned e_lo, unsigned e_hi) {
        const int n = q_n;
        if (n == 0) return;                                             // block-uniform
        const int b_first = (int)(e_lo / NC), nb = (int)(e_hi / NC) - b_first + 1;
        const int ntab = nb * C;
        const bool agg = ntab <= SCAN_TAB;
        if (agg) {
            for (int t = threadIdx.x; t < ntab; t += 256) tab_cnt[t] = 0;
            __syncthreads();
        }
        for (int t = threadIdx.x; t < n; t += 256) {
            const unsigned elem = q_elem[t];
            const float sc = sigmoid_cr(q_val[t]);
            unsigned tag = 0xFFFFFFFFu;
            if (sc > p.score_thr) {
                const int c = (int)(elem % (unsigned)C);
                const int b = (int)(elem / NC);
                if (agg) {
                    const int hidx = (b - b_first) * C + c;
                    tag = ((unsigned)hidx << 16) | (unsigned)atomicAdd(&tab_cnt[hidx], 1);
                } else {
                    tag = (unsigned)atomicAdd(&p.counts[b * C + c], 1);     // slot in the global list, directly
                }
            }
            q_slot[t] = tag;
            q_val[t] = sc;
        }
        __syncthreads();
        if (agg) {
            for (int t = threadIdx.x; t < ntab; t += 256) {
                const int cnt = tab_cnt[t];
                if (cnt) tab_base[t] = atomicAdd(&p.counts[b_first * C + t], cnt);
            }
            __syncthreads();
        }
        for (int t = threadIdx.x; t < n; t += 256) {
            const unsigned tag = q_slot[t];
            if (tag == 0xFFFFFFFFu) continue;
            const unsigned elem = q_elem[t];
            const int c = (int)(elem % (unsigned)C);
            const unsigned row = elem / (unsigned)C;
            const int i = (int)(row % (unsigned)p.N), b = (int)(row / (unsigned)p.N);
            const int slot = agg ? tab_base[tag >> 16] + (int)(tag & 0xFFFFu) : (int)tag;
            p.keys[((long long)b * C + c) * p.N + slot] = ((u64)__float_as_uint(q_val[t]) << 32) | (u64)(0xFFFFFFFFu - (unsigned)i);
            const v4f code = *(const v4f *)(p.codes + ((long long)b * p.N + i) * 4);
            const v4f anc = *(const v4f *)(p.anchors + (long long)i * 4);
            *(v4f *)(p.dec + ((long long)b * p.N + i) * 4) = decode_clip(code, anc);
        }
        __syncthreads();
        if (threadIdx.x == 0) q_n = 0;
        __syncthreads();
    };
    if (p.scan_fused) {
        // The logits convolution marked the octets that hold a candidate (igemm.hip / igemm16.hip / igemm_lat.hip epilogues):
        // read the bitmap (B*N*C/64 bytes) and the marked octets only.  Same filter, same candidates.  Dense phases per pass
        // of 128 bitmap words: (1) every thread queues the marked octets of its word, (2) the octet queue is read densely,
        // 256 octets at a time -- all logit loads of a round in flight together -- and the values above the bound go to the
        // candidate queue, (3) drain.  The words are cleared on the way: the bitmap is clean again for the next forward's
        // logits convolution without a memset.
        const long long nwords = ((long long)p.B * p.N * C / 8 + 31) / 32;
        for (long long w0 = (long long)blockIdx.x * 128; w0 < nwords; w0 += (long long)gridDim.x * 128) {
            if (threadIdx.x == 0) oq_n = 0;
            __syncthreads();
            const long long w = w0 + threadIdx.x;
            unsigned bits = (threadIdx.x < 128 && w < nwords) ? p.scan_bits[w] : 0u;
            if (bits) {
                p.scan_bits[w] = 0u;
                int k = atomicAdd(&oq_n, __builtin_popcount(bits));
                while (bits) {
                    const int bpos = __builtin_ctz(bits);
                    bits &= bits - 1;
                    oq[k++] = (unsigned)(threadIdx.x * 32 + bpos);      // octet index relative to this pass
                }
            }
            __syncthreads();
            const int no = oq_n;
            const long long last = (w0 + 128) * 256 - 1, tot_e = (long long)p.B * NC - 1;
            const unsigned e_lo = (unsigned)(w0 * 256), e_hi = (unsigned)(last < tot_e ? last : tot_e);
            for (int o0 = 0; o0 < no; o0 += 256) {                      // <= 256 * 8 candidates per round
                const int t = o0 + threadIdx.x;
                if (t < no) {
                    const long long e0 = (w0 * 32 + oq[t]) * 8;
                    const v4f x0 = *(const v4f *)(p.logits + e0), x1 = *(const v4f *)(p.logits + e0 + 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (x0[j] >= p.logit_lo) push((unsigned)(e0 + j), x0[j]);
                        if (x1[j] >= p.logit_lo) push((unsigned)(e0 + 4 + j), x1[j]);
                    }
                }
                // the decision must be block-uniform AND separated by a barrier from the next round's pushes (a fast wave's
                // atomicAdd on q_n could otherwise flip it for a slow wave: divergent barriers inside drain)
                const bool full = __syncthreads_or(q_n > 2048);
                if (full || o0 + 256 >= no) drain(e_lo, e_hi);
            }
        }
        return;
    }
    const long long stride = (long long)gridDim.x * 256 * SCAN_U;
    const long long per = vec ? 4 : 1;                                  // elements per unit
    unsigned e_lo = 0xFFFFFFFFu;                                        // first element of the oldest undrained iteration
    for (long long base = (long long)blockIdx.x * 256 * SCAN_U; base < total; base += stride) {
        if (e_lo == 0xFFFFFFFFu) e_lo = (unsigned)(base * per);
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
        // the decision must be block-uniform AND separated by a barrier from the next iteration's pushes
        const long long endu = base + 256 * SCAN_U < total ? base + 256 * SCAN_U : total;
        if (__syncthreads_or(q_n > 2048)) { drain(e_lo, (unsigned)(endu * per - 1)); e_lo = 0xFFFFFFFFu; }
    }
    // the queue may still hold candidates of iterations since the last drain: their elements lie between e_lo and the end
    __syncthreads();
    if (q_n > 0) drain(e_lo == 0xFFFFFFFFu ? 0u : e_lo, (unsigned)(total * per - 1));
}

// Wave-wide maximum of a 32-bit value on the DPP path (row shifts inside the rows of 16 lanes, then the gfx9 row broadcasts:
// lane 63 ends up with the maximum of all 64 lanes): seven cross-lane moves fused into the VALU instead of six
// ds_bpermute round trips through the LDS crossbar.  All 64 lanes must be active.
__device__ __forceinline__ unsigned wave_max_u32(unsigned v)
{
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x111, 0xf, 0xf, false));     // row_shr:1
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x112, 0xf, 0xf, false));     // row_shr:2
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x114, 0xf, 0xf, false));     // row_shr:4
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x118, 0xf, 0xf, false));     // row_shr:8: lane 15 of a row = the row's max
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x142, 0xa, 0xf, false));     // row_bcast:15 into rows 1, 3
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x143, 0xc, 0xf, false));     // row_bcast:31 into rows 2, 3
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// ... of a 64-bit key = (score bits << 32 | ~anchor): the largest score first, then, among the lanes that hold it, the
// largest low word (the smallest anchor) -- two 32-bit reductions give exactly the 64-bit maximum.  The result is
// wave-uniform.  (The first form, six butterfly steps of two ds_bpermute each, was ~0.3 us of every NMS round.)
__device__ __forceinline__ u64 wave_max_u64(u64 v)
{
    const unsigned hi = (unsigned)(v >> 32), lo = (unsigned)(v & 0xFFFFFFFFu);
    const unsigned bh = wave_max_u32(hi);
    const unsigned bl = wave_max_u32(hi == bh ? lo : 0u);
    return ((u64)bh << 32) | bl;
}

#define NMS_R 8      // candidates per thread kept in registers
#define NMS_MID 256  // threads of the per-pair kernel: lists up to 64 * NMS_R stay in wave 0, up to NMS_MID * NMS_R in the block

// The box of the round's winner without a trip to memory: exactly one (lane, slot) of the wave holds `best` (keys are
// unique: the anchor index is part of the key); that lane selects its box, a ballot names it, v_readlane broadcasts it.
// (The first form re-read dec[] from global memory every round: a dependent ~1.2 us load in each of up to 25 rounds.)
template <int R>
__device__ __forceinline__ v4f winner_box(const u64 (&key)[R], const v4f (&box)[R], u64 best, bool &found)
{
    v4f mine = {0.f, 0.f, 0.f, 0.f};
    found = false;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const bool hit = key[r] == best;
        found |= hit;
#pragma unroll
        for (int e = 0; e < 4; ++e) mine[e] = hit ? box[r][e] : mine[e];
    }
    return mine;
}
__device__ __forceinline__ v4f wave_broadcast_box(v4f mine, bool found)
{
    const unsigned long long m = __ballot(found);
    const int src = m ? (int)__builtin_ctzll(m) : 0;
    v4f wb;
#pragma unroll
    for (int e = 0; e < 4; ++e) wb[e] = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(mine[e]), src));
    return wb;
}

// Greedy NMS of one (image, class) list by ONE wavefront with R candidates per lane in registers.  keys and boxes are
// immutable; liveness is one bit per register slot.  (A version that zeroed key[r] under `key == best || iou > thr` was
// miscompiled by hipcc 7.2: the kill of the IoU branch was dropped -- keep this form branch-free.)
// kept0: boxes an earlier step already kept for this list (an exact prefix of its greedy NMS; this call continues behind them);
// kb (LDS, or null): receives the boxes kept HERE, kb[0] = the first new one.
template <int R>
__device__ __forceinline__ int nms_one_wave(const PostArgs &p, const u64 *keys, const float *dec, int n, int lane, float *ob, float *os,
                                            int kept0 = 0, v4f *kb = nullptr)
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
    int kept = kept0;
    while (kept < p.max_per_class) {
        u64 best = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const u64 k = ((alive >> r) & 1u) ? key[r] : 0ull;
            best = k > best ? k : best;
        }
        best = wave_max_u64(best);
        if (best == 0) break;
        bool found;
        const v4f mine = winner_box<R>(key, box, best, found);
        const v4f wb = wave_broadcast_box(mine, found);
        if (lane == 0) {
            *(v4f *)(ob + kept * 4) = wb;
            os[kept] = __uint_as_float((unsigned)(best >> 32));
            if (kb) kb[kept - kept0] = wb;
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

// Greedy NMS of one list by a block of NT threads, NMS_R candidates per thread in registers, one barrier per round: each
// wave's lane 0 posts (wave best, its box) in LDS, every thread takes the block's best and its box from there
// (double-buffered by round parity).  Returns the number of boxes kept (block-uniform).
template <int NT>
__device__ __forceinline__ int nms_block(const PostArgs &p, const u64 *keys, const float *dec, int n, int tid, float *ob, float *os,
                                         u64 (*wbest)[NT / 64], v4f (*wbox)[NT / 64], int kept0 = 0)
{
    const int lane = tid & 63, wave = tid >> 6;
    u64 key[NMS_R];
    v4f box[NMS_R];
    unsigned alive = 0;
#pragma unroll
    for (int r = 0; r < NMS_R; ++r) {
        const int i = tid + NT * r;
        const bool ok = i < n;
        key[r] = ok ? keys[ok ? i : 0] : 0ull;
        const unsigned anchor = 0xFFFFFFFFu - (unsigned)(key[r] & 0xFFFFFFFFu);
        box[r] = *(const v4f *)(dec + (long long)(ok ? anchor : 0u) * 4);
        alive |= ok ? (1u << r) : 0u;
    }
    int kept = kept0;
    while (kept < p.max_per_class) {        // uniform trip count: `best` is block-uniform
        u64 best = 0;
#pragma unroll
        for (int r = 0; r < NMS_R; ++r) {
            const u64 k = ((alive >> r) & 1u) ? key[r] : 0ull;
            best = k > best ? k : best;
        }
        best = wave_max_u64(best);
        bool found;
        const v4f mine = winner_box<NMS_R>(key, box, best, found);
        const v4f wv = wave_broadcast_box(mine, found & (best != 0));
        const int buf = kept & 1;
        if (lane == 0) { wbest[buf][wave] = best; wbox[buf][wave] = wv; }
        __syncthreads();
        v4f wb = wbox[buf][0];
        best = wbest[buf][0];
#pragma unroll
        for (int w = 1; w < NT / 64; ++w) {
            const u64 o = wbest[buf][w];
            const v4f ov = wbox[buf][w];
            const bool gt = o > best;
            best = gt ? o : best;
#pragma unroll
            for (int e = 0; e < 4; ++e) wb[e] = gt ? ov[e] : wb[e];
        }
        if (best == 0) break;
        if (tid == 0) {
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
    return kept;
}

// Greedy NMS of one list of ANY length by a block of NT threads with the keys in global memory: one pass per kept box that
// kills the winner's victims (zeroing their keys: the list is this forward's scratch) and takes the arg-max of the survivors
// for the next round on the way.  Exact for every n; the path of the lists neither the register forms nor the top-score
// trials of post_nms_kernel settle (beyond NMS_MID * NMS_R candidates with massive ties or massive suppression: rare, and
// then as many rounds as the class has objects).  Thread t revisits only the slots it wrote itself.
template <int NT>
__device__ __forceinline__ int nms_global(const PostArgs &p, u64 *keys, const float *dec, int n, int tid, float *ob, float *os,
                                          u64 (*wbest)[NT / 64], int kept0 = 0)
{
    const int lane = tid & 63, wave = tid >> 6;
    u64 best = 0;
    for (int i = tid; i < n; i += NT) {
        const u64 k = keys[i];
        best = k > best ? k : best;
    }
    int kept = kept0;
    while (kept < p.max_per_class) {        // uniform trip count: `best` is block-uniform after the reduction
        best = wave_max_u64(best);
        const int buf = kept & 1;
        if (lane == 0) wbest[buf][wave] = best;
        __syncthreads();
#pragma unroll
        for (int w = 0; w < NT / 64; ++w) {
            const u64 o = wbest[buf][w];
            best = o > best ? o : best;
        }
        if (best == 0) break;
        const v4f wb = *(const v4f *)(dec + (long long)(0xFFFFFFFFu - (unsigned)(best & 0xFFFFFFFFu)) * 4);
        if (tid == 0) {
            *(v4f *)(ob + kept * 4) = wb;
            os[kept] = __uint_as_float((unsigned)(best >> 32));
        }
        if (++kept >= p.max_per_class) break;
        u64 next = 0;
        for (int i = tid; i < n; i += NT) {
            const u64 k = keys[i];
            const unsigned anchor = k ? 0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFu) : 0u;   // dead: any mapped box
            const v4f bx = *(const v4f *)(dec + (long long)anchor * 4);
            const bool kill = (k == best) | iou_greater(bx, wb, p.iou_thr);
            if (k != 0 && kill) keys[i] = 0;
            const u64 live = kill ? 0ull : k;
            next = live > next ? live : next;
        }
        best = next;
    }
    return kept;
}

// The pass that lets a long list go on behind a top-score trial (post_nms_kernel): drops the trial's candidates (score bin >= cut)
// and every candidate one of the trial's kept boxes kb[0 .. nkb) suppresses, and compacts the rest in place; *new_n (LDS, zeroed by
// the caller) receives the new length.  Chunks of NMS_MID x 16 keys through registers: a chunk's loads complete (barrier) before
// any survivor is written, and a survivor's slot lies below the chunk's end, so nothing unread is overwritten.  (Register budget:
// the kernel keeps two waves per SIMD at 226 VGPRs with 16 keys per thread and chunk inlined here; 32 keys, or this pass behind a
// real call, or a loop that repeats the step, each took it past 256.)
__device__ __forceinline__ void nms_suppress_compact(u64 *keys, const float *dec, int n, int tid, const v4f *kb, int nkb, int cut, unsigned lo_bits,
                                                  int shift, float iou_thr, int *new_n)
{
    constexpr int KR = 16;
    const int lane = tid & 63;
#pragma unroll 1
    for (int base = 0; base < n; base += NMS_MID * KR) {
        u64 ck[KR];
#pragma unroll
        for (int r = 0; r < KR; ++r) {
            const int i = base + tid + NMS_MID * r;
            ck[r] = i < n ? keys[i] : 0ull;
        }
        __syncthreads();
#pragma unroll 1                    // (one group of NMS_R boxes in registers at a time)
        for (int g = 0; g < KR / NMS_R; ++g) {
            u64 kg[NMS_R];              // keys g * NMS_R .. of the chunk, picked with compile-time register indices (no scratch)
#pragma unroll
            for (int r = 0; r < NMS_R; ++r) {
                u64 v = ck[r];
#pragma unroll
                for (int q = 1; q < KR / NMS_R; ++q) v = g == q ? ck[q * NMS_R + r] : v;
                kg[r] = v;
            }
            v4f bx[NMS_R];
            unsigned rem = 0;
#pragma unroll
            for (int r = 0; r < NMS_R; ++r) {
                const u64 kk = kg[r];
                const bool below = kk != 0ull && (int)(((unsigned)(kk >> 32) - lo_bits) >> shift) < cut;
                const unsigned anchor = below ? 0xFFFFFFFFu - (unsigned)(kk & 0xFFFFFFFFu) : 0u;
                bx[r] = *(const v4f *)(dec + (long long)anchor * 4);
                rem |= below ? (1u << r) : 0u;
            }
            unsigned sup = 0;
            for (int j = 0; j < nkb; ++j) {
                const v4f wb = kb[j];
#pragma unroll
                for (int r = 0; r < NMS_R; ++r) sup |= iou_greater(bx[r], wb, iou_thr) ? (1u << r) : 0u;
            }
            const unsigned live = rem & ~sup;
#pragma unroll
            for (int r = 0; r < NMS_R; ++r) {
                const bool keepit = (live >> r) & 1u;
                const unsigned long long m = __ballot(keepit);
                int wbase = 0;
                if (lane == 0 && m) wbase = atomicAdd(new_n, (int)__popcll(m));
                wbase = __builtin_amdgcn_readfirstlane(wbase);
                if (keepit) keys[wbase + (int)__popcll(m & ((1ull << lane) - 1ull))] = kg[r];
            }
        }
        __syncthreads();
    }
}

// K9c, one block of NMS_MID threads per (image, class) pair.
//   n <= fast_max (64 * NMS_R)      wave 0 alone, everything in registers, no barrier (the other waves leave at once)
//   n <= mid_max (NMS_MID * NMS_R)  the block's four waves (nms_block)
//   longer                          top-score trials first (below), then nms_global: the keys stay in global memory
// (Measured and not adopted, round 2: ONE wave with 32 candidates per lane for lists of 513 .. 2 048 -- 0.22 -> 0.28 ms:
//  32 IoU tests per lane and round.  Four waves keep 8 per lane.)
__global__ __launch_bounds__(NMS_MID) void post_nms_kernel(const PostArgs p)
{
    __shared__ u64 wbest[2][NMS_MID / 64];
    __shared__ v4f wbox[2][NMS_MID / 64];
    const int bc = blockIdx.x;              // b*C + c
    const int b = bc / p.C;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int n = p.counts[bc];
    if (n > p.N) n = p.N;
    u64 *keys = p.keys + (long long)bc * p.N;
    const float *dec = p.dec + (long long)b * p.N * 4;
    float *ob = p.cls_boxes + (long long)bc * p.max_per_class * 4;
    float *os = p.cls_scores + (long long)bc * p.max_per_class;
    if (n > p.mid_max && p.fast_max < 64 * NMS_R) {    // a lowered hand-over point (tests): straight to the global-memory rounds,
        const int kept = nms_global<NMS_MID>(p, keys, dec, n, tid, ob, os, wbest);      // no trial
        if (tid == 0) p.cls_counts[bc] = kept;
        return;
    }
    int K = 0;                              // boxes kept so far (block-uniform): an exact prefix of the list's greedy NMS
    if (n > p.fast_max) {                   // block-uniform
    // A long list: greedy NMS consumes candidates in descending score order and stops at max_per_class kept boxes, so first
    // try the TOP scores alone -- a score histogram picks the largest score cut that leaves at most 64 * NMS_R candidates,
    // those are compacted into LDS and wave 0 runs the one-wave rounds on them.  If that keeps max_per_class boxes the result
    // is exactly the full list's (every candidate below the cut has a strictly smaller key than every one above it, and the
    // greedy order never reached them).
    // (Bench frames: ONE class holds 4 085 of a frame's 6 503 candidates; on the 1 024-thread in-register kernel its 25 rounds
    //  took 98 us -- 8 IoU tests x 16 waves on one CU per round -- and decided the post-processing's time.)
    //
    // Otherwise -- massive suppression: the top scores all sit in a few clusters -- the trial's K kept boxes are still an EXACT
    // PREFIX of the list's greedy NMS: every candidate above the cut was either kept or suppressed, and what greedy NMS does next
    // is take the candidates below the cut in descending order, each suppressed by any box kept before it.  Lists beyond the four
    // waves' registers go on from there instead of starting over (round 6): ONE pass over the list drops the candidates of the
    // trial and those the new kept boxes suppress (the keys go through registers in chunks of NMS_MID x 32 -- every load of a chunk
    // in flight together -- and the survivors are compacted in place), and the exact rounds below finish the shorter list from
    // kept0 = K: the clustered lists that used to cost one dependent pass over global memory PER ROUND (8 160 candidates in
    // 24 clusters: 389 us; 36 000: 1.4 ms) settle after one trial and one pass.  Where no trial ran (massive ties; more than
    // NMS_KB boxes per class) the rounds below run on the whole list as before.
    // (Repeating the step on the shorter list -- a loop around this block -- cost the kernel its second wave per SIMD: 342
    //  registers against 227; one step covers the clustered case the path exists for.)
    constexpr int NMS_KB = 64;
    __shared__ v4f kb[NMS_KB];
    __shared__ int new_n;
    {
        constexpr int TCAP = 64 * NMS_R, TCAP_S = 64 * 2, NBIN = 2048;
        __shared__ unsigned hist[NBIN];
        __shared__ u64 chunk[TCAP];
        __shared__ int chunk_n, cut_bin[2], trial_kept;
        const unsigned lo_bits = __float_as_uint(p.score_thr > 0.0f ? p.score_thr : 0.0f);
        int shift = 0;
        while (((0x3F800000u - lo_bits) >> shift) >= (unsigned)NBIN) ++shift;
        // The list's keys once into registers (up to 32 per thread, all loads in flight together): the histogram and the
        // compaction passes below then read registers.  (As two loops over global memory each pass was a chain of n / 256
        // dependent L2 round trips: 16 of them for the bench frames' class of 4 085 candidates, twice.)
        constexpr int KR = 32;
        u64 kreg[KR];
        const bool inreg = n <= NMS_MID * KR;                   // block-uniform
        if (inreg) {
#pragma unroll
            for (int r = 0; r < KR; ++r) {
                const int i = tid + NMS_MID * r;
                kreg[r] = i < n ? keys[i] : 0ull;               // (a key is never 0: its score exceeds a positive threshold)
            }
        }
        for (int i = tid; i < NBIN; i += NMS_MID) hist[i] = 0;
        if (tid == 0) trial_kept = -1;
        __syncthreads();
        if (inreg) {
#pragma unroll
            for (int r = 0; r < KR; ++r)
                if (kreg[r] != 0ull) atomicAdd(&hist[((unsigned)(kreg[r] >> 32) - lo_bits) >> shift], 1u);
        } else {
            for (int i = tid; i < n; i += NMS_MID) atomicAdd(&hist[((unsigned)(keys[i] >> 32) - lo_bits) >> shift], 1u);
        }
        __syncthreads();
        // cut_bin[k] = the smallest bin whose suffix count (candidates in bins >= it) is still <= cap_k, for the two caps
        // 128 and 512.  Wave 0, lane l owns bins 32 l .. 32 l + 31: lane totals, a suffix scan over the lanes, and the one lane
        // whose range holds a cut walks its 32 bins.  (One thread walking the histogram down from the top took 85 us: scores
        // crowd just above the threshold, so the walk covered nearly all 2 048 bins, a dependent LDS read each.)
        if (wave == 0) {
            static_assert(NBIN == 64 * 32, "one lane per 32 bins");
            unsigned mine = 0;
            for (int k = 0; k < 32; ++k) mine += hist[lane * 32 + k];
            unsigned above = mine;                              // becomes the count in bins >= 32 * lane
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const unsigned o = __shfl_down(above, off, 64);
                above += (lane + off < 64) ? o : 0u;
            }
            const unsigned higher = above - mine;               // count in bins >= 32 * (lane + 1)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const unsigned cap = k == 0 ? (unsigned)TCAP_S : (unsigned)TCAP;
                if (lane == 0) cut_bin[k] = above <= cap ? 0 : NBIN;
                if (higher <= cap && above > cap) {             // exactly one lane, unless every bin fits (cut 0, set above)
                    unsigned cum = higher;
                    int bin = lane * 32 + 32;
                    while (bin > lane * 32 && cum + hist[bin - 1] <= cap) { cum += hist[bin - 1]; --bin; }
                    cut_bin[k] = bin;
                }
            }
        }
        __syncthreads();
        // two trials: the top <= 128 candidates with 2 per lane (a round costs a quarter of an 8-per-lane round: on the bench
        // frames the class that holds two thirds of a frame's candidates fills its 25 boxes from them), then the top <= 512.
        // A list the block's registers hold (n <= mid_max) only runs a trial that can fill the class: its exact rounds below
        // start over anyway; a longer list runs every trial, whose kept boxes it goes on from.
        const bool progressive = p.max_per_class <= NMS_KB;
        int ran_cb = -1;                            // the cut of the last trial that ran (block-uniform), -1: none
#pragma unroll 1
        for (int k = 0; k < 2; ++k) {
            const int cb = cut_bin[k];
            if (k == 1 && cb == cut_bin[0]) break;          // the same candidates again
            if (tid == 0) chunk_n = 0;
            __syncthreads();
            if (inreg) {
#pragma unroll
                for (int r = 0; r < KR; ++r) {
                    const u64 kk = kreg[r];
                    if (kk != 0ull && (int)(((unsigned)(kk >> 32) - lo_bits) >> shift) >= cb) chunk[atomicAdd(&chunk_n, 1)] = kk;
                }
            } else {
                for (int i = tid; i < n; i += NMS_MID) {
                    const u64 kk = keys[i];
                    if ((int)(((unsigned)(kk >> 32) - lo_bits) >> shift) >= cb) chunk[atomicAdd(&chunk_n, 1)] = kk;
                }
            }
            __syncthreads();
            const int nn = chunk_n;
            // (fewer than the boxes still missing can never fill the class; cb == 0: the cut kept everything)
            if (cb > 0 && nn > 0 && (nn >= p.max_per_class - K || progressive)) {
                if (wave == 0) {
                    const int kept = k == 0 ? nms_one_wave<2>(p, chunk, dec, nn, lane, ob, os, K, kb)
                                            : nms_one_wave<NMS_R>(p, chunk, dec, nn, lane, ob, os, K, kb);
                    if (lane == 0) trial_kept = kept;
                }
                __syncthreads();
                ran_cb = cb;
                if (trial_kept >= p.max_per_class) {    // block-uniform
                    if (tid == 0) p.cls_counts[bc] = trial_kept;
                    return;
                }
            }
            __syncthreads();
        }
        if (progressive && ran_cb >= 0) {
        // ---- the list goes on behind the trial: drop the trial's candidates (bin >= ran_cb) and everything one of its kept
        // boxes kb[0 .. Kt - K) suppresses, compact the rest in place
        const int Kt = trial_kept;
        if (tid == 0) new_n = 0;
        __syncthreads();
        nms_suppress_compact(keys, dec, n, tid, kb, Kt - K, ran_cb, lo_bits, shift, p.iou_thr, &new_n);
        n = new_n;                          // (the survivors' writes are visible to the block: the pass ends in a barrier)
        K = Kt;
        __syncthreads();
        }
    }
    }
    // the exact rounds, from K kept boxes on: one wave's registers, the block's, or the list in global memory
    // (the 1 024-thread kernels that took such lists from a work list are gone: most forwards have none, and their launch --
    //  144 blocks that look at an empty list and leave -- cost its 5 us all the same)
    int kept = K;
    if (n <= p.fast_max) {                  // block-uniform
        if (wave != 0) return;
        if (n > 0) kept = nms_one_wave<NMS_R>(p, keys, dec, n, lane, ob, os, K);
    } else {
        kept = n > p.mid_max ? nms_global<NMS_MID>(p, keys, dec, n, tid, ob, os, wbest, K)
                             : nms_block<NMS_MID>(p, keys, dec, n, tid, ob, os, wbest, wbox, K);
    }
    if (tid == 0) p.cls_counts[bc] = kept;
}

__global__ __launch_bounds__(256) void post_pack_kernel(const PostArgs p)
{
    extern __shared__ int pre[];            // C+1 exclusive prefix
    const int b = blockIdx.x, C = p.C, mp = p.max_per_class, T = C * mp;
    const int tid = threadIdx.x, lane = tid & 63;
    // exclusive prefix of the per-class counts: every thread loads a count, waves scan theirs with shuffles, wave 0 adds the
    // wave totals of each 256-class chunk (one thread walking the 80 counts was 80 dependent loads: 10 us of a batch-1 forward)
    __shared__ int wtot[4];
    int carry = 0;
    for (int c0 = 0; c0 < C; c0 += 256) {
        const int c = c0 + tid;
        const int v = c < C ? p.cls_counts[b * C + c] : 0;
        int inc = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(inc, off, 64);
            inc += lane >= off ? o : 0;
        }
        if (lane == 63) wtot[tid >> 6] = inc;
        __syncthreads();
        int base = carry;
        for (int w = 0; w < (tid >> 6); ++w) base += wtot[w];
        if (c < C) pre[c] = base + inc - v;
        carry += wtot[0] + wtot[1] + wtot[2] + wtot[3];
        __syncthreads();
    }
    const long long rs = p.out_stride;                       // 0: dense tensors
    // model.py:67-68 `boxes /= box_scaler`: the batch's, or -- frames of different sizes in one batch -- this image's
    const float bsy = p.per_image_scaler ? p.scaler_img[b][0] : p.box_scaler[0], bsx = p.per_image_scaler ? p.scaler_img[b][1] : p.box_scaler[1];
    const float bsy2 = p.per_image_scaler ? p.scaler_img[b][0] : p.box_scaler[2], bsx2 = p.per_image_scaler ? p.scaler_img[b][1] : p.box_scaler[3];
    if (tid == 0) { pre[C] = carry; p.num[rs ? b * rs : (long long)b] = carry; }
    __syncthreads();
    const int total = pre[C];
    float *boxes = p.boxes + (rs ? b * rs : (long long)b * T * 4);
    float *scores = p.scores + (rs ? b * rs : (long long)b * T);
    int32_t *labels = p.labels + (rs ? b * rs : (long long)b * T);
    // rows as 16-byte stores when the box block is 16-byte aligned (always, for the packed output block of ssd.py): the block
    // may live in pinned HOST memory (Engine.detect_host at small batches), where every partial write is a PCIe transaction
    const bool vec = ((reinterpret_cast<uintptr_t>(boxes) | reinterpret_cast<uintptr_t>(p.cls_boxes)) & 15) == 0;
    for (int t = tid; t < T; t += blockDim.x) {
        const int c = t / mp, j = t - c * mp;
        if (j < pre[c + 1] - pre[c]) {
            const int d = pre[c] + j;
            const long long src = (long long)(b * C + c) * mp + j;
            if (vec) {
                const v4f q = *(const v4f *)(p.cls_boxes + src * 4);
                *(v4f *)(boxes + d * 4) = v4f{q[0] / bsy, q[1] / bsx, q[2] / bsy2, q[3] / bsx2};
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) boxes[d * 4 + k] = p.cls_boxes[src * 4 + k] / (k == 0 ? bsy : (k == 1 ? bsx : (k == 2 ? bsy2 : bsx2)));
            }
            scores[d] = p.cls_scores[src];
            labels[d] = c;
        }
    }
    for (int d = total + tid; d < T; d += blockDim.x) {
        if (vec) *(v4f *)(boxes + d * 4) = v4f{0.f, 0.f, 0.f, 0.f};
        else {
#pragma unroll
            for (int k = 0; k < 4; ++k) boxes[d * 4 + k] = 0.0f;
        }
        scores[d] = 0.0f;
        labels[d] = 0;
    }
    if (p.self_clean) {
        // the last kernel of the post-processing leaves the counters as the next forward's scan expects them (the plan
        // zeroed them once when it was built): no memset launches in front of the scan
        for (int c = threadIdx.x; c < C; c += blockDim.x) p.counts[b * C + c] = 0;
    }
}

static size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }
size_t post_scan_bitmap_bytes(int B, int N, int C) { return (((size_t)B * N * C / 8 + 31) / 32) * 4 + 4; }

size_t post_keys_bytes(int B, int N, int C) { return align_up((size_t)B * C * N * sizeof(u64)); }

size_t post_workspace_bytes(int B, int N, int C, int mp, bool with_keys)
{
    size_t s = 0;
    if (with_keys) s += post_keys_bytes(B, N, C);
    s += align_up((size_t)B * C * sizeof(int));
    s += align_up((size_t)B * N * 4 * sizeof(float));
    s += align_up((size_t)B * C * mp * 4 * sizeof(float));
    s += align_up((size_t)B * C * mp * sizeof(float));
    s += align_up((size_t)B * C * sizeof(int));
    s += align_up(post_scan_bitmap_bytes(B, N, C));        // candidate-octet bitmap of the fused scan
    return s;
}

// keys_elsewhere: the candidate keys -- the bulk of the workspace, and the one part with no state between forwards -- live in
// memory the caller lends (plan.hip: the backbone's ping-pong block, dead by then); `ws` then holds the rest only
void post_carve(PostArgs &p, void *ws, void *keys_elsewhere)
{
    unsigned char *q = (unsigned char *)ws;
    const size_t B = p.B, N = p.N, C = p.C, mp = p.max_per_class;
    if (keys_elsewhere) p.keys = (u64 *)keys_elsewhere;
    else { p.keys = (u64 *)q;   q += align_up(B * C * N * sizeof(u64)); }
    p.counts = (int *)q;        q += align_up(B * C * sizeof(int));
    p.dec = (float *)q;         q += align_up(B * N * 4 * sizeof(float));
    p.cls_boxes = (float *)q;   q += align_up(B * C * mp * 4 * sizeof(float));
    p.cls_scores = (float *)q;  q += align_up(B * C * mp * sizeof(float));
    p.cls_counts = (int *)q;    q += align_up(B * C * sizeof(int));
    p.scan_bits = (unsigned *)q;
}

hipError_t launch_postprocess(const PostArgs &pin, hipStream_t s)
{
    PostArgs p = pin;
    // lists up to fast_max candidates stay in one wave's registers, up to mid_max in one 256-thread block; the caller may
    // lower the hand-over point (tests route every list through the global-memory rounds): then longer lists go straight to
    // that kernel as before.  0 / out of range = the defaults
    if (p.fast_max < 1 || p.fast_max > 64 * NMS_R) {
        p.fast_max = pin.fast_max == -1 ? 0 : 64 * NMS_R;
        p.mid_max = pin.fast_max == -1 ? 0 : NMS_MID * NMS_R;
    } else {
        p.mid_max = p.fast_max < 64 * NMS_R ? p.fast_max : NMS_MID * NMS_R;
    }
    if (p.B < 1 || p.N < 1 || p.C < 1 || p.max_per_class < 1) return hipErrorInvalidValue;
    if ((long long)p.B * p.N * p.C >= (1LL << 32)) return hipErrorInvalidValue;   // 32-bit element index in the scan queue
    hipError_t e;
    if (!p.self_clean) {         // a caller's workspace (ssd_postprocess): nothing is known about its contents
        e = hipMemsetAsync(p.counts, 0, (size_t)p.B * p.C * sizeof(int), s);
        if (e != hipSuccess) return e;
    }
    const long long units = (long long)p.B * p.N * ((p.C & 3) ? p.C : p.C / 4);
    long long blocks = (units + 256 * SCAN_U - 1) / (256 * SCAN_U);
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (p.scan_fused) {          // bitmap walk: one resident round of blocks (72 KB of LDS each: two per CU)
        const long long nwords = ((long long)p.B * p.N * p.C / 8 + 31) / 32;
        blocks = (nwords + 255) / 256;
        if (blocks > 512) blocks = 512;
    }
    hipLaunchKernelGGL(post_scan_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p);
    hipLaunchKernelGGL(post_nms_kernel, dim3((unsigned)(p.B * p.C)), dim3(NMS_MID), 0, s, p);
    hipLaunchKernelGGL(post_pack_kernel, dim3((unsigned)p.B), dim3(256), (p.C + 1) * sizeof(int), s, p);
    return hipGetLastError();
}
