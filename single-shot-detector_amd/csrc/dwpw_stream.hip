// Depthwise 3x3 (+ batch norm + activation) -> pointwise 1x1 (+ batch norm + activation) in ONE kernel, for any
// channel count: the MobileNet block (mobilenet_v1.py:59-67, depthwise_conv.py:5-26) and the depthwise ->
// conv1x1_after pairs of shufflenet_v2.py:118-137.  The depthwise result never goes to memory.
//
//   work item   one 8x8 (stride 1) or 4x8 (stride 2) tile of output positions of one image x BN output channels;
//               a block walks a sequence of tiles (persistent), K is streamed in 32-channel slices:
//               iteration = (tile, slice)
//   input       LDS-staged 2-D patch per slice: (TY*s+2) x (TX*s+2) pixels x 32 channels (128 B = one line per
//               pixel), filled by LDS-DMA (buffer_load_dwordx4 ... lds: no staging registers); the convolution's
//               zero padding and the ragged image edge are the buffer range check (out-of-range lanes deliver
//               zeros to the LDS).  1.56 (stride 1) / 4.8 of 4 (stride 2) input lines per output position instead
//               of 4.5 / 9 loads through L1.  Depthwise weights + batch norm of the slice travel the same way.
//   pipeline    patch(it+2), weights(it+1) and the 1x1 weight slice B(it+1) are in flight while iteration `it`
//               computes: counted s_waitcnt vmcnt(N) + raw s_barrier, never vmcnt(0) in the loop, and the stream
//               continues across tiles (no per-tile prologue).  Two blocks per CU: one block's depthwise (VALU + LDS)
//               phase runs under the other's MFMA phase.
//   numerics    depthwise: per output one (ky,kx)-ordered fmaf chain from +0, batch norm (x-mean)*sf+beta separately
//               rounded (the code of depthwise_kernel); 1x1: v_mfma_f32_32x32x2_f32 over the slices in channel order
//               = the k-ordered chain of igemm.hip.  Bit-identical to the two-kernel pair and to the oracle.
//   operands    the 1x1 WEIGHTS are the MFMA's A operand and the depthwise result its B operand (fma(w, x, acc) and
//               fma(x, w, acc) are the same bits), so an accumulator tile has the output channel on its rows
//               (registers) and the position on its columns (lanes): a lane holds runs of 4 consecutive channels of
//               ONE position -- 16-byte stores straight from the accumulators, no LDS transpose.
//   epilogue    batch norm + activation per channel (parameters from LDS), then 16-B stores into dense rows
//               [M][out_rs] (out_rs >= Cout: a ShuffleNet stage's last unit writes its channels straight into the first
//               half of the stage output's rows).  ShuffleNet's concat_shuffle_split (shufflenet_v2.py:94-115) is not
//               here: producers store dense, the consumer's loads gather (sn_pw.hip).
#include "ssd_internal.h"
#include <cstdio>

typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void *lds_ptr_t;

#define WAIT_VM_CASE(n) case n: asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); break;
// s_waitcnt vmcnt(n) for a wave-uniform run-time n (the immediate must be a constant): loads, LDS-DMA and stores
// count together in issue order, so n = the number of YOUNGER vector-memory instructions allowed to stay in flight
static __device__ __forceinline__ void wait_vmcnt(int n)
{
    switch (n) {
        WAIT_VM_CASE(0) WAIT_VM_CASE(1) WAIT_VM_CASE(2) WAIT_VM_CASE(3) WAIT_VM_CASE(4) WAIT_VM_CASE(5) WAIT_VM_CASE(6) WAIT_VM_CASE(7)
        WAIT_VM_CASE(8) WAIT_VM_CASE(9) WAIT_VM_CASE(10) WAIT_VM_CASE(11) WAIT_VM_CASE(12) WAIT_VM_CASE(13) WAIT_VM_CASE(14) WAIT_VM_CASE(15)
        WAIT_VM_CASE(16) WAIT_VM_CASE(17) WAIT_VM_CASE(18) WAIT_VM_CASE(19) WAIT_VM_CASE(20) WAIT_VM_CASE(21) WAIT_VM_CASE(22) WAIT_VM_CASE(23)
        WAIT_VM_CASE(24) WAIT_VM_CASE(25) WAIT_VM_CASE(26) WAIT_VM_CASE(27) WAIT_VM_CASE(28) WAIT_VM_CASE(29) WAIT_VM_CASE(30) WAIT_VM_CASE(31)
        WAIT_VM_CASE(32) WAIT_VM_CASE(33) WAIT_VM_CASE(34) WAIT_VM_CASE(35) WAIT_VM_CASE(36) WAIT_VM_CASE(37) WAIT_VM_CASE(38) WAIT_VM_CASE(39)
        WAIT_VM_CASE(40)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

template <int STRIDE, int WN>
__global__ __launch_bounds__(256, (STRIDE == 1 && WN == 1) ? 3 : 2) void dwpw_stream_kernel(const DwPwSArgs a)
{
    constexpr int TY = STRIDE == 1 ? 8 : 4, TX = 8, BM = TY * TX;
    constexpr int WAVES_M = STRIDE == 1 ? 2 : 1, WAVES_N = 4 / WAVES_M;
    constexpr int BN = WAVES_N * WN * 32;
    constexpr int PH = (TY - 1) * STRIDE + 3, PW = (TX - 1) * STRIDE + 3, NSLOT = PH * PW;
    constexpr int NPI = (NSLOT + 7) / 8;             // patch DMA instructions (8 pixels x 128 B each): 13 / 20
    constexpr int KP = (NPI + 3) / 4;                // ... per wave, at most
    constexpr int NBW = BN / 32;                     // B-slice DMA instructions per wave (BN rows x 128 B over 4 waves)
    constexpr int PATCH_BYTES = NPI * 1024, W_BYTES = 2048, A_BYTES = BM * 128, B_BYTES = BN * 128, P_BYTES = BN * 16;
    constexpr int OFF_W = 2 * PATCH_BYTES, OFF_A = OFF_W + W_BYTES, OFF_B = OFF_A + A_BYTES, OFF_P = OFF_B + 2 * B_BYTES;
    constexpr int NSTORE = WN * 4;                   // epilogue stores per wave and tile
    constexpr unsigned OOB = 0x80000000u;
    static_assert(OFF_P + P_BYTES <= 80 * 1024, "two blocks per CU");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];     // the ONE shared array of this kernel

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // scalar: branches and wait counts on it are wave-uniform
    const int wave_m = wave / WAVES_N, wave_n = wave % WAVES_N;
    // blocks b, b+8, ... share an XCD (its L2): there, consecutive blocks take the n-tiles of one m-tile sequence
    const int xcd = blockIdx.x & 7, kk = blockIdx.x >> 3;
    const int tile_n = kk % a.n_tiles;
    const int Q = (int)gridDim.x / a.n_tiles;        // m-tile sequences (a multiple of 8)
    const int q = (kk / a.n_tiles) * 8 + xcd;
    if (q >= a.m_tiles) return;                      // whole block, before any barrier
    const int my_tiles = (a.m_tiles - q + Q - 1) / Q;
    const int K = a.K, KC = K >> 5, H = a.H, W = a.W, OW = a.OW, OH = a.OH;
    const int T = my_tiles * KC;                     // iterations of this block
    const int tiles_img = a.tiles_y * a.tiles_x;

    const __amdgpu_buffer_rsrc_t irsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.in, 0, (int)((long long)a.B * H * W * K * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.dwpack, 0, KC * 1536, 0x00020000);
    const __amdgpu_buffer_rsrc_t brsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.wt, 0, (int)((long long)a.wt_rows * K * 4), 0x00020000);   // rows beyond wt_rows read as zeros
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.out, 0, a.out_bytes, 0x00020000);

    // ---- DMA bookkeeping.  Patch instruction i (this wave issues i = wave, wave+4, ...) fills slots 8i..8i+7:
    // lane -> slot 8i + (lane >> 3) = patch pixel (r, c), 16-B chunk lane & 7 of its 32 channels.
    int prel[KP], pr[KP], pc[KP];
#pragma unroll
    for (int k = 0; k < KP; ++k) {
        const int slot = (wave + 4 * k) * 8 + (lane >> 3);
        const bool ok = wave + 4 * k < NPI && slot < NSLOT;
        // Stride 2: a patch row is kept de-interleaved in LDS -- its even columns first (9 of them), then the odd ones.  The
        // depthwise phase reads columns 2 tx + kx for eight tx at once: in patch order all eight pixels have the same parity
        // of their 128-byte slot index, i.e. they share one half of the 256-byte bank row, and a 16-lane group of a
        // ds_read_b128 (4 pixels x 4 chunks) hits every bank twice (SQ_LDS_BANK_CONFLICT = 20 % of the LDS cycles,
        // profiles/r02_f32_pmc_summary.txt).  De-interleaved, consecutive tx alternate between the halves: conflict-free.
        const int r = slot / PW, lc = slot - r * PW;             // lc: position inside the LDS row, c: patch column
        const int c = STRIDE == 2 ? (lc < (PW + 1) / 2 ? 2 * lc : 2 * (lc - (PW + 1) / 2) + 1) : lc;
        pr[k] = ok ? r : (1 << 20);                  // a row that fails every bounds check
        pc[k] = c;
        prel[k] = ((r * W + c) * K + (lane & 7) * 4) * 4;
    }
    const int np_w = (NPI - wave + 3) / 4;           // patch instructions this wave issues per iteration
    // B slice: rows n = (wave * NBW + k) * 8 + (lane >> 3) of the block's BN, LDS slot lane & 7 <- source chunk slot ^ ((n >> 1) & 7)
    int boff[NBW];
#pragma unroll
    for (int k = 0; k < NBW; ++k) {
        const int n = (wave * NBW + k) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((n >> 1) & 7);
        boff[k] = ((tile_n * BN + n) * K + chunk * 4) * 4;
    }
    // depthwise weights + batch norm of a slice: 96 chunks (9 taps, mean, sf, beta x 8 chunks), waves 0 and 1
    const int wchunk = wave * 64 + lane;
    const unsigned woff = (wave < 2 && wchunk < 96) ? (unsigned)(wchunk * 16) : OOB;

    // ---- tile cursors (scalar): the block's tiles are q, q + Q, q + 2Q, ...; a cursor steps slice by slice and, at
    // the end of a tile, by Q tiles = (qb images, qy tile rows, qx tile columns) with two carries -- no division in
    // the loop.  `cp` runs with the patch stream (two iterations ahead), `ce` with the iteration being computed.
    struct Cur { int b, ty, tx, s; };
    const int qb = Q / tiles_img, qy = (Q - qb * tiles_img) / a.tiles_x, qx = Q - qb * tiles_img - qy * a.tiles_x;
    auto advance = [&](Cur &c) {
        if (++c.s == KC) {
            c.s = 0;
            c.tx += qx;
            int cy = 0;
            if (c.tx >= a.tiles_x) { c.tx -= a.tiles_x; cy = 1; }
            c.ty += qy + cy;
            int cb = 0;
            if (c.ty >= a.tiles_y) { c.ty -= a.tiles_y; cb = 1; }
            c.b += qb + cb;
        }
    };
    Cur cp, ce;
    cp.b = q / tiles_img;
    cp.ty = (q - cp.b * tiles_img) / a.tiles_x;
    cp.tx = q - cp.b * tiles_img - cp.ty * a.tiles_x;
    cp.s = 0;
    ce = cp;
    auto dma_patch = [&](int iter, const Cur &c) {
        const int y0 = c.ty * TY * STRIDE - a.pad, x0 = c.tx * TX * STRIDE - a.pad;
        const int base = ((c.b * H + y0) * W + x0) * K * 4;
        unsigned char *dst = lds + (iter & 1) * PATCH_BYTES;
        const int so = c.s * 128;                    // (a plain local as builtin argument: see the note in dma_b)
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            if (wave + 4 * k < NPI) {                // wave-uniform
                const int iy = y0 + pr[k], ix = x0 + pc[k];
                const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
                const int off = ok ? base + prel[k] : (int)OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(irsrc, (lds_ptr_t)(dst + (wave + 4 * k) * 1024), 16, off, so, 0, 0);
            }
        }
    };
    auto dma_w = [&](int s) {
        if (wave < 2)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (lds_ptr_t)(lds + OFF_W + wave * 1024), 16, (int)woff, s * 1536, 0, 0);
    };
    auto dma_b = [&](int iter, int s) {
        unsigned char *dst = lds + OFF_B + (iter & 1) * B_BYTES + wave * NBW * 1024;
#pragma unroll
        for (int k = 0; k < NBW; ++k) {
            // (hipcc 7.2: a captured-array element passed straight to the builtin made the HOST pass drop the kernel's
            //  stub without a diagnostic -- libssd_hip.so then failed to load with an undefined symbol; keep the local)
            const int o = boff[k];
            __builtin_amdgcn_raw_ptr_buffer_load_lds(brsrc, (lds_ptr_t)(dst + k * 1024), 16, o, s * 128, 0, 0);
        }
    };

    // ---- epilogue parameters of the block's BN channels -> LDS (the block keeps its n-tile): mean | sf | beta.
    // Ordinary loads and LDS writes, all retired before the DMA pipeline starts.
    {
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

    int roff[4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
        roff[g] = (lane & 31) * 128 + (((2 * g + (lane >> 5)) ^ (((lane & 31) >> 1) & 7)) << 4);
    v16f acc[WN];
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;

    // depthwise item of this thread: channels 4*c4.., output column tx, output row(s) ty0 (+1 for stride 1)
    const int c4 = lane & 7, tx = lane >> 3;
    const int ty0 = STRIDE == 1 ? wave * 2 : wave;
    constexpr int NOUT = STRIDE == 1 ? 2 : 1;        // outputs per thread (vertically adjacent: shared patch rows)
    auto pcol = [](int col) { return STRIDE == 2 ? ((col & 1) ? (PW + 1) / 2 + (col >> 1) : (col >> 1)) : col; };   // LDS position of a patch column
    constexpr int NROW = STRIDE == 1 ? 4 : 3;
    // epilogue position of this lane: accumulator column lane & 31 = tile row wave_m * 32 + (lane & 31) = (ty, tx)
    const int ety = wave_m * 4 + ((lane & 31) >> 3), etx = lane & 7, eh = lane >> 5;

    // ---- prologue of the stream
    dma_w(0);
    dma_patch(0, cp);
    advance(cp);
    dma_b(0, 0);
    if (T > 1) { dma_patch(1, cp); advance(cp); }
    bool prev_last = false;
#ifdef SSD_DIAG    // timing ablations (results wrong): 1 no patch DMA in the loop, 2 no B / weight DMA, 4 no depthwise math, 8 no MFMA, 16 no stores
    const int abl = a.abl;
#else
    constexpr int abl = 0;
#endif
#ifdef SSD_DIAG    // libssd_hip_diag.so: per-block cycle totals of the loop's phases (thread 0 writes 8 int64 per block)
    long long ph[7] = {0, 0, 0, 0, 0, 0, 0}, tprev = 0;
    auto stamp = [&](int i) {
        if (a.ts) {
            __builtin_amdgcn_sched_barrier(0);
            const long long t = (long long)__builtin_amdgcn_s_memtime();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (i >= 0) ph[i] += t - tprev;
            tprev = t;
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    stamp(-1);
#define STAMP(i) stamp(i)
#else
#define STAMP(i)
#endif
#ifdef SSD_DIAG
    if ((abl & 32) && (blockIdx.x >= gridDim.x / 2)) { for (int z = 0; z < ((abl >> 8) & 255); ++z) __builtin_amdgcn_s_sleep(127); }   // experiment: start stagger
    if (abl & 64) { const int ph8 = (blockIdx.x >> 3) & 7; for (int z = 0; z < ph8 * ((abl >> 8) & 255); ++z) __builtin_amdgcn_s_sleep(2); }   // 8 start phases, 128-cycle units
#endif
    // a.dact / a.act are uniform: "ReLU, then min with 6 or +inf" when an activation follows, nothing otherwise -- as plain
    // selects on precomputed uniform bounds (lo = 0 or -inf, hi = 6 or +inf) instead of reading the codes per value
    const bool dact_on = a.dact >= 1, pact_on = a.act >= 1;
    const float dact_hi = a.dact == 2 ? 6.0f : __builtin_inff(), pact_hi = a.act == 2 ? 6.0f : __builtin_inff();
    for (int it = 0; it < T; ++it) {
        // landed after this wait: patch(it), B(it), weights(it) -- everything older than patch(it+1) and the stores of
        // the previous iteration's epilogue
        if (abl & 19) wait_vmcnt(0);
        else wait_vmcnt((it + 1 < T ? np_w : 0) + (prev_last ? NSTORE : 0));
        STAMP(0);                                                       // waiting for patch / weights / B of this iteration
        __builtin_amdgcn_s_barrier();                                   // (1) every wave's share has landed; MFMA(it-1) is over
        __builtin_amdgcn_sched_barrier(0);
        STAMP(1);
        const int s_next = ce.s + 1 == KC ? 0 : ce.s + 1;
        if (it + 1 < T && !(abl & 2)) dma_b(it + 1, s_next);
        __builtin_amdgcn_sched_barrier(0);
        {   // ---- depthwise 3x3 + batch norm + activation of this slice -> A image (64 or 32 rows x 128 B, swizzled)
            const unsigned char *pb_ = lds + (it & 1) * PATCH_BYTES + c4 * 16;
            const unsigned char *wb = lds + OFF_W + c4 * 16;
            v4f wv[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) wv[t] = *(const v4f *)(wb + t * 128);
            const v4f dm = *(const v4f *)(wb + 9 * 128), ds = *(const v4f *)(wb + 10 * 128), db = *(const v4f *)(wb + 11 * 128);
            v4f x[NROW][3];
#pragma unroll
            for (int rr = 0; rr < NROW; ++rr)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
                    x[rr][kx] = *(const v4f *)(pb_ + ((ty0 * STRIDE + rr) * PW + pcol(tx * STRIDE + kx)) * 128);
#pragma unroll
            for (int o = 0; o < NOUT; ++o) {
                v4f v = {0.0f, 0.0f, 0.0f, 0.0f};
                if (!(abl & 4))
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaf(x[o + ky][kx][e], wv[ky * 3 + kx][e], v[e]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float t = (v[e] - dm[e]) * ds[e];
                    v[e] = t + db[e];
                    if (dact_on) v[e] = __builtin_amdgcn_fmed3f(v[e], 0.0f, dact_hi);
                }
                const int r = (ty0 + o) * TX + tx;
                *(v4f *)(lds + OFF_A + r * 128 + ((c4 ^ ((r >> 1) & 7)) << 4)) = v;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // this wave's A rows are written, its patch / weight reads done
        STAMP(2);                                                       // B issue + depthwise phase
        __builtin_amdgcn_s_barrier();                                   // (2) A complete; patch buffer it&1 and the weight buffer are free
        __builtin_amdgcn_sched_barrier(0);
        STAMP(3);
        if (it + 1 < T && !(abl & 2)) dma_w(s_next);
        if (it + 2 < T && !(abl & 1)) { dma_patch(it + 2, cp); advance(cp); }
        __builtin_amdgcn_sched_barrier(0);
        STAMP(4);                                                       // issuing weights(it+1), patch(it+2)
        if (!(abl & 8))
        {   // ---- 1x1, transposed: acc[channel][position] += W slice (BN x 32) * A^T (32 x BM)
            const unsigned char *ab = lds + OFF_A + wave_m * 4096;
            const unsigned char *bb = lds + OFF_B + (it & 1) * B_BYTES + wave_n * WN * 4096;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const v4f af = *(const v4f *)(ab + roff[g]);
                v4f bf[WN];
#pragma unroll
                for (int j = 0; j < WN; ++j) bf[j] = *(const v4f *)(bb + j * 4096 + roff[g]);
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int j = 0; j < WN; ++j)
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[j][t], af[t], acc[j], 0, 0, 0);
            }
        }
        STAMP(5);                                                       // MFMA phase
        prev_last = ce.s == KC - 1;
        if (prev_last && !(abl & 16)) {
            // ---- epilogue.  acc[j][r]: channel (wave_n * WN + j) * 32 + (r & 3) + 8 (r >> 2) + 4 (lane >> 5) of the block's
            // BN, position = tile row wave_m * 32 + (lane & 31)
            const int oy = ce.ty * TY + ety, ox = ce.tx * TX + etx;
            const bool ok = oy < OH && ox < OW;
            const int pos = (ce.b * OH + oy) * OW + ox;
            const float *pp = (const float *)(lds + OFF_P);
#pragma unroll
            for (int j = 0; j < WN; ++j) {
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const int cl = (wave_n * WN + j) * 32 + 8 * m + 4 * eh;      // first of the lane's 4 consecutive channels
                    const v4f mean = *(const v4f *)(pp + cl), sf = *(const v4f *)(pp + BN + cl), beta = *(const v4f *)(pp + 2 * BN + cl);
                    v4f v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float x = acc[j][4 * m + e];
                        const float t = (x - mean[e]) * sf[e];
                        x = t + beta[e];
                        if (pact_on) x = __builtin_amdgcn_fmed3f(x, 0.0f, pact_hi);
                        v[e] = x;
                        acc[j][4 * m + e] = 0.0f;
                    }
                    const int n = tile_n * BN + cl;
                    const unsigned o = (ok && n < a.Cout) ? (unsigned)((pos * a.out_rs + n) * 4) : OOB;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, v), orsrc, (int)o, 0, 0);
                }
            }
        }
        advance(ce);
        STAMP(6);                                                       // epilogue (issue of the stores)
    }
#ifdef SSD_DIAG
    if (a.ts && tid == 0) {
        for (int i = 0; i < 7; ++i) a.ts[(long long)blockIdx.x * 8 + i] = ph[i];
        a.ts[(long long)blockIdx.x * 8 + 7] = T;
    }
#endif
}

int dwpws_tile_m(int stride) { return stride == 1 ? 64 : 32; }
int dwpws_tile_n(int stride, int CoutP) { return stride == 1 && CoutP <= 64 ? 64 : 128; }

template <int STRIDE, int WN>
static hipError_t launch_s(const DwPwSArgs &a, hipStream_t s)
{
    constexpr int TY = STRIDE == 1 ? 8 : 4, TX = 8;
    constexpr int WAVES_N = STRIDE == 1 ? 2 : 4, BN = WAVES_N * WN * 32;
    constexpr int NPI = (((TY - 1) * STRIDE + 3) * ((TX - 1) * STRIDE + 3) + 7) / 8;
    constexpr int lds_bytes = 2 * NPI * 1024 + 2048 + TY * TX * 128 + 2 * BN * 128 + BN * 16;
    constexpr int PER_CU = (STRIDE == 1 && WN == 1) ? 3 : 2;        // resident blocks per CU (LDS)
    static std::atomic<unsigned> attr_done{0};
    auto k = dwpw_stream_kernel<STRIDE, WN>;
    {
        hipError_t e = ssd_allow_lds((const void *)k, lds_bytes, attr_done);
        if (e != hipSuccess) return e;
    }
#ifdef SSD_DIAG
    {
        static bool said = false;
        int nb = 0;
        if (!said && hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k, 256, lds_bytes) == hipSuccess) {
            fprintf(stderr, "[diag] dwpw_stream_kernel<%d,%d>: %d bytes of LDS, occupancy API: %d blocks per CU\n", STRIDE, WN, lds_bytes, nb);
            said = true;
        }
    }
#endif
    // persistent grid: one resident round of blocks; m-tile sequences in multiples of 8 (one per XCD)
    int Q = 256 * PER_CU / a.n_tiles / 8 * 8;
    const int need = (a.m_tiles + 7) / 8 * 8;
    if (Q > need) Q = need;
    if (Q < 8) Q = 8;
    hipLaunchKernelGGL(k, dim3((unsigned)(Q * a.n_tiles)), dim3(256), lds_bytes, s, a);
    return hipGetLastError();
}

hipError_t launch_dwpw_stream(int stride, const DwPwSArgs &a, hipStream_t s)
{
    // host-side checks of everything the kernel assumes
    if (!a.in || !a.dwpack || !a.wt || !a.mean || !a.sf || !a.beta || !a.out) return hipErrorInvalidValue;
    if (stride != 1 && stride != 2) return hipErrorInvalidValue;
    if (a.K < 32 || a.K % 32 != 0 || a.B < 1 || a.H < 1 || a.W < 1 || a.OH < 1 || a.OW < 1 || a.pad < 0 || a.pad > 1) return hipErrorInvalidValue;
    if ((a.OH - 1) * stride + 2 - a.pad > a.H + 1 || (a.OW - 1) * stride + 2 - a.pad > a.W + 1) return hipErrorInvalidValue;
    const int BN = dwpws_tile_n(stride, a.Cout), TY = stride == 1 ? 8 : 4;
    if (a.Cout < 4 || a.Cout % 4 != 0 || a.wt_rows < a.Cout || a.n_tiles != (a.Cout + BN - 1) / BN || a.n_tiles > 64) return hipErrorInvalidValue;
    if (a.tiles_y != (a.OH + TY - 1) / TY || a.tiles_x != (a.OW + 7) / 8 || a.m_tiles != a.B * a.tiles_y * a.tiles_x) return hipErrorInvalidValue;
    // 32-bit byte offsets inside the buffer resources
    if ((long long)a.B * a.H * a.W * a.K * 4 >= (1LL << 31) || (long long)a.wt_rows * a.K * 4 >= (1LL << 31) || a.out_bytes <= 0) return hipErrorInvalidValue;
    if (a.out_rs < a.Cout || (long long)a.B * a.OH * a.OW * a.out_rs * 4 >= (1LL << 31)) return hipErrorInvalidValue;
    if (((long long)a.B * a.OH * a.OW - 1) * a.out_rs * 4 + (long long)a.Cout * 4 > (long long)a.out_bytes) return hipErrorInvalidValue;
    if (stride == 1) return BN == 64 ? launch_s<1, 1>(a, s) : launch_s<1, 2>(a, s);
    return launch_s<2, 1>(a, s);
}
