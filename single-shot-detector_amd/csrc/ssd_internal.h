// Internal declarations shared by the HIP translation units of libssd_hip.so.
// gfx950 (MI355X) only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// ---------------------------------------------------------------------------------------
// Physical channel order of every fp32 activation in HBM: NHWC with the channels of each
// group of 8 stored as logical [0,2,4,6,1,3,5,7].  The MFMA kernel reads 16 B (4 physical
// channels) per lane; lanes 0-31 take physical 0..3, lanes 32-63 physical 4..7 of an
// octet, and v_mfma_f32_32x32x2_f32 #j multiplies physical j (k=0) then physical 4+j
// (k=1): with this storage order the accumulation chain runs over logical channels
// 0,1,2,...,7 -- the order the oracle (and an HWIO kernel) defines.
// phys -> logical:  p<4 ? 2p : 2(p-4)+1 ;  logical -> phys:  l even ? l/2 : 4+(l-1)/2
// ---------------------------------------------------------------------------------------
static inline int ssd_phys_of_logical(int l) { int o = l & ~7, r = l & 7; return o + ((r & 1) ? 4 + (r >> 1) : (r >> 1)); }
static inline int ssd_logical_of_phys(int p) { int o = p & ~7, r = p & 7; return o + (r < 4 ? 2 * r : 2 * (r - 4) + 1); }

#define SSD_MAX_LEVELS 10     // 5 pyramid levels; 10 = both head towers' levels in one launch (plan.hip)

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, DEVICE): the attribute belongs to the device's copy of
// the function, so a second engine on another GPU of the same process needs its own call.  `done` is one word per
// kernel instance, one bit per device.
#include <atomic>
static inline hipError_t ssd_allow_lds(const void *fn, int bytes, std::atomic<unsigned> &done)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 32 && ((done.load(std::memory_order_relaxed) >> dev) & 1u)) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess && dev >= 0 && dev < 32) done.fetch_or(1u << dev, std::memory_order_relaxed);
    return e;
}

// floor(n / d) for 0 <= n < 2^31 without a division (Granlund-Montgomery: l = ceil(log2 d), mag = ceil(2^(31+l) / d) < 2^32,
// q = mulhi(n, mag) >> (l - 1); d == 1: sh < 0, q = n).  The compiler's run-time division is ~22 vector instructions, this
// is 2-3; the kernels' prologues and epilogues are paid in issue slots.  Host side fills, device side divides.
struct UDiv { unsigned mag; int sh; };
static inline UDiv ssd_udiv_make(unsigned d)
{
    UDiv u;
    if (d <= 1) { u.mag = 0; u.sh = -1; return u; }
    int l = 0;
    while ((1u << l) < d) ++l;
    u.mag = (unsigned)((((unsigned long long)1 << (31 + l)) + d - 1) / d);
    u.sh = l - 1;
    return u;
}

struct IgemmLevel {
    int H, W;              // input spatial size
    int OH, OW;            // output spatial size
    int M;                 // rows of this level = B*OH*OW
    int tile_begin;        // first M-tile of this level
    int param_off;         // float offset into mean/sf/beta/bias for this level
    int out_rstride;       // floats between consecutive output positions
    long long in_off;      // float offset of the level's input tensor [B,H,W,Cin]
    long long out_off;     // float offset of output (image 0, position 0)
    long long out_bstride; // floats between images in the output
    long long res_off;     // float offset of the coarse tensor [B,OH/2,OW/2,Cout] (upsample-add)
    long long wt_off;      // float offset of THIS level's kernel inside `wt` (0: one kernel shared by all levels, the head towers;
                           // grouped launches -- fpn p3 + p4 + p5 at batch 1 -- give every level its own [taps][CoutPad][Cin])
    UDiv dP, dOW;          // division by OH*OW and by OW
    int stride, pad;       // this level's convolution geometry (grouped launches mix them: fpn p7 -- stride 2, explicit pad -- rides
                           // in the p3 + p4 + p5 launch); the launch-wide IgemmArgs::stride / pad where the plan gives none
};

struct IgemmArgs {
    const float *in;
    const float *wt;       // [taps][CoutPad][Cin] physical-k order (transposed weights)
    const float *wt_lat;   // nullable: the same kernel in igemm_lat.hip's lane-order pieces ([tap][CoutPad/16][Cin/32][2][64 lanes][4])
    float *out;
    float *out2;           // optional second output = relu(raw accumulator) (fpn p6 -> p7 input)
    const float *mean, *sf, *beta;  // batch norm (nullable as a group)
    const float *bias;     // nullable
    const float *res;      // nullable: coarser map added after nearest x2 upsampling
    int B, Cin, Cout, CoutPad;
    int taps;              // 1 or 9
    int stride, pad;
    int act;
    int nlevels;
    int n_tiles_n;
    UDiv dN;               // division by n_tiles_n
    int n_major;           // igemm_lat.hip: 1 = tile index = tile_n * tiles_m + tile_m (the position tiles of one channel tile are
    int tiles_m;           //   neighbours: an XCD streams few channel tiles' weights), else tile_m * n_tiles_n + tile_n
    UDiv dM;               // division by tiles_m
    int dense_out;         // 1: out_bstride == OH*OW*out_rstride for every level
    // precision mode f16x3 (igemm.hip, "S16"): formats of the activations this launch touches
    int in_fmt;            // 1: `in` rows and `wt` rows are split-fp16 (h|l per octet), MFMA f16 x3;
                           // 2: `wt` rows split-fp16, `in` rows fp32 and split while staged (1x1, 128x128 / 64x64 tiles)
    int out_fmt;           // 1: `out` (and `out2`) rows are written split-fp16
    int res_fmt;           // 1: `res` rows are split-fp16
    float acc_scale;       // in_fmt = 1: 2^-s undoing the power-of-two scale of the packed weights
    int *flags;            // nullable: bit 0 set when an S16 output had to be clamped to the fp16 range
    // igemm16 bias form (class logits): the first half of the post-processing's threshold scan, fused into the
    // epilogue -- one bit per octet of 8 consecutive logits that holds a value >= scan_lo (postprocess.hip); nullable
    float scan_lo;
    unsigned *scan_bits;
    long long *ts;         // diagnostics (ssd_bench_conv tile 17): per-block phase timestamps, else null
    IgemmLevel lv[SSD_MAX_LEVELS];
};

// tile variants of the implicit-GEMM kernel: BM x BN
enum IgemmTile { IGEMM_128x128 = 0, IGEMM_128x64 = 1, IGEMM_128x32 = 2, IGEMM_128x256 = 3, IGEMM_256x128 = 4, IGEMM_64x64 = 5, IGEMM_128x96 = 6,
                 IGEMM_64x64D = 7 /* 64x64 with the wide tiles' two register sets: loads three K-steps ahead */ };
int igemm_tile_bm(int tile);
int igemm_tile_bn(int tile);
hipError_t launch_igemm(int tile, const IgemmArgs &a, int total_tiles_m, hipStream_t s);
// igemm_lat.hip: the latency form for small launches (v_mfma_f32_16x16x4_f32, one wave per block, no LDS): wave tile
// PT x 16 positions by CT x 16 channels; tile_begin of the levels counts PT*16-row tiles, n_tiles_n = CoutPad / (CT*16)
// ..._Wn_: n waves per block share the positions through the block's LDS image; block tile PT x 16 positions by n x CT x 16 channels
enum IgemmLatTile { IGEMM_LAT_1x1 = 20, IGEMM_LAT_1x2 = 21, IGEMM_LAT_2x1 = 22, IGEMM_LAT_2x2 = 23,
                    IGEMM_LAT_W2_1x1 = 24, IGEMM_LAT_W4_1x1 = 25, IGEMM_LAT_W4_2x1 = 26, IGEMM_LAT_W4_1x2 = 27,
                    // the one-wave 16x16 tile with its K-step interleaved (every MFMA of the one dependent chain followed by its
                    // share of the staging work), 4 / 8 / 16 K-steps of operands in flight, and the tile order that keeps the
                    // position tiles of ONE channel tile on one XCD (its weights then pass through that L2 once)
                    IGEMM_LAT_1x1_IL = 28, IGEMM_LAT_1x1_D8 = 29, IGEMM_LAT_1x1_D16 = 30, IGEMM_LAT_1x1_NM = 31, IGEMM_LAT_1x1_D8_NM = 32 };
#ifdef SSD_DIAG
static inline bool igemm_is_lat(int tile) { return tile >= IGEMM_LAT_1x1 && tile <= 36; }      // 33 .. 36: ablations of the interleaved K-step
#else
static inline bool igemm_is_lat(int tile) { return (tile >= IGEMM_LAT_1x1 && tile <= IGEMM_LAT_W4_1x2) || tile == IGEMM_LAT_1x1_D16; }   // (28, 29, 31, 32: diag build)
#endif
int igemm_lat_bm(int tile);
int igemm_lat_bn(int tile);
bool igemm_lat_n_major(int tile);
bool igemm_lat_supports(const IgemmArgs &a);
hipError_t launch_igemm_lat(int tile, const IgemmArgs &a, int total_tiles_m, hipStream_t s);
// igemm16.hip: 256 x 256 tiles, one block per CU, S16 in / S16 out with batch norm (towers, FPN outputs);
// tile_begin of the levels counts 256-row tiles, n_tiles_n = CoutPad / 256
#define IGEMM16_TILE 100
hipError_t launch_igemm16(const IgemmArgs &a, int total_tiles_m, hipStream_t s);

// depthwise -> pointwise, K streamed in 32-channel slices, LDS-DMA staged input patches (dwpw_stream.hip) ----------
struct DwPwSArgs {
    const float *in;                       // [B,H,W,K] depthwise input, physical channel order, K % 32 == 0
    const float *dwpack;                   // [K/32][12][32]: per slice 9 taps, mean, sf, beta of the depthwise layer
    const float *wt;                       // [wt_rows][K] pointwise weights (igemm B layout, taps = 1), wt_rows >= Cout
    const float *mean, *sf, *beta;         // [>= Cout] pointwise batch norm
    float *out;                            // dense rows [M][out_rs]; channels [0, Cout) of a row are written
    int out_rs;                            // floats between consecutive output rows (>= Cout)
    int B, H, W, K, OH, OW, Cout, wt_rows; // Cout: physical output channels (incl. pad channels inside octets)
    int pad;                               // depthwise pad_beg (1 for stride 1, 0 for stride 2 on even sizes)
    int dact, act;                         // activation after the depthwise / the pointwise batch norm
    int tiles_y, tiles_x, m_tiles, n_tiles;// output tiles per image (8x8 or 4x8 positions), B*tiles_y*tiles_x, ceil(Cout / BN)
    int out_bytes;                         // extent of the destination allocation from `out` (buffer range check)
    long long *ts;                         // diagnostics build only: per-block phase cycle totals [blocks][8], else null
    int abl;                               // diagnostics build only: timing-ablation mask (results wrong), else 0
};
int dwpws_tile_m(int stride);
int dwpws_tile_n(int stride, int CoutP);
hipError_t launch_dwpw_stream(int stride, const DwPwSArgs &a, hipStream_t s);

// 1x1 convolution whose input rows are gathered from several dense producer tensors (sn_pw.hip: ShuffleNet's
// conv1x1_before with concat_shuffle_split folded into its loads) ------------------------------------------------------
struct PwGArgs {
    const float *base;                     // one allocation that holds every source row
    int base_bytes;                        // its extent (< 2 GiB): the buffer range of the gather
    const int *src;                        // [K] device table: byte offset (a multiple of 4, relative to `base`) of physical input
                                           // channel k in row 0 of its producer tensor, or -1 = a zero channel
    int rs;                                // bytes between consecutive rows (positions) of every producer tensor
    const float *wt;                       // [wt_rows][K] weights (igemm B layout, taps = 1), wt_rows >= Cout
    const float *mean, *sf, *beta;         // [>= Cout] batch norm
    float *out;                            // dense rows [M][out_rs]; channels [0, Cout) of a row are written
    int out_rs, out_bytes;                 // floats between rows; extent of the destination from `out`
    int M, K, Cout, wt_rows, act;          // rows, physical input channels (K % 32 == 0), physical output channels (% 4 == 0)
    int m_tiles, n_tiles;                  // ceil(M / 64), ceil(Cout / pw_gather_tile_n(Cout))
};
int pw_gather_tile_n(int CoutP);
bool pw_gather_supports(int K, int CoutP, long long M, int rs, long long base_bytes, long long out_bytes);
hipError_t launch_pw_gather(const PwGArgs &a, hipStream_t s);

// MobileNet's first three layers in one launch (front.hip): first convolution 3x3 stride 2 on the uint8 frame (3 -> 32) ->
// depthwise 3x3 -> pointwise 32 -> 64, each with its batch norm and activation; the 32-channel tensor stays in LDS --------
// a batch of frames of DIFFERENT sizes that resize to the same [H,W] (ssd_forward_mixed): per frame its byte offset in the image
// block, its size, its resize target and the two scale factors; the table travels in kernel arguments, at most SSD_MIXED_MAX frames
#define SSD_MIXED_MAX 64
struct FrameGeom { unsigned off; int srcH, srcW, nh, nw; float hs, ws; };
struct MixedGeom { FrameGeom f[SSD_MIXED_MAX]; };
struct FrontArgs {
    const uint8_t *img;                    // [B,H,W,3] uint8 frames at the network's input size (identity resize), H and W even
    const float *w0, *m0, *s0, *b0;        // first convolution: [27][32] weights (physical output order), batch norm [32]
    const float *dwpack;                   // [12][32]: 9 taps, mean, sf, beta of the depthwise layer (DwW::pack, one slice)
    const float *wt;                       // [>= 64][32] pointwise weights (igemm B layout, taps = 1)
    const float *mean, *sf, *beta;         // [64] pointwise batch norm
    float *out;                            // [B,H/2,W/2,64] dense
    int B, H, W;
    int act0, dact, act;
    int tiles_y, tiles_x;                  // ceil((H/2) / front_tile_y()), ceil((W/2) / front_tile_x())
    int resized;                           // 1: img holds [B,srcH,srcW,3] frames that are resized to [nh,nw] and padded to [H,W] on the fly
    int srcH, srcW, nh, nw;                //    (resize_keeping_aspect_ratio; needs front_gen_supports: the width does not shrink)
    const MixedGeom *mixed;                // non-null: frames of different sizes, entries mixed_first .. + B - 1 (img = the offsets' base)
    int mixed_first;
};
bool front_supports(int B, int H, int W, int C0, int K, int Cout);
bool front_gen_supports(int B, int srcH, int srcW, int nh, int nw);
bool front_mixed_supports(const MixedGeom &mg, int first, int B, int H, int W, unsigned *end_bytes = nullptr);
int front_tile_y();
int front_tile_x();
hipError_t launch_front(const FrontArgs &q, hipStream_t s);
// ShuffleNet's first convolution (3 -> C0 physical channels: 24, or 32 with the network's zero pad channels; batch norm,
// activation) + 3x3 stride-2 max pool in one launch (front.hip): img [B,H,W,3] uint8 at the network's input size (H, W
// multiples of 4), w0 [27][C0] -> out [B,H/4,W/4,C0]
bool front_pool_supports(int B, int H, int W, int C0);
hipError_t launch_front_pool(const uint8_t *img, int B, int H, int W, const float *w0, int C0, const float *m0, const float *s0, const float *b0,
                             int act0, float *out, hipStream_t s, const int *src = nullptr /* {srcH, srcW, nh, nw}: resized frames */,
                             const MixedGeom *mixed = nullptr, int first = 0);

// elementwise / memory-bound kernels -----------------------------------------------------
// source image [B,srcH,srcW,3] is NN-resized to [nh,nw], zero padded to [H,W] (even), normalised and convolved
hipError_t launch_first_conv(const uint8_t *img, int B, int srcH, int srcW, int nh, int nw, int H, int W,
                             const float *w /*[27][Cout] phys-n*/, int Cout, const float *mean, const float *sf,
                             const float *beta, int act, float *out, hipStream_t s,
                             int variant = 0 /* resized frames: 0 auto (one lane per output pixel, K1d) | 1 the thread-per-4-channels kernel K1 */);
// ... the same for a batch of frames of DIFFERENT sizes that resize to the same [H,W] (ssd_forward_mixed): frame b of the launch is
// geometry entry first + b -- its byte offset in `img`, its size, its resize target and the two scale factors -- and the table
// travels in the kernel's arguments (no upload, nothing to keep alive): at most SSD_MIXED_MAX frames per batch
hipError_t launch_first_conv_mixed(const uint8_t *img, const MixedGeom &mg, int first, int B, int H, int W, const float *w, int Cout,
                                   const float *mean, const float *sf, const float *beta, int act, float *out, hipStream_t s, int variant = 0);
hipError_t launch_depthwise(const float *in, int B, int H, int W, int C, const float *w /*[9][C]*/,
                            int stride, int pad, int OH, int OW, const float *mean, const float *sf,
                            const float *beta, int act, float *out, hipStream_t s, int out16 = 0 /* 1: S16 rows */,
                            int *flags = nullptr /* out16: bit 0 set when a value left the fp16 range */);
hipError_t launch_maxpool(const float *in, int B, int H, int W, int C, float *out, hipStream_t s);
// x4 = up2(x5) + l4, x3 = up2(x4) + x3 (in place); x5 [B,H3/4,W3/4,C], l4 / x4 [B,H3/2,W3/2,C], x3 [B,H3,W3,C] (elementwise.hip)
hipError_t launch_fpn_merge(const float *x5, const float *l4, float *x4, float *x3, int B, int H3, int W3, int C, hipStream_t s);
// out[r][j] = (tab[2j]==0 ? x : y)[r][tab[2j+1]] (0 if tab[2j] < 0); tab is device memory
hipError_t launch_gather_channels(const float *x, int xs, const float *y, int ys, long long rows, const int *tab,
                                  int Cout, float *out, hipStream_t s);
// channel permutation logical<->physical (+ zero padding to Cpad) for the stage entry points
// to_phys: 0 physical fp32 -> logical, 1 logical -> physical fp32,
//          2 physical S16 -> logical fp32, 3 logical fp32 -> physical S16
hipError_t launch_permute_channels(const float *in, long long rows, int C, int Cpad, int to_phys, float *out,
                                   hipStream_t s, int split = 0 /* to_phys = 0: two-part physical rows, first part `split` logical channels */);
// out[r][c] = base[r * rs + src[c]] (byte offsets, src[c] < 0: 0), c < Cw: rows gathered from several dense tensors of one allocation
hipError_t launch_gather_rows(const float *base, const int *src, int rs, long long rows, int Cw, float *out, int ors, hipStream_t s);

// post-processing ------------------------------------------------------------------------
struct PostArgs {
    const float *logits, *codes, *anchors;
    int B, N, C;
    float score_thr, iou_thr;
    float logit_lo;        // conservative logit bound below which sigmoid(x) <= score_thr
    int max_per_class;
    int fast_max;          // candidate lists up to this length are processed in one wave's registers
    int mid_max;           // ... up to this length in one 256-thread block (set by launch_postprocess)
    int self_clean;        // 1: the workspace belongs to a layer plan that zeroed counts / scan_bits once; the kernels leave
                           // them zeroed for the next forward (no memset launches).  0: a caller's workspace, cleared per call
    float box_scaler[4];
    int per_image_scaler;                       // 1: image b divides by scaler_img[b] = {y, x} (a batch of frames of different sizes)
    float scaler_img[SSD_MIXED_MAX][2];
    float *boxes; int32_t *labels; float *scores; int32_t *num;
    long long out_stride;       // 32-bit words between the outputs of consecutive images, the same for all four pointers: 0 = the
                                // four dense tensors (boxes T*4, labels / scores T, num 1), else the record stride (ssd_forward_records)
    // workspace carve-up
    unsigned long long *keys;   // [B][C][N]
    int *counts;                // [B][C]
    float *dec;                 // [B][N][4] decoded+clipped boxes of candidate anchors
    float *cls_boxes;           // [B][C][max][4]
    float *cls_scores;          // [B][C][max]
    int *cls_counts;            // [B][C]
    // scan_fused: the logits convolution's epilogue (igemm16.hip) has marked in scan_bits (one bit per 8 consecutive
    // elements of [B][N][C]) every octet that holds a logit >= logit_lo; post_scan_kernel then reads the bitmap and
    // the marked octets instead of all logits.  The bitmap is cleared again at the end of the post-processing.
    unsigned *scan_bits;
    int scan_fused;
};
size_t post_workspace_bytes(int B, int N, int C, int max_per_class, bool with_keys = true);
size_t post_keys_bytes(int B, int N, int C);
size_t post_scan_bitmap_bytes(int B, int N, int C);
void post_carve(PostArgs &p, void *ws, void *keys_elsewhere = nullptr);
hipError_t launch_postprocess(const PostArgs &p, hipStream_t s);
