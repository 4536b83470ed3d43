// ssd_finalize's work: the reference's TF variables (by name) -> the packed device weights of the HIP kernels:
// physical channel order and padding (ssd_internal.h), transposed [tap][CoutPad][Cin] conv kernels (+ their split-fp16
// form for precision mode f16x3), batch-norm scale factors, slice-major depthwise packs.
#include "host.h"

#include <cmath>
#include <cstdio>
#include <cstring>

// physical position p -> logical channel (or -1 for a pad channel)
std::vector<int> phys_map(int C, int Cp)
{
    std::vector<int> m(Cp);
    for (int p = 0; p < Cp; ++p) {
        int l = ssd_logical_of_phys(p);
        m[p] = l < C ? l : -1;
    }
    return m;
}
std::vector<int> ident_map(int C, int Cp)
{
    std::vector<int> m(Cp);
    for (int p = 0; p < Cp; ++p) m[p] = p < C ? p : -1;
    return m;
}

std::vector<int> twopart_map(int D, int Dp)
{
    std::vector<int> m(2 * Dp, -1);
    for (int part = 0; part < 2; ++part)
        for (int p = 0; p < Dp; ++p) {
            const int l = ssd_logical_of_phys(p);
            if (l < D) m[part * Dp + p] = part * D + l;
        }
    return m;
}

// batch_norm_relu (layer_utils.py:5-12): sf = gamma * rsqrt(var + 1e-3)
static void bn_pack(const float *gamma, const float *beta, const float *mean, const float *var,
                    const std::vector<int> &outmap, BnHost &o)
{
    const float eps = 1e-3f;
    for (int p : outmap) {
        if (p < 0) { o.mean.push_back(0.f); o.sf.push_back(0.f); o.beta.push_back(0.f); continue; }
        o.mean.push_back(mean[p]);
        float s = 1.0f / sqrtf(var[p] + eps);
        o.sf.push_back(gamma[p] * s);
        o.beta.push_back(beta[p]);
    }
}

static int pick_tile(const ssd_handle *h, int CoutP)
{
    if (CoutP <= 32) return IGEMM_128x32;
    if (CoutP <= 64) return IGEMM_128x64;
    // an output width that 96 divides but 128 does not (the 480 = 6 x 80 class logits): no padded columns
    // (option igemm_96 = 0 keeps the padded 128-wide tiles: tests / A-B runs)
    if (ssd_opt(h, OPT_IGEMM_96, 1) && CoutP % 128 != 0 && CoutP % 96 == 0) return IGEMM_128x96;
    return IGEMM_128x128;
}

// w: HWIO [k,k,Cin_l,Cout_l] -> wt [taps][CoutPad][CinP]
int pack_conv(const ssd_handle *h, DevPool &pool, const float *w, int k, int Cin_l, int Cout_l, const std::vector<int> &inmap,
                     const std::vector<int> &outmap, ConvW &cw)
{
    cw.taps = k * k;
    cw.CinP = (int)inmap.size();
    cw.CoutP = (int)outmap.size();
    cw.tile = g_force_tile >= 0 ? g_force_tile : pick_tile(h, cw.CoutP);
    cw.CoutPad = round_up(cw.CoutP, igemm_tile_bn(cw.tile));
    cw.Cin_l = Cin_l;
    cw.Cout_l = Cout_l;
    std::vector<float> t((size_t)cw.taps * cw.CoutPad * cw.CinP, 0.0f);
    for (int tap = 0; tap < cw.taps; ++tap)
        for (int n = 0; n < cw.CoutP; ++n) {
            if (outmap[n] < 0) continue;
            float *dst = &t[((size_t)tap * cw.CoutPad + n) * cw.CinP];
            for (int p = 0; p < cw.CinP; ++p)
                if (inmap[p] >= 0) dst[p] = w[((size_t)tap * Cin_l + inmap[p]) * Cout_l + outmap[n]];
        }
    SSDCHK(pool.upload(&cw.wt, t));
    {   // igemm_lat.hip: per (tap, 16-channel tile, K-step of 32 channels) two 1-KB pieces in MFMA lane order -- lane
        // (i = l & 15, kk = l >> 4) holds the weights of channel row i for k = 4 t + kk, t = 4 hf .. 4 hf + 3
        const int KC = cw.CinP / 32, NT = cw.CoutPad / 16;
        std::vector<float> wl(t.size());
        for (int tap = 0; tap < cw.taps; ++tap)
            for (int ct = 0; ct < NT; ++ct)
                for (int kc = 0; kc < KC; ++kc)
                    for (int hf = 0; hf < 2; ++hf)
                        for (int l = 0; l < 64; ++l)
                            for (int e = 0; e < 4; ++e) {
                                const int i = l & 15, kk = l >> 4, cl = 4 * (4 * hf + e) + kk;      // logical channel within the K-step
                                const int phys = kc * 32 + ssd_phys_of_logical(cl);
                                wl[((((size_t)(tap * NT + ct) * KC + kc) * 2 + hf) * 64 + l) * 4 + e] =
                                    t[((size_t)tap * cw.CoutPad + ct * 16 + i) * cw.CinP + phys];
                            }
        SSDCHK(pool.upload(&cw.wlat, wl));
    }
    // split-fp16 rows of w * 2^s (igemm.hip "S16"): per octet of 8 input channels 8 halves h, then 8 halves
    // l = f16(w*2^s - h).  s puts the largest magnitude into [2^8, 2^9): every l of a weight within 2^-10 of
    // the largest is a normal half, and the scale is undone exactly in the epilogue (acc * 2^-s).
    float mx = 0.0f;
    for (float v : t) mx = fmaxf(mx, fabsf(v));
    int sh = 0;
    if (mx > 0.0f && std::isfinite(mx)) sh = 8 - ilogbf(mx);
    sh = sh > 100 ? 100 : (sh < -100 ? -100 : sh);
    cw.scale16 = ldexpf(1.0f, -sh);
    std::vector<uint16_t> t16(t.size() * 2);
    for (size_t r = 0; r < t.size() / 8; ++r)
        for (int e = 0; e < 8; ++e) {
            const float x = ldexpf(t[r * 8 + e], sh);
            const _Float16 hh = (_Float16)x;
            const _Float16 ll = (_Float16)(x - (float)hh);
            memcpy(&t16[r * 16 + e], &hh, 2);
            memcpy(&t16[r * 16 + 8 + e], &ll, 2);
        }
    uint16_t *d16 = nullptr;
    SSDCHK(pool.upload(&d16, t16));
    cw.wt16 = (float *)d16;
    cw.CoutPad16 = cw.CoutPad;
    if (cw.CoutP >= 256 && cw.CoutPad % 256 != 0) {
        cw.CoutPad16 = round_up(cw.CoutP, 256);
        const size_t rowh = (size_t)cw.CinP * 2;        // halves per row
        std::vector<uint16_t> w16((size_t)cw.taps * cw.CoutPad16 * rowh, 0);
        for (int tap = 0; tap < cw.taps; ++tap)
            memcpy(&w16[(size_t)tap * cw.CoutPad16 * rowh], &t16[(size_t)tap * cw.CoutPad * rowh], (size_t)cw.CoutPad * rowh * 2);
        uint16_t *dw = nullptr;
        SSDCHK(pool.upload(&dw, w16));
        cw.wt16w = (float *)dw;
    }
    return SSD_OK;
}

int upload_bn(DevPool &pool, const BnHost &b, ConvW &cw)
{
    SSDCHK(pool.upload(&cw.mean, b.mean));
    SSDCHK(pool.upload(&cw.sf, b.sf));
    return pool.upload(&cw.beta, b.beta);
}

// slice-major copy of a depthwise layer's parameters for the streaming fused kernel
int pack_dw(DevPool &pool, const std::vector<float> &w9, const std::vector<float> &mean, const std::vector<float> &sf,
                   const std::vector<float> &beta, DwW &d)
{
    if (d.Cp % 32) return SSD_OK;
    const int KC = d.Cp / 32;
    std::vector<float> p((size_t)KC * 12 * 32);
    for (int s = 0; s < KC; ++s)
        for (int c = 0; c < 32; ++c) {
            for (int t = 0; t < 9; ++t) p[((size_t)s * 12 + t) * 32 + c] = w9[(size_t)t * d.Cp + s * 32 + c];
            p[((size_t)s * 12 + 9) * 32 + c] = mean[s * 32 + c];
            p[((size_t)s * 12 + 10) * 32 + c] = sf[s * 32 + c];
            p[((size_t)s * 12 + 11) * 32 + c] = beta[s * 32 + c];
        }
    return pool.upload(&d.pack, p);
}

static const Tensor *getvar(ssd_handle *h, const std::string &n, std::initializer_list<int64_t> shape)
{
    auto it = h->vars.find(n);
    if (it == h->vars.end()) { ssd_fail(SSD_ERR_WEIGHT, "missing variable " + n); return nullptr; }
    const Tensor &t = it->second;
    std::vector<int64_t> want(shape);
    if (t.shape != want) {
        std::string s = "variable " + n + " has shape [";
        for (auto d : t.shape) s += std::to_string(d) + ",";
        s += "] expected [";
        for (auto d : want) s += std::to_string(d) + ",";
        ssd_fail(SSD_ERR_WEIGHT, s + "]");
        return nullptr;
    }
    return &t;
}

static int get_bn(ssd_handle *h, const std::string &scope, int C, const std::vector<int> &outmap, BnHost &o)
{
    const Tensor *g = getvar(h, scope + "/gamma", {C}), *b = getvar(h, scope + "/beta", {C});
    const Tensor *m = getvar(h, scope + "/moving_mean", {C}), *v = getvar(h, scope + "/moving_variance", {C});
    if (!g || !b || !m || !v) return SSD_ERR_WEIGHT;
    bn_pack(g->data.data(), b->data.data(), m->data.data(), v->data.data(), outmap, o);
    return SSD_OK;
}

// dense conv + optional BN, standard physical maps on both sides
// (in_split > 0: the input is a ShuffleNet stage output in two-part rows, in_split channels per half)
static int load_conv(ssd_handle *h, const std::string &wname, const std::string &bnscope, int k, int Cin, int Cout,
                     ConvW &cw, bool out_identity = false, int in_split = 0)
{
    const Tensor *w = getvar(h, wname, {k, k, Cin, Cout});
    if (!w) return SSD_ERR_WEIGHT;
    std::vector<int> inmap = in_split > 0 ? twopart_map(in_split, round_up(in_split, 32)) : phys_map(Cin, round_up(Cin, 32));
    std::vector<int> outmap = out_identity ? ident_map(Cout, Cout) : phys_map(Cout, round_up(Cout, 32));
    SSDCHK(pack_conv(h, h->wpool, w->data.data(), k, Cin, Cout, inmap, outmap, cw));
    if (!bnscope.empty()) {
        BnHost b;
        SSDCHK(get_bn(h, bnscope, Cout, outmap, b));
        SSDCHK(upload_bn(h->wpool, b, cw));
    }
    return SSD_OK;
}

static int load_dw(ssd_handle *h, const std::string &scope, const std::string &bnname, int C, DwW &d, int in_split = 0)
{
    const Tensor *w = getvar(h, scope + "/depthwise_weights", {3, 3, C, 1});
    if (!w) return SSD_ERR_WEIGHT;
    d.Cp = in_split > 0 ? 2 * round_up(in_split, 32) : round_up(C, 32);
    std::vector<int> map = in_split > 0 ? twopart_map(in_split, round_up(in_split, 32)) : phys_map(C, d.Cp);
    std::vector<float> t((size_t)9 * d.Cp, 0.0f);
    for (int tap = 0; tap < 9; ++tap)
        for (int p = 0; p < d.Cp; ++p)
            if (map[p] >= 0) t[(size_t)tap * d.Cp + p] = w->data[(size_t)tap * C + map[p]];
    SSDCHK(h->wpool.upload(&d.w, t));
    BnHost b;
    SSDCHK(get_bn(h, scope + "/" + bnname, C, map, b));
    SSDCHK(h->wpool.upload(&d.mean, b.mean));
    SSDCHK(h->wpool.upload(&d.sf, b.sf));
    SSDCHK(h->wpool.upload(&d.beta, b.beta));
    return pack_dw(h->wpool, t, b.mean, b.sf, b.beta, d);
}

static int load_first(ssd_handle *h, const std::string &scope, const std::string &bnname, int Cout)
{
    const Tensor *w = getvar(h, scope + "/weights", {3, 3, 3, Cout});
    if (!w) return SSD_ERR_WEIGHT;
    const int Cp = round_up(Cout, 32);
    std::vector<int> map = phys_map(Cout, Cp);
    std::vector<float> t((size_t)27 * Cp, 0.0f);
    for (int r = 0; r < 27; ++r)
        for (int p = 0; p < Cp; ++p)
            if (map[p] >= 0) t[(size_t)r * Cp + p] = w->data[(size_t)r * Cout + map[p]];
    SSDCHK(h->wpool.upload(&h->first.w, t));
    BnHost b;
    SSDCHK(get_bn(h, scope + "/" + bnname, Cout, map, b));
    SSDCHK(h->wpool.upload(&h->first.mean, b.mean));
    SSDCHK(h->wpool.upload(&h->first.sf, b.sf));
    SSDCHK(h->wpool.upload(&h->first.beta, b.beta));
    h->first.Cp = Cp;
    h->firstCp = Cp;
    return SSD_OK;
}

// mobilenet_v1.py:52-58
const int MB_STRIDE[13] = {1, 2, 1, 2, 1, 2, 1, 1, 1, 1, 1, 2, 1};
static const int MB_FILT[13] = {64, 128, 128, 256, 256, 512, 512, 512, 512, 512, 512, 1024, 1024};
static int mb_depth(int x, float dm) { int v = (int)(x * dm); return v > 8 ? v : 8; }

static int shuffle_initial_depth(float dm)
{
    // shufflenet_v2.py:22 `possibilities`, keyed by str(depth_multiplier) (model.py:29)
    if (dm == 0.5f) return 48;
    if (dm == 1.0f) return 116;
    if (dm == 1.5f) return 176;
    if (dm == 2.0f) return 224;
    return -1;
}

static int finalize_mobilenet(ssd_handle *h)
{
    const float dm = h->cfg.depth_multiplier;
    int c = mb_depth(32, dm);
    SSDCHK(load_first(h, "MobilenetV1/Conv2d_0", "BatchNorm", c));
    h->firstAct = SSD_ACT_RELU6;
    h->dw.resize(13);
    h->pw.resize(13);
    for (int i = 0; i < 13; ++i) {
        char s[96];
        snprintf(s, sizeof s, "MobilenetV1/Conv2d_%d_depthwise", i + 1);
        SSDCHK(load_dw(h, s, "BatchNorm", c, h->dw[i]));
        const int f = mb_depth(MB_FILT[i], dm);
        snprintf(s, sizeof s, "MobilenetV1/Conv2d_%d_pointwise", i + 1);
        SSDCHK(load_conv(h, std::string(s) + "/weights", std::string(s) + "/BatchNorm", 1, c, f, h->pw[i]));
        c = f;
        if (i == 4) h->c_ch[0] = c;
        if (i == 10) h->c_ch[1] = c;
        if (i == 12) h->c_ch[2] = c;
    }
    return SSD_OK;
}

// ShuffleNet layer order (execution order used by the plan):
//   per stage: unit_1 {before, dw, after, second dw, second after}, units 2..n {before, dw, after}
//   then Conv5.
static int finalize_shufflenet(ssd_handle *h)
{
    const int D0 = shuffle_initial_depth(h->cfg.depth_multiplier);
    if (D0 < 0) return ssd_fail(SSD_ERR_INVALID, "shufflenet depth_multiplier must be 0.5, 1.0, 1.5 or 2.0");
    SSDCHK(load_first(h, "ShuffleNetV2/Conv1", "batch_norm", 24));
    h->firstAct = SSD_ACT_RELU;
    const int units[3] = {4, 8, 4};
    // A stage's output is kept in TWO-PART rows [x half | y half], each half (D channels) in its own standard layout over Dp
    // physical channels: the last unit's conv1x1_after then stores its channels as one dense run per position (plan.hip), and
    // the stage's consumers -- the next stage's unit_1 (conv1x1_before, second_branch depthwise + conv1x1_after), the FPN
    // lateral, Conv5 -- get their weights packed for that layout here.  Their k-ordered accumulation still visits the
    // logical channels in ascending order (the pad channels of the x half, zeros, sit between 2 D - 1's halves).
    int cin = 24, out = D0, split = 0;          // split: channels per half of this stage's input rows (0: standard rows)
    for (int st = 0; st < 3; ++st) {
        const int D = out / 2;
        char base[64];
        snprintf(base, sizeof base, "ShuffleNetV2/Stage%d", st + 2);
        std::string u1 = std::string(base) + "/unit_1";
        ConvW cw; DwW d;
        SSDCHK(load_conv(h, u1 + "/conv1x1_before/weights", u1 + "/conv1x1_before/batch_norm", 1, cin, cin, cw, false, split)); h->pw.push_back(cw);
        SSDCHK(load_dw(h, u1 + "/depthwise", "batch_norm", cin, d)); h->dw.push_back(d);
        cw = ConvW();
        SSDCHK(load_conv(h, u1 + "/conv1x1_after/weights", u1 + "/conv1x1_after/batch_norm", 1, cin, D, cw)); h->pw.push_back(cw);
        d = DwW();
        SSDCHK(load_dw(h, u1 + "/second_branch/depthwise", "batch_norm", cin, d, split)); h->dw.push_back(d);
        cw = ConvW();
        SSDCHK(load_conv(h, u1 + "/second_branch/conv1x1_after/weights", u1 + "/second_branch/conv1x1_after/batch_norm", 1, cin, D, cw, false, split)); h->pw.push_back(cw);
        for (int j = 2; j <= units[st]; ++j) {
            std::string u = std::string(base) + "/unit_" + std::to_string(j);
            cw = ConvW();
            SSDCHK(load_conv(h, u + "/conv1x1_before/weights", u + "/conv1x1_before/batch_norm", 1, D, D, cw)); h->pw.push_back(cw);
            // this layer exists ONLY as the gathering kernel (sn_pw.hip: concat_shuffle_split folded into its loads): a width that
            // kernel does not take must fail HERE, with the layer's name, not at the first forward (the size-dependent limits -- 2 GiB
            // per stage allocation -- are select_plans' sub-batch split)
            if (!pw_gather_supports(cw.CinP, cw.CoutP, 1, cw.CinP * 4, cw.CinP * 4, cw.CoutP * 4))
                return ssd_fail(SSD_ERR_INVALID, "ssd_finalize: " + u + "/conv1x1_before (" + std::to_string(D) + " -> " + std::to_string(D) +
                                                 " channels) is not a shape of the gathering 1x1 kernel (sn_pw.hip: 32 <= padded input channels <= 512)");
            d = DwW();
            SSDCHK(load_dw(h, u + "/depthwise", "batch_norm", D, d)); h->dw.push_back(d);
            cw = ConvW();
            SSDCHK(load_conv(h, u + "/conv1x1_after/weights", u + "/conv1x1_after/batch_norm", 1, D, D, cw)); h->pw.push_back(cw);
        }
        cin = out;
        split = D;
        if (st == 0) { h->c_ch[0] = out; h->c_split[0] = D; }
        if (st == 1) { h->c_ch[1] = out; h->c_split[1] = D; }
        out *= 2;
    }
    const int fin = h->cfg.depth_multiplier == 2.0f ? 2048 : 1024;
    ConvW cw;
    SSDCHK(load_conv(h, "ShuffleNetV2/Conv5/weights", "ShuffleNetV2/Conv5/batch_norm", 1, cin, fin, cw, false, split));
    h->pw.push_back(cw);
    h->c_ch[2] = fin;
    return SSD_OK;
}

static int finalize_fpn_heads(ssd_handle *h)
{
    // feature_extractor.py:55-74
    for (int i = 0; i < 3; ++i) {
        char n[48];
        snprintf(n, sizeof n, "fpn/lateral%d/kernel", i + 3);
        SSDCHK(load_conv(h, n, "", 1, h->c_ch[i], 256, h->lat[i], false, h->c_split[i]));
    }
    for (int i = 0; i < 5; ++i) {
        char n[48], b[48];
        snprintf(n, sizeof n, "fpn/p%d/kernel", i + 3);
        snprintf(b, sizeof b, "fpn/p%d_batch_norm", i + 3);
        SSDCHK(load_conv(h, n, b, 3, i == 3 ? h->c_ch[2] : 256, 256, h->pconv[i]));
    }
    {   // p3 | p4 | p5 | p7 again behind one pointer each (same shapes: 3x3, 256 -> 256, batch norm + ReLU): at batch 1 the
        // convolutions run as ONE launch whose levels carry their own kernel offset (plan.hip, IgemmLevel::wt_off) and
        // geometry (p7: stride 2, explicit pad)
        const int src_of[4] = {0, 1, 2, 4};
        ConvW &g = h->pgroup;
        g = h->pconv[0];
        g.wt16 = g.wt16w = nullptr;            // (exact fp32 only: the split-fp16 packs carry a per-convolution scale)
        const bool have_lat = h->pconv[0].wlat && h->pconv[1].wlat && h->pconv[2].wlat && h->pconv[4].wlat;
        g.wlat = nullptr;
        const size_t wn = (size_t)g.taps * g.CoutPad * g.CinP, pn = (size_t)g.CoutP;
        for (int i = 1; i < 4; ++i)
            if (h->pconv[src_of[i]].CoutPad != g.CoutPad || h->pconv[src_of[i]].CinP != g.CinP || h->pconv[src_of[i]].CoutP != g.CoutP)
                return ssd_fail(SSD_ERR_WEIGHT, "fpn p3 / p4 / p5 / p7 kernels differ in shape");
        SSDCHK(h->wpool.alloc((void **)&g.wt, 4 * wn * sizeof(float)));
        SSDCHK(h->wpool.alloc((void **)&g.mean, 4 * pn * sizeof(float)));
        SSDCHK(h->wpool.alloc((void **)&g.sf, 4 * pn * sizeof(float)));
        SSDCHK(h->wpool.alloc((void **)&g.beta, 4 * pn * sizeof(float)));
        if (have_lat) SSDCHK(h->wpool.alloc((void **)&g.wlat, 4 * wn * sizeof(float)));
        for (int i = 0; i < 4; ++i) {
            const ConvW &src = h->pconv[src_of[i]];
            if (have_lat) HIPCHK(hipMemcpy(g.wlat + i * wn, src.wlat, wn * sizeof(float), hipMemcpyDeviceToDevice));
            HIPCHK(hipMemcpy(g.wt + i * wn, src.wt, wn * sizeof(float), hipMemcpyDeviceToDevice));
            HIPCHK(hipMemcpy(g.mean + i * pn, src.mean, pn * sizeof(float), hipMemcpyDeviceToDevice));
            HIPCHK(hipMemcpy(g.sf + i * pn, src.sf, pn * sizeof(float), hipMemcpyDeviceToDevice));
            HIPCHK(hipMemcpy(g.beta + i * pn, src.beta, pn * sizeof(float), hipMemcpyDeviceToDevice));
        }
    }
    // box_predictor.py:107-155: conv weights shared across levels, batch norm per level
    const char *nets[2] = {"box_net", "class_net"};
    const int A = 6, C = h->cfg.num_classes;
    for (int t = 0; t < 2; ++t) {
        for (int i = 0; i < 4; ++i) {
            char n[64];
            snprintf(n, sizeof n, "%s/conv3x3_%d/kernel", nets[t], i);
            ConvW &cw = h->tower[t][i];
            SSDCHK(load_conv(h, n, "", 3, 256, 256, cw));
            BnHost b;
            std::vector<int> outmap = phys_map(256, 256);
            for (int l = 3; l <= 7; ++l) {
                char s[80];
                snprintf(s, sizeof s, "%s/batch_norm_%d_for_level_%d", nets[t], i, l);
                SSDCHK(get_bn(h, s, 256, outmap, b));
            }
            SSDCHK(upload_bn(h->wpool, b, cw));
        }
        const int Cout = t == 0 ? 4 * A : C * A;
        const std::string scope = std::string(nets[t]) + (t == 0 ? "/encoded_boxes" : "/logits");
        ConvW &cw = h->final_[t];
        SSDCHK(load_conv(h, scope + "/kernel", "", 3, 256, Cout, cw, /*out_identity=*/true));
        const Tensor *bias = getvar(h, scope + "/bias", {Cout});
        if (!bias) return SSD_ERR_WEIGHT;
        std::vector<float> bpad(bias->data);
        bpad.resize((size_t)round_up(Cout, 4), 0.0f);     // the epilogue reads parameters 4 at a time
        SSDCHK(h->wpool.upload(&cw.bias, bpad));
    }
    return SSD_OK;
}

int finalize_weights(ssd_handle *h)
{
    int r = h->cfg.backbone == SSD_BACKBONE_MOBILENET ? finalize_mobilenet(h) : finalize_shufflenet(h);
    if (r == SSD_OK) r = finalize_fpn_heads(h);
    if (r != SSD_OK) {
        h->wpool.free_all();
        h->dw.clear(); h->pw.clear();
    }
    return r;
}
