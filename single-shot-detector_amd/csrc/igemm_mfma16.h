// The epilogue of the kernel built on v_mfma_f32_16x16x4_f32 (igemm_lat.hip): the TRANSPOSED
// product -- weights as the MFMA's A operand, activations as its B operand, so that D's rows are channels and its columns
// positions: lane l = (column i = l & 15, row group kk = l >> 4) holds channels 4*kk .. 4*kk+3 of position i of every 16x16
// accumulator -- four CONSECUTIVE physical channels of one position: batch norm with vector parameter loads and 16-byte
// stores straight from the accumulators, no transpose through LDS.  Same separately rounded operations as igemm.hip.
#pragma once
#include "ssd_internal.h"

typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned udivl(unsigned n, UDiv u) { return u.sh < 0 ? n : __umulhi(n, u.mag) >> u.sh; }

// acc[c][p]: channels n_first + c*16 + 4*kk + r (r = register), position m_first + p*16 + i of level L
template <int PT, int CT>
__device__ __forceinline__ void epilogue_16x16(const IgemmArgs &a, const IgemmLevel &L, const v4f (&acc)[CT][PT], int m_first, int n_first, int lane)
{
    constexpr unsigned OOB = 0x80000000u;
    const int i = lane & 15, kk = lane >> 4;
    const int M = L.M, P = L.OH * L.OW, OW = L.OW;
    const bool has_bn = a.mean != nullptr;
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(a.out + L.out_off), 0, (int)OOB, 0x00020000);
    const __amdgpu_buffer_rsrc_t o2rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)((a.out2 ? a.out2 : a.out) + L.out_off), 0, (int)OOB, 0x00020000);
    const int bstride = (int)L.out_bstride, rstride = L.out_rstride;
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        const int col = n_first + c * 16 + 4 * kk;
        const bool colok = col < a.Cout;                      // Cout % 4 == 0 (host check): the lane's four channels exist together
        v4f mean = {0.f, 0.f, 0.f, 0.f}, sf = {1.f, 1.f, 1.f, 1.f}, beta = {0.f, 0.f, 0.f, 0.f}, bias = {0.f, 0.f, 0.f, 0.f};
        if (colok && has_bn) {
            mean = *(const v4f *)(a.mean + L.param_off + col);
            sf = *(const v4f *)(a.sf + L.param_off + col);
            beta = *(const v4f *)(a.beta + L.param_off + col);
        }
        if (colok && a.bias) bias = *(const v4f *)(a.bias + L.param_off + col);
#pragma unroll
        for (int p = 0; p < PT; ++p) {
            const int m = m_first + p * 16 + i;
            const bool ok = m < M && colok;
            const int mm = m < M ? m : 0;
            const int b = (int)udivl((unsigned)mm, L.dP), pp = mm - b * P;
            const unsigned o = ok ? (unsigned)(b * bstride + pp * rstride + col) * 4u : OOB;
            const v4f raw = acc[c][p];
            v4f v = raw;
            if (has_bn) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float t = (v[e] - mean[e]) * sf[e];
                    v[e] = t + beta[e];
                }
            } else if (a.bias) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[e] + bias[e];
            } else if (a.res) {
                v4f rv = {0.f, 0.f, 0.f, 0.f};
                if (ok) {
                    const int oy = (int)udivl((unsigned)pp, L.dOW), ox = pp - oy * OW;
                    const int ch = L.OH >> 1, cw = OW >> 1;
                    rv = *(const v4f *)(a.res + L.res_off + (((long long)b * ch + (oy >> 1)) * cw + (ox >> 1)) * a.Cout + col);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = rv[e] + v[e];
            }
            // ReLU / ReLU6 as igemm.hip: `x > 0 ? x : 0` then `x < 6 ? x : 6` (a NaN comes out as 0)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (a.act >= 1) v[e] = v[e] > 0.0f ? v[e] : 0.0f;
                if (a.act == 2) v[e] = v[e] < 6.0f ? v[e] : 6.0f;
            }
            if (a.bias && a.scan_bits && o != OOB) {
                // first half of the post-processing's score filter (as igemm.hip's bias form): mark the octet of 8 consecutive
                // logits that holds a value at or above the conservative logit bound
                const float mx = __builtin_fmaxf(__builtin_fmaxf(v[0], v[1]), __builtin_fmaxf(v[2], v[3]));
                if (mx >= a.scan_lo) {
                    const unsigned oct = ((unsigned)L.out_off + (o >> 2)) >> 3;
                    atomicOr(a.scan_bits + (oct >> 5), 1u << (oct & 31));
                }
            }
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, v), orsrc, (int)o, 0, 0);
            if (a.out2) {                                     // fpn p6: relu(raw) feeds p7 (feature_extractor.py:60)
                v4f q;
#pragma unroll
                for (int e = 0; e < 4; ++e) q[e] = raw[e] > 0.0f ? raw[e] : 0.0f;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, q), o2rsrc, (int)o, 0, 0);
            }
        }
    }
}
