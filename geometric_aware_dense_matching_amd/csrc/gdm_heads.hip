// The per-point heads of GeoMatch.forward in ONE kernel (inference), gfx950.
//
// /root/reference/models/geoMatch.py:159-200 ends with nine 1x1 convolutions applied to every scene point independently:
//   rgbd_features   = feature_encoding_layer(rgbd_emb)         128 -> 128 (BN, ReLU) x3 -> 128 (no bias, no activation)
//   rgbd_normalized = normalize_feature_layer(rgbd_features)   128 -> 128 (BN, ReLU)
//   seg             = seg_layer(rgbd_emb + rgbd_normalized)    128 -> 128 (BN, ReLU) x3 -> 2
// As library calls that is 9 GEMM launches + 8 BN/activation launches + an add + a concat of the two embedding halves, each 5-20 us
// for ~1 GFLOP in total.  Here a workgroup takes 64 points of one crop and walks the whole chain: the activations of its points live
// in LDS as split-bf16 operand rows (ping-pong), every layer is W . X on the matrix cores (split-bf16 x3, fp32 accumulate -- the
// scheme of the matching / convolution kernels) with the layer's weight rows streamed straight from L2 into the A fragments, the
// folded BatchNorm + ReLU (+ the residual, + the rgbd_features output) is applied on the accumulators, and the result is split
// again into the other buffer.  Outputs: rgbd_features f32[B,128,N] and seg f32[B,c_last,N].
#include "gdm_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int HD_C = 128;                 // channels of every hidden layer
constexpr int HD_P = 64;                  // points per workgroup
constexpr int HD_ROWB = 512;              // operand row: 128 bf16 hi | 128 bf16 lo
constexpr int HD_BUF = HD_P * HD_ROWB;    // 32 KiB
constexpr int HD_MAXL = 12;

// 16-byte chunk `ch` (0-15 hi, 16-31 lo) of row r, XOR-swizzled: 16 consecutive rows' same chunk land in 16 different bank groups
__device__ __forceinline__ int hd_off(int r, int ch) { return r * HD_ROWB + (((ch & 16) | ((ch ^ r) & 15)) << 4); }

struct HeadArgs {
    const float* a;                       // f32[B, Ca, N]  first Ca input channels
    const float* b;                       // f32[B, 128 - Ca, N]
    const unsigned char* w[HD_MAXL];      // per layer: 128 packed rows of 512 B (gdm_conv1x1_pack_weight_hip(Cout, 128); rows >= Cout zero)
    const float* scale[HD_MAXL];          // folded BN (NULL = 1)
    const float* shift[HD_MAXL];          // folded BN / bias (NULL = 0)
    int act[HD_MAXL];                     // 0 none, 1 ReLU
    int nlayer;                           // hidden layers (all 128 -> 128); the last layer (128 -> c_last) follows them
    int feat_layer;                       // output of this layer (after its affine) is written to out_feat
    int res_layer;                        // the input x0 is added to the output of this layer (after its activation)
    const unsigned char* w_last;
    const float* shift_last;              // bias of the last layer (NULL = 0)
    int c_last;
    float* out_feat;                      // f32[B, 128, N] (NULL with feat_layer < 0)
    float* out_last;                      // f32[B, c_last, N] (unused with c_last == 0: the chain then ends with its hidden layers)
    int Ca, N;
    const float* ra;                      // the tensor added at res_layer, as (first rCa channels, the rest): the input itself by default
    const float* rb;
    int rCa;
};

__global__ __launch_bounds__(256, 2) void point_heads_kernel(HeadArgs A)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char rows[];       // 2 x HD_BUF
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l16 = lane & 15, kg = lane >> 4;
    const int b = blockIdx.y, n0 = blockIdx.x * HD_P;
    const int N = A.N, Ca = A.Ca, Cb = HD_C - A.Ca;

    auto x0_at = [&](int c, int p) -> float {                     // input channel c of point n0 + p (0 beyond N); no control flow
        const float* src = c < Ca ? A.a + ((long)b * Ca + c) * N : A.b + ((long)b * Cb + (c - Ca)) * N;
        const float v = src[min(n0 + p, N - 1)];
        return n0 + p < N ? v : 0.f;
    };

    auto r_at = [&](int c, int p) -> float {                      // residual source channel c of point n0 + p
        const int rCa = A.rCa;
        const float* src = c < rCa ? A.ra + ((long)b * rCa + c) * N : A.rb + ((long)b * (HD_C - rCa) + (c - rCa)) * N;
        const float v = src[min(n0 + p, N - 1)];
        return n0 + p < N ? v : 0.f;
    };

    // ---- input rows: thread = (point, 8-channel group), lanes = consecutive points (coalesced channel rows) ----
#pragma unroll
    for (int i = 0; i < HD_P * 16 / 256; ++i) {
        const int item = i * 256 + tid;
        const int p = item & (HD_P - 1), grp = item >> 6;
        unsigned hi[4], lo[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) gdm_split2(x0_at(grp * 8 + 2 * j, p), x0_at(grp * 8 + 2 * j + 1, p), hi[j], lo[j]);
        *reinterpret_cast<u32x4*>(rows + hd_off(p, grp)) = u32x4{hi[0], hi[1], hi[2], hi[3]};
        *reinterpret_cast<u32x4*>(rows + hd_off(p, 16 + grp)) = u32x4{lo[0], lo[1], lo[2], lo[3]};
    }

    // ---- a layer: the wave owns output channels [32 wave, 32 wave + 32) (two 16-row blocks) x the 64 points (four 16-column blocks) ----
    u32x4 wh[2][4], wl[2][4];                                     // A fragments: lane (row l16, k-group kg), k-step S = channels 32 S + 8 kg ..
    auto weights_load = [&](const unsigned char* w, int nblk) {
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            if (cb >= nblk) break;
            const unsigned char* r = w + (long)(32 * wave + 16 * cb + l16) * HD_ROWB;
#pragma unroll
            for (int S = 0; S < 4; ++S) {
                wh[cb][S] = *reinterpret_cast<const u32x4*>(r + (4 * S + kg) * 16);
                wl[cb][S] = *reinterpret_cast<const u32x4*>(r + 256 + (4 * S + kg) * 16);
            }
        }
    };
    f32x4 acc[2][4];
    auto mma = [&](const unsigned char* src, int nblk) {
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) acc[cb][pb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int S = 0; S < 4; ++S)
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) {
                const bf16x8 xh = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(src + hd_off(16 * pb + l16, 4 * S + kg)));
                const bf16x8 xl = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(src + hd_off(16 * pb + l16, 16 + 4 * S + kg)));
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    if (cb >= nblk) break;
                    const bf16x8 ah = __builtin_bit_cast(bf16x8, wh[cb][S]);
                    const bf16x8 al = __builtin_bit_cast(bf16x8, wl[cb][S]);
                    acc[cb][pb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, xl, acc[cb][pb], 0, 0, 0);
                    acc[cb][pb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, xh, acc[cb][pb], 0, 0, 0);
                    acc[cb][pb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, xh, acc[cb][pb], 0, 0, 0);
                }
            }
    };

    weights_load(A.w[0], 2);
#pragma unroll 1
    for (int l = 0; l < A.nlayer; ++l) {
        __syncthreads();                                           // rows of buffer l & 1 complete; the other buffer's readers are done
        const unsigned char* src = rows + (l & 1) * HD_BUF;
        unsigned char* dst = rows + ((l + 1) & 1) * HD_BUF;
        mma(src, 2);
        // the next layer's weights go out now: in flight during this epilogue and the barrier
        const bool last_next = l + 1 == A.nlayer;
        if (!last_next) weights_load(A.w[l + 1], 2);
        else if (wave == 0 && A.c_last > 0) weights_load(A.w_last, 1);
        const float* sc = A.scale[l];
        const float* sh = A.shift[l];
        const int act = A.act[l];
        const bool feat = l == A.feat_layer, res = l == A.res_layer;
        // accumulator tile: lane column l16 = point 16 pb + l16, registers r = channels 32 wave + 16 cb + 4 kg + r
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            const int c0 = 32 * wave + 16 * cb + 4 * kg;
            float s4[4], h4[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s4[r] = sc ? sc[c0 + r] : 1.f;
                h4[r] = sh ? sh[c0 + r] : 0.f;
            }
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) {
                const int p = 16 * pb + l16;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float o = acc[cb][pb][r] * s4[r] + h4[r];
                    if (act == 1) o = fmaxf(o, 0.f);
                    if (res) o = r_at(c0 + r, p) + o;              // rgbd_emb + rgbd_normalized (geoMatch.py:178)
                    v[r] = o;
                }
                if (feat && A.out_feat && n0 + p < N) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) A.out_feat[((long)b * HD_C + c0 + r) * N + n0 + p] = v[r];
                }
                unsigned h0, l0, h1, l1;
                gdm_split2(v[0], v[1], h0, l0);
                gdm_split2(v[2], v[3], h1, l1);
                // channels c0 .. c0+3: chunk c0 / 8, bytes 8 (kg & 1) .. of it
                const int ch = (32 * wave + 16 * cb) / 8 + (kg >> 1);
                *reinterpret_cast<uint2*>(dst + hd_off(p, ch) + 8 * (kg & 1)) = make_uint2(h0, h1);
                *reinterpret_cast<uint2*>(dst + hd_off(p, 16 + ch) + 8 * (kg & 1)) = make_uint2(l0, l1);
            }
        }
    }
    __syncthreads();
    // ---- the last layer: c_last <= 16 output channels, one 16-row block on wave 0 ----
    if (wave == 0 && A.c_last > 0) {
        mma(rows + (A.nlayer & 1) * HD_BUF, 1);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = 4 * kg + r;
            if (c >= A.c_last) continue;
            const float bias = A.shift_last ? A.shift_last[c] : 0.f;
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) {
                const int p = 16 * pb + l16;
                if (n0 + p < N) A.out_last[((long)b * A.c_last + c) * N + n0 + p] = acc[0][pb][r] + bias;
            }
        }
    }
}

} // namespace

/* see include/gdm.h */
extern "C" int gdm_point_heads2_hip(const float* a, const float* b, int Ca, const float* ra, const float* rb, int rCa, int B, int N, int nlayer,
                                    const void* const* w, const float* const* scale, const float* const* shift, const int* act, int feat_layer,
                                    int res_layer, const void* w_last, const float* shift_last, int c_last, float* out_feat, float* out_last,
                                    void* stream);

extern "C" int gdm_point_heads_hip(const float* a, const float* b, int Ca, int B, int N, int nlayer, const void* const* w,
                                   const float* const* scale, const float* const* shift, const int* act, int feat_layer, int res_layer,
                                   const void* w_last, const float* shift_last, int c_last, float* out_feat, float* out_last, void* stream)
{
    GDM_CHECK_ARG(w_last && out_feat && out_last && c_last >= 1, "gdm_point_heads_hip: NULL pointer / c_last=%d", c_last);
    return gdm_point_heads2_hip(a, b, Ca, a, b, Ca, B, N, nlayer, w, scale, shift, act, feat_layer, res_layer, w_last, shift_last, c_last,
                                out_feat, out_last, stream);
}

/* see include/gdm.h */
extern "C" int gdm_point_heads2_hip(const float* a, const float* b, int Ca, const float* ra, const float* rb, int rCa, int B, int N, int nlayer,
                                    const void* const* w, const float* const* scale, const float* const* shift, const int* act, int feat_layer,
                                    int res_layer, const void* w_last, const float* shift_last, int c_last, float* out_feat, float* out_last,
                                    void* stream)
{
    GDM_CHECK_ARG(a && w && scale && shift && act, "gdm_point_heads2_hip: NULL pointer");
    GDM_CHECK_ARG(Ca >= 8 && Ca <= HD_C && Ca % 8 == 0 && (b || Ca == HD_C), "gdm_point_heads2_hip: Ca=%d (a multiple of 8, with b unless 128)", Ca);
    GDM_CHECK_ARG(B >= 1 && B <= 65535 && N >= 1 && nlayer >= 1 && nlayer <= HD_MAXL && c_last >= 0 && c_last <= 16,
                  "gdm_point_heads2_hip: B=%d N=%d nlayer=%d c_last=%d", B, N, nlayer, c_last);
    GDM_CHECK_ARG(c_last == 0 || (w_last && out_last), "gdm_point_heads2_hip: last layer without weights / output");
    GDM_CHECK_ARG(feat_layer >= -1 && feat_layer < nlayer && res_layer >= -1 && res_layer < nlayer, "gdm_point_heads2_hip: bad layer index");
    GDM_CHECK_ARG(feat_layer < 0 || out_feat, "gdm_point_heads2_hip: feat_layer without out_feat");
    if (!ra) { ra = a; rb = b; rCa = Ca; }                         // the residual source defaults to the input
    GDM_CHECK_ARG(res_layer < 0 || (rCa >= 8 && rCa <= HD_C && rCa % 8 == 0 && (rb || rCa == HD_C)),
                  "gdm_point_heads2_hip: residual source rCa=%d (a multiple of 8, with rb unless 128)", rCa);
    HeadArgs A;
    A.a = a; A.b = b; A.Ca = Ca; A.N = N; A.nlayer = nlayer; A.feat_layer = feat_layer; A.res_layer = res_layer;
    A.ra = ra; A.rb = rb; A.rCa = rCa;
    for (int l = 0; l < HD_MAXL; ++l) {
        A.w[l] = l < nlayer ? (const unsigned char*)w[l] : nullptr;
        A.scale[l] = l < nlayer ? scale[l] : nullptr;
        A.shift[l] = l < nlayer ? shift[l] : nullptr;
        A.act[l] = l < nlayer ? act[l] : 0;
        GDM_CHECK_ARG(l >= nlayer || (w[l] && act[l] >= 0 && act[l] <= 1), "gdm_point_heads2_hip: layer %d: NULL weights or act=%d", l, l < nlayer ? act[l] : 0);
    }
    A.w_last = (const unsigned char*)w_last; A.shift_last = shift_last; A.c_last = c_last; A.out_feat = out_feat; A.out_last = out_last;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)point_heads_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * HD_BUF);
        attr = true;
    }
    hipLaunchKernelGGL(point_heads_kernel, dim3(gdm_cdiv(N, HD_P), B), dim3(256), 2 * HD_BUF, (hipStream_t)stream, A);
    return gdm_launch_status("point_heads_kernel");
}
