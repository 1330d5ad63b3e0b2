// SplineConv message + mean aggregation for the object-model (mesh) branch, gfx950.
//
// Replaces the torch_spline_conv CUDA ops behind torch_geometric.nn.SplineConv as used by
// /root/reference/models/SplineCNN.py:136-140,234-239 (dim=3, kernel_size=5, degree=1, open
// splines, aggr='mean', root weight + bias).  torch_geometric / torch_spline_conv are third-party
// and not vendored in the reference (README.md:24-25); the arithmetic follows the published
// operator (Fey et al., SplineCNN, CVPR 2018; torch_spline_conv basis/weighting):
//   for edge e = (j -> i) with pseudo-coordinate u in [0,1]^3 and each of the 2^3 corners s:
//     v_d = u_d * (ksize - degree) ; k_d = bit d of s
//     wi  = sum_d ((floor(v_d) + k_d) mod ksize) * ksize^d ;  b = prod_d (k_d ? frac(v_d) : 1-frac(v_d))
//   msg_e = sum_s b_s * (x_j W[wi_s])          out_i = mean_{e -> i} msg_e + x_i W_root + bias
//
// Formulation for the GPU: XW = X @ [W_0 | ... | W_124] is ONE dense GEMM (rocBLAS/hipBLASLt via
// torch.matmul, M x Cin x 125*Cout) and this kernel does the sparse part: per target vertex, for
// each incoming edge gather the 8 rows XW[j, wi_s, :] (512 B each, contiguous), weight, average,
// add root term and bias, optional ReLU.  Edges are in CSR form sorted by target, so the mean needs
// no atomics and is bitwise reproducible.
#include "gdm_common.h"
#include <math.h>

namespace {

template <bool RELU>
__global__ __launch_bounds__(128) void spline_aggregate_kernel(const float* __restrict__ xw,     // [M, KS^3, C]
                                                               const int32_t* __restrict__ rowptr, // [M+1]
                                                               const int32_t* __restrict__ src,    // [E] neighbour j per edge
                                                               const float* __restrict__ attr,   // [E,3] pseudo in [0,1]
                                                               const float* __restrict__ root,   // [M,C] x W_root (may be NULL)
                                                               const float* __restrict__ bias,   // [C] (may be NULL)
                                                               int C, int KS, float* __restrict__ out)
{
    const int i = blockIdx.x;
    const int nk = KS * KS * KS;
    const int e0 = rowptr[i], e1 = rowptr[i + 1];
    for (int o = threadIdx.x; o < C; o += blockDim.x) {
        float acc = 0.f;
        for (int e = e0; e < e1; ++e) {
            const int j = src[e];
            float fr[3];
            int fl[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const float v = attr[3 * e + d] * (float)(KS - 1);     // open spline, degree 1
                const float f = floorf(v);
                fl[d] = (int)f;
                fr[d] = v - f;
            }
            const float* base = xw + (long)j * nk * C + o;
            float m = 0.f;
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                int wi = 0, off = 1;
                float b = 1.f;
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    const int kd = (s >> d) & 1;
                    wi += ((fl[d] + kd) % KS) * off;
                    off *= KS;
                    b *= kd ? fr[d] : 1.f - fr[d];
                }
                m += b * base[(long)wi * C];
            }
            acc += m;
        }
        const int deg = e1 - e0;
        float r = deg > 0 ? acc / (float)deg : 0.f;
        if (root) r += root[(long)i * C + o];
        if (bias) r += bias[o];
        if (RELU) r = fmaxf(r, 0.f);
        out[(long)i * C + o] = r;
    }
}

// grad_xw[j, wi_s, o] += b_s * grad_out[i, o] / deg(i)   (grad_xw zeroed by the caller)
__global__ __launch_bounds__(128) void spline_aggregate_bwd_kernel(const float* __restrict__ go, const int32_t* __restrict__ rowptr,
                                                                   const int32_t* __restrict__ src, const float* __restrict__ attr,
                                                                   int C, int KS, float* __restrict__ gxw)
{
    const int i = blockIdx.x;
    const int nk = KS * KS * KS;
    const int e0 = rowptr[i], e1 = rowptr[i + 1];
    const int deg = e1 - e0;
    if (deg <= 0) return;
    for (int o = threadIdx.x; o < C; o += blockDim.x) {
        const float g = go[(long)i * C + o] / (float)deg;
        for (int e = e0; e < e1; ++e) {
            const int j = src[e];
            float fr[3];
            int fl[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const float v = attr[3 * e + d] * (float)(KS - 1);
                const float f = floorf(v);
                fl[d] = (int)f;
                fr[d] = v - f;
            }
            float* base = gxw + (long)j * nk * C + o;
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                int wi = 0, off = 1;
                float b = 1.f;
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    const int kd = (s >> d) & 1;
                    wi += ((fl[d] + kd) % KS) * off;
                    off *= KS;
                    b *= kd ? fr[d] : 1.f - fr[d];
                }
                if (b != 0.f) atomicAdd(&base[(long)wi * C], b * g);
            }
        }
    }
}

// Few input channels (the first mesh layer: 9 -> 128): the dense form would write and re-read an [M, 125*C] table (524 MB at
// M = 8192) for a GEMM with K = 9.  Here the message is formed directly, out_i = mean_e sum_s b_s * (x_j . W[wi_s]) + x_i . W_root + bias:
// thread = output channel, the CIN inputs of x_j are wave-uniform, W[wi][q][:] rows are contiguous over the output channel.
template <bool RELU, int CIN_MAX>
__global__ __launch_bounds__(128) void spline_direct_kernel(const float* __restrict__ x,        // [M, Cin]
                                                            const float* __restrict__ w,        // [KS^3, Cin, C]
                                                            const int32_t* __restrict__ rowptr, const int32_t* __restrict__ src,
                                                            const float* __restrict__ attr, const float* __restrict__ root_t,  // [Cin, C]
                                                            const float* __restrict__ bias, int Cin, int C, int KS, float* __restrict__ out)
{
    const int i = blockIdx.x;
    const int e0 = rowptr[i], e1 = rowptr[i + 1];
    for (int o = threadIdx.x; o < C; o += blockDim.x) {
        float acc = 0.f;
        for (int e = e0; e < e1; ++e) {
            const int j = src[e];
            float xj[CIN_MAX];
#pragma unroll
            for (int q = 0; q < CIN_MAX; ++q) xj[q] = q < Cin ? x[(long)j * Cin + q] : 0.f;
            float fr[3];
            int fl[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const float v = attr[3 * e + d] * (float)(KS - 1);     // open spline, degree 1
                const float f = floorf(v);
                fl[d] = (int)f;
                fr[d] = v - f;
            }
            float m = 0.f;
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                int wi = 0, off = 1;
                float b = 1.f;
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    const int kd = (s >> d) & 1;
                    wi += ((fl[d] + kd) % KS) * off;
                    off *= KS;
                    b *= kd ? fr[d] : 1.f - fr[d];
                }
                const float* wr = w + (long)wi * Cin * C + o;
                float dot = 0.f;
#pragma unroll
                for (int q = 0; q < CIN_MAX; ++q)
                    if (q < Cin) dot = fmaf(xj[q], wr[(long)q * C], dot);
                m += b * dot;
            }
            acc += m;
        }
        const int deg = e1 - e0;
        float r = deg > 0 ? acc / (float)deg : 0.f;
        if (root_t) {
            float dot = 0.f;
            for (int q = 0; q < Cin; ++q) dot = fmaf(x[(long)i * Cin + q], root_t[(long)q * C + o], dot);
            r += dot;
        }
        if (bias) r += bias[o];
        if (RELU) r = fmaxf(r, 0.f);
        out[(long)i * C + o] = r;
    }
}

// The four channels [o, o+4) of vertex i as their half of a 16-byte group of the packed split-bf16 operand the next layer's grouped GEMM
// reads (conv_pack_act_kernel's layout for a [1, C, 1, M] map: planes of 3 x (M + 2) zero-bordered pixels, hi plane q / lo plane 16 + q of
// every 128-channel chunk, 8 channels = 16 bytes per pixel): the pack launch between two SplineConv layers is then not needed.
__device__ __forceinline__ void spline_store_packed(unsigned char* __restrict__ out_pk, int M, int i, int o, const float4 r)
{
    const long plane = 3L * (M + 2);
    const int chunk = o >> 7, q = (o & 127) >> 3, half = (o & 7) >> 2;
    unsigned hi0, lo0, hi1, lo1;
    gdm_split2(r.x, r.y, hi0, lo0);
    gdm_split2(r.z, r.w, hi1, lo1);
    unsigned char* p = out_pk + (((long)chunk * 32 + q) * plane + (M + 2) + i + 1) * 16 + half * 8;
    *reinterpret_cast<uint2*>(p) = make_uint2(hi0, hi1);
    *reinterpret_cast<uint2*>(p + 16 * plane * 16) = make_uint2(lo0, lo1);
}

// The same with FOUR output channels per thread (C % 4 == 0): the kernel is bound by its weight loads (4 edges x 8 corners x Cin rows
// per output channel, L2 hits), and a 16-byte load costs the memory path what a 4-byte one does -- a quarter of the load instructions
// and of the threads.  A 128-thread block owns 512 / C vertices.
template <bool RELU, int CIN_MAX>
__global__ __launch_bounds__(128) void spline_direct_vec_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                const int32_t* __restrict__ rowptr, const int32_t* __restrict__ src,
                                                                const float* __restrict__ attr, const float* __restrict__ root_t,
                                                                const float* __restrict__ bias, int M, int Cin, int C, int KS,
                                                                float* __restrict__ out, float* __restrict__ out_t,
                                                                unsigned char* __restrict__ out_pk)
{
    const int tpv = C / 4;                                        // threads per vertex
    const int i = blockIdx.x * (128 / tpv) + threadIdx.x / tpv;
    const int o = (threadIdx.x % tpv) * 4;
    if (i >= M) return;
    const int e0 = rowptr[i], e1 = rowptr[i + 1];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int e = e0; e < e1; ++e) {
        const int j = src[e];
        float xj[CIN_MAX];
#pragma unroll
        for (int q = 0; q < CIN_MAX; ++q) xj[q] = q < Cin ? x[(long)j * Cin + q] : 0.f;
        float fr[3];
        int fl[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float v = attr[3 * e + d] * (float)(KS - 1);     // open spline, degree 1
            const float f = floorf(v);
            fl[d] = (int)f;
            fr[d] = v - f;
        }
        float4 m = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            int wi = 0, off = 1;
            float b = 1.f;
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const int kd = (s >> d) & 1;
                wi += ((fl[d] + kd) % KS) * off;
                off *= KS;
                b *= kd ? fr[d] : 1.f - fr[d];
            }
            const float* wr = w + (long)wi * Cin * C + o;
            float4 dot = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int q = 0; q < CIN_MAX; ++q)
                if (q < Cin) {
                    const float4 wv = *reinterpret_cast<const float4*>(wr + (long)q * C);
                    dot.x = fmaf(xj[q], wv.x, dot.x); dot.y = fmaf(xj[q], wv.y, dot.y);
                    dot.z = fmaf(xj[q], wv.z, dot.z); dot.w = fmaf(xj[q], wv.w, dot.w);
                }
            m.x += b * dot.x; m.y += b * dot.y; m.z += b * dot.z; m.w += b * dot.w;
        }
        acc.x += m.x; acc.y += m.y; acc.z += m.z; acc.w += m.w;
    }
    const int deg = e1 - e0;
    float4 r = deg > 0 ? make_float4(acc.x / (float)deg, acc.y / (float)deg, acc.z / (float)deg, acc.w / (float)deg)
                       : make_float4(0.f, 0.f, 0.f, 0.f);
    if (root_t) {
        float4 dot = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int q = 0; q < Cin; ++q) {
            const float xv = x[(long)i * Cin + q];
            const float4 wv = *reinterpret_cast<const float4*>(root_t + (long)q * C + o);
            dot.x = fmaf(xv, wv.x, dot.x); dot.y = fmaf(xv, wv.y, dot.y); dot.z = fmaf(xv, wv.z, dot.z); dot.w = fmaf(xv, wv.w, dot.w);
        }
        r.x += dot.x; r.y += dot.y; r.z += dot.z; r.w += dot.w;
    }
    if (bias) {
        const float4 bv = *reinterpret_cast<const float4*>(bias + o);
        r.x += bv.x; r.y += bv.y; r.z += bv.z; r.w += bv.w;
    }
    if (RELU) { r.x = fmaxf(r.x, 0.f); r.y = fmaxf(r.y, 0.f); r.z = fmaxf(r.z, 0.f); r.w = fmaxf(r.w, 0.f); }
    if (out) *reinterpret_cast<float4*>(out + (long)i * C + o) = r;
    if (out_t) {                                                  // channel-major copy [C, M]: what the next layer's GEMM and 1x1 layers read
        out_t[(long)(o + 0) * M + i] = r.x; out_t[(long)(o + 1) * M + i] = r.y;
        out_t[(long)(o + 2) * M + i] = r.z; out_t[(long)(o + 3) * M + i] = r.w;
    }
    if (out_pk) spline_store_packed(out_pk, M, i, o, r);
}

// Aggregation for the edge-grouped form: Y f32[R,128-wide rows of C] holds x_j . W[wi] for every (source, kernel index) pair that
// some edge needs (gdm_gemm_grouped_hip); pos i32[E,8] = row of Y for (edge, corner), basis f32[E,8] the corner weights.
template <bool RELU>
__global__ __launch_bounds__(128) void spline_pairs_aggregate_kernel(const float* __restrict__ Y, const int32_t* __restrict__ rowptr,
                                                                     const int32_t* __restrict__ pos, const float* __restrict__ basis,
                                                                     const float* __restrict__ root, const float* __restrict__ bias,
                                                                     int C, float* __restrict__ out)
{
    const int i = blockIdx.x;
    const int e0 = rowptr[i], e1 = rowptr[i + 1];
    for (int o = threadIdx.x; o < C; o += blockDim.x) {
        float acc = 0.f;
        for (int e = e0; e < e1; ++e) {
            float m = 0.f;
#pragma unroll
            for (int s = 0; s < 8; ++s) m += basis[8 * e + s] * Y[(long)pos[8 * e + s] * C + o];
            acc += m;
        }
        const int deg = e1 - e0;
        float r = deg > 0 ? acc / (float)deg : 0.f;
        if (root) r += root[(long)i * C + o];
        if (bias) r += bias[o];
        if (RELU) r = fmaxf(r, 0.f);
        out[(long)i * C + o] = r;
    }
}

// four output channels per thread (C % 4 == 0): 16-byte loads of the Y rows, a block owns 512 / C vertices
template <bool RELU>
__global__ __launch_bounds__(128) void spline_pairs_aggregate_vec_kernel(const float* __restrict__ Y, const int32_t* __restrict__ rowptr,
                                                                         const int32_t* __restrict__ pos, const float* __restrict__ basis,
                                                                         const float* __restrict__ root, const float* __restrict__ bias,
                                                                         int M, int C, float* __restrict__ out, float* __restrict__ out_t,
                                                                         unsigned char* __restrict__ out_pk)
{
    const int tpv = C / 4;
    const int i = blockIdx.x * (128 / tpv) + threadIdx.x / tpv;
    const int o = (threadIdx.x % tpv) * 4;
    if (i >= M) return;
    const int e0 = rowptr[i], e1 = rowptr[i + 1];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int e = e0; e < e1; ++e) {
        float4 m = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const float b = basis[8 * e + s];
            const float4 y = *reinterpret_cast<const float4*>(Y + (long)pos[8 * e + s] * C + o);
            m.x += b * y.x; m.y += b * y.y; m.z += b * y.z; m.w += b * y.w;
        }
        acc.x += m.x; acc.y += m.y; acc.z += m.z; acc.w += m.w;
    }
    const int deg = e1 - e0;
    float4 r = deg > 0 ? make_float4(acc.x / (float)deg, acc.y / (float)deg, acc.z / (float)deg, acc.w / (float)deg)
                       : make_float4(0.f, 0.f, 0.f, 0.f);
    if (root) {
        const float4 v = *reinterpret_cast<const float4*>(root + (long)i * C + o);
        r.x += v.x; r.y += v.y; r.z += v.z; r.w += v.w;
    }
    if (bias) {
        const float4 v = *reinterpret_cast<const float4*>(bias + o);
        r.x += v.x; r.y += v.y; r.z += v.z; r.w += v.w;
    }
    if (RELU) { r.x = fmaxf(r.x, 0.f); r.y = fmaxf(r.y, 0.f); r.z = fmaxf(r.z, 0.f); r.w = fmaxf(r.w, 0.f); }
    if (out) *reinterpret_cast<float4*>(out + (long)i * C + o) = r;
    if (out_t) {
        out_t[(long)(o + 0) * M + i] = r.x; out_t[(long)(o + 1) * M + i] = r.y;
        out_t[(long)(o + 2) * M + i] = r.z; out_t[(long)(o + 3) * M + i] = r.w;
    }
    if (out_pk) spline_store_packed(out_pk, M, i, o, r);
}

} // namespace

extern "C" int gdm_spline_pairs_aggregate3_hip(const float* Y, const int32_t* rowptr, const int32_t* pos, const float* basis,
                                               const float* root, const float* bias, int M, int C, int relu, float* out, float* out_t,
                                               void* out_packed, void* stream)
{
    unsigned char* out_pk = (unsigned char*)out_packed;
    GDM_CHECK_ARG(Y && rowptr && pos && basis && (out || out_t), "gdm_spline_pairs_aggregate_hip: NULL pointer");
    GDM_CHECK_ARG(!out_pk || (C % 128 == 0 && C <= 512), "gdm_spline_pairs_aggregate3_hip: the packed output needs C = 128, 256 or 512");
    GDM_CHECK_ARG(M >= 1 && C >= 1, "gdm_spline_pairs_aggregate_hip: bad shape");
    if (C % 4 == 0 && C <= 512 && 512 % C == 0 && ((uintptr_t)Y & 15) == 0 && ((uintptr_t)out & 15) == 0 && (!root || ((uintptr_t)root & 15) == 0) &&
        (!bias || ((uintptr_t)bias & 15) == 0)) {
        const dim3 grid(gdm_cdiv(M, 128 / (C / 4)));
        if (relu)
            hipLaunchKernelGGL(spline_pairs_aggregate_vec_kernel<true>, grid, dim3(128), 0, (hipStream_t)stream, Y, rowptr, pos, basis, root, bias, M, C, out, out_t, out_pk);
        else
            hipLaunchKernelGGL(spline_pairs_aggregate_vec_kernel<false>, grid, dim3(128), 0, (hipStream_t)stream, Y, rowptr, pos, basis, root, bias, M, C, out, out_t, out_pk);
        return gdm_launch_status("spline_pairs_aggregate_vec_kernel");
    }
    GDM_CHECK_ARG(out && !out_t && !out_pk, "gdm_spline_pairs_aggregate2_hip: the channel-major / packed outputs need C %% 4 == 0, 512 %% C == 0 and 16-byte aligned buffers");
    if (relu)
        hipLaunchKernelGGL(spline_pairs_aggregate_kernel<true>, dim3(M), dim3(128), 0, (hipStream_t)stream, Y, rowptr, pos, basis, root, bias, C, out);
    else
        hipLaunchKernelGGL(spline_pairs_aggregate_kernel<false>, dim3(M), dim3(128), 0, (hipStream_t)stream, Y, rowptr, pos, basis, root, bias, C, out);
    return gdm_launch_status("spline_pairs_aggregate_kernel");
}

extern "C" int gdm_spline_pairs_aggregate2_hip(const float* Y, const int32_t* rowptr, const int32_t* pos, const float* basis,
                                               const float* root, const float* bias, int M, int C, int relu, float* out, float* out_t,
                                               void* stream)
{
    return gdm_spline_pairs_aggregate3_hip(Y, rowptr, pos, basis, root, bias, M, C, relu, out, out_t, nullptr, stream);
}

extern "C" int gdm_spline_pairs_aggregate_hip(const float* Y, const int32_t* rowptr, const int32_t* pos, const float* basis,
                                              const float* root, const float* bias, int M, int C, int relu, float* out, void* stream)
{
    return gdm_spline_pairs_aggregate2_hip(Y, rowptr, pos, basis, root, bias, M, C, relu, out, nullptr, stream);
}

extern "C" int gdm_spline_direct2_hip(const float* x, const float* weight, const int32_t* rowptr, const int32_t* src, const float* attr,
                                     const float* root_t, const float* bias, int M, int Cin, int C, int kernel_size, int relu,
                                     float* out, float* out_t, void* stream);
extern "C" int gdm_spline_direct3_hip(const float* x, const float* weight, const int32_t* rowptr, const int32_t* src, const float* attr,
                                     const float* root_t, const float* bias, int M, int Cin, int C, int kernel_size, int relu,
                                     float* out, float* out_t, void* out_packed, void* stream);

extern "C" int gdm_spline_direct_hip(const float* x, const float* weight, const int32_t* rowptr, const int32_t* src, const float* attr,
                                     const float* root_t, const float* bias, int M, int Cin, int C, int kernel_size, int relu,
                                     float* out, void* stream)
{
    return gdm_spline_direct2_hip(x, weight, rowptr, src, attr, root_t, bias, M, Cin, C, kernel_size, relu, out, nullptr, stream);
}

extern "C" int gdm_spline_direct2_hip(const float* x, const float* weight, const int32_t* rowptr, const int32_t* src, const float* attr,
                                     const float* root_t, const float* bias, int M, int Cin, int C, int kernel_size, int relu,
                                     float* out, float* out_t, void* stream)
{
    return gdm_spline_direct3_hip(x, weight, rowptr, src, attr, root_t, bias, M, Cin, C, kernel_size, relu, out, out_t, nullptr, stream);
}

extern "C" int gdm_spline_direct3_hip(const float* x, const float* weight, const int32_t* rowptr, const int32_t* src, const float* attr,
                                     const float* root_t, const float* bias, int M, int Cin, int C, int kernel_size, int relu,
                                     float* out, float* out_t, void* out_packed, void* stream)
{
    unsigned char* out_pk = (unsigned char*)out_packed;
    GDM_CHECK_ARG(x && weight && rowptr && src && attr && (out || out_t), "gdm_spline_direct_hip: NULL pointer");
    GDM_CHECK_ARG(!out_pk || (C % 128 == 0 && C <= 512), "gdm_spline_direct3_hip: the packed output needs C = 128, 256 or 512");
    GDM_CHECK_ARG(M >= 1 && C >= 1 && kernel_size >= 2 && Cin >= 1 && Cin <= 16, "gdm_spline_direct_hip: bad shape M=%d Cin=%d (<= 16) C=%d ks=%d", M, Cin, C, kernel_size);
    const bool vec = C % 4 == 0 && C <= 512 && 512 % C == 0 && ((uintptr_t)weight & 15) == 0 && ((uintptr_t)out & 15) == 0 &&
                     (!root_t || ((uintptr_t)root_t & 15) == 0) && (!bias || ((uintptr_t)bias & 15) == 0);
    if (vec) {
        const int vpb = 128 / (C / 4);                            // vertices per block
        const dim3 grid(gdm_cdiv(M, vpb));
        if (relu)
            hipLaunchKernelGGL((spline_direct_vec_kernel<true, 16>), grid, dim3(128), 0, (hipStream_t)stream, x, weight, rowptr, src, attr, root_t,
                               bias, M, Cin, C, kernel_size, out, out_t, out_pk);
        else
            hipLaunchKernelGGL((spline_direct_vec_kernel<false, 16>), grid, dim3(128), 0, (hipStream_t)stream, x, weight, rowptr, src, attr, root_t,
                               bias, M, Cin, C, kernel_size, out, out_t, out_pk);
        return gdm_launch_status("spline_direct_vec_kernel");
    }
    GDM_CHECK_ARG(out && !out_t && !out_pk, "gdm_spline_direct2_hip: the channel-major / packed outputs need C %% 4 == 0, 512 %% C == 0 and 16-byte aligned buffers");
    if (relu)
        hipLaunchKernelGGL((spline_direct_kernel<true, 16>), dim3(M), dim3(128), 0, (hipStream_t)stream, x, weight, rowptr, src, attr, root_t, bias, Cin, C, kernel_size, out);
    else
        hipLaunchKernelGGL((spline_direct_kernel<false, 16>), dim3(M), dim3(128), 0, (hipStream_t)stream, x, weight, rowptr, src, attr, root_t, bias, Cin, C, kernel_size, out);
    return gdm_launch_status("spline_direct_kernel");
}

extern "C" int gdm_spline_aggregate_hip(const float* xw, const int32_t* rowptr, const int32_t* src, const float* attr,
                                        const float* root, const float* bias, int M, int C, int kernel_size, int relu,
                                        float* out, void* stream)
{
    GDM_CHECK_ARG(xw && rowptr && src && attr && out, "gdm_spline_aggregate_hip: NULL pointer");
    GDM_CHECK_ARG(M >= 1 && C >= 1 && kernel_size >= 2, "gdm_spline_aggregate_hip: bad shape M=%d C=%d ks=%d", M, C, kernel_size);
    if (relu)
        hipLaunchKernelGGL(spline_aggregate_kernel<true>, dim3(M), dim3(128), 0, (hipStream_t)stream, xw, rowptr, src, attr, root, bias, C, kernel_size, out);
    else
        hipLaunchKernelGGL(spline_aggregate_kernel<false>, dim3(M), dim3(128), 0, (hipStream_t)stream, xw, rowptr, src, attr, root, bias, C, kernel_size, out);
    return gdm_launch_status("spline_aggregate_kernel");
}

extern "C" int gdm_spline_aggregate_bwd_hip(const float* grad_out, const int32_t* rowptr, const int32_t* src, const float* attr,
                                            int M, int C, int kernel_size, float* grad_xw, void* stream)
{
    GDM_CHECK_ARG(grad_out && rowptr && src && attr && grad_xw, "gdm_spline_aggregate_bwd_hip: NULL pointer");
    GDM_CHECK_ARG(M >= 1 && C >= 1 && kernel_size >= 2, "gdm_spline_aggregate_bwd_hip: bad shape");
    hipLaunchKernelGGL(spline_aggregate_bwd_kernel, dim3(M), dim3(128), 0, (hipStream_t)stream, grad_out, rowptr, src, attr, C, kernel_size, grad_xw);
    return gdm_launch_status("spline_aggregate_bwd_kernel");
}
