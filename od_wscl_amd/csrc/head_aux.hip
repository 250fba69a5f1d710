// head_aux.hip -- operand plumbing between ROIPool and fc6 for the stacked clean + DropBlock pass
// (ROIWeakRegHead.forward, roi_heads/weak_head/weak_head.py:107-112: forward(), then forward_dropblock() and
// forward_neck() on the SAME pooled features; DropBlock2D.forward, modeling/dropblock/drop_block.py:29-71).
//
// The reference (and a straight PyTorch rendition) moves the 200 MB pooled tensor nine times between the pool and
// the first GEMM: x*block, *numel, /sum, flatten, cat(clean, aug), the bf16 cast -- and as often again on the way
// back.  Here both directions are one pass each:
//   stack   : pooled fp32 (P, C, S) --> bf16 (2P, ld): row p = x, row P+p = ((x * block) * numel) / sum
//   unstack : dX (2P, ld) bf16|fp32 --> d pooled fp32 (P, C, S) = dX[p] + ((dX[P+p] * block) * numel) / sum
// `block` is the (P, S) keep mask after dilation (S = 7*7), `block_sum` its sum on the device (no host sync).
#include "odw_common.h"

namespace {

__device__ __forceinline__ unsigned short f2bf(float f) {
    unsigned int u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf2f(unsigned int h) { return __uint_as_float(h << 16); }

constexpr int kMaxS = 256;      // spatial cells per ROI held in LDS (7x7 = 49, 14x14 = 196)

// one workgroup per ROI; 4 consecutive elements per thread (row length C*S is a multiple of 4)
__global__ __launch_bounds__(256) void stack_clean_aug_kernel(const float* __restrict__ pooled,
                                                              const float* __restrict__ block,
                                                              const float* __restrict__ block_sum, int P, int CS, int S,
                                                              float numel, unsigned short* __restrict__ out, int ld) {
    __shared__ float keep[kMaxS];
    const int p = blockIdx.x;
    for (int s = threadIdx.x; s < S; s += blockDim.x) keep[s] = block[(size_t)p * S + s];
    __syncthreads();
    const float sum = *block_sum;
    const float4* src = reinterpret_cast<const float4*>(pooled + (size_t)p * CS);
    uint2* clean = reinterpret_cast<uint2*>(out + (size_t)p * ld);
    uint2* aug = reinterpret_cast<uint2*>(out + (size_t)(P + p) * ld);
    for (int q = threadIdx.x; q < CS / 4; q += blockDim.x) {
        const float4 v = src[q];
        const int s0 = (q * 4) % S;
        const float x[4] = {v.x, v.y, v.z, v.w};
        unsigned short c[4], a[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            int s = s0 + t;
            s = s >= S ? s - S : s;
            c[t] = f2bf(x[t]);
            a[t] = f2bf(((x[t] * keep[s]) * numel) / sum);           // the reference's evaluation order (:49-50)
        }
        clean[q] = make_uint2((unsigned)c[0] | ((unsigned)c[1] << 16), (unsigned)c[2] | ((unsigned)c[3] << 16));
        aug[q] = make_uint2((unsigned)a[0] | ((unsigned)a[1] << 16), (unsigned)a[2] | ((unsigned)a[3] << 16));
    }
    // zero the row padding (ld > CS) so that the GEMM's K tail reads zeros
    for (int k = CS + threadIdx.x; k < ld; k += blockDim.x) {
        out[(size_t)p * ld + k] = 0;
        out[(size_t)(P + p) * ld + k] = 0;
    }
}

template <bool DX_F32>
__global__ __launch_bounds__(256) void unstack_clean_aug_kernel(const void* __restrict__ dXv, int ld,
                                                                const float* __restrict__ block,
                                                                const float* __restrict__ block_sum, int P, int CS,
                                                                int S, float numel, float* __restrict__ dpooled) {
    __shared__ float keep[kMaxS];
    const int p = blockIdx.x;
    for (int s = threadIdx.x; s < S; s += blockDim.x) keep[s] = block[(size_t)p * S + s];
    __syncthreads();
    const float sum = *block_sum;
    float4* dst = reinterpret_cast<float4*>(dpooled + (size_t)p * CS);
    for (int q = threadIdx.x; q < CS / 4; q += blockDim.x) {
        float c[4], a[4];
        if (DX_F32) {
            const float* dX = reinterpret_cast<const float*>(dXv);
            const float4 vc = *reinterpret_cast<const float4*>(dX + (size_t)p * ld + q * 4);
            const float4 va = *reinterpret_cast<const float4*>(dX + (size_t)(P + p) * ld + q * 4);
            c[0] = vc.x; c[1] = vc.y; c[2] = vc.z; c[3] = vc.w;
            a[0] = va.x; a[1] = va.y; a[2] = va.z; a[3] = va.w;
        } else {
            const unsigned short* dX = reinterpret_cast<const unsigned short*>(dXv);
            const uint2 vc = *reinterpret_cast<const uint2*>(dX + (size_t)p * ld + q * 4);
            const uint2 va = *reinterpret_cast<const uint2*>(dX + (size_t)(P + p) * ld + q * 4);
            c[0] = bf2f(vc.x & 0xffff); c[1] = bf2f(vc.x >> 16); c[2] = bf2f(vc.y & 0xffff); c[3] = bf2f(vc.y >> 16);
            a[0] = bf2f(va.x & 0xffff); a[1] = bf2f(va.x >> 16); a[2] = bf2f(va.y & 0xffff); a[3] = bf2f(va.y >> 16);
        }
        const int s0 = (q * 4) % S;
        float r[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            int s = s0 + t;
            s = s >= S ? s - S : s;
            r[t] = c[t] + ((a[t] * keep[s]) * numel) / sum;
        }
        dst[q] = make_float4(r[0], r[1], r[2], r[3]);
    }
}

}  // namespace

ODW_EXPORT int odw_stack_clean_aug(const float* pooled, const float* block, const float* block_sum, int P, int C,
                                   int S, void* out_bf16, int ld, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    ODW_REQUIRE(P >= 0 && C > 0 && S > 0 && S <= kMaxS, "stack_clean_aug: bad dims P=%d C=%d S=%d", P, C, S);
    if (P == 0) return ODW_OK;
    const long cs = (long)C * S;
    ODW_REQUIRE(pooled && block && block_sum && out_bf16, "stack_clean_aug: null pointer");
    ODW_REQUIRE(cs % 4 == 0 && ld >= cs && ld % 4 == 0 && S >= 4, "stack_clean_aug: C*S=%ld must be a multiple of 4 and fit ld=%d", cs, ld);
    ODW_REQUIRE((((uintptr_t)pooled) & 15) == 0 && (((uintptr_t)out_bf16) & 7) == 0, "stack_clean_aug: alignment");
    stack_clean_aug_kernel<<<P, 256, 0, stream>>>(pooled, block, block_sum, P, (int)cs, S, (float)((double)P * S),
                                                  (unsigned short*)out_bf16, ld);
    ODW_CHECK_LAUNCH("stack_clean_aug_kernel");
    return ODW_OK;
}

ODW_EXPORT int odw_unstack_clean_aug_bwd(const void* dX, int dx_is_f32, int ld, const float* block,
                                         const float* block_sum, int P, int C, int S, float* dpooled, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    ODW_REQUIRE(P >= 0 && C > 0 && S > 0 && S <= kMaxS, "unstack_clean_aug_bwd: bad dims P=%d C=%d S=%d", P, C, S);
    if (P == 0) return ODW_OK;
    const long cs = (long)C * S;
    ODW_REQUIRE(dX && block && block_sum && dpooled, "unstack_clean_aug_bwd: null pointer");
    ODW_REQUIRE(cs % 4 == 0 && ld >= cs && ld % 4 == 0 && S >= 4, "unstack_clean_aug_bwd: C*S=%ld must be a multiple of 4 and fit ld=%d", cs, ld);
    ODW_REQUIRE((((uintptr_t)dX) & 15) == 0 && (((uintptr_t)dpooled) & 15) == 0, "unstack_clean_aug_bwd: alignment");
    const float numel = (float)((double)P * S);
    if (dx_is_f32)
        unstack_clean_aug_kernel<true><<<P, 256, 0, stream>>>(dX, ld, block, block_sum, P, (int)cs, S, numel, dpooled);
    else
        unstack_clean_aug_kernel<false><<<P, 256, 0, stream>>>(dX, ld, block, block_sum, P, (int)cs, S, numel, dpooled);
    ODW_CHECK_LAUNCH("unstack_clean_aug_kernel");
    return ODW_OK;
}
