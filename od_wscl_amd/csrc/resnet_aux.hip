// resnet_aux.hip -- the pieces of the ResNet-C5 bodies (wetectron/modeling/backbone/resnet.py:258-406) that are not a
// GEMM or a 3x3 implicit GEMM, on NHWC bf16 activations:
//   add_relu        out = relu(a + b)                the residual junction of a bottleneck (:368-373)
//   relu_bwd        g = dout where out > 0           its gradient (the same tensor goes to both branches)
//   stem_conv7x7    7x7 / stride 2 / pad 3 convolution of the fp32 NCHW image + frozen batch-norm + ReLU (:381-403):
//                   3 input channels -- 147 MACs per output, direct form; weights + the affine in LDS
//   maxpool3x3s2    3x3 / stride 2 / pad 1 max pool (:404)
// The stem and layer1 are frozen in every shipped config (FREEZE_CONV_BODY_AT 2), so the last two are forward-only.
#include "odw_common.h"
#include "odw_planes.h"

namespace {

__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ unsigned short f2bf(float f) {
    unsigned int u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

// 8 bf16 per lane (16-byte accesses); BWD: a = dout, b = out
template <bool BWD>
__global__ __launch_bounds__(256) void add_relu_kernel(const uint4* __restrict__ a, const uint4* __restrict__ b,
                                                       uint4* __restrict__ out, size_t n8) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 va = a[i], vb = b[i];
        const unsigned wa[4] = {va.x, va.y, va.z, va.w}, wb[4] = {vb.x, vb.y, vb.z, vb.w};
        unsigned wo[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (BWD) {      // keep dout where out > 0 (out is a ReLU output: positive <=> non-zero magnitude, sign clear)
                const unsigned lo = (wb[q] & 0x7fffu) && !(wb[q] & 0x8000u) ? (wa[q] & 0xffffu) : 0u;
                const unsigned hi = (wb[q] & 0x7fff0000u) && !(wb[q] & 0x80000000u) ? (wa[q] & 0xffff0000u) : 0u;
                wo[q] = lo | hi;
            } else {
                const float s0 = bf2f((unsigned short)(wa[q] & 0xffff)) + bf2f((unsigned short)(wb[q] & 0xffff));
                const float s1 = bf2f((unsigned short)(wa[q] >> 16)) + bf2f((unsigned short)(wb[q] >> 16));
                wo[q] = (unsigned)f2bf(s0 > 0.0f ? s0 : 0.0f) | ((unsigned)f2bf(s1 > 0.0f ? s1 : 0.0f) << 16);
            }
        }
        out[i] = make_uint4(wo[0], wo[1], wo[2], wo[3]);
    }
}

// One thread = one output pixel x 8 output channels.  w_lds[(ky*7+kx)*3 + ci][co] fp32 (147 x Co), scale/shift (Co).
template <bool OUT_F32>
__global__ __launch_bounds__(256) void stem_conv7x7_kernel(const float* __restrict__ img, const float* __restrict__ w,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           int B, int H, int W, int Ho, int Wo, int Co,
                                                           void* __restrict__ out) {
    extern __shared__ float w_lds[];                // 147*Co weights, then Co scale, Co shift
    float* s_scale = w_lds + 147 * Co;
    float* s_shift = s_scale + Co;
    for (int i = threadIdx.x; i < 147 * Co; i += blockDim.x) {
        const int k = i / Co, co = i - k * Co;      // k = (ky*7+kx)*3 + ci ; source layout (Co, 3, 7, 7)
        const int ci = k % 3, kk = k / 3;
        w_lds[i] = w[((size_t)co * 3 + ci) * 49 + kk];
    }
    for (int i = threadIdx.x; i < Co; i += blockDim.x) { s_scale[i] = scale[i]; s_shift[i] = shift[i]; }
    __syncthreads();
    const int groups = Co / 8;
    const size_t total = (size_t)B * Ho * Wo * groups;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const int g = (int)(t % groups);
        size_t p = t / groups;
        const int xo = (int)(p % Wo); p /= Wo;
        const int yo = (int)(p % Ho);
        const int b = (int)(p / Ho);
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const float* base = img + (size_t)b * 3 * H * W;
        for (int ky = 0; ky < 7; ++ky) {
            const int y = yo * 2 - 3 + ky;
            if ((unsigned)y >= (unsigned)H) continue;
            for (int kx = 0; kx < 7; ++kx) {
                const int x = xo * 2 - 3 + kx;
                if ((unsigned)x >= (unsigned)W) continue;
                const float* wk = w_lds + ((ky * 7 + kx) * 3) * Co + g * 8;
#pragma unroll
                for (int ci = 0; ci < 3; ++ci) {
                    const float v = base[((size_t)ci * H + y) * W + x];
#pragma unroll
                    for (int q = 0; q < 8; ++q) acc[q] = fmaf(v, wk[ci * Co + q], acc[q]);
                }
            }
        }
        unsigned wo[4];
        float rf[8];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float r0 = acc[2 * q] * s_scale[g * 8 + 2 * q] + s_shift[g * 8 + 2 * q];
            float r1 = acc[2 * q + 1] * s_scale[g * 8 + 2 * q + 1] + s_shift[g * 8 + 2 * q + 1];
            r0 = r0 > 0.0f ? r0 : 0.0f;
            r1 = r1 > 0.0f ? r1 : 0.0f;
            rf[2 * q] = r0; rf[2 * q + 1] = r1;
            wo[q] = (unsigned)f2bf(r0) | ((unsigned)f2bf(r1) << 16);
        }
        if (OUT_F32) {
            float4* o = reinterpret_cast<float4*>(out) + 2 * t;
            o[0] = make_float4(rf[0], rf[1], rf[2], rf[3]);
            o[1] = make_float4(rf[4], rf[5], rf[6], rf[7]);
        } else {
            reinterpret_cast<uint4*>(out)[t] = make_uint4(wo[0], wo[1], wo[2], wo[3]);
        }
    }
}


// 3x3 / stride 1 / pad 1 convolution of the fp32 NCHW image (3 channels) + bias + ReLU -> NHWC bf16: the first layer of
// the VGG body (modeling/backbone/vgg16.py:58-60).  On the 128x128 MFMA tile that layer is K = 27 padded to 128 and
// runs at 14 TF (90 us at 608x608, + 11 us for the layout pass of the image); it is 47 MB of output and 0.64 GFLOP.
// (First version, one thread per pixel x 8 channels: 80 us, issue-bound on its 27 loads per 216 FMAs.)
// One thread = one pixel x 8 output channels; weights [tap*3 + ci][co] fp32 and the bias in LDS.
// PLANES: the result leaves as bf16 planes of the fp32 values (precision mode "bf16x2f": the input operand of the
// next convolution, T column blocks of `block` channels holding plane pat[t]) instead of rounded bf16 -- the kernel's
// arithmetic is an fp32 fmaf chain either way, so the split-precision modes need no MFMA pass for the 3-channel layer.
template <bool PLANES>
__global__ __launch_bounds__(256) void stem_conv3x3_kernel(const float* __restrict__ img, const float* __restrict__ w,
                                                           const float* __restrict__ bias, int B, int H, int W, int Co,
                                                           unsigned short* __restrict__ out, odwpl::Pattern pat, int ld,
                                                           int block) {
    extern __shared__ float w_lds[];                // 27*Co weights, then Co bias
    float* s_bias = w_lds + 27 * Co;
    for (int i = threadIdx.x; i < 27 * Co; i += blockDim.x) {
        const int k = i / Co, co = i - k * Co;      // k = tap*3 + ci ; source layout (Co, 3, 3, 3)
        const int ci = k % 3, tap = k / 3;
        w_lds[i] = w[((size_t)co * 3 + ci) * 9 + tap];
    }
    for (int i = threadIdx.x; i < Co; i += blockDim.x) s_bias[i] = bias ? bias[i] : 0.0f;
    __syncthreads();
    // one thread = one pixel, ALL output channels: the 27 inputs are loaded once (zero in the padding) and every weight
    // is an LDS broadcast -- 27 loads per 27*Co FMAs instead of per 27*8
    const size_t total = (size_t)B * H * W;
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (size_t)gridDim.x * blockDim.x) {
        const int x0 = (int)(p % W);
        const int y0 = (int)((p / W) % H);
        const int b = (int)(p / ((size_t)W * H));
        const float* base = img + (size_t)b * 3 * H * W;
        float v[27];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int y = y0 - 1 + ky, x = x0 - 1 + kx;
                const bool ok = (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
#pragma unroll
                for (int ci = 0; ci < 3; ++ci) v[(ky * 3 + kx) * 3 + ci] = ok ? base[((size_t)ci * H + y) * W + x] : 0.0f;
            }
        uint4* o = reinterpret_cast<uint4*>(out + p * (PLANES ? (size_t)ld : (size_t)Co));
        bool need_lo = false;
        if (PLANES)
            for (int t = 0; t < pat.T; ++t) need_lo |= pat.p[t] == 2;
        for (int g = 0; g < Co / 8; ++g) {
            float acc[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) acc[q] = s_bias[g * 8 + q];
#pragma unroll
            for (int k = 0; k < 27; ++k) {
                const float* wk = w_lds + k * Co + g * 8;
#pragma unroll
                for (int q = 0; q < 8; ++q) acc[q] = fmaf(v[k], wk[q], acc[q]);
            }
            if (PLANES) {
                unsigned hi[4], mid[4], lo[4];
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    odwpl::split2(fmaxf(acc[2 * q], 0.0f), fmaxf(acc[2 * q + 1], 0.0f), need_lo, hi[q], mid[q], lo[q]);
                for (int t = 0; t < pat.T; ++t) {
                    const int pl = pat.p[t];
                    o[(t * block) / 8 + g] = pl == 0 ? make_uint4(hi[0], hi[1], hi[2], hi[3])
                                           : (pl == 1 ? make_uint4(mid[0], mid[1], mid[2], mid[3])
                                           : (pl == 2 ? make_uint4(lo[0], lo[1], lo[2], lo[3]) : make_uint4(0, 0, 0, 0)));
                }
                continue;
            }
            unsigned wo[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float r0 = acc[2 * q] > 0.0f ? acc[2 * q] : 0.0f, r1 = acc[2 * q + 1] > 0.0f ? acc[2 * q + 1] : 0.0f;
                wo[q] = (unsigned)f2bf(r0) | ((unsigned)f2bf(r1) << 16);
            }
            o[g] = make_uint4(wo[0], wo[1], wo[2], wo[3]);
        }
    }
}

// The planes form of the layer for Co = 64 (VGG's conv1_1): 16 lanes share a pixel, each owns 4 output channels whose
// 108 weights stay in REGISTERS for the whole kernel, and walks runs of 4 consecutive pixels of a row with a sliding
// 3 x 6 x 3 input window (13.5 loads per 108 FMAs).  The 16 lanes of a pixel write 128 contiguous bytes per plane
// block.  Same fmaf chain per output as stem_conv3x3_kernel (bias first, then tap-major, channel-minor): same bits.
// (The one-thread-per-pixel form above keeps the weights in LDS -- one broadcast ds_read_b128 per 8 FMAs -- and stores
// 16-byte pieces 384 bytes apart: 145 us at 608 x 608 in this mode, against 142 MB of output.)
__global__ __launch_bounds__(256, 2) void stem_conv3x3_planes64_kernel(const float* __restrict__ img, const float* __restrict__ w,
                                                                       const float* __restrict__ bias, int B, int H, int W,
                                                                       unsigned short* __restrict__ out, odwpl::Pattern pat,
                                                                       int ld, int block) {
    const int cg = threadIdx.x & 15;                 // channels 4 cg .. 4 cg + 3
    float wr[27][4], bs[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        bs[q] = bias ? bias[cg * 4 + q] : 0.0f;
#pragma unroll
        for (int k = 0; k < 27; ++k)                 // k = tap*3 + ci ; source layout (Co, 3, 3, 3)
            wr[k][q] = w[((size_t)(cg * 4 + q) * 3 + (k % 3)) * 9 + (k / 3)];
    }
    bool need_lo = false;
    for (int t = 0; t < pat.T; ++t) need_lo |= pat.p[t] == 2;
    const int runs_w = (W + 3) / 4;
    const long long runs = (long long)B * H * runs_w;
    for (long long r = (long long)blockIdx.x * 16 + (threadIdx.x >> 4); r < runs; r += (long long)gridDim.x * 16) {
        const int xr = (int)(r % runs_w), y0 = (int)((r / runs_w) % H), b = (int)(r / ((long long)runs_w * H));
        const int x0 = xr * 4;
        const float* base = img + (size_t)b * 3 * H * W;
        float win[3][6][3];                          // [ky][column x0 - 1 + j][ci], zero in the padding
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const int y = y0 - 1 + ky, x = x0 - 1 + j;
                const bool ok = (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
#pragma unroll
                for (int ci = 0; ci < 3; ++ci) win[ky][j][ci] = ok ? base[((size_t)ci * H + y) * W + x] : 0.0f;
            }
#pragma unroll
        for (int px = 0; px < 4; ++px) {
            float acc[4] = {bs[0], bs[1], bs[2], bs[3]};
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int ci = 0; ci < 3; ++ci) {
                        const float v = win[ky][px + kx][ci];
#pragma unroll
                        for (int q = 0; q < 4; ++q) acc[q] = fmaf(v, wr[(ky * 3 + kx) * 3 + ci][q], acc[q]);
                    }
            if (x0 + px < W) {
                unsigned hi[2], mid[2], lo[2];
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    odwpl::split2(fmaxf(acc[2 * q], 0.0f), fmaxf(acc[2 * q + 1], 0.0f), need_lo, hi[q], mid[q], lo[q]);
                unsigned short* o = out + ((size_t)(b * H + y0) * W + x0 + px) * (size_t)ld + cg * 4;
                for (int t = 0; t < pat.T; ++t) {
                    const int pl = pat.p[t];
                    *reinterpret_cast<uint2*>(o + (size_t)t * block) = pl == 0 ? make_uint2(hi[0], hi[1])
                                                                     : (pl == 1 ? make_uint2(mid[0], mid[1])
                                                                     : (pl == 2 ? make_uint2(lo[0], lo[1]) : make_uint2(0, 0)));
                }
            }
        }
    }
}

// NHWC bf16 3x3 / stride 2 / pad 1 max pool, 8 channels per thread (padding never wins: -inf)
__global__ __launch_bounds__(256) void maxpool3x3s2_kernel(const uint4* __restrict__ X, int B, int H, int W, int C8, int Ho,
                                                           int Wo, uint4* __restrict__ Y) {
    const size_t total = (size_t)B * Ho * Wo * C8;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C8);
        size_t q = i / C8;
        const int xo = (int)(q % Wo); q /= Wo;
        const int yo = (int)(q % Ho);
        const int b = (int)(q / Ho);
        float best[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) best[k] = -__builtin_inff();
        for (int dy = 0; dy < 3; ++dy) {
            const int y = yo * 2 - 1 + dy;
            if ((unsigned)y >= (unsigned)H) continue;
            for (int dx = 0; dx < 3; ++dx) {
                const int x = xo * 2 - 1 + dx;
                if ((unsigned)x >= (unsigned)W) continue;
                const uint4 v = X[(((size_t)b * H + y) * W + x) * C8 + c];
                const unsigned wv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    best[2 * k] = fmaxf(best[2 * k], bf2f((unsigned short)(wv[k] & 0xffff)));
                    best[2 * k + 1] = fmaxf(best[2 * k + 1], bf2f((unsigned short)(wv[k] >> 16)));
                }
            }
        }
        Y[i] = make_uint4((unsigned)f2bf(best[0]) | ((unsigned)f2bf(best[1]) << 16),
                          (unsigned)f2bf(best[2]) | ((unsigned)f2bf(best[3]) << 16),
                          (unsigned)f2bf(best[4]) | ((unsigned)f2bf(best[5]) << 16),
                          (unsigned)f2bf(best[6]) | ((unsigned)f2bf(best[7]) << 16));
    }
}

// ---- fp32 NHWC forms (split precision modes keep activations in fp32 between kernels, csrc/split.hip) ---------
template <bool BWD>
__global__ __launch_bounds__(256) void add_relu_f32_kernel(const float4* __restrict__ a, const float4* __restrict__ b,
                                                           float4* __restrict__ out, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 va = a[i], vb = b[i];
        float4 o;
        if (BWD) {      // a = dout, b = out
            o.x = vb.x > 0.0f ? va.x : 0.0f; o.y = vb.y > 0.0f ? va.y : 0.0f;
            o.z = vb.z > 0.0f ? va.z : 0.0f; o.w = vb.w > 0.0f ? va.w : 0.0f;
        } else {
            const float s0 = va.x + vb.x, s1 = va.y + vb.y, s2 = va.z + vb.z, s3 = va.w + vb.w;
            o.x = s0 > 0.0f ? s0 : 0.0f; o.y = s1 > 0.0f ? s1 : 0.0f;
            o.z = s2 > 0.0f ? s2 : 0.0f; o.w = s3 > 0.0f ? s3 : 0.0f;
        }
        out[i] = o;
    }
}

__global__ __launch_bounds__(256) void maxpool3x3s2_f32_kernel(const float* __restrict__ X, int B, int H, int W, int C,
                                                               int Ho, int Wo, float* __restrict__ Y) {
    const size_t total = (size_t)B * Ho * Wo * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        size_t q = i / C;
        const int xo = (int)(q % Wo); q /= Wo;
        const int yo = (int)(q % Ho);
        const int b = (int)(q / Ho);
        float best = -__builtin_inff();
        for (int dy = 0; dy < 3; ++dy) {
            const int y = yo * 2 - 1 + dy;
            if ((unsigned)y >= (unsigned)H) continue;
            for (int dx = 0; dx < 3; ++dx) {
                const int x = xo * 2 - 1 + dx;
                if ((unsigned)x >= (unsigned)W) continue;
                best = fmaxf(best, X[(((size_t)b * H + y) * W + x) * C + c]);
            }
        }
        Y[i] = best;
    }
}

int blocks_for(size_t n) {
    size_t b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 16384 ? 16384 : b));
}

}  // namespace

ODW_EXPORT int odw_add_relu_bf16(const void* a, const void* b, void* out, int64_t n, void* stream_) {
    ODW_REQUIRE(n >= 0 && n % 8 == 0, "add_relu: n=%lld must be a multiple of 8", (long long)n);
    if (n == 0) return ODW_OK;
    ODW_REQUIRE(a && b && out && ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)out)) & 15) == 0, "add_relu: pointers");
    add_relu_kernel<false><<<blocks_for((size_t)n / 8), 256, 0, (hipStream_t)stream_>>>((const uint4*)a, (const uint4*)b,
                                                                                        (uint4*)out, (size_t)n / 8);
    ODW_CHECK_LAUNCH("add_relu_kernel");
    return ODW_OK;
}

ODW_EXPORT int odw_relu_bwd_bf16(const void* dout, const void* out, void* g, int64_t n, void* stream_) {
    ODW_REQUIRE(n >= 0 && n % 8 == 0, "relu_bwd: n=%lld must be a multiple of 8", (long long)n);
    if (n == 0) return ODW_OK;
    ODW_REQUIRE(dout && out && g && ((((uintptr_t)dout) | ((uintptr_t)out) | ((uintptr_t)g)) & 15) == 0, "relu_bwd: pointers");
    add_relu_kernel<true><<<blocks_for((size_t)n / 8), 256, 0, (hipStream_t)stream_>>>((const uint4*)dout, (const uint4*)out,
                                                                                       (uint4*)g, (size_t)n / 8);
    ODW_CHECK_LAUNCH("add_relu_kernel");
    return ODW_OK;
}

ODW_EXPORT int odw_stem_conv7x7_bn_relu(const float* img_nchw, const float* weight, const float* scale, const float* shift,
                                        int B, int H, int W, int Co, void* out_nhwc_bf16, void* stream_) {
    ODW_REQUIRE(B > 0 && H > 0 && W > 0 && Co > 0 && Co % 8 == 0 && Co <= 128, "stem_conv7x7: bad dims");
    ODW_REQUIRE(img_nchw && weight && scale && shift && out_nhwc_bf16 && (((uintptr_t)out_nhwc_bf16) & 15) == 0,
                "stem_conv7x7: pointers");
    const int Ho = (H + 2 * 3 - 7) / 2 + 1, Wo = (W + 2 * 3 - 7) / 2 + 1;
    const size_t total = (size_t)B * Ho * Wo * (Co / 8);
    const size_t lds = (size_t)(147 + 2) * Co * sizeof(float);
    stem_conv7x7_kernel<false><<<blocks_for(total) > 2048 ? 2048 : blocks_for(total), 256, lds, (hipStream_t)stream_>>>(
        img_nchw, weight, scale, shift, B, H, W, Ho, Wo, Co, out_nhwc_bf16);
    ODW_CHECK_LAUNCH("stem_conv7x7_kernel");
    return ODW_OK;
}

ODW_EXPORT int odw_stem_conv7x7_bn_relu_f32(const float* img_nchw, const float* weight, const float* scale, const float* shift,
                                            int B, int H, int W, int Co, float* out_nhwc, void* stream_) {
    ODW_REQUIRE(B > 0 && H > 0 && W > 0 && Co > 0 && Co % 8 == 0 && Co <= 128, "stem_conv7x7_f32: bad dims");
    ODW_REQUIRE(img_nchw && weight && scale && shift && out_nhwc && (((uintptr_t)out_nhwc) & 15) == 0, "stem_conv7x7_f32: pointers");
    const int Ho = (H + 2 * 3 - 7) / 2 + 1, Wo = (W + 2 * 3 - 7) / 2 + 1;
    const size_t total = (size_t)B * Ho * Wo * (Co / 8);
    const size_t lds = (size_t)(147 + 2) * Co * sizeof(float);
    stem_conv7x7_kernel<true><<<blocks_for(total) > 2048 ? 2048 : blocks_for(total), 256, lds, (hipStream_t)stream_>>>(
        img_nchw, weight, scale, shift, B, H, W, Ho, Wo, Co, out_nhwc);
    ODW_CHECK_LAUNCH("stem_conv7x7_kernel");
    return ODW_OK;
}

ODW_EXPORT int odw_maxpool3x3s2_nhwc_bf16(const void* X, int B, int H, int W, int C, void* Y, void* stream_) {
    ODW_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && X && Y, "maxpool3x3s2: bad arguments");
    ODW_REQUIRE((((uintptr_t)X) & 15) == 0 && (((uintptr_t)Y) & 15) == 0, "maxpool3x3s2: 16-byte alignment");
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const size_t n = (size_t)B * Ho * Wo * (C / 8);
    maxpool3x3s2_kernel<<<blocks_for(n), 256, 0, (hipStream_t)stream_>>>((const uint4*)X, B, H, W, C / 8, Ho, Wo, (uint4*)Y);
    ODW_CHECK_LAUNCH("maxpool3x3s2_kernel");
    return ODW_OK;
}

ODW_EXPORT int odw_stem_conv3x3_bias_relu(const float* img_nchw, const float* weight, const float* bias, int B, int H, int W,
                                          int Co, void* out_nhwc_bf16, void* stream_) {
    ODW_REQUIRE(B > 0 && H > 0 && W > 0 && Co > 0 && Co % 8 == 0 && Co <= 256, "stem_conv3x3: bad dims");
    ODW_REQUIRE(img_nchw && weight && out_nhwc_bf16 && (((uintptr_t)out_nhwc_bf16) & 15) == 0, "stem_conv3x3: pointers");
    const size_t total = (size_t)B * H * W;
    const size_t lds = (size_t)28 * Co * sizeof(float);
    const int grid = blocks_for(total) > 4096 ? 4096 : blocks_for(total);
    odwpl::Pattern none;
    none.T = 0;
    stem_conv3x3_kernel<false><<<grid, 256, lds, (hipStream_t)stream_>>>(img_nchw, weight, bias, B, H, W, Co,
                                                                        (unsigned short*)out_nhwc_bf16, none, 0, 0);
    ODW_CHECK_LAUNCH("stem_conv3x3_kernel");
    return ODW_OK;
}

ODW_EXPORT int odw_stem_conv3x3_bias_relu_planes(const float* img_nchw, const float* weight, const float* bias, int B, int H,
                                                 int W, int Co, const int* pattern, int T, void* out_planes, int64_t ld,
                                                 int block, void* stream_) {
    odwpl::Pattern pat;
    ODW_REQUIRE(odwpl::pattern_ok(pattern, T, pat), "stem_conv3x3_planes: pattern = up to %d plane codes in 0..3", odwpl::kMaxTerms);
    ODW_REQUIRE(B > 0 && H > 0 && W > 0 && Co > 0 && Co % 8 == 0 && Co <= 256, "stem_conv3x3_planes: bad dims");
    ODW_REQUIRE(block >= Co && block % 8 == 0 && ld >= (int64_t)T * block && ld % 8 == 0 && ld < (1ll << 31),
                "stem_conv3x3_planes: block / ld");
    ODW_REQUIRE(img_nchw && weight && out_planes && (((uintptr_t)out_planes) & 15) == 0, "stem_conv3x3_planes: pointers");
    const size_t total = (size_t)B * H * W;
    const size_t lds = (size_t)28 * Co * sizeof(float);
    const int grid = blocks_for(total) > 4096 ? 4096 : blocks_for(total);
    static const bool old_form = getenv("ODW_STEM_OLD") != nullptr;      // (comparison runs)
    if (Co == 64 && !old_form) {
        const long long runs = (long long)B * H * ((W + 3) / 4);
        const long long want = (runs + 15) / 16;
        stem_conv3x3_planes64_kernel<<<(int)(want < 2 * ODW_NUM_CU ? want : 2 * ODW_NUM_CU), 256, 0, (hipStream_t)stream_>>>(
            img_nchw, weight, bias, B, H, W, (unsigned short*)out_planes, pat, (int)ld, block);
        ODW_CHECK_LAUNCH("stem_conv3x3_planes64_kernel");
        return ODW_OK;
    }
    stem_conv3x3_kernel<true><<<grid, 256, lds, (hipStream_t)stream_>>>(img_nchw, weight, bias, B, H, W, Co,
                                                                       (unsigned short*)out_planes, pat, (int)ld, block);
    ODW_CHECK_LAUNCH("stem_conv3x3_kernel");
    return ODW_OK;
}

ODW_EXPORT int odw_add_relu_f32(const float* a, const float* b, float* out, int64_t n, void* stream_) {
    ODW_REQUIRE(n >= 0 && n % 4 == 0, "add_relu_f32: n=%lld must be a multiple of 4", (long long)n);
    if (n == 0) return ODW_OK;
    ODW_REQUIRE(a && b && out && ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)out)) & 15) == 0, "add_relu_f32: pointers");
    add_relu_f32_kernel<false><<<blocks_for((size_t)n / 4), 256, 0, (hipStream_t)stream_>>>((const float4*)a, (const float4*)b,
                                                                                            (float4*)out, (size_t)n / 4);
    ODW_CHECK_LAUNCH("add_relu_f32_kernel");
    return ODW_OK;
}

ODW_EXPORT int odw_relu_bwd_f32(const float* dout, const float* out, float* g, int64_t n, void* stream_) {
    ODW_REQUIRE(n >= 0 && n % 4 == 0, "relu_bwd_f32: n=%lld must be a multiple of 4", (long long)n);
    if (n == 0) return ODW_OK;
    ODW_REQUIRE(dout && out && g && ((((uintptr_t)dout) | ((uintptr_t)out) | ((uintptr_t)g)) & 15) == 0, "relu_bwd_f32: pointers");
    add_relu_f32_kernel<true><<<blocks_for((size_t)n / 4), 256, 0, (hipStream_t)stream_>>>((const float4*)dout, (const float4*)out,
                                                                                           (float4*)g, (size_t)n / 4);
    ODW_CHECK_LAUNCH("add_relu_f32_kernel");
    return ODW_OK;
}

ODW_EXPORT int odw_maxpool3x3s2_nhwc_f32(const float* X, int B, int H, int W, int C, float* Y, void* stream_) {
    ODW_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && X && Y, "maxpool3x3s2_f32: bad arguments");
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const size_t n = (size_t)B * Ho * Wo * C;
    maxpool3x3s2_f32_kernel<<<blocks_for(n), 256, 0, (hipStream_t)stream_>>>(X, B, H, W, C, Ho, Wo, Y);
    ODW_CHECK_LAUNCH("maxpool3x3s2_f32_kernel");
    return ODW_OK;
}
