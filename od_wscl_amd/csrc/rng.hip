// rng.hip -- random tensors of the hot path as pure functions of (key, element index):
// dropout (vgg16.py:124,127), DropBlock centres (drop_block.py:42), the noise view
// (vgg16.py:177-180).  Dropout never stores a mask: backward re-derives it from the counter.
#include "odw_common.h"
#include "odw_rng.h"

namespace {

__global__ void uniform_kernel(float* __restrict__ out, size_t n, uint32_t k0, uint32_t k1, uint32_t off) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = odw_uniform((uint32_t)i + off, k0, k1);
}

// element 2k = r cos t, 2k+1 = r sin t  (utils/rng.py normal)
__global__ void normal_kernel(float* __restrict__ out, size_t n, uint32_t k0, uint32_t k1) {
    const size_t pairs = (n + 1) / 2;
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < pairs; p += (size_t)gridDim.x * blockDim.x) {
        float u1 = 1.0f - odw_uniform((uint32_t)(2 * p), k0, k1);
        float t = 6.283185307179586f * odw_uniform((uint32_t)(2 * p + 1), k0, k1);
        float r = sqrtf(-2.0f * logf(u1));
        out[2 * p] = r * cosf(t);
        if (2 * p + 1 < n) out[2 * p + 1] = r * sinf(t);
    }
}

// out = x * [u >= p] * scale ; rows may come from several logical draws:
// element (row, col) of draw d uses index (row - row0[d]) * cols + col under key[d].
__global__ void dropout_kernel(const float* __restrict__ x, float* __restrict__ out, size_t n, uint32_t k0,
                               uint32_t k1, float p, float scale) {
    const size_t n4 = n / 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 v = reinterpret_cast<const float4*>(x)[i];
        const uint32_t b = (uint32_t)(4 * i);
        v.x = odw_uniform(b + 0, k0, k1) >= p ? v.x * scale : 0.0f;
        v.y = odw_uniform(b + 1, k0, k1) >= p ? v.y * scale : 0.0f;
        v.z = odw_uniform(b + 2, k0, k1) >= p ? v.z * scale : 0.0f;
        v.w = odw_uniform(b + 3, k0, k1) >= p ? v.w * scale : 0.0f;
        reinterpret_cast<float4*>(out)[i] = v;
    }
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = odw_uniform((uint32_t)i, k0, k1) >= p ? x[i] * scale : 0.0f;
}

// out = x + N(0,1)*x = x * (1 + z)   (noise_pool)
__global__ void noise_kernel(const float* __restrict__ x, float* __restrict__ out, size_t n, uint32_t k0,
                             uint32_t k1) {
    const size_t pairs = (n + 1) / 2;
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < pairs; p += (size_t)gridDim.x * blockDim.x) {
        float u1 = 1.0f - odw_uniform((uint32_t)(2 * p), k0, k1);
        float t = 6.283185307179586f * odw_uniform((uint32_t)(2 * p + 1), k0, k1);
        float r = sqrtf(-2.0f * logf(u1));
        float z0 = r * cosf(t), z1 = r * sinf(t);
        out[2 * p] = z0 * x[2 * p] + x[2 * p];
        if (2 * p + 1 < n) out[2 * p + 1] = z1 * x[2 * p + 1] + x[2 * p + 1];
    }
}

// DropBlock2D's keep mask in one pass (modeling/dropblock/drop_block.py:38-47 of the reference: Bernoulli(gamma) block
// centres from ONE uniform draw over (n, h, w), dilated by a block_size max-pool with padding block_size / 2 (cropped
// to h x w for even sizes), inverted) and its sum.  A cell re-derives the <= block_size^2 draws of its window from the
// counter-based stream (element index = its position in the (n, h, w) draw) instead of reading a materialised mask:
// uniform -> compare -> copy -> max_pool2d -> 1 - m -> sum were six torch launches at the head of every step.
// The sum is a sum of 0/1 values: exact in fp32 in any order (n h w < 2^24), so float atomics are deterministic here.
__global__ __launch_bounds__(256) void dropblock_keep_kernel(int n, int h, int w, int bs, float gamma, uint32_t k0, uint32_t k1,
                                                             float* __restrict__ keep, float* __restrict__ keep_sum) {
    const int hw = h * w, total = n * hw;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float v = 0.0f;
    if (i < total) {
        const int img = i / hw, r = i - img * hw, y = r / w, x = r - y * w;
        const int y0 = y - bs / 2, x0 = x - bs / 2;
        bool dropped = false;
        for (int dy = 0; dy < bs; ++dy) {
            const int yy = y0 + dy;
            if (yy < 0 || yy >= h) continue;
            for (int dx = 0; dx < bs; ++dx) {
                const int xx = x0 + dx;
                if (xx < 0 || xx >= w) continue;
                dropped |= odw_uniform((uint32_t)(img * hw + yy * w + xx), k0, k1) < gamma;
            }
        }
        v = dropped ? 0.0f : 1.0f;
        keep[i] = v;
    }
    // workgroup count -> one atomic
    __shared__ float part[4];
    float s = v;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(keep_sum, part[0] + part[1] + part[2] + part[3]);
}

int grid_for(size_t n, int per) {
    size_t g = (n / per + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

}  // namespace

ODW_EXPORT int odw_rng_uniform(float* out, int64_t n, uint32_t k0, uint32_t k1, uint32_t offset, void* stream_) {
    ODW_REQUIRE(n >= 0 && n < (1ll << 32), "rng_uniform: n out of range");
    if (n == 0) return ODW_OK;
    ODW_REQUIRE(out, "rng_uniform: null pointer");
    uniform_kernel<<<grid_for(n, 1), 256, 0, (hipStream_t)stream_>>>(out, (size_t)n, k0, k1, offset);
    ODW_CHECK_LAUNCH("uniform_kernel");
    return ODW_OK;
}

ODW_EXPORT int odw_dropblock_keep_mask(int n, int h, int w, int block_size, float gamma, uint32_t k0, uint32_t k1, float* keep,
                                       float* keep_sum, void* stream_) {
    ODW_REQUIRE(n >= 0 && h > 0 && w > 0 && block_size >= 1 && block_size <= 15, "dropblock_keep_mask: bad dims n=%d h=%d w=%d bs=%d", n, h, w, block_size);
    ODW_REQUIRE((long long)n * h * w < (1ll << 24), "dropblock_keep_mask: %lld cells (the fp32 count is exact below 2^24)", (long long)n * h * w);
    ODW_REQUIRE(keep_sum, "dropblock_keep_mask: null pointer");
    hipStream_t stream = (hipStream_t)stream_;
    ODW_CHECK_HIP(hipMemsetAsync(keep_sum, 0, sizeof(float), stream), "dropblock_keep_mask memset");
    if (n == 0) return ODW_OK;
    ODW_REQUIRE(keep, "dropblock_keep_mask: null pointer");
    const int total = n * h * w;
    dropblock_keep_kernel<<<(total + 255) / 256, 256, 0, stream>>>(n, h, w, block_size, gamma, k0, k1, keep, keep_sum);
    ODW_CHECK_LAUNCH("dropblock_keep_kernel");
    return ODW_OK;
}

ODW_EXPORT int odw_rng_normal(float* out, int64_t n, uint32_t k0, uint32_t k1, void* stream_) {
    ODW_REQUIRE(n >= 0 && n < (1ll << 32), "rng_normal: n out of range");
    if (n == 0) return ODW_OK;
    ODW_REQUIRE(out, "rng_normal: null pointer");
    normal_kernel<<<grid_for(n, 2), 256, 0, (hipStream_t)stream_>>>(out, (size_t)n, k0, k1);
    ODW_CHECK_LAUNCH("normal_kernel");
    return ODW_OK;
}

ODW_EXPORT int odw_dropout(const float* x, float* out, int64_t n, uint32_t k0, uint32_t k1, float p,
                           void* stream_) {
    ODW_REQUIRE(n >= 0 && n < (1ll << 32), "dropout: n out of range");
    ODW_REQUIRE(p >= 0.0f && p < 1.0f, "dropout: p=%f", p);
    if (n == 0) return ODW_OK;
    ODW_REQUIRE(x && out, "dropout: null pointer");
    ODW_REQUIRE((((uintptr_t)x) & 15) == 0 && (((uintptr_t)out) & 15) == 0, "dropout: 16-byte alignment");
    dropout_kernel<<<grid_for(n, 4), 256, 0, (hipStream_t)stream_>>>(x, out, (size_t)n, k0, k1, p, 1.0f / (1.0f - p));
    ODW_CHECK_LAUNCH("dropout_kernel");
    return ODW_OK;
}

ODW_EXPORT int odw_noise_mul(const float* x, float* out, int64_t n, uint32_t k0, uint32_t k1, void* stream_) {
    ODW_REQUIRE(n >= 0 && n < (1ll << 32), "noise_mul: n out of range");
    if (n == 0) return ODW_OK;
    ODW_REQUIRE(x && out, "noise_mul: null pointer");
    noise_kernel<<<grid_for(n, 2), 256, 0, (hipStream_t)stream_>>>(x, out, (size_t)n, k0, k1);
    ODW_CHECK_LAUNCH("noise_kernel");
    return ODW_OK;
}
