// odw_fixed.h -- order-independent (deterministic) scatter-add on LDS: fixed-point accumulation.
//
// Measured on gfx950 (tools/exp/lds_atomic_rate.hip, 512 workgroups x 1024 threads on 76x76 planes):
//     ds_add_f32  0.20 T updates/s chip-wide      ds_add_u32  3.9 T/s      ds_add_u64  2.0 T/s
// -- the float LDS atomic is 10-19x slower than the integer ones.  Every scatter-add of the path (ROIPool / ROIAlign
// backward: csrc/cuda/ROIPool_cuda.cu:80-108, ROIAlign_cuda.cu:178-254 use float atomicAdd) therefore accumulates
// in 64-bit fixed point: one power-of-two scale per launch (2^40 / the power of two above max|gradient|, found by
// an atomicMax pre-pass whose result does not depend on the order either), 64-bit integer LDS atomics, one
// conversion back.  Integer addition is associative, so the result is bit-identical from run to run -- which the
// reference's own backward is not -- and it is the exact sum to ~2^-40 of the largest term.
// No overflow: |term| <= 2^40 and a cell receives fewer than 2^22 terms.
#pragma once
#include <hip/hip_runtime.h>

namespace odwfx {

__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

// max |x| as the bit pattern of a non-negative float (bit patterns of non-negative floats order like the values;
// NaN sorts above inf).  out must be zeroed before the launch.  BF16: x holds bf16 values (n of them, n % 8 == 0).
// rows_dev (optional): x holds rows of row_elems values and only the first *rows_dev of them exist (a device-resident count).
template <bool BF16>
__global__ __launch_bounds__(256) void absmax_kernel(const void* __restrict__ xv, size_t n, unsigned* __restrict__ out,
                                                     const int* __restrict__ rows_dev = nullptr, size_t row_elems = 0) {
    if (rows_dev) { const size_t live = (size_t)(*rows_dev > 0 ? *rows_dev : 0) * row_elems; n = live < n ? live : n; }
    unsigned m = 0;
    if (BF16) {
        const uint4* x = reinterpret_cast<const uint4*>(xv);
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n / 8; i += (size_t)gridDim.x * blockDim.x) {
            const uint4 v = x[i];
            const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                m = max(m, (w[q] & 0x7fffu) << 16);
                m = max(m, w[q] & 0x7fff0000u);
            }
        }
    } else {
        const float4* x = reinterpret_cast<const float4*>(xv);
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n / 4; i += (size_t)gridDim.x * blockDim.x) {
            const float4 v = x[i];
            m = max(m, __float_as_uint(v.x) & 0x7fffffffu);
            m = max(m, __float_as_uint(v.y) & 0x7fffffffu);
            m = max(m, __float_as_uint(v.z) & 0x7fffffffu);
            m = max(m, __float_as_uint(v.w) & 0x7fffffffu);
        }
        if (blockIdx.x == 0 && threadIdx.x < (n & 3))
            m = max(m, __float_as_uint(reinterpret_cast<const float*>(xv)[(n & ~(size_t)3) + threadIdx.x]) & 0x7fffffffu);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, off, 64));
    // one global atomic per WORKGROUP (8192 same-address atomics -- one per wave of a 2048-block grid -- serialise in
    // L2 and cost more than the 100 MB scan itself: 117 us measured)
    __shared__ unsigned wmax[4];
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3]));
        if (m) atomicMax(out, m);
    }
}

struct Scale {
    float to_fixed, to_float;
    int state;          // 0 = all-zero input, 1 = finite, 2 = inf / nan present
};

__device__ __forceinline__ Scale scale_of(unsigned absmax_bits) {
    Scale s;
    s.to_fixed = 0.0f; s.to_float = 0.0f;
    // classified on the BIT PATTERN: the pre-pass sorts NaN (0x7fc00000) above inf, and a float comparison with a NaN
    // maximum is false both ways -- `!(amax > 0)` used to file a diverged step under "all-zero gradient"
    if (absmax_bits >= 0x7f800000u) { s.state = 2; return s; }
    if (absmax_bits == 0u) { s.state = 0; return s; }
    int e;
    frexpf(__uint_as_float(absmax_bits), &e);                // amax < 2^e
    e = e < -86 ? -86 : e;                                   // (denormal-sized gradients: 2^(40-e) must stay finite)
    s.to_fixed = ldexpf(1.0f, 40 - e);
    s.to_float = ldexpf(1.0f, e - 40);
    s.state = 1;
    return s;
}

__device__ __forceinline__ void add(long long* cell, float v, float to_fixed) {
    atomicAdd(reinterpret_cast<unsigned long long*>(cell), (unsigned long long)__float2ll_rn(v * to_fixed));
}

}  // namespace odwfx
