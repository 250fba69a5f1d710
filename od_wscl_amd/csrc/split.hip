// split.hip -- fp32-grade products on the bf16 matrix cores by operand splitting ("bf16x3").
//
// The reference computes every Linear and convolution in fp32 (config/defaults.py:559 DTYPE float32;
// modeling/backbone/vgg16.py:107-193).  gfx950's fp32-input MFMA peaks at 157 TFLOP/s, its bf16 MFMA at 2.5 PFLOP/s
// with exact bf16 x bf16 products and fp32 accumulation -- so an fp32 value is carried as THREE bf16 planes
//     x = hi + mid + lo,   hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid)     (24 significand bits)
// and a product a.b as the six plane products of total order <= 2
//     hi.hi + hi.mid + hi.lo + mid.hi + mid.mid + lo.hi                                  (dropped terms <= 2^-24 |a.b|)
// laid out along the REDUCTION axis: operand A holds the planes [hi hi hi mid mid lo], operand B [hi mid lo hi mid hi],
// each `block` elements wide, and the unchanged MFMA kernels (gemm_nt_bf16_*, conv3x3_glds_kernel) run over
// K' = T * block.  417 TFLOP/s-equivalent instead of 157, and the tuned tile pipeline instead of a second one.
// The two kernels here produce those operands from fp32 tensors:
//   split_rows : out[r][t*block + c] = plane_{pat[t]}(in[r][c])           (activations, weights: reduction along c)
//   split_cols : out[c][t*block + r] = plane_{pat[t]}(in[r][c])           (weight gradients: reduction along r)
// pat[t] in {0,1,2} = hi/mid/lo, 3 = a zero block (the convolution wants a power-of-two channel count: T = 8).
#include "odw_common.h"

namespace {

constexpr int kMaxTerms = 8;

struct Pattern { int T; int p[kMaxTerms]; };

__device__ __forceinline__ unsigned short f2bf(float f) {   // round-to-nearest-even
    unsigned int u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float((unsigned int)h << 16); }

// the three planes (+ a zero) of one value; both subtractions are exact in fp32
__device__ __forceinline__ void planes(float x, unsigned short (&p)[4]) {
    p[0] = f2bf(x);
    p[1] = p[2] = p[3] = 0;
    if ((__float_as_uint(x) & 0x7f800000u) == 0x7f800000u) return;      // inf / nan live in the hi plane alone
    const float r1 = x - bf2f(p[0]);
    p[1] = f2bf(r1);
    const float r2 = r1 - bf2f(p[1]);
    p[2] = f2bf(r2);
}

// two values -> one packed pair of plane p's bf16 (v_cvt_pk_bf16_f32 rounds to nearest even like f2bf)
__device__ __forceinline__ unsigned pk(float a, float b) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ float finite_or_zero(float x, float r) {      // inf / nan live in the hi plane alone
    return (__float_as_uint(x) & 0x7f800000u) == 0x7f800000u ? 0.0f : r;
}

// one thread = 4 consecutive columns of one row: one 16-byte load (lanes contiguous), T 8-byte stores -- the shape that
// reaches the copy ceiling for these bytes (tools/exp/split_rate.hip: 6.06 TB/s on the 4096 x 25088 fc6 weight against
// 5.3 for 8 columns per thread; the round-2 kernel, software rounding + 64-bit index division, ran at 0.85 TB/s).
// The pattern is wave-uniform, so picking a plane is a scalar branch.
__global__ __launch_bounds__(256) void split_rows_kernel(const float* __restrict__ in, long long ld_in, int R, int Cc,
                                                         Pattern pat, unsigned short* __restrict__ out,
                                                         long long ld_out, int block, const int* __restrict__ r_dev) {
    if (r_dev) { const int rd = *r_dev; R = rd < R ? rd : R; }      // rows that exist (device-resident count; R = the capacity)
    const unsigned chunks = (unsigned)block / 4u;
    const unsigned total = (unsigned)R * chunks;
    bool need_lo = false;
    for (int t = 0; t < pat.T; ++t) need_lo |= pat.p[t] == 2;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const unsigned r = i / chunks;
        const int c0 = (int)(i - r * chunks) * 4;
        const float* src = in + (long long)r * ld_in + c0;
        float v[4];
        if (c0 + 4 <= Cc && ((reinterpret_cast<uintptr_t>(src) & 15) == 0)) {
            const float4 a = *reinterpret_cast<const float4*>(src);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = c0 + j < Cc ? src[j] : 0.0f;
        }
        uint2 hi, mid, lo = make_uint2(0, 0);
        hi.x = pk(v[0], v[1]); hi.y = pk(v[2], v[3]);
        float r1[4];                        // both subtractions are exact in fp32
        r1[0] = finite_or_zero(v[0], v[0] - __uint_as_float(hi.x << 16));
        r1[1] = finite_or_zero(v[1], v[1] - __uint_as_float(hi.x & 0xffff0000u));
        r1[2] = finite_or_zero(v[2], v[2] - __uint_as_float(hi.y << 16));
        r1[3] = finite_or_zero(v[3], v[3] - __uint_as_float(hi.y & 0xffff0000u));
        mid.x = pk(r1[0], r1[1]); mid.y = pk(r1[2], r1[3]);
        if (need_lo) {
            lo.x = pk(r1[0] - __uint_as_float(mid.x << 16), r1[1] - __uint_as_float(mid.x & 0xffff0000u));
            lo.y = pk(r1[2] - __uint_as_float(mid.y << 16), r1[3] - __uint_as_float(mid.y & 0xffff0000u));
        }
        unsigned short* dst = out + (long long)r * ld_out + c0;
        for (int t = 0; t < pat.T; ++t) {
            const int p = pat.p[t];
            const uint2 o = p == 0 ? hi : (p == 1 ? mid : (p == 2 ? lo : make_uint2(0, 0)));
            *reinterpret_cast<uint2*>(dst + (long long)t * block) = o;
        }
    }
}

// 64 x 64 tile through LDS (fp32, padded rows): coalesced 256-byte reads along c, 16-byte writes along r
__global__ __launch_bounds__(256) void split_cols_kernel(const float* __restrict__ in, long long ld_in, int R, int Cc,
                                                         Pattern pat, unsigned short* __restrict__ out,
                                                         long long ld_out, int block) {
    __shared__ float tile[64][65];
    const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;         // 64 x 4
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int r = r0 + ty + 4 * k, c = c0 + tx;
        tile[ty + 4 * k][tx] = (r < R && c < Cc) ? in[(long long)r * ld_in + c] : 0.0f;
    }
    __syncthreads();
    // 64 output rows (c) x 8 chunks of 8 consecutive r: two items per thread
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int item = threadIdx.x + it * 256;
        const int c = item & 63, q = item >> 6;                      // lanes walk c: LDS column stride 65 = conflict-free
        if (c0 + c >= Cc || r0 + q * 8 >= block) continue;
        unsigned short pl[8][4];
#pragma unroll
        for (int j = 0; j < 8; ++j) planes(tile[q * 8 + j][c], pl[j]);
        unsigned short* dst = out + (long long)(c0 + c) * ld_out + r0 + q * 8;
        for (int t = 0; t < pat.T; ++t) {
            const int p = pat.p[t];
            uint4 o;
            o.x = (unsigned)pl[0][p] | ((unsigned)pl[1][p] << 16);
            o.y = (unsigned)pl[2][p] | ((unsigned)pl[3][p] << 16);
            o.z = (unsigned)pl[4][p] | ((unsigned)pl[5][p] << 16);
            o.w = (unsigned)pl[6][p] | ((unsigned)pl[7][p] << 16);
            *reinterpret_cast<uint4*>(dst + (long long)t * block) = o;
        }
    }
}

// dZ = dY * [Y != 0] * scale in fp32 and db[n] += sum_m dZ[m][n]: the backward prologue of a Linear whose ReLU /
// dropout were fused into the forward epilogue (the saved output is zero exactly where either one cut).  A workgroup
// owns 64 columns x one chunk of rows; the bias gradient is reduced in two fixed-order stages (column sums of each
// row chunk, then the chunks in order), so every entry has one writer and one summation order: deterministic.
constexpr int kMaskRows = 256;          // rows per workgroup

template <bool Y_BF16>
__global__ __launch_bounds__(256) void linear_bwd_mask_kernel(const float* __restrict__ dY, long long ld_dy,
                                                              const void* __restrict__ Yv, long long ld_y, int M, int N,
                                                              float scale, float* __restrict__ dZ, long long ld_z,
                                                              float* __restrict__ part) {
    __shared__ float red[4][64];
    const int n = blockIdx.x * 64 + (threadIdx.x & 63);
    const int ry = threadIdx.x >> 6;
    const int m0 = blockIdx.y * kMaskRows, m1 = min(M, m0 + kMaskRows);
    float acc = 0.0f;
    if (n < N) {
        for (int m = m0 + ry; m < m1; m += 4) {
            float v = dY[(long long)m * ld_dy + n];
            if (Yv) {
                const bool on = Y_BF16
                    ? (reinterpret_cast<const unsigned short*>(Yv)[(long long)m * ld_y + n] & 0x7fff) != 0
                    : reinterpret_cast<const float*>(Yv)[(long long)m * ld_y + n] != 0.0f;
                v = on ? v * scale : 0.0f;
            }
            dZ[(long long)m * ld_z + n] = v;
            acc += v;
        }
    }
    red[ry][threadIdx.x & 63] = acc;
    __syncthreads();
    if (part && ry == 0 && n < N)
        part[(long long)blockIdx.y * N + n] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

__global__ __launch_bounds__(256) void bias_grad_finish_kernel(const float* __restrict__ part, int chunks, int N,
                                                               float* __restrict__ db) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float acc = 0.0f;
    for (int c = 0; c < chunks; ++c) acc += part[(long long)c * N + n];
    db[n] += acc;
}

bool pattern_ok(const int* pattern, int T, Pattern& pat) {
    if (!pattern || T < 1 || T > kMaxTerms) return false;
    pat.T = T;
    for (int t = 0; t < kMaxTerms; ++t) pat.p[t] = 3;
    for (int t = 0; t < T; ++t) {
        if (pattern[t] < 0 || pattern[t] > 3) return false;
        pat.p[t] = pattern[t];
    }
    return true;
}

// ---- cell-major planes of an operand of the first head Linear (round 4) ---------------------------------------------
// fc6 reduces over k = c * S + s (channel-major: the reference flattens (C, 7, 7), vgg16.py:121).  The shared clean +
// DropBlock forward (gemm_bf16.hip: gemm_nt_cm_kernel) walks the reduction CELL by cell -- the DropBlock mask zeroes
// whole cells of a ROI -- so its operands hold k' = s * C + c, as TWO stored planes [hi | mid] (the kernel's K-tile map
// reads hi twice: the three products hi.hi + hi.mid + mid.hi of precision.py without a duplicated hi plane).
// One workgroup = (row, 64 channels): 64 * S consecutive floats in (coalesced), S runs of 64 bf16 = 128 bytes out per plane.
__global__ __launch_bounds__(512) void split_rows_cm_kernel(const float* __restrict__ in, long long ld_in, int C, int S,
                                                            unsigned short* __restrict__ out, long long ld_out,
                                                            long long mid_off) {
    __shared__ __attribute__((aligned(16))) float s_val[64 * 64];
    const int r = blockIdx.x, c0 = blockIdx.y * 64;
    const float* src = in + (long long)r * ld_in + (long long)c0 * S;
    const int n4 = 16 * S;                                 // 64 * S / 4
    if ((reinterpret_cast<uintptr_t>(src) & 15) == 0) {
        for (int i = threadIdx.x; i < n4; i += 512)
            reinterpret_cast<float4*>(s_val)[i] = reinterpret_cast<const float4*>(src)[i];
    } else {
        for (int i = threadIdx.x; i < 64 * S; i += 512) s_val[i] = src[i];
    }
    __syncthreads();
    const int bin = threadIdx.x >> 3, cg = threadIdx.x & 7;
    if (bin >= S) return;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = s_val[(cg * 8 + j) * S + bin];
    unsigned hi[4], mid[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        hi[j] = pk(v[2 * j], v[2 * j + 1]);
        const float r0 = finite_or_zero(v[2 * j], v[2 * j] - __uint_as_float(hi[j] << 16));
        const float r1 = finite_or_zero(v[2 * j + 1], v[2 * j + 1] - __uint_as_float(hi[j] & 0xffff0000u));
        mid[j] = pk(r0, r1);
    }
    unsigned short* dst = out + (long long)r * ld_out + (long long)bin * C + c0 + cg * 8;
    *reinterpret_cast<uint4*>(dst) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
    *reinterpret_cast<uint4*>(dst + mid_off) = make_uint4(mid[0], mid[1], mid[2], mid[3]);
}

}  // namespace

static int split_rows_launch(const float* in, int64_t ld_in, int R, int Cc, const int* pattern, int T, void* out,
                             int64_t ld_out, int block, void* stream_, const int* r_dev);

ODW_EXPORT int odw_split_rows_bf16(const float* in, int64_t ld_in, int R, int Cc, const int* pattern, int T, void* out,
                                   int64_t ld_out, int block, void* stream_) {
    return split_rows_launch(in, ld_in, R, Cc, pattern, T, out, ld_out, block, stream_, nullptr);
}

// the same with the number of rows on the device (R_cap bounds the launch; rows >= *r_dev are neither read nor written)
ODW_EXPORT int odw_split_rows_bf16_dyn(const float* in, int64_t ld_in, int R_cap, int Cc, const int* pattern, int T, void* out,
                                       int64_t ld_out, int block, const int* r_dev, void* stream_) {
    ODW_REQUIRE(r_dev, "split_rows_dyn: r_dev is null");
    return split_rows_launch(in, ld_in, R_cap, Cc, pattern, T, out, ld_out, block, stream_, r_dev);
}

static int split_rows_launch(const float* in, int64_t ld_in, int R, int Cc, const int* pattern, int T, void* out,
                             int64_t ld_out, int block, void* stream_, const int* r_dev) {
    Pattern pat;
    ODW_REQUIRE(pattern_ok(pattern, T, pat), "split_rows: pattern = up to %d plane codes in 0..3", kMaxTerms);
    ODW_REQUIRE(R >= 0 && Cc >= 0 && ld_in >= Cc && block >= Cc && block % 8 == 0 && ld_out >= (int64_t)T * block && ld_out % 8 == 0,
                "split_rows: bad dims R=%d C=%d block=%d", R, Cc, block);
    if (R == 0 || block == 0) return ODW_OK;
    ODW_REQUIRE(in && out && (((uintptr_t)out) & 15) == 0, "split_rows: pointers");
    const long long total = (long long)R * (block / 4);
    ODW_REQUIRE(total < (1ll << 31), "split_rows: R * block / 4 must stay below 2^31");
    long long blocks = (total + 255) / 256;
    if (r_dev && blocks > 4096) blocks = 4096;      // (rows on the device: a bounded grid; the kernel strides over what exists)
    split_rows_kernel<<<(int)(blocks > 65536 ? 65536 : blocks), 256, 0, (hipStream_t)stream_>>>(
        in, ld_in, R, Cc, pat, (unsigned short*)out, ld_out, block, r_dev);
    ODW_CHECK_LAUNCH("split_rows_kernel");
    return ODW_OK;
}

ODW_EXPORT int odw_split_rows_cm(const float* in, int64_t ld_in, int R, int C, int S, void* out, int64_t ld_out,
                                 int64_t mid_off, void* stream_) {
    ODW_REQUIRE(R >= 0 && C > 0 && C % 64 == 0 && S >= 1 && S <= 64 && ld_in >= (int64_t)C * S && mid_off >= (int64_t)C * S &&
                mid_off % 8 == 0 && ld_out >= mid_off + (int64_t)C * S && ld_out % 8 == 0,
                "split_rows_cm: bad dims R=%d C=%d S=%d", R, C, S);
    if (R == 0) return ODW_OK;
    ODW_REQUIRE(in && out && (((uintptr_t)out) & 15) == 0, "split_rows_cm: pointers");
    split_rows_cm_kernel<<<dim3((unsigned)R, (unsigned)(C / 64)), 512, 0, (hipStream_t)stream_>>>(
        in, ld_in, C, S, (unsigned short*)out, ld_out, mid_off);
    ODW_CHECK_LAUNCH("split_rows_cm_kernel");
    return ODW_OK;
}

ODW_EXPORT int odw_split_cols_bf16(const float* in, int64_t ld_in, int R, int Cc, const int* pattern, int T, void* out,
                                   int64_t ld_out, int block, void* stream_) {
    Pattern pat;
    ODW_REQUIRE(pattern_ok(pattern, T, pat), "split_cols: pattern = up to %d plane codes in 0..3", kMaxTerms);
    ODW_REQUIRE(R >= 0 && Cc >= 0 && ld_in >= Cc && block >= R && block % 8 == 0 && ld_out >= (int64_t)T * block && ld_out % 8 == 0,
                "split_cols: bad dims R=%d C=%d block=%d", R, Cc, block);
    if (Cc == 0 || block == 0) return ODW_OK;
    ODW_REQUIRE(in && out && (((uintptr_t)out) & 15) == 0, "split_cols: pointers");
    dim3 grid((Cc + 63) / 64, (block + 63) / 64);
    split_cols_kernel<<<grid, 256, 0, (hipStream_t)stream_>>>(in, ld_in, R, Cc, pat, (unsigned short*)out, ld_out, block);
    ODW_CHECK_LAUNCH("split_cols_kernel");
    return ODW_OK;
}

ODW_EXPORT int64_t odw_linear_bwd_mask_workspace(int M, int N) {
    return M > 0 && N > 0 ? (int64_t)((M + kMaskRows - 1) / kMaskRows) * N * 4 : 0;
}

ODW_EXPORT int odw_linear_bwd_mask_f32(const float* dY, int64_t ld_dy, const void* Y, int y_is_bf16, int64_t ld_y, int M,
                                       int N, float scale, float* dZ, int64_t ld_z, float* db, void* workspace,
                                       int64_t workspace_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    ODW_REQUIRE(M >= 0 && N >= 0 && ld_dy >= N && ld_z >= N && (!Y || ld_y >= N), "linear_bwd_mask: bad dims");
    if (M == 0 || N == 0) return ODW_OK;
    ODW_REQUIRE(dY && dZ, "linear_bwd_mask: null pointer");
    ODW_REQUIRE(!db || (workspace && workspace_bytes >= odw_linear_bwd_mask_workspace(M, N)),
                "linear_bwd_mask: the bias gradient needs odw_linear_bwd_mask_workspace(M, N) bytes");
    const int chunks = (M + kMaskRows - 1) / kMaskRows;
    const dim3 grid((N + 63) / 64, chunks);
    float* part = db ? (float*)workspace : nullptr;
    if (y_is_bf16)
        linear_bwd_mask_kernel<true><<<grid, 256, 0, stream>>>(dY, ld_dy, Y, ld_y, M, N, scale, dZ, ld_z, part);
    else
        linear_bwd_mask_kernel<false><<<grid, 256, 0, stream>>>(dY, ld_dy, Y, ld_y, M, N, scale, dZ, ld_z, part);
    ODW_CHECK_LAUNCH("linear_bwd_mask_kernel");
    if (db) {
        bias_grad_finish_kernel<<<(N + 255) / 256, 256, 0, stream>>>(part, chunks, N, db);
        ODW_CHECK_LAUNCH("bias_grad_finish_kernel");
    }
    return ODW_OK;
}
