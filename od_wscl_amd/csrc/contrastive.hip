// contrastive.hip -- proposal x proposal similarity and the SupConLossV2
// (NT-Xent style) contrastive loss, forward + backward, for gfx950.
//
// Reference behaviour:
//   sim_mat = E E^T                      roi_heads/weak_head/loss.py:319
//   SupConLossV2.forward                 roi_heads/sim_head/sim_loss.py:49-80
//   (its backward is torch autograd; the closed form is in SURVEY.md s8a)
//
// MI355X structure.  Everything is built on ONE wave-level primitive: a 32x32
// tile of X Y^T for 128-d rows on the exact-fp32 matrix pipe
// (v_mfma_f32_32x32x2_f32, bit-identical to an fmaf chain).  Because a dot
// product does not care in which order k is visited, lanes 0-31 walk
// k = 0..63 and lanes 32-63 walk k = 64..127: every lane reads ONE contiguous
// 256-byte half-row with 16-byte loads straight into MFMA operand registers --
// no LDS staging, no transposes.
//
//   * pairwise_sim: upper-triangular 64x64 block tiles only; the mirrored
//     tile is written through a padded LDS transpose so both stores are
//     128-byte coalesced.  HBM traffic = E once + S once.
//   * supcon: "flash" form -- S is never materialised.  The tile is computed
//     TRANSPOSED (rows j in registers, column i = lane) so the row-i softmax
//     statistics (running max, A_i = sum over same-label j, B_i = sum over all
//     j != i) are lane-local; the two half-waves are merged once at the end.
//     Backward recomputes the tile, forms H = G + G^T in registers and feeds
//     it straight back into the matrix pipe as the A operand of H F (the C
//     layout of one MFMA is the A layout of the next when k is walked in the
//     same permuted order), accumulating dF in 64 accumulator registers.
//     The j range is split across waves (grid.y) for occupancy; partials are
//     merged by tiny deterministic combine kernels -- no atomics anywhere.
#include "odw_common.h"
#include <stdlib.h>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kD = 128;      // embedding width served by the MFMA path (Sim_Net output, sim_net.py:14)
constexpr int kHalf = 64;    // floats per lane per row

// row index (0..31) held in accumulator register k of a 32x32 MFMA C tile
__device__ __forceinline__ int crow(int k, int half) { return (k & 3) + 8 * (k >> 2) + 4 * half; }

// this lane's 64-float half of row `row` of X (zeros past nrows)
__device__ __forceinline__ void load_half_row(const float* __restrict__ X, int row, int nrows, int half,
                                              float (&v)[kHalf]) {
    if (row < nrows) {
        const float4* p = reinterpret_cast<const float4*>(X + (size_t)row * kD + half * kHalf);
#pragma unroll
        for (int q = 0; q < kHalf / 4; ++q) {
            float4 t = p[q];
            v[4 * q + 0] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
        }
    } else {
#pragma unroll
        for (int q = 0; q < kHalf; ++q) v[q] = 0.0f;
    }
}

// acc[m][n] = sum_k A[m][k] B[n][k] with lane (l&31) supplying row m of A and row n of B
__device__ __forceinline__ f32x16 tile_dot(const float (&a)[kHalf], const float (&b)[kHalf]) {
    f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < kHalf; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[t], acc, 0, 0, 0);
    return acc;
}

// ------------------------------------------------------------------ pairwise
__global__ __launch_bounds__(256) void pairwise_sim_kernel(const float* __restrict__ E, int P,
                                                           float* __restrict__ S) {
    const int bi = blockIdx.y, bj = blockIdx.x;
    if (bj < bi) return;
    __shared__ float tr[4][32 * 33];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int half = lane >> 5, c = lane & 31;
    const int I = bi * 64 + (wave >> 1) * 32, J = bj * 64 + (wave & 1) * 32;
    float a[kHalf], b[kHalf];
    load_half_row(E, I + c, P, half, a);
    load_half_row(E, J + c, P, half, b);
    const f32x16 acc = tile_dot(a, b);   // rows: I + crow, col: J + c
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int r = I + crow(k, half);
        if (r < P && J + c < P) S[(size_t)r * P + J + c] = acc[k];
    }
    if (bi != bj) {  // block-uniform
        float* t = tr[wave];
#pragma unroll
        for (int k = 0; k < 16; ++k) t[crow(k, half) * 33 + c] = acc[k];
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int cc = 2 * it + half;  // column of the direct tile = row of the mirror
            if (J + cc < P && I + c < P) S[(size_t)(J + cc) * P + I + c] = t[c * 33 + cc];
        }
    }
}

// ------------------------------------------------------------------ pairwise, split-bf16 form
// The same E E^T on the bf16 matrix cores at fp32 grade: every fp32 value is carried as three bf16 planes
// hi + mid + lo (csrc/split.hip) and a product as the six plane products of order <= 2 -- 6 x 2.5 PF-class MFMAs
// instead of one 157 TF-class fp32 MFMA chain: the 32x32 tile costs 48 x 32 = 1536 MFMA cycles instead of 4096, and
// more to the point the kernel becomes what its roofline says it is: a 4 P^2-byte WRITE of S (64 MB at P = 4000).
//   * a pre-pass splits E once into planes (512 k values; in the main kernel the ~25 VALU operations per value,
//     repeated by every workgroup that stages the row, were the whole run time: 64 of 73 us);
//   * one 4-wave workgroup = one 128x128 block of the upper triangle (diagonal blocks last), each wave a 64x64
//     quarter = 2x2 MFMA tiles, 64 accumulator registers; two workgroups per CU;
//   * the 128 + 128 rows are staged 32 k at a time: 192 contiguous bytes per row -> LDS (80-byte row pitch: the
//     16-lane phases of ds_read_b128 hit 16 distinct bank groups), the next stage's global loads in flight under
//     the current stage's 48 MFMAs per wave;
//   * the mirrored block is written from the same accumulators: the C layout gives a lane 4 consecutive rows of one
//     column = 4 consecutive COLUMNS of one row of the mirror -> 16-byte stores; the direct block leaves as 128-byte
//     row segments.  Diagonal 32x32 tiles write their upper triangle twice, so S == S^T bit for bit.
// Measured (tools/pairwise_bench.py, P = 4000): 38 us against 73 us for the exact-fp32 chain and 57 us for rocBLAS;
// with the stores suppressed 25 us, with the compute suppressed 18 us (the store pattern alone reaches 0.45-0.59 of
// the 8 TB/s roofline; a plain fill of S takes 10.8 us = 0.73).  What is left is a fixed ~14 us (two launches and one
// block's dependent chain of four staged loads -- the grid is a single round) and the L2 -> LDS fill rate of the
// 135 MB of plane reads; non-temporal stores of S were tried and are worse (63 us: partial lines no longer merge).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned short ps_f2bf(float f) {
    unsigned int u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

// 4 fp32 -> 3 planes x 4 bf16 (8 bytes each)
__device__ __forceinline__ void ps_split4(const float4 v, uint2 (&pl)[3]) {
    const float x[4] = {v.x, v.y, v.z, v.w};
    unsigned short h[4], m[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        h[i] = ps_f2bf(x[i]);
        const float r1 = x[i] - __uint_as_float((unsigned)h[i] << 16);
        m[i] = ps_f2bf(r1);
        const float r2 = r1 - __uint_as_float((unsigned)m[i] << 16);
        l[i] = ps_f2bf(r2);
    }
    pl[0] = make_uint2((unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16));
    pl[1] = make_uint2((unsigned)m[0] | ((unsigned)m[1] << 16), (unsigned)m[2] | ((unsigned)m[3] << 16));
    pl[2] = make_uint2((unsigned)l[0] | ((unsigned)l[1] << 16), (unsigned)l[2] | ((unsigned)l[3] << 16));
}

constexpr int kPsPitch = 80;                         // bytes per staged row: 32 k x 2 B + 16 B pad
constexpr int kPsRows = 128;                         // rows per side of a workgroup's block
constexpr int kPsPlane = kPsRows * kPsPitch;         // one plane of one side
constexpr int kPsSide = 3 * kPsPlane;                // 30720 B; two sides = 61440 B -> two workgroups per CU

// pre-pass: E (P x 128 fp32) -> E3 (P x 384 bf16), row layout [stage 0..3][plane hi|mid|lo][32 k]: what one staging
// step of the main kernel reads for a row is 192 contiguous bytes.  512 k values in all: the split costs ~25 VALU
// operations per value, which is why it is NOT done in the main kernel (each row is staged by ~60 workgroups).
__global__ __launch_bounds__(256) void pairwise_split_kernel(const float* __restrict__ E, int P, unsigned short* __restrict__ E3) {
    const int id = blockIdx.x * blockDim.x + threadIdx.x;          // (row, 8-k chunk)
    const int row = id >> 4, c = id & 15;
    if (row >= P) return;
    const float4* src = reinterpret_cast<const float4*>(E + (size_t)row * kD + c * 8);
    uint2 a[3], b[3];
    ps_split4(src[0], a);
    ps_split4(src[1], b);
    unsigned short* dst = E3 + (size_t)row * 384 + (c >> 2) * 96 + (c & 3) * 8;
#pragma unroll
    for (int p = 0; p < 3; ++p) *reinterpret_cast<uint4*>(dst + p * 32) = make_uint4(a[p].x, a[p].y, b[p].x, b[p].y);
}

__global__ __launch_bounds__(256, 2) void pairwise_sim_split_kernel(const unsigned short* __restrict__ E3, int P,
                                                                 float* __restrict__ S, int nb, int dbg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];          // 2 * kPsSide
    // linear block id -> (bi <= bj): off-diagonal blocks first (row bi holds nb - 1 - bi of them), the nb lighter
    // diagonal blocks last (they fill the tail of the grid)
    int bi, bj;
    {
        const int noff = nb * (nb - 1) / 2;
        int t = blockIdx.x;
        if (t >= noff) {
            bi = bj = t - noff;
        } else {
            const float n2 = 2.0f * nb - 1.0f;
            bi = (int)((n2 - sqrtf(n2 * n2 - 8.0f * (float)t)) * 0.5f);
            bi = bi < 0 ? 0 : (bi >= nb - 1 ? nb - 2 : bi);
            while (bi > 0 && bi * (nb - 1) - (bi * (bi - 1)) / 2 > t) --bi;
            while ((bi + 1) * (nb - 1) - ((bi + 1) * bi) / 2 <= t) ++bi;
            bj = bi + 1 + (t - (bi * (nb - 1) - (bi * (bi - 1)) / 2));
        }
    }
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wy = wave >> 1, wx = wave & 1;
    const int I = bi * kPsRows, J = bj * kPsRows;

    // staging: 256 rows (A side then B side) x 12 uint4 (3 planes x 4 chunks of 8 k) per stage = 3072 / 256 threads
    uint4 pre[12];
    auto load_stage = [&](int s) {
#pragma unroll
        for (int it = 0; it < 12; ++it) {
            const int id = tid + it * 256;
            const int r = id / 12, q = id - r * 12;
            const int grow = (r < kPsRows ? I + r : J + r - kPsRows);
            pre[it] = grow < P ? *reinterpret_cast<const uint4*>(E3 + (size_t)grow * 384 + s * 96 + q * 8) : make_uint4(0, 0, 0, 0);
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x16){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const bool idle = bi == bj && wy > wx;              // the strictly-lower quarter of a diagonal block

    load_stage(0);
#pragma unroll 1
    for (int s = 0; s < ((dbg & 2) ? 0 : 4); ++s) {
#pragma unroll
        for (int it = 0; it < 12; ++it) {
            const int id = tid + it * 256;
            const int r = id / 12, q = id - r * 12;
            const int side = r >= kPsRows, rr = r - side * kPsRows;
            *reinterpret_cast<uint4*>(lds + side * kPsSide + (q >> 2) * kPsPlane + rr * kPsPitch + (q & 3) * 16) = pre[it];
        }
        __syncthreads();
        if (s < 3) load_stage(s + 1);                    // in flight under the MFMAs below
        if (!idle) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int c = (2 * ks + half) * 16;      // this lane's 8 k of the 32-k row
                bf16x8 fa[2][3], fb[2][3];
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int p = 0; p < 3; ++p) {
                        fa[r][p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(
                            lds + p * kPsPlane + (wy * 64 + r * 32 + l31) * kPsPitch + c));
                        fb[r][p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(
                            lds + kPsSide + p * kPsPlane + (wx * 64 + r * 32 + l31) * kPsPitch + c));
                    }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        f32x16 a = acc[i][j];
                        // smallest terms first: lo.hi, hi.lo, mid.mid, mid.hi, hi.mid, hi.hi
                        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][2], fb[j][0], a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][2], a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][1], a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][0], a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][1], a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][0], a, 0, 0, 0);
                        acc[i][j] = a;
                    }
            }
        }
        __syncthreads();
    }
    if (idle) return;
    if ((dbg & 1) && acc[0][0][0] != 12345.0f) return;

    const bool vec = (P & 3) == 0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int I0 = I + wy * 64 + i * 32, J0 = J + wx * 64 + j * 32;
            if (J0 < I0 || I0 >= P || J0 >= P) continue;          // strictly-lower tiles of a diagonal block; past the end
            const int col = J0 + l31;
            if (I0 == J0) {
                // diagonal tile: (r, c) and (c, r) were accumulated in different term orders -- write the upper
                // triangle to both places so that S is exactly symmetric
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const int r = crow(k, half);
                    if (r <= l31 && col < P) {
                        S[(size_t)(I0 + r) * P + col] = acc[i][j][k];
                        S[(size_t)col * P + I0 + r] = acc[i][j][k];
                    }
                }
                continue;
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) {                        // direct tile: 128-byte row segments
                const int r = I0 + crow(k, half);
                if (r < P && col < P) S[(size_t)r * P + col] = acc[i][j][k];
            }
            if (col < P) {                                        // mirror: this lane's row `col`, 4 x 4 consecutive columns
                float* dst = S + (size_t)col * P + I0 + 4 * half;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c0 = I0 + 8 * q + 4 * half;
                    if (vec && c0 + 3 < P) {
                        *reinterpret_cast<float4*>(dst + 8 * q) = make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1],
                                                                              acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (c0 + e < P) dst[8 * q + e] = acc[i][j][4 * q + e];
                    }
                }
            }
        }
}

// any D (multiple of 4): plain wave-per-row kernel, used when D != 128
__global__ void pairwise_sim_generic(const float* __restrict__ E, int P, int D, float* __restrict__ S) {
    const size_t total = (size_t)P * P;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (size_t)gridDim.x * blockDim.x) {
        int i = (int)(t / P), j = (int)(t - (size_t)i * P);
        const float4* x = reinterpret_cast<const float4*>(E + (size_t)i * D);
        const float4* y = reinterpret_cast<const float4*>(E + (size_t)j * D);
        float acc = 0.0f;
        for (int q = 0; q < D / 4; ++q) {
            float4 u = x[q], v = y[q];
            acc = fmaf(u.x, v.x, acc); acc = fmaf(u.y, v.y, acc);
            acc = fmaf(u.z, v.z, acc); acc = fmaf(u.w, v.w, acc);
        }
        S[t] = acc;
    }
}

// -------------------------------------------------------------------- supcon
// partial statistics of rows i over the j blocks of one split.
// part layout: [nsplit][3][Npad], Npad = 32*ceil(N/32)
__global__ __launch_bounds__(64) void supcon_stats_kernel(const float* __restrict__ F,
                                                          const int* __restrict__ labels, int N,
                                                          float inv_tau, float* __restrict__ part) {
    const int lane = threadIdx.x, half = lane >> 5, c = lane & 31;
    const int nblk = (N + 31) / 32, npad = nblk * 32;
    const int I = blockIdx.x * 32, i = I + c;
    const int sp = blockIdx.y, nsplit = gridDim.y;
    float b[kHalf];
    load_half_row(F, i, N, half, b);
    const int yi = i < N ? labels[i] : -1;
    float m = -__builtin_inff(), A = 0.0f, Bs = 0.0f;
    for (int jb = sp; jb < nblk; jb += nsplit) {
        const int J = jb * 32;
        float a[kHalf];
        load_half_row(F, J + c, N, half, a);
        const f32x16 acc = tile_dot(a, b);  // acc[k] = f_{J+crow(k)} . f_i
        float tmax = -__builtin_inff();
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int j = J + crow(k, half);
            if (j < N) tmax = fmaxf(tmax, acc[k] * inv_tau);
        }
        if (tmax > m) {
            const float sc = expf(m - tmax);  // m = -inf -> 0
            A *= sc; Bs *= sc; m = tmax;
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int j = J + crow(k, half);
            if (j < N && j != i) {
                const float e = expf(acc[k] * inv_tau - m);
                Bs += e;
                if (labels[j] == yi) A += e;
            }
        }
    }
    // merge the two half-waves (same i, disjoint j sets)
    const float m2 = __shfl_xor(m, 32), A2 = __shfl_xor(A, 32), B2 = __shfl_xor(Bs, 32);
    const float M = fmaxf(m, m2);
    const float s1 = (m == -__builtin_inff()) ? 0.0f : expf(m - M);
    const float s2 = (m2 == -__builtin_inff()) ? 0.0f : expf(m2 - M);
    if (half == 0 && i < N) {
        float* p = part + (size_t)sp * 3 * npad;
        p[i] = M;
        p[npad + i] = A * s1 + A2 * s2;
        p[2 * npad + i] = Bs * s1 + B2 * s2;
    }
}

// merge splits -> stats[3][Npad] (m, A, B), per-row loss, mean loss
__global__ __launch_bounds__(256) void supcon_combine_kernel(const float* __restrict__ part, int nsplit,
                                                             const float* __restrict__ w, int N,
                                                             float* __restrict__ stats,
                                                             float* __restrict__ loss) {
    const int npad = ((N + 31) / 32) * 32;
    __shared__ float red[256];
    float local = 0.0f;
    for (int i = threadIdx.x; i < N; i += blockDim.x) {
        float M = -__builtin_inff();
        for (int s = 0; s < nsplit; ++s) M = fmaxf(M, part[(size_t)s * 3 * npad + i]);
        float A = 0.0f, B = 0.0f;
        for (int s = 0; s < nsplit; ++s) {
            const float* p = part + (size_t)s * 3 * npad;
            const float ms = p[i];
            const float sc = (ms == -__builtin_inff()) ? 0.0f : expf(ms - M);
            A += p[npad + i] * sc;
            B += p[2 * npad + i] * sc;
        }
        stats[i] = M; stats[npad + i] = A; stats[2 * npad + i] = B;
        local += -logf(A / B) * w[i];   // sim_loss.py:76-78
    }
    red[threadIdx.x] = local;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = red[0] / (float)N;
}

// partial dF of rows i over the j blocks of one split.  dpart: [nsplit][Npad][128]
__global__ __launch_bounds__(64) void supcon_grad_kernel(const float* __restrict__ F,
                                                         const int* __restrict__ labels,
                                                         const float* __restrict__ w,
                                                         const float* __restrict__ stats, int N,
                                                         float inv_tau, float out_scale,
                                                         float* __restrict__ dpart) {
    const int lane = threadIdx.x, half = lane >> 5, c = lane & 31;
    const int nblk = (N + 31) / 32, npad = nblk * 32;
    const int I = blockIdx.x * 32, i = I + c;
    const int sp = blockIdx.y, nsplit = gridDim.y;
    const float invN = 1.0f / (float)N;
    float b[kHalf];
    load_half_row(F, i, N, half, b);
    const bool vi = i < N;
    const int yi = vi ? labels[i] : -1;
    const float mi = vi ? stats[i] : 0.0f;
    const float wi = vi ? w[i] * invN : 0.0f;
    const float ciA = vi ? wi / stats[npad + i] : 0.0f;
    const float ciB = vi ? wi / stats[2 * npad + i] : 0.0f;
    f32x16 out[4];
#pragma unroll
    for (int dc = 0; dc < 4; ++dc) out[dc] = (f32x16){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int jb = sp; jb < nblk; jb += nsplit) {
        const int J = jb * 32;
        float a[kHalf];
        load_half_row(F, J + c, N, half, a);
        const f32x16 acc = tile_dot(a, b);
        float H[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int j = J + crow(k, half);
            float h = 0.0f;
            if (vi && j < N && j != i) {
                const float s = acc[k] * inv_tau;
                const bool same = labels[j] == yi;
                const float wj = w[j] * invN;
                // G_ij + G_ji: e_ij (w_i/N)(1/B_i - [same]/A_i) + e_ji (w_j/N)(1/B_j - [same]/A_j)
                const float g1 = expf(s - mi) * (ciB - (same ? ciA : 0.0f));
                const float g2 = expf(s - stats[j]) *
                                 (wj / stats[2 * npad + j] - (same ? wj / stats[npad + j] : 0.0f));
                h = g1 + g2;
            }
            H[k] = h;
        }
        // dF_i += sum_j H_ij f_j : A operand = H (already in A layout), B operand = rows of F
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int j = J + crow(k, half);
            const float* fj = F + (size_t)j * kD + c;
#pragma unroll
            for (int dc = 0; dc < 4; ++dc) {
                const float bv = j < N ? fj[dc * 32] : 0.0f;
                out[dc] = __builtin_amdgcn_mfma_f32_32x32x2f32(H[k], bv, out[dc], 0, 0, 0);
            }
        }
    }
    float* dp = dpart + (size_t)sp * npad * kD;
#pragma unroll
    for (int dc = 0; dc < 4; ++dc)
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int r = I + crow(k, half);
            if (r < N) dp[(size_t)r * kD + dc * 32 + c] = out[dc][k] * out_scale;
        }
}

__global__ void supcon_grad_combine(const float* __restrict__ dpart, int nsplit, int N, int npad,
                                    float* __restrict__ dF) {
    const int total = N * kD / 4;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
        float4 acc = reinterpret_cast<const float4*>(dpart)[t];
        for (int s = 1; s < nsplit; ++s) {
            float4 v = reinterpret_cast<const float4*>(dpart + (size_t)s * npad * kD)[t];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        reinterpret_cast<float4*>(dF)[t] = acc;
    }
}

int supcon_nsplit(int N) {
    const int nblk = (N + 31) / 32;
    int s = 1024 / (nblk > 0 ? nblk : 1);
    if (s > 16) s = 16;
    if (s > nblk) s = nblk;
    if (s < 1) s = 1;
    return s;
}

}  // namespace

ODW_EXPORT int64_t odw_pairwise_sim_workspace(int P, int D) {
    return D == kD && P > 0 ? odw_align_up((int64_t)P * 384 * 2, 256) : 0;
}

ODW_EXPORT int odw_pairwise_sim_ws(const float* E, int P, int D, float* S, void* workspace, int64_t workspace_bytes,
                                   void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    ODW_REQUIRE(P >= 0 && D > 0 && D % 4 == 0, "pairwise_sim: bad dims P=%d D=%d", P, D);
    if (P == 0) return ODW_OK;
    ODW_REQUIRE(E && S, "pairwise_sim: null pointer");
    ODW_REQUIRE((((uintptr_t)E) & 15) == 0, "pairwise_sim: E must be 16-byte aligned");
    static const bool fp32_chain = getenv("ODW_PAIRWISE_FP32") != nullptr;      // force the exact-fp32 MFMA form (comparison)
    if (D == kD && !fp32_chain && workspace && workspace_bytes >= odw_pairwise_sim_workspace(P, D) &&
        (((uintptr_t)S) & 15) == 0 && (((uintptr_t)workspace) & 15) == 0) {
        // split-bf16 form: planes once, then the 128 x 128 blocks of the upper triangle
        unsigned short* E3 = (unsigned short*)workspace;
        pairwise_split_kernel<<<(P * 16 + 255) / 256, 256, 0, stream>>>(E, P, E3);
        ODW_CHECK_LAUNCH("pairwise_split_kernel");
        const int nb = (P + kPsRows - 1) / kPsRows;
#ifdef ODW_EXPERIMENTS      // timing experiments that skip compute or stores (WRONG results): experiment builds only
        static const int dbg = getenv("ODW_PAIRWISE_DBG") ? atoi(getenv("ODW_PAIRWISE_DBG")) : 0;
#else
        const int dbg = 0;
#endif
        ODW_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(pairwise_sim_split_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kPsSide), "pairwise attr");
        pairwise_sim_split_kernel<<<nb * (nb + 1) / 2, 256, 2 * kPsSide, stream>>>(E3, P, S, nb, dbg);
        ODW_CHECK_LAUNCH("pairwise_sim_split_kernel");
    } else if (D == kD) {
        const int nb = (P + 63) / 64;
        pairwise_sim_kernel<<<dim3(nb, nb), 256, 0, stream>>>(E, P, S);
        ODW_CHECK_LAUNCH("pairwise_sim_kernel");
    } else {
        size_t total = (size_t)P * P;
        int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
        pairwise_sim_generic<<<grid, 256, 0, stream>>>(E, P, D, S);
        ODW_CHECK_LAUNCH("pairwise_sim_generic");
    }
    return ODW_OK;
}

ODW_EXPORT int odw_pairwise_sim(const float* E, int P, int D, float* S, void* stream_) {
    return odw_pairwise_sim_ws(E, P, D, S, nullptr, 0, stream_);          // no workspace: the exact-fp32 MFMA chain
}

ODW_EXPORT int64_t odw_supcon_workspace(int N) {
    if (N < 1) N = 1;
    const int64_t npad = ((N + 31) / 32) * 32;
    const int64_t ns = supcon_nsplit(N);
    return odw_align_up(3 * npad * 4, 256) + odw_align_up(ns * 3 * npad * 4, 256) +
           odw_align_up(ns * npad * kD * 4, 256);
}

ODW_EXPORT int odw_supcon_v2(const float* F, const int32_t* labels, const float* w, int N, int D, float tau,
                             float grad_scale, float* loss, float* dF, void* workspace,
                             int64_t workspace_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    ODW_REQUIRE(N >= 1, "supcon_v2: N=%d (the reference would take the mean of an empty tensor)", N);
    ODW_REQUIRE(D == kD, "supcon_v2: D=%d unsupported (Sim_Net emits 128-d embeddings)", D);
    ODW_REQUIRE(tau > 0.0f, "supcon_v2: temperature must be > 0");
    ODW_REQUIRE(F && labels && w && loss, "supcon_v2: null pointer");
    ODW_REQUIRE((((uintptr_t)F) & 15) == 0 && (dF == nullptr || (((uintptr_t)dF) & 15) == 0),
                "supcon_v2: F/dF must be 16-byte aligned");
    if (!workspace || workspace_bytes < odw_supcon_workspace(N)) {
        odw_set_error("supcon_v2: workspace %lld < %lld bytes", (long long)workspace_bytes,
                      (long long)odw_supcon_workspace(N));
        return ODW_EWORKSPACE;
    }
    const int nblk = (N + 31) / 32, npad = nblk * 32;
    const int ns = supcon_nsplit(N);
    unsigned char* p = (unsigned char*)workspace;
    float* stats = (float*)p; p += odw_align_up((int64_t)3 * npad * 4, 256);
    float* part = (float*)p;  p += odw_align_up((int64_t)ns * 3 * npad * 4, 256);
    float* dpart = (float*)p;
    const float inv_tau = 1.0f / tau;
    supcon_stats_kernel<<<dim3(nblk, ns), 64, 0, stream>>>(F, labels, N, inv_tau, part);
    ODW_CHECK_LAUNCH("supcon_stats_kernel");
    supcon_combine_kernel<<<1, 256, 0, stream>>>(part, ns, w, N, stats, loss);
    ODW_CHECK_LAUNCH("supcon_combine_kernel");
    if (dF) {
        supcon_grad_kernel<<<dim3(nblk, ns), 64, 0, stream>>>(F, labels, w, stats, N, inv_tau,
                                                             grad_scale * inv_tau, dpart);
        ODW_CHECK_LAUNCH("supcon_grad_kernel");
        int total = N * kD / 4;
        supcon_grad_combine<<<(total + 255) / 256, 256, 0, stream>>>(dpart, ns, N, npad, dF);
        ODW_CHECK_LAUNCH("supcon_grad_combine");
    }
    return ODW_OK;
}
