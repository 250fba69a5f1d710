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
#include "odw_rng.h"
#include "odw_planes.h"

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
template <bool OUT_F32>
__global__ __launch_bounds__(256) void stack_clean_aug_kernel(const float* __restrict__ pooled,
                                                              const float* __restrict__ block,
                                                              const float* __restrict__ block_sum, int P, int CS, int S,
                                                              float numel, void* __restrict__ outv, int ld) {
    __shared__ float keep[kMaxS];
    const int p = blockIdx.x;
    for (int s = threadIdx.x; s < S; s += blockDim.x) keep[s] = block[(size_t)p * S + s];
    __syncthreads();
    const float sum = *block_sum;
    const float4* src = reinterpret_cast<const float4*>(pooled + (size_t)p * CS);
    if (OUT_F32) {          // fp32 operand (the split kernels of csrc/split.hip lay it out for the matrix cores)
        float* out = reinterpret_cast<float*>(outv);
        float4* clean = reinterpret_cast<float4*>(out + (size_t)p * ld);
        float4* aug = reinterpret_cast<float4*>(out + (size_t)(P + p) * ld);
        for (int q = threadIdx.x; q < CS / 4; q += blockDim.x) {
            const float4 v = src[q];
            const int s0 = (q * 4) % S;
            const float x[4] = {v.x, v.y, v.z, v.w};
            float a[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                int s = s0 + t;
                s = s >= S ? s - S : s;
                a[t] = ((x[t] * keep[s]) * numel) / sum;
            }
            clean[q] = v;
            aug[q] = make_float4(a[0], a[1], a[2], a[3]);
        }
        for (int k = CS + threadIdx.x; k < ld; k += blockDim.x) {
            out[(size_t)p * ld + k] = 0.0f;
            out[(size_t)(P + p) * ld + k] = 0.0f;
        }
        return;
    }
    unsigned short* out = reinterpret_cast<unsigned short*>(outv);
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


// ---- the contrastive views of the sampled rows (loss.py:292-305: drop_pool + noise_pool of pooled[rows]) ----------
// For one (image, class): k sampled proposals -> 2k rows of the bf16 GEMM operand, rows [out_row0, +k) the
// DropBlock(block 1) view ((x * keep) * numel) / sum with keep = !(u < gamma) drawn per (row, cell), rows
// [out_row0 + k, +k) the noise view z*x + x with z ~ N(0,1) per element (Box-Muller on pairs, as noise_kernel).
// The PyTorch rendition is 14 launches per class (gather, uniform, compare, cast, max-pool, 1-x, mul, mul, sum,
// div, noise, cat, cast) in a phase where the GPU is drained and waits for each of them.
__global__ __launch_bounds__(256) void rows_keep_sum_kernel(int n, float gamma, uint32_t k0, uint32_t k1,
                                                            float* __restrict__ sum_out) {
    __shared__ float red[256];
    float acc = 0.0f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) acc += odw_uniform((uint32_t)i, k0, k1) < gamma ? 0.0f : 1.0f;
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) *sum_out = red[0];          // a count <= 2^24: exact in fp32 whatever the order
}

__device__ __forceinline__ void normal_pair(uint32_t p, uint32_t k0, uint32_t k1, float& z0, float& z1) {
    const float u1 = 1.0f - odw_uniform(2 * p, k0, k1);
    const float t = 6.283185307179586f * odw_uniform(2 * p + 1, k0, k1);
    const float r = sqrtf(-2.0f * logf(u1));
    z0 = r * cosf(t);
    z1 = r * sinf(t);
}

constexpr int kRowSlices = 8;          // workgroups per row of rows_drop_noise_kernel (blockIdx.y)

// BWD = false: out rows from pooled;  BWD = true: dpooled[rows[r]] += d(drop row) and d(noise row) folded back
// (body: r = the sampled row's position inside its (image, class) group of k rows; src_row = the row of `pooled` / `dpooled`
// it belongs to; row0 = the group's first row in the stacked views.  The per-group launch below and the grouped launch
// whose group boundaries live on the device share it.)
template <bool BWD, bool DX_F32, bool SRC_BF16 = false, bool OUT_F32 = false, bool STORE = false>
__device__ __forceinline__ void rows_drop_noise_body(const float* __restrict__ pooled, const void* __restrict__ dXv,
                                                     const int r, const size_t src_row, int k, int CS,
                                                     int S, float gamma, uint32_t kd0, uint32_t kd1,
                                                     uint32_t kn0, uint32_t kn1, const float sum,
                                                     unsigned short* __restrict__ out, int ld, int row0,
                                                     float* __restrict__ dpooled, float* keep) {
    for (int s = threadIdx.x; s < S; s += blockDim.x)
        keep[s] = odw_uniform((uint32_t)(r * S + s), kd0, kd1) < gamma ? 0.0f : 1.0f;
    __syncthreads();
    const float numel = (float)((double)k * S);
    // blockIdx.y = a slice of the row: k ~ 200 rows alone leave most of the chip idle behind one 256-thread block per CU
    // (37 us per launch on 225 x 25088 elements; the Box-Muller pair of every element is what a thread waits on)
    const int per = (CS / 4 + gridDim.y - 1) / gridDim.y;
    const int q_end = min(CS / 4, (int)(blockIdx.y + 1) * per);
    for (int q = blockIdx.y * per + threadIdx.x; q < q_end; q += blockDim.x) {
        const int s0 = (q * 4) % S;
        const uint32_t e = (uint32_t)r * (uint32_t)CS + (uint32_t)q * 4u;       // element index in the (k, C, S) draw
        float z[4];
        normal_pair(e / 2, kn0, kn1, z[0], z[1]);
        normal_pair(e / 2 + 1, kn0, kn1, z[2], z[3]);
        if (!BWD) {
            float x[4];
            if (SRC_BF16) {          // rows of the stacked bf16 operand itself (row stride ld)
                const uint2 v = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(pooled) + src_row * ld + q * 4);
                x[0] = bf2f(v.x & 0xffff); x[1] = bf2f(v.x >> 16); x[2] = bf2f(v.y & 0xffff); x[3] = bf2f(v.y >> 16);
            } else {
                const float4 v = *reinterpret_cast<const float4*>(pooled + src_row * CS + q * 4);
                x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w;
            }
            if (OUT_F32) {
                float df[4], nf[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    int sidx = s0 + t;
                    sidx = sidx >= S ? sidx - S : sidx;
                    df[t] = ((x[t] * keep[sidx]) * numel) / sum;
                    nf[t] = z[t] * x[t] + x[t];
                }
                float* of = reinterpret_cast<float*>(out);
                *reinterpret_cast<float4*>(of + (size_t)(row0 + r) * ld + q * 4) = make_float4(df[0], df[1], df[2], df[3]);
                *reinterpret_cast<float4*>(of + (size_t)(row0 + k + r) * ld + q * 4) = make_float4(nf[0], nf[1], nf[2], nf[3]);
                continue;
            }
            unsigned short d[4], n[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                int sidx = s0 + t;
                sidx = sidx >= S ? sidx - S : sidx;
                d[t] = f2bf(((x[t] * keep[sidx]) * numel) / sum);
                n[t] = f2bf(z[t] * x[t] + x[t]);
            }
            *reinterpret_cast<uint2*>(out + (size_t)(row0 + r) * ld + q * 4) =
                make_uint2((unsigned)d[0] | ((unsigned)d[1] << 16), (unsigned)d[2] | ((unsigned)d[3] << 16));
            *reinterpret_cast<uint2*>(out + (size_t)(row0 + k + r) * ld + q * 4) =
                make_uint2((unsigned)n[0] | ((unsigned)n[1] << 16), (unsigned)n[2] | ((unsigned)n[3] << 16));
        } else {
            float gd[4], gn[4];
            if (DX_F32) {
                const float* dX = reinterpret_cast<const float*>(dXv);
                const float4 a = *reinterpret_cast<const float4*>(dX + (size_t)(row0 + r) * ld + q * 4);
                const float4 b = *reinterpret_cast<const float4*>(dX + (size_t)(row0 + k + r) * ld + q * 4);
                gd[0] = a.x; gd[1] = a.y; gd[2] = a.z; gd[3] = a.w;
                gn[0] = b.x; gn[1] = b.y; gn[2] = b.z; gn[3] = b.w;
            } else {
                const unsigned short* dX = reinterpret_cast<const unsigned short*>(dXv);
                const uint2 a = *reinterpret_cast<const uint2*>(dX + (size_t)(row0 + r) * ld + q * 4);
                const uint2 b = *reinterpret_cast<const uint2*>(dX + (size_t)(row0 + k + r) * ld + q * 4);
                gd[0] = bf2f(a.x & 0xffff); gd[1] = bf2f(a.x >> 16); gd[2] = bf2f(a.y & 0xffff); gd[3] = bf2f(a.y >> 16);
                gn[0] = bf2f(b.x & 0xffff); gn[1] = bf2f(b.x >> 16); gn[2] = bf2f(b.y & 0xffff); gn[3] = bf2f(b.y >> 16);
            }
            float4* dst = reinterpret_cast<float4*>(dpooled + src_row * CS + q * 4);
            // rows of one launch are distinct; launches are stream-ordered.  STORE: the row is this launch's alone (an entry
            // of the pooling node's side buffer): written, not added to -- the buffer needs no zero fill and is not read
            float4 acc = STORE ? make_float4(0.0f, 0.0f, 0.0f, 0.0f) : *dst;
            float add[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                int sidx = s0 + t;
                sidx = sidx >= S ? sidx - S : sidx;
                add[t] = ((gd[t] * keep[sidx]) * numel) / sum + (z[t] * gn[t] + gn[t]);
            }
            acc.x += add[0]; acc.y += add[1]; acc.z += add[2]; acc.w += add[3];
            *dst = acc;
        }
    }
    if (!BWD && OUT_F32) {
        float* of = reinterpret_cast<float*>(out);
        for (int c = CS + threadIdx.x; c < ld; c += blockDim.x) {
            of[(size_t)(row0 + r) * ld + c] = 0.0f;
            of[(size_t)(row0 + k + r) * ld + c] = 0.0f;
        }
    } else if (!BWD) {
        for (int c = CS + threadIdx.x; c < ld; c += blockDim.x) {
            out[(size_t)(row0 + r) * ld + c] = 0;
            out[(size_t)(row0 + k + r) * ld + c] = 0;
        }
    }
}

template <bool BWD, bool DX_F32, bool SRC_BF16 = false, bool OUT_F32 = false, bool STORE = false>
__global__ __launch_bounds__(256) void rows_drop_noise_kernel(const float* __restrict__ pooled, const void* __restrict__ dXv,
                                                              const int* __restrict__ rows, int row_base, int k, int CS,
                                                              int S, float gamma, uint32_t kd0, uint32_t kd1,
                                                              uint32_t kn0, uint32_t kn1,
                                                              const float* __restrict__ keep_sum,
                                                              unsigned short* __restrict__ out, int ld, int row0,
                                                              float* __restrict__ dpooled) {
    __shared__ float keep[kMaxS];
    const int r = blockIdx.x;
    rows_drop_noise_body<BWD, DX_F32, SRC_BF16, OUT_F32, STORE>(pooled, dXv, r, (size_t)(row_base + rows[r]), k, CS, S, gamma, kd0,
                                                                kd1, kn0, kn1, *keep_sum, out, ld, row0, dpooled, keep);
}

// ---- grouped forms (round 6): every (image, class) group of the step in ONE launch, group boundaries on the device ----
// The loss's first selection kernel (discover.hip: discover_iou) leaves the sampled rows of each positive class in device
// memory; odw_loss_lists_a (loss_lists.hip) turns their counts into k[g] and the entry prefix e0[g].  Nothing of that is
// known to the host when these kernels are launched: the grid covers the CAPACITY (every entry that could exist), an entry
// e < *n_entries finds its group by walking the prefix, and the per-group constants the host does know -- the
// counter-based keys of the group's draws -- come from a table uploaded before the step's first read would have happened.
constexpr int kViewGrid = 512;        // entries a grouped launch puts in its grid (the kernels stride over the rest)
struct ViewGroups {
    const int* n_entries;       // [1]   number of sampled rows over all groups (E1)
    const int* e0;              // [G+1] entry prefix: group g owns entries [e0[g], e0[g+1]); its views are rows [2 e0[g], 2 e0[g+1])
    const uint32_t* keys;       // [G][4] kd0, kd1 (drop mask draw), kn0, kn1 (noise draw)
    const int* src_row;         // [E1]  row of the source (a proposal of the concatenated batch) of every entry
    const float* keep_sum;      // [G]   sum of the group's keep mask (rows_keep_sum_grouped_kernel)
    int G;
};
__device__ __forceinline__ int view_group_of(const ViewGroups& vg, int e) {
    int g = 0;
    while (g + 1 < vg.G && e >= vg.e0[g + 1]) ++g;
    return g;
}

__global__ __launch_bounds__(256) void rows_keep_sum_grouped_kernel(ViewGroups vg, int S, float gamma, float* __restrict__ sum_out) {
    __shared__ float red[256];
    const int g = blockIdx.x;
    const int n = (vg.e0[g + 1] - vg.e0[g]) * S;
    const uint32_t k0 = vg.keys[4 * g], k1 = vg.keys[4 * g + 1];
    float acc = 0.0f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) acc += odw_uniform((uint32_t)i, k0, k1) < gamma ? 0.0f : 1.0f;
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) sum_out[g] = red[0];          // a count <= 2^24: exact in fp32 whatever the order
}

// backward of the grouped views into the pooling node's side buffer: entry e's row is *dst_off + e (STORE: written, not added)
template <bool DX_F32>
__global__ __launch_bounds__(256) void rows_views_bwd_store_grouped_kernel(const void* __restrict__ dXv, int ld, ViewGroups vg,
                                                                           int CS, int S, float gamma,
                                                                           const int* __restrict__ dst_off,
                                                                           float* __restrict__ extra) {
    __shared__ float keep[kMaxS];
    const int n_e = *vg.n_entries;
    for (int e = blockIdx.x; e < n_e; e += gridDim.x) {       // (a bounded grid strides over the entries that exist)
        const int g = view_group_of(vg, e);
        const int e0 = vg.e0[g], k = vg.e0[g + 1] - e0;
        const uint32_t* ky = vg.keys + 4 * g;
        rows_drop_noise_body<true, DX_F32, false, false, true>(nullptr, dXv, e - e0, (size_t)((dst_off ? *dst_off : 0) + e), k, CS, S,
                                                               gamma, ky[0], ky[1], ky[2], ky[3], vg.keep_sum[g], nullptr, ld, 2 * e0,
                                                               extra, keep);
        __syncthreads();
    }
}

// ---- row-wise L2 normalisation of the 128-d embeddings (Sim_Net.forward, sim_head/sim_net.py:25-26: F.normalize) ------
// y = x / max(||x||, eps); backward dx = (g - y (g.y)) / max(||x||, eps).  One wavefront per row; the PyTorch
// rendition is 3 launches forward and ~12 backward, twice per step, in the latency-bound part of the step.
template <bool BWD>
__global__ __launch_bounds__(256) void l2norm_rows_kernel(const float* __restrict__ a, const float* __restrict__ y_in,
                                                          const float* __restrict__ norm_in, int R, int D, float eps,
                                                          float* __restrict__ out, float* __restrict__ norm_out,
                                                          const int* __restrict__ r_dev) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r_dev) { const int rd = *r_dev; R = rd < R ? rd : R; }      // rows that exist (device-resident count)
    if (row >= R) return;
    const float* src = a + (size_t)row * D;
    float acc = 0.0f;
    if (!BWD) {
        for (int j = lane; j < D; j += 64) acc += src[j] * src[j];
    } else {
        const float* y = y_in + (size_t)row * D;
        for (int j = lane; j < D; j += 64) acc += src[j] * y[j];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (!BWD) {
        const float n = sqrtf(acc);
        const float d = fmaxf(n, eps);
        for (int j = lane; j < D; j += 64) out[(size_t)row * D + j] = src[j] / d;
        if (lane == 0) norm_out[row] = n;
    } else {
        const float n = norm_in[row];
        const float d = fmaxf(n, eps);
        const float* y = y_in + (size_t)row * D;
        // below eps the forward is x / eps (no dependence on the norm): the projection term drops out
        const float dot = n > eps ? acc : 0.0f;
        for (int j = lane; j < D; j += 64) out[(size_t)row * D + j] = (src[j] - y[j] * dot) / d;
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
    stack_clean_aug_kernel<false><<<P, 256, 0, stream>>>(pooled, block, block_sum, P, (int)cs, S, (float)((double)P * S),
                                                         out_bf16, ld);
    ODW_CHECK_LAUNCH("stack_clean_aug_kernel");
    return ODW_OK;
}

ODW_EXPORT int odw_stack_clean_aug_f32(const float* pooled, const float* block, const float* block_sum, int P, int C,
                                       int S, float* out, int ld, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    ODW_REQUIRE(P >= 0 && C > 0 && S > 0 && S <= kMaxS, "stack_clean_aug_f32: bad dims P=%d C=%d S=%d", P, C, S);
    if (P == 0) return ODW_OK;
    const long cs = (long)C * S;
    ODW_REQUIRE(pooled && block && block_sum && out, "stack_clean_aug_f32: null pointer");
    ODW_REQUIRE(cs % 4 == 0 && ld >= cs && ld % 4 == 0 && S >= 4, "stack_clean_aug_f32: C*S=%ld must be a multiple of 4 and fit ld=%d", cs, ld);
    ODW_REQUIRE((((uintptr_t)pooled) & 15) == 0 && (((uintptr_t)out) & 15) == 0, "stack_clean_aug_f32: alignment");
    stack_clean_aug_kernel<true><<<P, 256, 0, stream>>>(pooled, block, block_sum, P, (int)cs, S, (float)((double)P * S), out, ld);
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

ODW_EXPORT int odw_rows_drop_noise(const void* pooled, int src_is_bf16, const int* rows, int row_base, int k, int C,
                                   int S, float gamma, uint32_t kd0, uint32_t kd1, uint32_t kn0, uint32_t kn1,
                                   float* keep_sum, void* out_bf16, int ld, int out_row0, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    ODW_REQUIRE(k >= 0 && C > 0 && S >= 4 && S <= kMaxS && row_base >= 0 && out_row0 >= 0, "rows_drop_noise: bad dims");
    if (k == 0) return ODW_OK;
    const long cs = (long)C * S;
    ODW_REQUIRE(pooled && rows && keep_sum && out_bf16, "rows_drop_noise: null pointer");
    ODW_REQUIRE(cs % 4 == 0 && ld >= cs && ld % 4 == 0 && (long)k * cs < (1ll << 32), "rows_drop_noise: C*S=%ld, ld=%d", cs, ld);
    ODW_REQUIRE((((uintptr_t)pooled) & 15) == 0 && (((uintptr_t)out_bf16) & 7) == 0, "rows_drop_noise: alignment");
    rows_keep_sum_kernel<<<1, 256, 0, stream>>>(k * S, gamma, kd0, kd1, keep_sum);
    if (src_is_bf16)          // source rows have the same stride as the output (rows of one (2R x ld) operand)
        rows_drop_noise_kernel<false, false, true><<<dim3(k, kRowSlices), 256, 0, stream>>>((const float*)pooled, nullptr, rows, row_base, k,
                                                                          (int)cs, S, gamma, kd0, kd1, kn0, kn1, keep_sum,
                                                                          (unsigned short*)out_bf16, ld, out_row0, nullptr);
    else
        rows_drop_noise_kernel<false, false><<<dim3(k, kRowSlices), 256, 0, stream>>>((const float*)pooled, nullptr, rows, row_base, k, (int)cs,
                                                                    S, gamma, kd0, kd1, kn0, kn1, keep_sum,
                                                                    (unsigned short*)out_bf16, ld, out_row0, nullptr);
    ODW_CHECK_LAUNCH("rows_drop_noise_kernel");
    return ODW_OK;
}

ODW_EXPORT int odw_rows_drop_noise_f32(const float* pooled, const int* rows, int row_base, int k, int C, int S, float gamma,
                                       uint32_t kd0, uint32_t kd1, uint32_t kn0, uint32_t kn1, float* keep_sum, float* out,
                                       int ld, int out_row0, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    ODW_REQUIRE(k >= 0 && C > 0 && S >= 4 && S <= kMaxS && row_base >= 0 && out_row0 >= 0, "rows_drop_noise_f32: bad dims");
    if (k == 0) return ODW_OK;
    const long cs = (long)C * S;
    ODW_REQUIRE(pooled && rows && keep_sum && out, "rows_drop_noise_f32: null pointer");
    ODW_REQUIRE(cs % 4 == 0 && ld >= cs && ld % 4 == 0 && (long)k * cs < (1ll << 32), "rows_drop_noise_f32: C*S=%ld, ld=%d", cs, ld);
    ODW_REQUIRE((((uintptr_t)pooled) & 15) == 0 && (((uintptr_t)out) & 15) == 0, "rows_drop_noise_f32: alignment");
    rows_keep_sum_kernel<<<1, 256, 0, stream>>>(k * S, gamma, kd0, kd1, keep_sum);
    rows_drop_noise_kernel<false, false, false, true><<<dim3(k, kRowSlices), 256, 0, stream>>>(pooled, nullptr, rows, row_base, k, (int)cs, S, gamma,
                                                                             kd0, kd1, kn0, kn1, keep_sum,
                                                                             reinterpret_cast<unsigned short*>(out), ld, out_row0,
                                                                             nullptr);
    ODW_CHECK_LAUNCH("rows_drop_noise_kernel");
    return ODW_OK;
}

// ---- the same two views written as the OPERAND of the first head Linear in the "bf16x2f" mode (round 5) ---------------
// There the pooling kernel leaves the clean rows as two cell-major planes [hi | mid] (k' = cell * C + channel: what
// gemm_nt_cm_kernel sweeps) and the views went pooled32 (an fp32 copy of all P rows that only these ~10^2-10^3 sampled rows
// ever read: 200 MB of the pooling kernel's 700) -> fp32 views -> split_rows_cm -> cell-major planes.  This kernel reads the
// sampled rows FROM the clean planes (x = hi + mid: the 16 leading bits of the pooled value, which is all the three plane
// products of the forward see of any operand) and writes what the Linear reads: the views' cell-major planes for the forward
// and their channel-major hi plane (the bf16 the single-plane backward transposes) -- no fp32 copy of the pooled rows, no fp32
// views, no split pass.  Same draws as rows_drop_noise_kernel (element index in the (k, C, S) draw, Box-Muller on pairs),
// same evaluation order.  A workgroup = (sampled row, 64 channels): the slice is staged channel-major in LDS so that
// the draw runs over pairs of consecutive elements and both layouts leave as full 16-byte vectors.
constexpr int kViewCh = 64;
__device__ __forceinline__ void rows_views_cm_body(const unsigned short* __restrict__ src, long long ld_src,
                                                   long long src_mid, const int r, const size_t src_row, const int c0,
                                                   int k, int C, int S, float gamma, uint32_t kd0, uint32_t kd1,
                                                   uint32_t kn0, uint32_t kn1, const float sum,
                                                   unsigned short* __restrict__ out_cm, long long ld_cm,
                                                   long long cm_mid, unsigned short* __restrict__ out_hi,
                                                   long long ld_hi, int row0, float* vlds, float* keep) {
    const int n = kViewCh * S;
    float* xs = vlds;
    float* dl = vlds + n;
    float* nl = vlds + 2 * n;
    for (int s = threadIdx.x; s < S; s += blockDim.x)
        keep[s] = odw_uniform((uint32_t)(r * S + s), kd0, kd1) < gamma ? 0.0f : 1.0f;
    for (int t = threadIdx.x; t < S * 8; t += blockDim.x) {
        const int cell = t >> 3, cg = t & 7;
        const unsigned short* p = src + src_row * ld_src + (size_t)cell * C + c0 + cg * 8;
        const uint4 h = *reinterpret_cast<const uint4*>(p), m = *reinterpret_cast<const uint4*>(p + src_mid);
        const unsigned hw[4] = {h.x, h.y, h.z, h.w}, mw[4] = {m.x, m.y, m.z, m.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            xs[(cg * 8 + 2 * j) * S + cell] = bf2f(hw[j] & 0xffffu) + bf2f(mw[j] & 0xffffu);
            xs[(cg * 8 + 2 * j + 1) * S + cell] = bf2f(hw[j] >> 16) + bf2f(mw[j] >> 16);
        }
    }
    __syncthreads();
    const float numel = (float)((double)k * S);
    const int CS = C * S;
    for (int q = threadIdx.x; q < n / 4; q += blockDim.x) {
        const int l0 = q * 4, s0 = l0 % S;
        const uint32_t e = (uint32_t)r * (uint32_t)CS + (uint32_t)c0 * (uint32_t)S + (uint32_t)l0;
        float z[4];
        normal_pair(e / 2, kn0, kn1, z[0], z[1]);
        normal_pair(e / 2 + 1, kn0, kn1, z[2], z[3]);
        const float4 v = *reinterpret_cast<const float4*>(xs + l0);
        const float x[4] = {v.x, v.y, v.z, v.w};
        float df[4], nf[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            int sidx = s0 + t;
            sidx = sidx >= S ? sidx - S : sidx;
            df[t] = ((x[t] * keep[sidx]) * numel) / sum;
            nf[t] = z[t] * x[t] + x[t];
        }
        *reinterpret_cast<float4*>(dl + l0) = make_float4(df[0], df[1], df[2], df[3]);
        *reinterpret_cast<float4*>(nl + l0) = make_float4(nf[0], nf[1], nf[2], nf[3]);
        const size_t col = (size_t)c0 * S + l0;
        *reinterpret_cast<uint2*>(out_hi + (size_t)(row0 + r) * ld_hi + col) = make_uint2(odwpl::pk(df[0], df[1]), odwpl::pk(df[2], df[3]));
        *reinterpret_cast<uint2*>(out_hi + (size_t)(row0 + k + r) * ld_hi + col) = make_uint2(odwpl::pk(nf[0], nf[1]), odwpl::pk(nf[2], nf[3]));
    }
    __syncthreads();
    for (int t = threadIdx.x; t < S * 8; t += blockDim.x) {
        const int cell = t >> 3, cg = t & 7;
#pragma unroll
        for (int view = 0; view < 2; ++view) {
            const float* vv = view ? nl : dl;
            unsigned hi[4], mid[4], lo[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                odwpl::split2(vv[(cg * 8 + 2 * j) * S + cell], vv[(cg * 8 + 2 * j + 1) * S + cell], false, hi[j], mid[j], lo[j]);
            unsigned short* dst = out_cm + (size_t)(row0 + (view ? k : 0) + r) * ld_cm + (size_t)cell * C + c0 + cg * 8;
            *reinterpret_cast<uint4*>(dst) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
            *reinterpret_cast<uint4*>(dst + cm_mid) = make_uint4(mid[0], mid[1], mid[2], mid[3]);
        }
    }
}

__global__ __launch_bounds__(256) void rows_views_cm_kernel(const unsigned short* __restrict__ src, long long ld_src,
                                                            long long src_mid, const int* __restrict__ rows, int row_base,
                                                            int k, int C, int S, float gamma, uint32_t kd0, uint32_t kd1,
                                                            uint32_t kn0, uint32_t kn1, const float* __restrict__ keep_sum,
                                                            unsigned short* __restrict__ out_cm, long long ld_cm,
                                                            long long cm_mid, unsigned short* __restrict__ out_hi,
                                                            long long ld_hi, int row0) {
    extern __shared__ __attribute__((aligned(16))) float vlds[];     // x | drop view | noise view, each [64][S]
    __shared__ float keep[kMaxS];
    const int r = blockIdx.x;
    rows_views_cm_body(src, ld_src, src_mid, r, (size_t)(row_base + rows[r]), blockIdx.y * kViewCh, k, C, S, gamma, kd0, kd1, kn0,
                       kn1, *keep_sum, out_cm, ld_cm, cm_mid, out_hi, ld_hi, row0, vlds, keep);
}

__global__ __launch_bounds__(256) void rows_views_cm_grouped_kernel(const unsigned short* __restrict__ src, long long ld_src,
                                                                    long long src_mid, ViewGroups vg, int C, int S, float gamma,
                                                                    unsigned short* __restrict__ out_cm, long long ld_cm,
                                                                    long long cm_mid, unsigned short* __restrict__ out_hi,
                                                                    long long ld_hi) {
    extern __shared__ __attribute__((aligned(16))) float vlds[];
    __shared__ float keep[kMaxS];
    const int n_e = *vg.n_entries;
    for (int e = blockIdx.x; e < n_e; e += gridDim.x) {
        const int g = view_group_of(vg, e);
        const int e0 = vg.e0[g], k = vg.e0[g + 1] - e0;
        const uint32_t* ky = vg.keys + 4 * g;
        rows_views_cm_body(src, ld_src, src_mid, e - e0, (size_t)vg.src_row[e], blockIdx.y * kViewCh, k, C, S, gamma, ky[0], ky[1], ky[2],
                           ky[3], vg.keep_sum[g], out_cm, ld_cm, cm_mid, out_hi, ld_hi, 2 * e0, vlds, keep);
        __syncthreads();
    }
}

ODW_EXPORT int odw_rows_views_cm(const void* src_cm, int64_t ld_src, int64_t src_mid, const int* rows, int row_base, int k, int C,
                                 int S, float gamma, uint32_t kd0, uint32_t kd1, uint32_t kn0, uint32_t kn1, float* keep_sum,
                                 void* out_cm, int64_t ld_cm, int64_t cm_mid, void* out_hi, int64_t ld_hi, int out_row0,
                                 void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    ODW_REQUIRE(k >= 0 && C > 0 && C % kViewCh == 0 && S >= 4 && S <= kMaxS && row_base >= 0 && out_row0 >= 0,
                "rows_views_cm: bad dims (k=%d, C=%d a multiple of 64, S=%d)", k, C, S);
    if (k == 0) return ODW_OK;
    const long long cs = (long long)C * S;
    ODW_REQUIRE(src_cm && rows && keep_sum && out_cm && out_hi, "rows_views_cm: null pointer");
    ODW_REQUIRE(src_mid >= cs && ld_src >= src_mid + cs && cm_mid >= cs && ld_cm >= cm_mid + cs && ld_hi >= cs && ld_src % 8 == 0 &&
                src_mid % 8 == 0 && ld_cm % 8 == 0 && cm_mid % 8 == 0 && ld_hi % 4 == 0 && (long long)k * cs < (1ll << 32),
                "rows_views_cm: planes [hi | mid] of %lld elements must fit the rows (ld_src=%lld src_mid=%lld ld_cm=%lld cm_mid=%lld "
                "ld_hi=%lld)", cs, (long long)ld_src, (long long)src_mid, (long long)ld_cm, (long long)cm_mid, (long long)ld_hi);
    ODW_REQUIRE((((uintptr_t)src_cm) & 15) == 0 && (((uintptr_t)out_cm) & 15) == 0 && (((uintptr_t)out_hi) & 7) == 0,
                "rows_views_cm: alignment");
    rows_keep_sum_kernel<<<1, 256, 0, stream>>>(k * S, gamma, kd0, kd1, keep_sum);
    const size_t lds = (size_t)3 * kViewCh * S * sizeof(float);
    ODW_REQUIRE(lds + kMaxS * sizeof(float) <= (size_t)ODW_LDS_BYTES, "rows_views_cm: S=%d cells per ROI need %zu bytes of LDS "
                "(a 64-channel slice of the row and of both views)", S, lds);
    ODW_CHECK_HIP(odw_set_max_lds(reinterpret_cast<const void*>(rows_views_cm_kernel), (int)lds), "rows_views_cm attr");
    rows_views_cm_kernel<<<dim3(k, C / kViewCh), 256, lds, stream>>>(
        (const unsigned short*)src_cm, ld_src, src_mid, rows, row_base, k, C, S, gamma, kd0, kd1, kn0, kn1, keep_sum,
        (unsigned short*)out_cm, ld_cm, cm_mid, (unsigned short*)out_hi, ld_hi, out_row0);
    ODW_CHECK_LAUNCH("rows_views_cm_kernel");
    return ODW_OK;
}

ODW_EXPORT int odw_rows_drop_noise_bwd(const void* dX, int dx_is_f32, int ld, int dx_row0, const int* rows, int row_base,
                                       int k, int C, int S, float gamma, uint32_t kd0, uint32_t kd1, uint32_t kn0,
                                       uint32_t kn1, const float* keep_sum, float* dpooled, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    ODW_REQUIRE(k >= 0 && C > 0 && S >= 4 && S <= kMaxS && row_base >= 0 && dx_row0 >= 0, "rows_drop_noise_bwd: bad dims");
    if (k == 0) return ODW_OK;
    const long cs = (long)C * S;
    ODW_REQUIRE(dX && rows && keep_sum && dpooled, "rows_drop_noise_bwd: null pointer");
    ODW_REQUIRE(cs % 4 == 0 && ld >= cs && ld % 4 == 0, "rows_drop_noise_bwd: C*S=%ld, ld=%d", cs, ld);
    ODW_REQUIRE((((uintptr_t)dX) & 15) == 0 && (((uintptr_t)dpooled) & 15) == 0, "rows_drop_noise_bwd: alignment");
    if (dx_is_f32)
        rows_drop_noise_kernel<true, true><<<dim3(k, kRowSlices), 256, 0, stream>>>(nullptr, dX, rows, row_base, k, (int)cs, S, gamma, kd0, kd1,
                                                                  kn0, kn1, keep_sum, nullptr, ld, dx_row0, dpooled);
    else
        rows_drop_noise_kernel<true, false><<<dim3(k, kRowSlices), 256, 0, stream>>>(nullptr, dX, rows, row_base, k, (int)cs, S, gamma, kd0, kd1,
                                                                   kn0, kn1, keep_sum, nullptr, ld, dx_row0, dpooled);
    ODW_CHECK_LAUNCH("rows_drop_noise_kernel");
    return ODW_OK;
}

ODW_EXPORT int odw_rows_drop_noise_bwd_store(const void* dX, int dx_is_f32, int ld, int dx_row0, const int* rows, int row_base,
                                             int k, int C, int S, float gamma, uint32_t kd0, uint32_t kd1, uint32_t kn0,
                                             uint32_t kn1, const float* keep_sum, float* dpooled, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    ODW_REQUIRE(k >= 0 && C > 0 && S >= 4 && S <= kMaxS && row_base >= 0 && dx_row0 >= 0, "rows_drop_noise_bwd_store: bad dims");
    if (k == 0) return ODW_OK;
    const long cs = (long)C * S;
    ODW_REQUIRE(dX && rows && keep_sum && dpooled, "rows_drop_noise_bwd_store: null pointer");
    ODW_REQUIRE(cs % 4 == 0 && ld >= cs && ld % 4 == 0, "rows_drop_noise_bwd_store: C*S=%ld, ld=%d", cs, ld);
    ODW_REQUIRE((((uintptr_t)dX) & 15) == 0 && (((uintptr_t)dpooled) & 15) == 0, "rows_drop_noise_bwd_store: alignment");
    if (dx_is_f32)
        rows_drop_noise_kernel<true, true, false, false, true><<<dim3(k, kRowSlices), 256, 0, stream>>>(
            nullptr, dX, rows, row_base, k, (int)cs, S, gamma, kd0, kd1, kn0, kn1, keep_sum, nullptr, ld, dx_row0, dpooled);
    else
        rows_drop_noise_kernel<true, false, false, false, true><<<dim3(k, kRowSlices), 256, 0, stream>>>(
            nullptr, dX, rows, row_base, k, (int)cs, S, gamma, kd0, kd1, kn0, kn1, keep_sum, nullptr, ld, dx_row0, dpooled);
    ODW_CHECK_LAUNCH("rows_drop_noise_kernel");
    return ODW_OK;
}

ODW_EXPORT int odw_l2norm_rows(const float* x, int R, int D, float eps, float* y, float* norm, void* stream_) {
    ODW_REQUIRE(R >= 0 && D >= 1, "l2norm_rows: bad dims R=%d D=%d", R, D);
    if (R == 0) return ODW_OK;
    ODW_REQUIRE(x && y && norm, "l2norm_rows: null pointer");
    l2norm_rows_kernel<false><<<(R + 3) / 4, 256, 0, (hipStream_t)stream_>>>(x, nullptr, nullptr, R, D, eps, y, norm, nullptr);
    ODW_CHECK_LAUNCH("l2norm_rows_kernel");
    return ODW_OK;
}

ODW_EXPORT int odw_l2norm_rows_bwd(const float* g, const float* y, const float* norm, int R, int D, float eps,
                                   float* dx, void* stream_) {
    ODW_REQUIRE(R >= 0 && D >= 1, "l2norm_rows_bwd: bad dims R=%d D=%d", R, D);
    if (R == 0) return ODW_OK;
    ODW_REQUIRE(g && y && norm && dx, "l2norm_rows_bwd: null pointer");
    l2norm_rows_kernel<true><<<(R + 3) / 4, 256, 0, (hipStream_t)stream_>>>(g, y, norm, R, D, eps, dx, nullptr, nullptr);
    ODW_CHECK_LAUNCH("l2norm_rows_kernel");
    return ODW_OK;
}

// ---- device-resident forms (round 6; see the grouped kernels above) ------------------------------------------------------
ODW_EXPORT int odw_l2norm_rows_dyn(const float* x, int R_cap, int D, float eps, float* y, float* norm, const int* r_dev,
                                   void* stream_) {
    ODW_REQUIRE(R_cap >= 0 && D >= 1 && r_dev, "l2norm_rows_dyn: bad arguments R_cap=%d D=%d", R_cap, D);
    if (R_cap == 0) return ODW_OK;
    ODW_REQUIRE(x && y && norm, "l2norm_rows_dyn: null pointer");
    l2norm_rows_kernel<false><<<(R_cap + 3) / 4, 256, 0, (hipStream_t)stream_>>>(x, nullptr, nullptr, R_cap, D, eps, y, norm, r_dev);
    ODW_CHECK_LAUNCH("l2norm_rows_kernel");
    return ODW_OK;
}

ODW_EXPORT int odw_l2norm_rows_bwd_dyn(const float* g, const float* y, const float* norm, int R_cap, int D, float eps, float* dx,
                                       const int* r_dev, void* stream_) {
    ODW_REQUIRE(R_cap >= 0 && D >= 1 && r_dev, "l2norm_rows_bwd_dyn: bad arguments R_cap=%d D=%d", R_cap, D);
    if (R_cap == 0) return ODW_OK;
    ODW_REQUIRE(g && y && norm && dx, "l2norm_rows_bwd_dyn: null pointer");
    l2norm_rows_kernel<true><<<(R_cap + 3) / 4, 256, 0, (hipStream_t)stream_>>>(g, y, norm, R_cap, D, eps, dx, nullptr, r_dev);
    ODW_CHECK_LAUNCH("l2norm_rows_kernel");
    return ODW_OK;
}

static bool view_groups_ok(int G, int E_cap, const int* n_entries, const int* e0, const uint32_t* keys, const int* src_row,
                           const float* keep_sum) {
    return G >= 1 && E_cap >= 1 && n_entries && e0 && keys && keep_sum && src_row;
}

// The drop view and the noise view of EVERY sampled row of the step (loss.py:292-305) as the operand of the first head
// Linear, one launch: odw_rows_views_cm with the groups' sizes on the device.  e0 (G + 1 entry prefix), n_entries, src_row
// come from odw_loss_lists_a; keys (G x 4: kd0 kd1 kn0 kn1) from the host; keep_sum (G floats) is written here.
// E_cap = the most entries that can exist (the launch covers them; rows [2 *n_entries, 2 E_cap) of the outputs stay unwritten).
ODW_EXPORT int odw_rows_views_cm_grouped(const void* src_cm, int64_t ld_src, int64_t src_mid, int G, int E_cap,
                                         const int* n_entries, const int* e0, const uint32_t* keys, const int* src_row, int C,
                                         int S, float gamma, float* keep_sum, void* out_cm, int64_t ld_cm, int64_t cm_mid,
                                         void* out_hi, int64_t ld_hi, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    ODW_REQUIRE(C > 0 && C % kViewCh == 0 && S >= 4 && S <= kMaxS, "rows_views_cm_grouped: bad dims (C=%d a multiple of 64, S=%d)", C, S);
    ODW_REQUIRE(view_groups_ok(G, E_cap, n_entries, e0, keys, src_row, keep_sum) && src_cm && out_cm && out_hi,
                "rows_views_cm_grouped: bad arguments (G=%d, E_cap=%d)", G, E_cap);
    const long long cs = (long long)C * S;
    ODW_REQUIRE(src_mid >= cs && ld_src >= src_mid + cs && cm_mid >= cs && ld_cm >= cm_mid + cs && ld_hi >= cs && ld_src % 8 == 0 &&
                src_mid % 8 == 0 && ld_cm % 8 == 0 && cm_mid % 8 == 0 && ld_hi % 4 == 0 && (long long)E_cap * cs < (1ll << 32),
                "rows_views_cm_grouped: planes [hi | mid] of %lld elements must fit the rows", cs);
    ODW_REQUIRE((((uintptr_t)src_cm) & 15) == 0 && (((uintptr_t)out_cm) & 15) == 0 && (((uintptr_t)out_hi) & 7) == 0,
                "rows_views_cm_grouped: alignment");
    ViewGroups vg;
    vg.n_entries = n_entries; vg.e0 = e0; vg.keys = keys; vg.src_row = src_row; vg.keep_sum = keep_sum; vg.G = G;
    rows_keep_sum_grouped_kernel<<<G, 256, 0, stream>>>(vg, S, gamma, keep_sum);
    const size_t lds = (size_t)3 * kViewCh * S * sizeof(float);
    ODW_REQUIRE(lds + kMaxS * sizeof(float) <= (size_t)ODW_LDS_BYTES, "rows_views_cm_grouped: S=%d cells per ROI need %zu bytes of LDS", S, lds);
    ODW_CHECK_HIP(odw_set_max_lds(reinterpret_cast<const void*>(rows_views_cm_grouped_kernel), (int)lds), "rows_views_cm_grouped attr");
    rows_views_cm_grouped_kernel<<<dim3(E_cap < kViewGrid ? E_cap : kViewGrid, C / kViewCh), 256, lds, stream>>>(
        (const unsigned short*)src_cm, ld_src, src_mid, vg, C, S, gamma, (unsigned short*)out_cm, ld_cm, cm_mid,
        (unsigned short*)out_hi, ld_hi);
    ODW_CHECK_LAUNCH("rows_views_cm_grouped_kernel");
    return ODW_OK;
}

// Backward of the grouped views into the pooling node's side buffer (one fp32 row of C*S per entry): entry e's gradient =
// the fold of its drop-view row and its noise-view row of dX, STORED at row *dst_off + e of `extra` (dst_off null = 0).
ODW_EXPORT int odw_rows_views_bwd_store_grouped(const void* dX, int dx_is_f32, int ld, int G, int E_cap, const int* n_entries,
                                                const int* e0, const uint32_t* keys, const float* keep_sum, int C, int S,
                                                float gamma, const int* dst_off, float* extra, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    ODW_REQUIRE(C > 0 && S >= 4 && S <= kMaxS && G >= 1 && E_cap >= 1 && n_entries && e0 && keys && keep_sum && dX && extra,
                "rows_views_bwd_store_grouped: bad arguments");
    const long cs = (long)C * S;
    ODW_REQUIRE(cs % 4 == 0 && ld >= cs && ld % 4 == 0, "rows_views_bwd_store_grouped: C*S=%ld, ld=%d", cs, ld);
    ODW_REQUIRE((((uintptr_t)dX) & 15) == 0 && (((uintptr_t)extra) & 15) == 0, "rows_views_bwd_store_grouped: alignment");
    ViewGroups vg;
    vg.n_entries = n_entries; vg.e0 = e0; vg.keys = keys; vg.src_row = nullptr; vg.keep_sum = keep_sum; vg.G = G;
    if (dx_is_f32)
        rows_views_bwd_store_grouped_kernel<true><<<dim3(E_cap < kViewGrid ? E_cap : kViewGrid, kRowSlices), 256, 0, stream>>>(dX, ld, vg, (int)cs, S, gamma, dst_off, extra);
    else
        rows_views_bwd_store_grouped_kernel<false><<<dim3(E_cap < kViewGrid ? E_cap : kViewGrid, kRowSlices), 256, 0, stream>>>(dX, ld, vg, (int)cs, S, gamma, dst_off, extra);
    ODW_CHECK_LAUNCH("rows_views_bwd_store_grouped_kernel");
    return ODW_OK;
}
