// conv_aux.hip -- layout / pooling helpers around the implicit-GEMM convolution (gemm_bf16.hip):
// the VGG16-OICR backbone (modeling/backbone/vgg16.py:58-104) runs in NHWC bf16 on the device.
//   * weight repacking: torch (Cout,Cin,3,3) fp32 -> [Cout][tap*Cp+ci] (forward) and
//     [Cin][tap*Cout+co] (input gradient) bf16, rows zero padded to a multiple of 64
//   * transposed im2col for the weight gradient (reduction over pixels needs pixel-contiguous operands)
//   * 2x2 max pooling forward / backward (first maximum wins, like torch) with the ReLU mask folded in
//   * NCHW fp32 <-> NHWC bf16 at the two ends of the backbone
#include "odw_common.h"
#include "odw_planes.h"

namespace {

__device__ __forceinline__ unsigned short f2bf(float f) {
    unsigned int u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float((unsigned int)h << 16); }

// wk[co][t*Cp + ci] (ld = ldk) and wd[ci][t*Co + co] (ld = ldd); either may be null.
__global__ void weight_prep_kernel(const float* __restrict__ w, int Co, int Ci, int Cp,
                                   unsigned short* __restrict__ wk, int ldk, unsigned short* __restrict__ wd, int ldd) {
    const int total_k = Co * ldk;
    const int total_d = wd ? Ci * ldd : 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total_k + total_d; i += gridDim.x * blockDim.x) {
        if (i < total_k) {
            if (!wk) continue;
            const int co = i / ldk, k = i - co * ldk;
            const int t = k / Cp, ci = k - t * Cp;
            wk[i] = (t < 9 && ci < Ci) ? f2bf(w[((size_t)co * Ci + ci) * 9 + t]) : (unsigned short)0;
        } else {
            const int j = i - total_k;
            const int ci = j / ldd, k = j - ci * ldd;
            const int t = k / Co, co = k - t * Co;
            wd[j] = (t < 9) ? f2bf(w[((size_t)co * Ci + ci) * 9 + t]) : (unsigned short)0;
        }
    }
}

// every layer of a body in ONE launch (blockIdx.y = layer): the per-layer launches were 9 x ~14 us of a few thousand
// elements each at the head of every forward
constexpr int kMaxPrepLayers = 32;
// T > 0: wk holds PLANES of the fp32 weight (csrc/split.hip): per tap T blocks of Cp channels, block tt = plane pat[tt]
struct PrepLayer { const float* w; unsigned short* wk; unsigned short* wd; int Co, Ci, Cp, ldk, ldd, T; int pat[4]; };
struct PrepBatch { PrepLayer l[kMaxPrepLayers]; };

// blockIdx.x walks the layer's work units: first Co units "wk row co" (the row's Ci x 9 floats are one contiguous,
// coalesced read; through LDS they leave as 9 runs of Cp bf16), then Ci x ceil(Co / 64) units "wd row ci, 64 output
// channels" (64 reads of 36 contiguous bytes -> 9 runs of 64 bf16).  Zero padding of both layouts included.
// Tiled form for the layers that matter (Ci and Co multiples of 32, no padding: every trainable layer of the VGG /
// ResNet bodies): a workgroup reads a (32 co) x (32 ci) x 9 tile of w ONCE -- 32 contiguous runs of 1152 bytes -- into
// LDS and writes both copies from it as 16-byte vectors: wk in 64-byte runs (32 ci of one tap and plane block), wd
// in 64-byte runs (32 co of one tap).  The row form below read every weight twice, the wd half of it as 36-byte
// gathers, and stored 2 bytes per lane: 172 us for the nine trainable VGG layers, on the critical path at the head of
// every step.  (64-ci tiles, two workgroups per CU: 150 us; 32-ci tiles, four per CU: 121 us; the grid cut from
// 2048 x layers -- most of them exiting at once -- to one workgroup per tile: 76 us.)
constexpr int kPrepTileCo = 32, kPrepTileCi = 32, kPrepPitch = kPrepTileCi * 9 + 1;      // 37 KB of LDS: four workgroups per CU
constexpr int kPrepC8 = kPrepTileCi / 8;
constexpr int kPrepLds = kPrepTileCo * kPrepPitch * 4;

__device__ __forceinline__ bool prep_tiled(const PrepLayer& L) {
    return L.Ci % kPrepTileCi == 0 && L.Co % kPrepTileCo == 0 && L.Cp == L.Ci &&
           (!L.wk || L.ldk == 9 * L.Cp * (L.T > 0 ? L.T : (L.T == -2 ? 2 : 1))) &&
           (!L.wd || L.ldd == 9 * L.Co);
}

__device__ __forceinline__ void prep_tile(const PrepLayer& L, int tile, float* sm) {
    const int tiles_ci = L.Ci / kPrepTileCi;
    const int co0 = (tile / tiles_ci) * kPrepTileCo, ci0 = (tile % tiles_ci) * kPrepTileCi;
    const int tid = threadIdx.x;
    __syncthreads();                                     // the previous tile's readers are done
    for (int q = tid; q < kPrepTileCo * (kPrepTileCi * 9 / 4); q += 256) {
        const int co = q / (kPrepTileCi * 9 / 4), k4 = q - co * (kPrepTileCi * 9 / 4);
        const float4 v = *reinterpret_cast<const float4*>(L.w + ((size_t)(co0 + co) * L.Ci + ci0) * 9 + 4 * k4);
        float* d = sm + co * kPrepPitch + 4 * k4;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    __syncthreads();
    const int T = L.T > 0 ? L.T : 1;
    if (L.wk) {
        // item = (co, tap, 8 consecutive ci): lanes walk the 8 chunks of a 128-byte run first
        for (int it = tid; it < kPrepTileCo * 9 * (kPrepTileCi / 8); it += 256) {
            const int c8 = it % kPrepC8, t = (it / kPrepC8) % 9, co = it / (9 * kPrepC8);
            const float* src = sm + co * kPrepPitch + c8 * 72 + t;
            unsigned pl[3][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                unsigned short h[2], m[2], l[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const float x = src[(2 * j + u) * 9];
                    h[u] = f2bf(x);
                    const float r1 = x - bf2f(h[u]);
                    m[u] = f2bf(r1);
                    l[u] = f2bf(r1 - bf2f(m[u]));
                }
                pl[0][j] = h[0] | ((unsigned)h[1] << 16);
                pl[1][j] = m[0] | ((unsigned)m[1] << 16);
                pl[2][j] = l[0] | ((unsigned)l[1] << 16);
            }
            if (L.T == -2) {
                // conv3x3_halo2_kernel's layout: per tap, per block of 32 channels, [hi 32 | mid 32] (T = -2)
                const int ci = ci0 + c8 * 8;
                unsigned short* d = L.wk + (size_t)(co0 + co) * L.ldk + (size_t)t * 2 * L.Cp + (ci >> 5) * 64 + (ci & 31);
                *reinterpret_cast<uint4*>(d) = make_uint4(pl[0][0], pl[0][1], pl[0][2], pl[0][3]);
                *reinterpret_cast<uint4*>(d + 32) = make_uint4(pl[1][0], pl[1][1], pl[1][2], pl[1][3]);
                continue;
            }
            unsigned short* row = L.wk + (size_t)(co0 + co) * L.ldk + ci0 + c8 * 8;
            for (int tt = 0; tt < T; ++tt) {
                const int p = L.T > 0 ? (L.pat[tt] & 3) : 0;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (p < 3) v = make_uint4(pl[p][0], pl[p][1], pl[p][2], pl[p][3]);
                *reinterpret_cast<uint4*>(row + (size_t)(t * T + tt) * L.Cp) = v;
            }
        }
    }
    if (L.wd) {
        // item = (ci, tap, 8 consecutive co): lanes walk the 4 chunks of a 64-byte run first
        for (int it = tid; it < kPrepTileCi * 9 * (kPrepTileCo / 8); it += 256) {
            const int c8 = it & 3, t = (it >> 2) % 9, ci = it / 36;
            const float* src = sm + (c8 * 8) * kPrepPitch + ci * 9 + t;
            unsigned v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                v[j] = f2bf(src[(2 * j) * kPrepPitch]) | ((unsigned)f2bf(src[(2 * j + 1) * kPrepPitch]) << 16);
            *reinterpret_cast<uint4*>(L.wd + (size_t)(ci0 + ci) * L.ldd + (size_t)t * L.Co + co0 + c8 * 8) = make_uint4(v[0], v[1], v[2], v[3]);
        }
    }
}

__global__ __launch_bounds__(256) void weight_prep_batch_kernel(PrepBatch b) {
    extern __shared__ __attribute__((aligned(16))) float sm[];        // kPrepLds bytes
    const PrepLayer L = b.l[blockIdx.y];
    if (prep_tiled(L)) {
        const int ntiles = (L.Co / kPrepTileCo) * (L.Ci / kPrepTileCi);
        for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) prep_tile(L, tile, sm);
        return;
    }
    const int nk = L.wk ? L.Co : 0;
    const int cochunks = (L.Co + 63) / 64;
    const int nd = L.wd ? L.Ci * cochunks : 0;
    for (int u = blockIdx.x; u < nk + nd; u += gridDim.x) {
        if (u < nk) {
            const int co = u;
            unsigned short* row = L.wk + (size_t)co * L.ldk;
            for (int c0 = 0; c0 < L.Cp; c0 += 512) {             // 512 input channels per pass
                const int nc = L.Ci - c0 < 512 ? (L.Ci - c0 > 0 ? L.Ci - c0 : 0) : 512;
                const float* src = L.w + ((size_t)co * L.Ci + c0) * 9;
                for (int k = threadIdx.x; k < nc * 9; k += 256) sm[k] = src[k];
                __syncthreads();
                const int np = L.Cp - c0 < 512 ? L.Cp - c0 : 512;
                if (L.T == -2) {
                    for (int k = threadIdx.x; k < 9 * np; k += 256) {
                        const int t = k / np, ci = k - t * np;
                        const float x = ci < nc ? sm[ci * 9 + t] : 0.0f;
                        const unsigned short hi = f2bf(x);
                        const int cc = c0 + ci;
                        unsigned short* d = row + (size_t)t * 2 * L.Cp + (cc >> 5) * 64 + (cc & 31);
                        d[0] = hi;
                        d[32] = f2bf(x - bf2f(hi));
                    }
                } else if (L.T > 0) {
                    for (int k = threadIdx.x; k < 9 * np; k += 256) {
                        const int t = k / np, ci = k - t * np;
                        const float x = ci < nc ? sm[ci * 9 + t] : 0.0f;
                        unsigned short pl[4];               // hi, mid, lo, zeros (both subtractions exact in fp32)
                        pl[0] = f2bf(x);
                        const float r1 = x - bf2f(pl[0]);
                        pl[1] = f2bf(r1);
                        pl[2] = f2bf(r1 - bf2f(pl[1]));
                        pl[3] = 0;
                        for (int tt = 0; tt < L.T; ++tt) row[(t * L.T + tt) * L.Cp + c0 + ci] = pl[L.pat[tt] & 3];
                    }
                } else
                for (int k = threadIdx.x; k < 9 * np; k += 256) {
                    const int t = k / np, ci = k - t * np;
                    row[t * L.Cp + c0 + ci] = ci < nc ? f2bf(sm[ci * 9 + t]) : (unsigned short)0;
                }
                __syncthreads();
            }
            for (int k = 9 * L.Cp * (L.T > 0 ? L.T : (L.T == -2 ? 2 : 1)) + threadIdx.x; k < L.ldk; k += 256) row[k] = 0;
        } else {
            const int v = u - nk, ci = v / cochunks, co0 = (v - ci * cochunks) * 64;
            const int nco = L.Co - co0 < 64 ? L.Co - co0 : 64;
            for (int k = threadIdx.x; k < nco * 9; k += 256) {
                const int c = k / 9, t = k - c * 9;
                sm[k] = L.w[((size_t)(co0 + c) * L.Ci + ci) * 9 + t];
            }
            __syncthreads();
            unsigned short* row = L.wd + (size_t)ci * L.ldd;
            for (int k = threadIdx.x; k < 9 * nco; k += 256) {
                const int t = k / nco, c = k - t * nco;
                row[t * L.Co + co0 + c] = f2bf(sm[c * 9 + t]);
            }
            if (co0 == 0)
                for (int k = 9 * L.Co + threadIdx.x; k < L.ldd; k += 256) row[k] = 0;
            __syncthreads();
        }
    }
}

// out[n] += sum_m X[m][n] (bf16 rows, fp32 sums): the bias gradient of a convolution whose weight gradient reads dZ as
// it is (odw_conv_wgrad_tn) -- linear_bwd_prep would write two copies of dZ only to produce these sums.
__global__ __launch_bounds__(256) void colsum_bf16_kernel(const unsigned short* __restrict__ X, int ld, int M, int N,
                                                          float* __restrict__ out) {
    __shared__ float sm[32][65];
    const int n0 = blockIdx.x * 64, m0 = blockIdx.y * 256;
    const int ch = threadIdx.x & 7, r = threadIdx.x >> 3;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (n0 + ch * 8 < N) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int m = m0 + r + 32 * k;
            if (m < M) {
                const uint4 v = *reinterpret_cast<const uint4*>(X + (size_t)m * ld + n0 + ch * 8);
                const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int q = 0; q < 4; ++q) { acc[2 * q] += __uint_as_float(w[q] << 16); acc[2 * q + 1] += __uint_as_float(w[q] & 0xffff0000u); }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) sm[r][ch * 8 + q] = acc[q];
    __syncthreads();
    if (threadIdx.x < 64 && n0 + (int)threadIdx.x < N) {
        float t = 0.0f;
#pragma unroll 8
        for (int k = 0; k < 32; ++k) t += sm[k][threadIdx.x];
        atomicAdd(out + n0 + threadIdx.x, t);
    }
}

// The same sums with ONE summation order (the float atomicAdd above makes the bias gradient differ from run to run): the
// rows are cut into `chunks` contiguous pieces, a workgroup adds its piece in a fixed order (8 rows in flight per
// thread group, 32 thread groups combined in order through LDS) and parks 64 column sums; a second small launch adds the
// parked sums of each column in chunk order.  (A single launch with a ticket counter -- the last block to arrive doing
// the final sum -- was tried: its device-scope __threadfence() is an L2 write-back on this multi-XCD part, 43-54 us
// per call against ~10 for the atomic form.)
__global__ __launch_bounds__(256) void colsum_bf16_part_kernel(const unsigned short* __restrict__ X, int ld, int M, int N,
                                                               int rows_per_chunk, float* __restrict__ part) {
    __shared__ float sm[32][65];
    const int n0 = blockIdx.x * 64;
    const int m0 = blockIdx.y * rows_per_chunk, m1 = min(M, m0 + rows_per_chunk);
    const int ch = threadIdx.x & 7, r = threadIdx.x >> 3;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (n0 + ch * 8 < N) {
        const unsigned short* col = X + n0 + ch * 8;
        for (int m = m0 + r; m < m1; m += 32 * 4) {
            uint4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = m + 32 * u < m1 ? *reinterpret_cast<const uint4*>(col + (size_t)(m + 32 * u) * ld) : make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const unsigned w[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
                for (int q = 0; q < 4; ++q) { acc[2 * q] += __uint_as_float(w[q] << 16); acc[2 * q + 1] += __uint_as_float(w[q] & 0xffff0000u); }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) sm[r][ch * 8 + q] = acc[q];
    __syncthreads();
    if (threadIdx.x < 64 && n0 + (int)threadIdx.x < N) {
        float t = 0.0f;
#pragma unroll 8
        for (int k = 0; k < 32; ++k) t += sm[k][threadIdx.x];
        part[(size_t)blockIdx.y * N + n0 + threadIdx.x] = t;
    }
}

__global__ __launch_bounds__(256) void colsum_finish_kernel(const float* __restrict__ part, int chunks, int N, float* __restrict__ out) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float t = 0.0f;
    for (int c0 = 0; c0 < chunks; c0 += 16) {        // 16 loads in flight, added in chunk order
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = c0 + u < chunks ? part[(size_t)(c0 + u) * N + n] : 0.0f;
#pragma unroll
        for (int u = 0; u < 16; ++u) t += v[u];
    }
    out[n] += t;
}

// dw[co][ci][t] = dwk[co][t*Cp + ci]
__global__ void wgrad_unpack_kernel(const float* __restrict__ dwk, int ld, int Co, int Ci, int Cp,
                                    float* __restrict__ dw) {
    const int total = Co * Ci * 9;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int t = i % 9, ci = (i / 9) % Ci, co = i / (9 * Ci);
        dw[i] = dwk[(size_t)co * ld + t * Cp + ci];
    }
}

// out[(t*C + ci)][m] = X[m + shift(t)][ci] (0 in the padding), m < ldm zero padded.
// LDS-free: a thread owns an 8(pixel) x 8(channel) block -- eight 16-byte loads (one per pixel), an in-register 8x8
// transposition of the 16-bit elements (32 v_perm_b32), eight 16-byte stores (one per channel).  A wave covers
// 16 pixel groups x 4 channel chunks: 64-byte read segments (L2 hits: every tap re-reads X) and 256-byte write runs
// (the writes are 9/10 of the traffic).  History: 32x32 LDS tiles with 2-byte accesses; 64x64 LDS tiles with 16-byte
// global accesses but 2-byte, 8-way bank-conflicted LDS reads: 88 us per launch inside the step (1 TB/s); this one
// 45-50 us.  Inside the step the saving is mostly absorbed: the old kernel was LDS-bound and overlapped the
// HBM-bound side-stream optimiser for free, now the weight-gradient GEMMs that follow share HBM with it instead
// (GPU-busy per step 9.11 -> 8.96 ms on the same box class).
__global__ __launch_bounds__(256) void im2col_t_kernel(const unsigned short* __restrict__ X, int M, int H, int W,
                                                       int C, int dil, unsigned short* __restrict__ out, int ldm, int cols) {
    const int t = blockIdx.z;
    const int ty9 = t / 3, tx9 = t - 3 * ty9;
    const int dh = (ty9 - 1) * dil, dw = (tx9 - 1) * dil;
    const int hw = H * W;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pg = lane & 15, ch = lane >> 4;                 // pixel group (8 pixels), channel chunk (8 channels)
    const int m0 = (blockIdx.y * 4 + wave) * 128 + pg * 8;    // block = 4 waves x 128 pixels
    const int c0 = blockIdx.x * 32 + ch * 8;
    if (c0 >= C || m0 >= cols) return;
    unsigned int r[8][4];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int m = m0 + q;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (m < M) {
            const int p = m % hw;
            const int y = p / W + dh, x = p % W + dw;
            if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W)
                v = *reinterpret_cast<const uint4*>(X + (size_t)(m + dh * W + dw) * C + c0);
        }
        r[q][0] = v.x; r[q][1] = v.y; r[q][2] = v.z; r[q][3] = v.w;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        // channel j of pixels (2p, 2p+1): low halves (selector 0x05040100) or high halves (0x07060302) of dword j/2
        const unsigned int sel = (j & 1) ? 0x07060302u : 0x05040100u;
        uint4 o;
        o.x = __builtin_amdgcn_perm(r[1][j >> 1], r[0][j >> 1], sel);
        o.y = __builtin_amdgcn_perm(r[3][j >> 1], r[2][j >> 1], sel);
        o.z = __builtin_amdgcn_perm(r[5][j >> 1], r[4][j >> 1], sel);
        o.w = __builtin_amdgcn_perm(r[7][j >> 1], r[6][j >> 1], sel);
        *reinterpret_cast<uint4*>(out + ((size_t)t * C + c0 + j) * ldm + m0) = o;
    }
}

// NHWC bf16 2x2/2 max pool, 8 channels per thread
__global__ void maxpool_fwd_kernel(const uint4* __restrict__ X, int B, int H, int W, int C8, uint4* __restrict__ Y) {
    const int Ho = H / 2, Wo = W / 2;
    const size_t total = (size_t)B * Ho * Wo * C8;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C8);
        size_t q = i / C8;
        const int xo = (int)(q % Wo); q /= Wo;
        const int yo = (int)(q % Ho);
        const int b = (int)(q / Ho);
        const size_t base = (((size_t)b * H + 2 * yo) * W + 2 * xo) * C8 + c;
        const uint4 v[4] = {X[base], X[base + C8], X[base + (size_t)W * C8], X[base + (size_t)W * C8 + C8]};
        uint4 o;
        unsigned int* op = reinterpret_cast<unsigned int*>(&o);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            unsigned int r = 0;
#pragma unroll
            for (int hlf = 0; hlf < 2; ++hlf) {
                float best = -__builtin_inff();
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const unsigned int word = reinterpret_cast<const unsigned int*>(&v[k])[j];
                    const float f = bf2f((unsigned short)(hlf ? word >> 16 : word & 0xffff));
                    best = f > best ? f : best;
                }
                r |= (unsigned int)f2bf(best) << (16 * hlf);
            }
            op[j] = r;
        }
        Y[i] = o;
    }
}

// dX (pre-pool, NHWC bf16) = unpool(dY) routed to the FIRST maximum of each window, times [X > 0]
// (X = the post-ReLU activation that was pooled: modeling/backbone/vgg16.py:61-80 conv,ReLU,...,MaxPool)
__global__ void maxpool_bwd_kernel(const unsigned short* __restrict__ X, const unsigned short* __restrict__ dY, int B,
                                   int H, int W, int C, unsigned short* __restrict__ dX) {
    const int Ho = H / 2, Wo = W / 2;
    const size_t total = (size_t)B * Ho * Wo * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        size_t q = i / C;
        const int xo = (int)(q % Wo); q /= Wo;
        const int yo = (int)(q % Ho);
        const int b = (int)(q / Ho);
        const size_t base = (((size_t)b * H + 2 * yo) * W + 2 * xo) * C + c;
        const size_t off[4] = {0, (size_t)C, (size_t)W * C, (size_t)W * C + C};
        float best = -__builtin_inff();
        int bi = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float f = bf2f(X[base + off[k]]);
            if (f > best) { best = f; bi = k; }
        }
        const unsigned short g = best > 0.0f ? dY[i] : (unsigned short)0;
#pragma unroll
        for (int k = 0; k < 4; ++k) dX[base + off[k]] = k == bi ? g : (unsigned short)0;
    }
}

// out[b][c][p] fp32 = in[b][p][c] bf16   (p = h*W+w), 32x32 tiles
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const unsigned short* __restrict__ in, int HW, int C,
                                                           float* __restrict__ out) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int c0 = blockIdx.x * 32, p0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const unsigned short* src = in + (size_t)b * HW * C;
    float* dst = out + (size_t)b * HW * C;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int p = p0 + ty + 8 * k, c = c0 + tx;
        tile[ty + 8 * k][tx] = (p < HW && c < C) ? bf2f(src[(size_t)p * C + c]) : 0.0f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c0 + ty + 8 * k, p = p0 + tx;
        if (c < C && p < HW) dst[(size_t)c * HW + p] = tile[tx][ty + 8 * k];
    }
}

// out[b][p][c] bf16 (c < Cp, zero padded) = in[b][c][p] fp32
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ in, int HW, int C, int Cp,
                                                           unsigned short* __restrict__ out) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float* src = in + (size_t)b * HW * C;
    unsigned short* dst = out + (size_t)b * HW * Cp;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c0 + ty + 8 * k, p = p0 + tx;
        tile[ty + 8 * k][tx] = (c < C && p < HW) ? src[(size_t)c * HW + p] : 0.0f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int p = p0 + ty + 8 * k, c = c0 + tx;
        if (p < HW && c < Cp) dst[(size_t)p * Cp + c] = f2bf(tile[tx][ty + 8 * k]);
    }
}

// ---- the same helpers on fp32 NHWC activations (the bf16x3 precision mode keeps activations in fp32 between
// kernels and splits them into bf16 planes right before each product: csrc/split.hip) -----------------------------
__global__ void maxpool_f32_fwd_kernel(const float* __restrict__ X, int B, int H, int W, int C, float* __restrict__ Y) {
    const int Ho = H / 2, Wo = W / 2;
    const size_t total = (size_t)B * Ho * Wo * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        size_t q = i / C;
        const int xo = (int)(q % Wo); q /= Wo;
        const int yo = (int)(q % Ho);
        const int b = (int)(q / Ho);
        const size_t base = (((size_t)b * H + 2 * yo) * W + 2 * xo) * C + c;
        const float v0 = X[base], v1 = X[base + C], v2 = X[base + (size_t)W * C], v3 = X[base + (size_t)W * C + C];
        float best = v0;
        best = v1 > best ? v1 : best;
        best = v2 > best ? v2 : best;
        best = v3 > best ? v3 : best;
        Y[i] = best;
    }
}

// 2x2 max pooling of an fp32 NHWC activation written straight as the next convolution's operand: the two bf16 planes
// [hi (C) | mid (C)] per pooled pixel (row stride ldy bf16 elements, mid plane ldy / 2 further) -- what
// maxpool_f32_fwd_kernel + split_rows_kernel produced in two passes (conv3x3_halo2_kernel, "bf16x2f").  4 channels per thread.
__global__ __launch_bounds__(256) void maxpool_f32_planes2_kernel(const float* __restrict__ X, int B, int H, int W, int C,
                                                                  unsigned short* __restrict__ Y, int ldy) {
    const int Ho = H / 2, Wo = W / 2, C4 = C / 4;
    const size_t total = (size_t)B * Ho * Wo * C4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4) * 4;
        size_t q = i / C4;
        const size_t opix = q;
        const int xo = (int)(q % Wo); q /= Wo;
        const int yo = (int)(q % Ho);
        const int b = (int)(q / Ho);
        const size_t base = (((size_t)b * H + 2 * yo) * W + 2 * xo) * C + c;
        const float4 v0 = *reinterpret_cast<const float4*>(X + base), v1 = *reinterpret_cast<const float4*>(X + base + C);
        const float4 v2 = *reinterpret_cast<const float4*>(X + base + (size_t)W * C);
        const float4 v3 = *reinterpret_cast<const float4*>(X + base + (size_t)W * C + C);
        float m[4] = {v0.x, v0.y, v0.z, v0.w};
        const float o[3][4] = {{v1.x, v1.y, v1.z, v1.w}, {v2.x, v2.y, v2.z, v2.w}, {v3.x, v3.y, v3.z, v3.w}};
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) m[e] = o[k][e] > m[e] ? o[k][e] : m[e];      // the same comparisons as maxpool_f32_fwd_kernel
        unsigned h0, m0, h1, m1, lo_;
        odwpl::split2(m[0], m[1], false, h0, m0, lo_);
        odwpl::split2(m[2], m[3], false, h1, m1, lo_);
        unsigned short* row = Y + opix * ldy + c;
        *reinterpret_cast<uint2*>(row) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(row + (ldy >> 1)) = make_uint2(m0, m1);
    }
}

__global__ void maxpool_f32_bwd_kernel(const float* __restrict__ X, const float* __restrict__ dY, int B, int H, int W,
                                       int C, float* __restrict__ dX) {
    const int Ho = H / 2, Wo = W / 2;
    const size_t total = (size_t)B * Ho * Wo * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        size_t q = i / C;
        const int xo = (int)(q % Wo); q /= Wo;
        const int yo = (int)(q % Ho);
        const int b = (int)(q / Ho);
        const size_t base = (((size_t)b * H + 2 * yo) * W + 2 * xo) * C + c;
        const size_t off[4] = {0, (size_t)C, (size_t)W * C, (size_t)W * C + C};
        float best = -__builtin_inff();
        int bi = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float f = X[base + off[k]];
            if (f > best) { best = f; bi = k; }
        }
        const float g = best > 0.0f ? dY[i] : 0.0f;        // X = the post-ReLU activation that was pooled
#pragma unroll
        for (int k = 0; k < 4; ++k) dX[base + off[k]] = k == bi ? g : 0.0f;
    }
}

// precision mode "bf16x2f": the pooled activation X was kept in fp32 by the split-precision forward (the routing must
// follow ITS first maximum: rounded to bf16, distinct values tie), the gradients travel as bf16
__global__ void maxpool_f32x_bf16_bwd_kernel(const float* __restrict__ X, const unsigned short* __restrict__ dY, int B, int H,
                                             int W, int C, unsigned short* __restrict__ dX) {
    const int Ho = H / 2, Wo = W / 2;
    const size_t total = (size_t)B * Ho * Wo * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        size_t q = i / C;
        const int xo = (int)(q % Wo); q /= Wo;
        const int yo = (int)(q % Ho);
        const int b = (int)(q / Ho);
        const size_t base = (((size_t)b * H + 2 * yo) * W + 2 * xo) * C + c;
        const size_t off[4] = {0, (size_t)C, (size_t)W * C, (size_t)W * C + C};
        float best = -__builtin_inff();
        int bi = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float f = X[base + off[k]];
            if (f > best) { best = f; bi = k; }
        }
        const unsigned short g = best > 0.0f ? dY[i] : (unsigned short)0;
#pragma unroll
        for (int k = 0; k < 4; ++k) dX[base + off[k]] = k == bi ? g : (unsigned short)0;
    }
}

// out[b][p][c] (c < Cp, zero padded) = in[b][c][p], both fp32; TO_NCHW: the inverse (Cp = C)
template <bool TO_NCHW>
__global__ __launch_bounds__(256) void layout_f32_kernel(const float* __restrict__ in, int HW, int C, int Cp,
                                                         float* __restrict__ out) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    if (TO_NCHW) {
        const float* src = in + (size_t)b * HW * Cp;
        float* dst = out + (size_t)b * HW * C;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int p = p0 + ty + 8 * k, c = c0 + tx;
            tile[ty + 8 * k][tx] = (p < HW && c < C) ? src[(size_t)p * Cp + c] : 0.0f;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = c0 + ty + 8 * k, p = p0 + tx;
            if (c < C && p < HW) dst[(size_t)c * HW + p] = tile[tx][ty + 8 * k];
        }
    } else {
        const float* src = in + (size_t)b * HW * C;
        float* dst = out + (size_t)b * HW * Cp;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = c0 + ty + 8 * k, p = p0 + tx;
            tile[ty + 8 * k][tx] = (c < C && p < HW) ? src[(size_t)c * HW + p] : 0.0f;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int p = p0 + ty + 8 * k, c = c0 + tx;
            if (p < HW && c < Cp) dst[(size_t)p * Cp + c] = tile[tx][ty + 8 * k];
        }
    }
}

int blocks_for(size_t n) { size_t g = (n + 255) / 256; return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g)); }

}  // namespace

ODW_EXPORT int odw_conv_weight_prep(const float* w, int Co, int Ci, int Cp, void* wk, int ldk, void* wd, int ldd,
                                    void* stream_) {
    ODW_REQUIRE(Co > 0 && Ci > 0 && Cp >= Ci, "conv_weight_prep: bad dims");
    ODW_REQUIRE(w && (wk || wd), "conv_weight_prep: null pointer");
    ODW_REQUIRE((!wk || ldk >= 9 * Cp) && (!wd || ldd >= 9 * Co), "conv_weight_prep: leading dimensions too small");
    size_t n = (size_t)Co * ldk + (wd ? (size_t)Ci * ldd : 0);
    weight_prep_kernel<<<blocks_for(n), 256, 0, (hipStream_t)stream_>>>(w, Co, Ci, Cp, (unsigned short*)wk, ldk,
                                                                        (unsigned short*)wd, ldd);
    ODW_CHECK_LAUNCH("weight_prep_kernel");
    return ODW_OK;
}

ODW_EXPORT int odw_conv_weight_prep_planes_batch(int n, const void* const* w, const int* Co, const int* Ci, const int* Cp,
                                                 void* const* wk, const int* ldk, void* const* wd, const int* ldd,
                                                 const int* T, const int* patterns, void* stream_);

// n layers at once; the eight arrays are HOST arrays of length n (pointers: device memory; wk[i] / wd[i] may be null)
ODW_EXPORT int odw_conv_weight_prep_batch(int n, const void* const* w, const int* Co, const int* Ci, const int* Cp,
                                          void* const* wk, const int* ldk, void* const* wd, const int* ldd, void* stream_) {
    return odw_conv_weight_prep_planes_batch(n, w, Co, Ci, Cp, wk, ldk, wd, ldd, nullptr, nullptr, stream_);
}

// The same with wk written as bf16 PLANES of the fp32 weights (split-precision forward operand, csrc/split.hip):
// T[i] (0 = plain bf16, else 1..4) blocks of Cp[i] channels per tap, block tt = plane patterns[4 i + tt] (0 hi, 1 mid,
// 2 lo, 3 zeros); ldk[i] >= 9 * T[i] * Cp[i].  wd (the input-gradient copy) stays single-plane bf16.  T == NULL: plain.
// T[i] == -2: hi and mid interleaved per block of 32 channels -- row = per tap, per block, [hi 32 | mid 32] (ldk >= 18 Cp).
ODW_EXPORT int odw_conv_weight_prep_planes_batch(int n, const void* const* w, const int* Co, const int* Ci, const int* Cp,
                                                 void* const* wk, const int* ldk, void* const* wd, const int* ldd,
                                                 const int* T, const int* patterns, void* stream_) {
    ODW_REQUIRE(n >= 0 && n <= kMaxPrepLayers, "conv_weight_prep_batch: %d layers (at most %d per call)", n, kMaxPrepLayers);
    if (n == 0) return ODW_OK;
    ODW_REQUIRE(w && Co && Ci && Cp && wk && ldk && wd && ldd, "conv_weight_prep_batch: null array");
    PrepBatch b;
    size_t most = 0;
    for (int i = 0; i < n; ++i) {
        ODW_REQUIRE(Co[i] > 0 && Ci[i] > 0 && Cp[i] >= Ci[i] && w[i] && (wk[i] || wd[i]), "conv_weight_prep_batch: layer %d", i);
        const int Ti = T ? T[i] : 0;
        // T = -2: the two planes interleaved per block of 32 channels, [hi 32 | mid 32] (conv3x3_halo2_kernel's operand)
        ODW_REQUIRE((Ti >= 0 && Ti <= 4 && (Ti == 0 || patterns)) || (Ti == -2 && Cp[i] % 32 == 0),
                    "conv_weight_prep_batch: layer %d: T = %d (0..4, or -2 with Cp a multiple of 32)", i, Ti);
        const int planes_i = Ti > 0 ? Ti : (Ti == -2 ? 2 : 1);
        ODW_REQUIRE((!wk[i] || ldk[i] >= 9 * Cp[i] * planes_i) && (!wd[i] || ldd[i] >= 9 * Co[i]),
                    "conv_weight_prep_batch: leading dimensions of layer %d too small", i);
        b.l[i].w = (const float*)w[i]; b.l[i].wk = (unsigned short*)wk[i]; b.l[i].wd = (unsigned short*)wd[i];
        b.l[i].Co = Co[i]; b.l[i].Ci = Ci[i]; b.l[i].Cp = Cp[i]; b.l[i].ldk = ldk[i]; b.l[i].ldd = ldd[i];
        b.l[i].T = Ti;
        for (int tt = 0; tt < 4; ++tt) b.l[i].pat[tt] = (Ti > 0 && tt < Ti) ? patterns[4 * i + tt] : 3;
        size_t e = (wk[i] ? (size_t)Co[i] : 0) + (wd[i] ? (size_t)Ci[i] * ((Co[i] + 63) / 64) : 0);
        const bool tiled = Ci[i] % kPrepTileCi == 0 && Co[i] % kPrepTileCo == 0 && Cp[i] == Ci[i] &&
                           (!wk[i] || ldk[i] == 9 * Cp[i] * planes_i) && (!wd[i] || ldd[i] == 9 * Co[i]);      // = prep_tiled()
        if (tiled) e = (size_t)(Co[i] / kPrepTileCo) * (Ci[i] / kPrepTileCi);      // one workgroup per tile, none idle
        most = e > most ? e : most;
    }
    const int gx = (int)(most < 2048 ? most : 2048);
    const hipError_t attr = odw_set_max_lds(reinterpret_cast<const void*>(weight_prep_batch_kernel),
                                                       kPrepLds);      // once
    ODW_CHECK_HIP(attr, "weight_prep attr");
    weight_prep_batch_kernel<<<dim3(gx, n), 256, kPrepLds, (hipStream_t)stream_>>>(b);
    ODW_CHECK_LAUNCH("weight_prep_batch_kernel");
    return ODW_OK;
}

ODW_EXPORT int odw_colsum_bf16(const void* X, int ld, int M, int N, float* out, void* stream_) {
    ODW_REQUIRE(M >= 0 && N >= 0 && ld >= N, "colsum_bf16: bad dims");
    if (M == 0 || N == 0) return ODW_OK;
    ODW_REQUIRE(X && out && N % 8 == 0 && ld % 8 == 0 && (((uintptr_t)X) & 15) == 0, "colsum_bf16: N, ld multiples of 8, X 16-byte aligned");
    colsum_bf16_kernel<<<dim3((N + 63) / 64, (M + 255) / 256), 256, 0, (hipStream_t)stream_>>>((const unsigned short*)X, ld, M, N, out);
    ODW_CHECK_LAUNCH("colsum_bf16_kernel");
    return ODW_OK;
}

namespace {
int colsum_chunks(int M, int N) {          // enough workgroups to fill the chip, at least 256 rows each
    const int groups = (N + 63) / 64;
    int c = (2 * ODW_NUM_CU + groups - 1) / groups;
    const int most = (M + 255) / 256;
    c = c > most ? most : c;
    return c < 1 ? 1 : (c > 64 ? 64 : c);
}
}  // namespace

ODW_EXPORT int64_t odw_colsum_workspace(int M, int N) {
    if (M <= 0 || N <= 0) return 0;
    return (int64_t)colsum_chunks(M, N) * N * 4;
}

// deterministic form: workspace = odw_colsum_workspace(M, N) bytes of scratch (no initial contents required)
ODW_EXPORT int odw_colsum_bf16_ws(const void* X, int ld, int M, int N, float* out, void* workspace, int64_t workspace_bytes,
                                  void* stream_) {
    ODW_REQUIRE(M >= 0 && N >= 0 && ld >= N, "colsum_bf16: bad dims");
    if (M == 0 || N == 0) return ODW_OK;
    ODW_REQUIRE(X && out && N % 8 == 0 && ld % 8 == 0 && (((uintptr_t)X) & 15) == 0, "colsum_bf16: N, ld multiples of 8, X 16-byte aligned");
    ODW_REQUIRE(workspace && workspace_bytes >= odw_colsum_workspace(M, N) && (((uintptr_t)workspace) & 15) == 0,
                "colsum_bf16_ws: workspace of odw_colsum_workspace(M, N) bytes, 16-byte aligned");
    const int chunks = colsum_chunks(M, N);
    const int rows = ((M + chunks - 1) / chunks + 31) / 32 * 32;
    float* part = (float*)workspace;
    colsum_bf16_part_kernel<<<dim3((N + 63) / 64, chunks), 256, 0, (hipStream_t)stream_>>>((const unsigned short*)X, ld, M, N, rows, part);
    ODW_CHECK_LAUNCH("colsum_bf16_part_kernel");
    colsum_finish_kernel<<<(N + 255) / 256, 256, 0, (hipStream_t)stream_>>>(part, chunks, N, out);
    ODW_CHECK_LAUNCH("colsum_finish_kernel");
    return ODW_OK;
}

ODW_EXPORT int odw_conv_wgrad_unpack(const float* dwk, int ld, int Co, int Ci, int Cp, float* dw, void* stream_) {
    ODW_REQUIRE(Co > 0 && Ci > 0 && Cp >= Ci && ld >= 9 * Cp && dwk && dw, "conv_wgrad_unpack: bad arguments");
    wgrad_unpack_kernel<<<blocks_for((size_t)Co * Ci * 9), 256, 0, (hipStream_t)stream_>>>(dwk, ld, Co, Ci, Cp, dw);
    ODW_CHECK_LAUNCH("wgrad_unpack_kernel");
    return ODW_OK;
}

ODW_EXPORT int odw_im2col_t_bf16(const void* X, int n_pix, int H, int W, int C, int dilation, void* out, int ldm,
                                 void* stream_) {
    return odw_im2col_t_bf16_part(X, n_pix, H, W, C, dilation, out, ldm, ldm, stream_);
}

ODW_EXPORT int odw_im2col_t_bf16_part(const void* X, int n_pix, int H, int W, int C, int dilation, void* out, int ldm,
                                      int cols, void* stream_) {
    ODW_REQUIRE(n_pix > 0 && H > 0 && W > 0 && C > 0 && cols >= n_pix && ldm >= cols && n_pix % (H * W) == 0 && X && out,
                "im2col_t: bad arguments");
    ODW_REQUIRE(C % 8 == 0 && ldm % 8 == 0 && cols % 8 == 0 && (((uintptr_t)X) & 15) == 0 && (((uintptr_t)out) & 15) == 0,
                "im2col_t: C=%d, ldm=%d and cols=%d must be multiples of 8, pointers 16-byte aligned", C, ldm, cols);
    dim3 grid((C + 31) / 32, (cols + 511) / 512, 9);
    im2col_t_kernel<<<grid, 256, 0, (hipStream_t)stream_>>>((const unsigned short*)X, n_pix, H, W, C, dilation,
                                                            (unsigned short*)out, ldm, cols);
    ODW_CHECK_LAUNCH("im2col_t_kernel");
    return ODW_OK;
}

ODW_EXPORT int odw_maxpool2x2_nhwc_bf16(const void* X, int B, int H, int W, int C, void* Y, void* stream_) {
    ODW_REQUIRE(B > 0 && H % 2 == 0 && W % 2 == 0 && C % 8 == 0 && X && Y, "maxpool2x2: bad arguments");
    ODW_REQUIRE((((uintptr_t)X) & 15) == 0 && (((uintptr_t)Y) & 15) == 0, "maxpool2x2: 16-byte alignment");
    size_t n = (size_t)B * (H / 2) * (W / 2) * (C / 8);
    maxpool_fwd_kernel<<<blocks_for(n), 256, 0, (hipStream_t)stream_>>>((const uint4*)X, B, H, W, C / 8, (uint4*)Y);
    ODW_CHECK_LAUNCH("maxpool_fwd_kernel");
    return ODW_OK;
}

ODW_EXPORT int odw_maxpool2x2_nhwc_bf16_bwd(const void* X, const void* dY, int B, int H, int W, int C, void* dX,
                                            void* stream_) {
    ODW_REQUIRE(B > 0 && H % 2 == 0 && W % 2 == 0 && C > 0 && X && dY && dX, "maxpool2x2_bwd: bad arguments");
    size_t n = (size_t)B * (H / 2) * (W / 2) * C;
    maxpool_bwd_kernel<<<blocks_for(n), 256, 0, (hipStream_t)stream_>>>((const unsigned short*)X,
                                                                        (const unsigned short*)dY, B, H, W, C,
                                                                        (unsigned short*)dX);
    ODW_CHECK_LAUNCH("maxpool_bwd_kernel");
    return ODW_OK;
}

ODW_EXPORT int odw_nhwc_bf16_to_nchw_f32(const void* in, int B, int HW, int C, float* out, void* stream_) {
    ODW_REQUIRE(B > 0 && HW > 0 && C > 0 && in && out, "nhwc_to_nchw: bad arguments");
    dim3 grid((C + 31) / 32, (HW + 31) / 32, B);
    nhwc_to_nchw_kernel<<<grid, 256, 0, (hipStream_t)stream_>>>((const unsigned short*)in, HW, C, out);
    ODW_CHECK_LAUNCH("nhwc_to_nchw_kernel");
    return ODW_OK;
}

ODW_EXPORT int odw_nchw_f32_to_nhwc_bf16(const float* in, int B, int HW, int C, int Cp, void* out, void* stream_) {
    ODW_REQUIRE(B > 0 && HW > 0 && C > 0 && Cp >= C && in && out, "nchw_to_nhwc: bad arguments");
    dim3 grid((HW + 31) / 32, (Cp + 31) / 32, B);
    nchw_to_nhwc_kernel<<<grid, 256, 0, (hipStream_t)stream_>>>(in, HW, C, Cp, (unsigned short*)out);
    ODW_CHECK_LAUNCH("nchw_to_nhwc_kernel");
    return ODW_OK;
}

ODW_EXPORT int odw_maxpool2x2_nhwc_f32(const float* X, int B, int H, int W, int C, float* Y, void* stream_) {
    ODW_REQUIRE(B > 0 && H % 2 == 0 && W % 2 == 0 && C > 0 && X && Y, "maxpool2x2_f32: bad arguments");
    maxpool_f32_fwd_kernel<<<blocks_for((size_t)B * (H / 2) * (W / 2) * C), 256, 0, (hipStream_t)stream_>>>(X, B, H, W, C, Y);
    ODW_CHECK_LAUNCH("maxpool_f32_fwd_kernel");
    return ODW_OK;
}

ODW_EXPORT int odw_maxpool2x2_nhwc_f32_planes2(const float* X, int B, int H, int W, int C, void* Y, int ldy, void* stream_) {
    ODW_REQUIRE(B > 0 && H % 2 == 0 && W % 2 == 0 && C > 0 && C % 4 == 0 && X && Y, "maxpool2x2_f32_planes2: bad arguments");
    ODW_REQUIRE(ldy >= 2 * C && ldy % 8 == 0 && (((uintptr_t)X) & 15) == 0 && (((uintptr_t)Y) & 7) == 0,
                "maxpool2x2_f32_planes2: row stride %d (>= 2 C, multiple of 8), aligned buffers", ldy);
    maxpool_f32_planes2_kernel<<<blocks_for((size_t)B * (H / 2) * (W / 2) * (C / 4)), 256, 0, (hipStream_t)stream_>>>(
        X, B, H, W, C, (unsigned short*)Y, ldy);
    ODW_CHECK_LAUNCH("maxpool_f32_planes2_kernel");
    return ODW_OK;
}

ODW_EXPORT int odw_maxpool2x2_nhwc_f32x_bf16_bwd(const float* X, const void* dY, int B, int H, int W, int C, void* dX,
                                                 void* stream_) {
    ODW_REQUIRE(B > 0 && H % 2 == 0 && W % 2 == 0 && C > 0 && X && dY && dX, "maxpool2x2_f32x_bf16_bwd: bad arguments");
    maxpool_f32x_bf16_bwd_kernel<<<blocks_for((size_t)B * (H / 2) * (W / 2) * C), 256, 0, (hipStream_t)stream_>>>(
        X, (const unsigned short*)dY, B, H, W, C, (unsigned short*)dX);
    ODW_CHECK_LAUNCH("maxpool_f32x_bf16_bwd_kernel");
    return ODW_OK;
}

ODW_EXPORT int odw_maxpool2x2_nhwc_f32_bwd(const float* X, const float* dY, int B, int H, int W, int C, float* dX,
                                           void* stream_) {
    ODW_REQUIRE(B > 0 && H % 2 == 0 && W % 2 == 0 && C > 0 && X && dY && dX, "maxpool2x2_f32_bwd: bad arguments");
    maxpool_f32_bwd_kernel<<<blocks_for((size_t)B * (H / 2) * (W / 2) * C), 256, 0, (hipStream_t)stream_>>>(X, dY, B, H, W, C, dX);
    ODW_CHECK_LAUNCH("maxpool_f32_bwd_kernel");
    return ODW_OK;
}

ODW_EXPORT int odw_nchw_f32_to_nhwc_f32(const float* in, int B, int HW, int C, int Cp, float* out, void* stream_) {
    ODW_REQUIRE(B > 0 && HW > 0 && C > 0 && Cp >= C && in && out, "nchw_to_nhwc_f32: bad arguments");
    dim3 grid((HW + 31) / 32, (Cp + 31) / 32, B);
    layout_f32_kernel<false><<<grid, 256, 0, (hipStream_t)stream_>>>(in, HW, C, Cp, out);
    ODW_CHECK_LAUNCH("layout_f32_kernel");
    return ODW_OK;
}

ODW_EXPORT int odw_nhwc_f32_to_nchw_f32(const float* in, int B, int HW, int C, int Cp, float* out, void* stream_) {
    ODW_REQUIRE(B > 0 && HW > 0 && C > 0 && Cp >= C && in && out, "nhwc_to_nchw_f32: bad arguments");
    dim3 grid((HW + 31) / 32, (C + 31) / 32, B);
    layout_f32_kernel<true><<<grid, 256, 0, (hipStream_t)stream_>>>(in, HW, C, Cp, out);
    ODW_CHECK_LAUNCH("layout_f32_kernel");
    return ODW_OK;
}
