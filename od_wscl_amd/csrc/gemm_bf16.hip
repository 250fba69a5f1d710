// gemm_bf16.hip -- the ROI-head GEMM of the hot path on the gfx950 matrix cores.
//
//   C[M,N] (+)= epilogue( sum_k A[M,K] * B[N,K] )        "NT": both operands K-contiguous bf16
//
// One kernel serves all three products of a Linear layer (vgg16.py:121-127 fc6/fc7,
// sim_net.py:12-16, roi_weak_predictors.py:158-165):
//   forward  Y  = X  W^T      A = X  (M x K),    B = W     (N x K)
//   dgrad    dX = dY W        A = dY (M x N),    B = W^T   (K x N)   (bf16 transposed shadow)
//   wgrad    dW = dY^T X      A = dY^T (N x M),  B = X^T   (K x M)   (bf16 transposed copies)
// so the MFMA operand fetch is always two 16-byte K-contiguous reads per lane.
//
// Structure (CDNA4): 128x128 output tile per 256-thread workgroup (4 waves as 2x2, 64x64 per
// wave = 2x2 v_mfma_f32_32x32x16_bf16 accumulators), BK = 64, LDS double buffer (64 KB ->
// 2 workgroups/CU, 512 tiles of fc6 = one wave of workgroups on 256 CUs).  Global -> register
// -> LDS staging with the NEXT tile's loads issued before the current tile's MFMAs (T14).
// LDS rows are 128 B; 16-byte chunk c of row r lives in slot c ^ ((r >> 1) & 7): ds_write_b128
// (8-lane groups = one row) and ds_read_b128 (16-lane groups = 16 rows) are both conflict-free.
// Workgroup -> tile mapping is XCD-aware: the 8 XCDs (block id mod 8) each own a contiguous
// band of N tiles, so a band of B stays in that XCD's L2 while A streams through the MALL.
// Epilogue fused: bias, ReLU, counter-based dropout (odw_rng.h), bf16 or fp32 store,
// optional accumulate (C += ...) for weight gradients shared by several passes.
#include "odw_common.h"
#include "odw_rng.h"
#include "odw_planes.h"
#include <stdlib.h>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int kThreads = 256;
#ifndef ODW_MAX_SEG
#define ODW_MAX_SEG 8
#endif
constexpr int kMaxSeg = ODW_MAX_SEG;      // stacked logical passes per launch (each has its own dropout key); more = more launches.  (4 until
                                // round 5: an image with three positive classes stacks 6 sampled-row views -- drop + noise per class,
                                // loss.py:292-305 -- and needed two launches of every Linear, each streaming the weight again)
constexpr int kChunksPerRow = BK / 8;                 // 16-byte chunks per LDS row
constexpr int kTileChunks = BM * kChunksPerRow;       // 1024 uint4 per operand tile
constexpr int kLoadsPerThread = kTileChunks / kThreads;  // 4

// (the absmax word lives in a base the pair kernel's narrow epilogue does not carry: EpilogueS keeps the argument block it was
// tuned with -- see below for what one more spilled register did to its counted waits)
template <bool DYN> struct EpilogueAbsmax {
    unsigned* absmax = nullptr;      // fp32 output only: atomicMax of the bit patterns of |value stored| lands here (epi_absmax_commit)
};
template <> struct EpilogueAbsmax<false> { static constexpr unsigned* absmax = nullptr; };
template <int NSEG, bool DYN = true>
struct EpilogueT : EpilogueAbsmax<DYN> {
    static constexpr int kSegs = NSEG;
    static constexpr bool kDyn = DYN;      // the dynamic-row fields below are honoured (false: the pair kernel, whose registers are spoken for)
    const float* bias;     // (N) or null
    int relu;
    float drop_p;          // 0 = no dropout
    int nseg;              // dropout row segments (each logical draw has its own key)
    int seg_row[NSEG];
    uint32_t seg_k0[NSEG], seg_k1[NSEG];
    int accumulate;        // fp32 output only: C += result
    float alpha;           // scales the product before bias
    const unsigned short* mask;   // optional bf16 (M x ldmask): result forced to 0 where mask == 0 (ReLU backward)
    int ldmask;
    int pm;                // tile-order group height (tile_coords); 0 = the kernel's default
    const int* row_ids;    // dropout of a gathered row subset: logical row of GEMM row m (with segment 0's key); null = m
    int kchunk;            // split-K: > 0 = this launch's blockIdx.y owns K range [y*kchunk, (y+1)*kchunk) and writes
    long long split_stride;  //          its partial product split_stride bytes further into C (an fp32 workspace)
    // ---- device-resident extents (round 6: the loss's control flow stays on the GPU, loss_lists.hip).  The launch is
    // sized for the CAPACITY of its operands; the number of rows that exist (m_dev) / the length of the reduction (k_dev)
    // is read from device memory when the kernel starts: workgroups past it exit, the K loop stops at it.  null = the
    // host's M / K are exact.
    const int* m_dev = nullptr;
    const int* k_dev = nullptr;
    const uint4* row_tab = nullptr;  // dropout draws of stacked passes whose boundaries live on the device: row m draws element
                           // (row_tab[m].x, column) of the stream keyed (row_tab[m].y, row_tab[m].z); replaces seg_* / row_ids
};
typedef EpilogueT<kMaxSeg> Epilogue;
// The pair form of gemm_nt_cm_kernel (two accumulator sets, 256 VGPRs, counted vmcnt waits) takes the two segments it
// needs and no more: with the eight-entry table the compiler spilled 16 SGPRs and one more VGPR INSIDE its main loop,
// and a scratch reload sits in the same vmcnt queue as the operand DMA the loop's counted waits are written for --
// four rows of the DropBlock half came out wrong (tests/test_pair_gpu.py caught it).
typedef EpilogueT<2, false> EpilogueS;

template <class EPO>
inline EPO epilogue_narrow(const Epilogue& e) {
    constexpr int NSEG = EPO::kSegs;
    EPO o;
    o.m_dev = e.m_dev; o.k_dev = e.k_dev; o.row_tab = e.row_tab;
    if constexpr (EPO::kDyn) o.absmax = e.absmax;
    o.bias = e.bias; o.relu = e.relu; o.drop_p = e.drop_p; o.nseg = e.nseg < NSEG ? e.nseg : NSEG;
    for (int i = 0; i < NSEG; ++i) { o.seg_row[i] = e.seg_row[i]; o.seg_k0[i] = e.seg_k0[i]; o.seg_k1[i] = e.seg_k1[i]; }
    o.accumulate = e.accumulate; o.alpha = e.alpha; o.mask = e.mask; o.ldmask = e.ldmask; o.pm = e.pm; o.row_ids = e.row_ids;
    o.kchunk = e.kchunk; o.split_stride = e.split_stride;
    return o;
}

// split-K entry of a DMA kernel: narrow the operands / output to this workgroup's K range
#define ODW_SPLITK_ENTER(ODW_TILE_M, ODW_TILE_N)                                                               \
    int kchunk_ = ep.kchunk;                                                               \
    if (ep.m_dev) {             /* rows that exist: workgroups of the capacity-sized grid past them exit */ \
        const int md_ = *ep.m_dev;                                                         \
        M = md_ < M ? md_ : M;                                                             \
        tiles_m = (M + ODW_TILE_M - 1) / ODW_TILE_M;                                       \
        if ((int)blockIdx.x >= tiles_m * tiles_n) return;                                  \
    }                                                                                      \
    if (ep.k_dev) {             /* reduction length on the device; the K slices are cut from IT (none is left empty) */ \
        const int kd_ = *ep.k_dev;                                                         \
        K = kd_ < K ? kd_ : K;                                                             \
        if (kchunk_ > 0) {                                                                 \
            kchunk_ = ((K + (int)gridDim.y - 1) / (int)gridDim.y + 63) / 64 * 64;          \
            kchunk_ = kchunk_ < 64 ? 64 : kchunk_;                                         \
        }                                                                                  \
    }                                                                                      \
    if (kchunk_ > 0) {                                                                     \
        const int ks_ = blockIdx.y * kchunk_;                                              \
        A += ks_; B += ks_;                                                                \
        K = K - ks_ < kchunk_ ? K - ks_ : kchunk_;                                         \
        K = K < 0 ? 0 : K;                                                                 \
        Cv = reinterpret_cast<char*>(Cv) + (long long)blockIdx.y * ep.split_stride;        \
    }                                                                                      \
    if (K <= 0) {               /* an empty reduction (a K slice past a device-resident length): the tile is zeros.  NOT through */ \
        /* the pipeline: with no K step nothing waits for the prologue's DMA, which would land in the LDS the epilogue stages through */ \
        int tm_, tn_;                                                                      \
        tile_coords<4>(blockIdx.x, tiles_m, tiles_n, tm_, tn_, ep.pm);                     \
        zero_tile<OUT_BF16>(Cv, ldc, M, N, tm_ * ODW_TILE_M, tn_ * ODW_TILE_N, ODW_TILE_M, ODW_TILE_N, ep.accumulate); \
        return;                                                                            \
    }
// (kernels without a split-K form)
#define ODW_DYN_ENTER(ODW_TILE_M)                                                                  \
    if (ep.m_dev) {                                                                        \
        const int md_ = *ep.m_dev;                                                         \
        M = md_ < M ? md_ : M;                                                             \
        tiles_m = (M + ODW_TILE_M - 1) / ODW_TILE_M;                                       \
        if ((int)blockIdx.x >= tiles_m * tiles_n) return;                                  \
    }                                                                                      \
    if (ep.k_dev) {                                                                        \
        const int kd_ = *ep.k_dev; K = kd_ < K ? kd_ : K;                                  \
        if (K <= 0) {                                                                      \
            int tm_, tn_;                                                                  \
            tile_coords<8>(blockIdx.x, tiles_m, tiles_n, tm_, tn_, ep.pm);                 \
            zero_tile<OUT_BF16>(Cv, ldc, M, N, tm_ * BM, tn_ * BN, BM, BN, ep.accumulate); \
            return;                                                                        \
        }                                                                                  \
    }

__device__ __forceinline__ int lds_slot(int row, int chunk) { return row * kChunksPerRow + (chunk ^ ((row >> 1) & 7)); }

__device__ __forceinline__ unsigned short f2bf(float f) {   // round-to-nearest-even
    unsigned int u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);  // NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}


// two floats -> packed bf16 (round-to-nearest-even) in ONE instruction (v_cvt_pk_bf16_f32, gfx950); the integer form
// above is ~8 VALU instructions per element, and the staged epilogues convert 128 elements per lane
typedef __bf16 odw_bf16x2 __attribute__((ext_vector_type(2)));
typedef float odw_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t f2bf_pk(float lo, float hi) {
    const odw_f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, odw_bf16x2));
}

// staging: thread t moves 16-byte chunks t, t+256, t+512, t+768 of each operand tile
__device__ __forceinline__ void load_tile(const unsigned short* __restrict__ A, int lda,
                                          const unsigned short* __restrict__ B, int ldb, int M, int N, int K,
                                          int m0, int n0, int kt, int tid, uint4 (&ra)[kLoadsPerThread],
                                          uint4 (&rb)[kLoadsPerThread]) {
    const int k0 = kt * BK;
#pragma unroll
    for (int i = 0; i < kLoadsPerThread; ++i) {
        const int id = tid + i * kThreads;
        const int row = id >> 3, c = id & 7;
        const int k = k0 + c * 8;
        const int gm = m0 + row, gn = n0 + row;
        uint4 va = make_uint4(0, 0, 0, 0), vb = make_uint4(0, 0, 0, 0);
        if (gm < M && k < K) va = *reinterpret_cast<const uint4*>(A + (size_t)gm * lda + k);
        if (gn < N && k < K) vb = *reinterpret_cast<const uint4*>(B + (size_t)gn * ldb + k);
        ra[i] = va;
        rb[i] = vb;
    }
}

__device__ __forceinline__ void store_tile(uint4* __restrict__ sa, uint4* __restrict__ sb, int tid,
                                           const uint4 (&ra)[kLoadsPerThread], const uint4 (&rb)[kLoadsPerThread]) {
#pragma unroll
    for (int i = 0; i < kLoadsPerThread; ++i) {
        const int id = tid + i * kThreads;
        const int row = id >> 3, c = id & 7;
        sa[lds_slot(row, c)] = ra[i];
        sb[lds_slot(row, c)] = rb[i];
    }
}


// max |x| over the values a launch STORES, for the consumer that scatters them in fixed point (odw_fixed.h: ROI pooling's
// backward scales by the power of two above the largest gradient).  The stand-alone pre-pass (odwfx::absmax_kernel) re-reads
// the 200 MB input gradient of fc6 for it -- 87 us per step; the product's epilogue has every value in registers.  A maximum
// does not depend on the order it is taken in, so the word ends up bit-identical to the pre-pass's.  One atomic per wave at most:
// the word only grows, so a wave whose maximum is not above what a (possibly stale, hence smaller) read returns has nothing to
// add (same-address atomics serialise in L2 at ~14 ns each; almost all are skipped after the first few workgroups).
__device__ __forceinline__ unsigned epi_absbits(float x) { return __float_as_uint(x) & 0x7fffffffu; }
__device__ __forceinline__ void epi_absmax_commit(unsigned m, unsigned* __restrict__ out) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, off, 64));
    if ((threadIdx.x & 63) == 0 && m > __atomic_load_n(out, __ATOMIC_RELAXED)) atomicMax(out, m);
}

template <bool OUT_BF16, int MI = 2, int NJ = 2>
__device__ __forceinline__ void store_tile_out(const f32x16 (&acc)[MI][NJ], void* __restrict__ Cv, int ldc, int M, int N,
                                               int m0, int n0, int wm, int wn, int half, int l31,
                                               const Epilogue& ep) {
    // ---- epilogue.  C layout of a 32x32 tile: col = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    unsigned amax = 0;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int n = n0 + wn * (32 * NJ) + j * 32 + l31;
        if (n >= N) continue;
        const float bias = ep.bias ? ep.bias[n] : 0.0f;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * (32 * MI) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (m >= M) continue;
                float v = acc[i][j][r] * ep.alpha + bias;
                if (ep.relu) v = fmaxf(v, 0.0f);
                if (ep.mask && (ep.mask[(size_t)m * ep.ldmask + n] & 0x7fff) == 0) v = 0.0f;
                if (ep.drop_p > 0.0f) {
                    int srow = ep.seg_row[0];
                    uint32_t k0 = ep.seg_k0[0], k1 = ep.seg_k1[0];
                    #pragma unroll
                    for (int sg = 1; sg < kMaxSeg; ++sg)
                        if (ep.nseg > sg && m >= ep.seg_row[sg]) { srow = ep.seg_row[sg]; k0 = ep.seg_k0[sg]; k1 = ep.seg_k1[sg]; }
                    uint32_t lrow = ep.row_ids ? (uint32_t)ep.row_ids[m] : (uint32_t)(m - srow);
                    if (ep.row_tab) { const uint4 rt = ep.row_tab[m]; lrow = rt.x; k0 = rt.y; k1 = rt.z; }
                    const uint32_t idx = lrow * (uint32_t)N + (uint32_t)n;
                    v = odw_uniform(idx, k0, k1) >= ep.drop_p ? v * (1.0f / (1.0f - ep.drop_p)) : 0.0f;
                }
                if (OUT_BF16) {
                    reinterpret_cast<unsigned short*>(Cv)[(size_t)m * ldc + n] = f2bf(v);
                } else {
                    float* c = reinterpret_cast<float*>(Cv) + (size_t)m * ldc + n;
                    v = ep.accumulate ? *c + v : v;
                    *c = v;
                    amax = max(amax, epi_absbits(v));
                }
            }
        }
    }
    if (!OUT_BF16 && ep.absmax) epi_absmax_commit(amax, ep.absmax);
}

// ---- XCD-aware, L2-patch tile mapping (correctness never depends on it) ------------------------
// Workgroup b is dispatched to XCD b % 8, so XCD x is handed one CONTIGUOUS chunk of the tile order and
// its 32 CUs run the chunk's first tiles together.  The order itself is grouped: PM row-tiles at a time,
// walking N inside the group, so the tiles that are co-resident on one XCD form a PM x (resident/PM) patch
// (~1024 x 1024 of C) instead of one tall column: per K-step they fetch 2 x 1024 rows of operands
// instead of 4096 + 256, which halves the XCD's L2 miss traffic (fc6: 65 % -> ~83 % L2 hits).
template <int PM_DEFAULT>
__device__ __forceinline__ void tile_coords(int b, int tiles_m, int tiles_n, int& tm, int& tn, int pm = 0) {
    const int PM = pm > 0 ? pm : PM_DEFAULT;
    const int nblk = tiles_m * tiles_n;
    const int q = nblk / 8, r = nblk % 8, xcd = b % 8, j = b / 8;
    const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;   // bijective
    const int per_group = PM * tiles_n;
    const int group = t / per_group, in_group = t - group * per_group;
    const int first = group * PM;
    const int gsize = tiles_m - first < PM ? tiles_m - first : PM;
    tm = first + in_group % gsize;
    tn = in_group / gsize;
}

// the (TM x TN) tile at (m0, n0) of C set to zero (an empty reduction; accumulate: C += 0 = nothing to do)
template <bool OUT_BF16>
__device__ __forceinline__ void zero_tile(void* __restrict__ Cv, int ldc, int M, int N, int m0, int n0, int TM, int TN, int accumulate) {
    if (accumulate) return;
    for (int i = threadIdx.x; i < TM * TN; i += blockDim.x) {
        const int m = m0 + i / TN, n = n0 + i % TN;
        if (m < M && n < N) {
            if (OUT_BF16) reinterpret_cast<unsigned short*>(Cv)[(size_t)m * ldc + n] = 0;
            else reinterpret_cast<float*>(Cv)[(size_t)m * ldc + n] = 0.0f;
        }
    }
}

template <bool OUT_BF16>
__global__ __launch_bounds__(kThreads, 2) void gemm_nt_bf16_kernel(
    const unsigned short* __restrict__ A, int lda, const unsigned short* __restrict__ B, int ldb, int M, int N,
    int K, void* __restrict__ Cv, int ldc, Epilogue ep, int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) uint4 lds[];   // [stage][A|B][kTileChunks]
    ODW_DYN_ENTER(BM);
    // ---- XCD-aware tile mapping (block b runs on XCD b % 8; correctness never depends on it)
    int tm, tn;
    tile_coords<8>(blockIdx.x, tiles_m, tiles_n, tm, tn, ep.pm);
    const int m0 = tm * BM, n0 = tn * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int half = lane >> 5, l31 = lane & 31;

    uint4 ra[kLoadsPerThread], rb[kLoadsPerThread];

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x16){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

    const int nk = (K + BK - 1) / BK;
    load_tile(A, lda, B, ldb, M, N, K, m0, n0, 0, tid, ra, rb);
    store_tile(lds, lds + kTileChunks, tid, ra, rb);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int stage = kt & 1;
        if (kt + 1 < nk) load_tile(A, lda, B, ldb, M, N, K, m0, n0, kt + 1, tid, ra, rb);   // in flight under the MFMAs
        const uint4* sa = lds + (size_t)stage * 2 * kTileChunks;
        const uint4* sb = sa + kTileChunks;
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            const int c = kk * 2 + half;
            bf16x8 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = wm * 64 + i * 32 + l31;
                fa[i] = __builtin_bit_cast(bf16x8, sa[lds_slot(row, c)]);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row = wn * 64 + j * 32 + l31;
                fb[j] = __builtin_bit_cast(bf16x8, sb[lds_slot(row, c)]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) {
            uint4* na = lds + (size_t)(stage ^ 1) * 2 * kTileChunks;
            store_tile(na, na + kTileChunks, tid, ra, rb);
        }
        __syncthreads();
    }

    store_tile_out<OUT_BF16>(acc, Cv, ldc, M, N, m0, n0, wm, wn, half, l31, ep);
}


// ---- LDS-DMA variant ---------------------------------------------------------------------
// Same tile / swizzle / MFMA schedule, but the operand tiles go HBM -> LDS directly with
// global_load_lds_dwordx4 (16 B per lane, no VGPR round trip, no ds_write pass).  The DMA writes
// 64 consecutive 16-byte slots per wave instruction (LDS image is lane-linear), so the XOR swizzle
// is applied to the per-lane SOURCE address: lane l fills physical slot (l & 7) of row (l >> 3)
// with logical chunk (l & 7) ^ f(row) -- still one full 128-byte line per row.
// Needs lda/ldb >= K rounded up to 64 with zero padding (the caller's scratch operands are
// allocated that way); out-of-range rows are clamped (their products are never stored).
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;

__device__ __forceinline__ void dma_tile(const unsigned short* __restrict__ G, int ld, int nrows, int row0, int k0,
                                         uint4* __restrict__ tile, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rbase = wave * 32 + i * 8;              // 8 rows x 8 slots per instruction
        const int row = rbase + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        int g = row0 + row;
        g = g < nrows ? g : nrows - 1;
        const unsigned short* src = G + (size_t)g * ld + k0 + c * 8;
        __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(tile + rbase * kChunksPerRow), 16, 0, 0);
    }
}

template <bool OUT_BF16>
__global__ __launch_bounds__(kThreads, 2) void gemm_nt_bf16_glds_kernel(
    const unsigned short* __restrict__ A, int lda, const unsigned short* __restrict__ B, int ldb, int M, int N,
    int K, void* __restrict__ Cv, int ldc, Epilogue ep, int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) uint4 lds[];
    ODW_DYN_ENTER(BM);
    int tm, tn;
    tile_coords<8>(blockIdx.x, tiles_m, tiles_n, tm, tn, ep.pm);
    const int m0 = tm * BM, n0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int half = lane >> 5, l31 = lane & 31;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x16){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

    const int nk = (K + BK - 1) / BK;
    dma_tile(A, lda, M, m0, 0, lds, wave, lane);
    dma_tile(B, ldb, N, n0, 0, lds + kTileChunks, wave, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int stage = kt & 1;
        uint4* sa = lds + (size_t)stage * 2 * kTileChunks;
        uint4* sb = sa + kTileChunks;
        if (kt + 1 < nk) {
            uint4* na = lds + (size_t)(stage ^ 1) * 2 * kTileChunks;
            dma_tile(A, lda, M, m0, (kt + 1) * BK, na, wave, lane);
            dma_tile(B, ldb, N, n0, (kt + 1) * BK, na + kTileChunks, wave, lane);
        }
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            const int c = kk * 2 + half;
            bf16x8 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = __builtin_bit_cast(bf16x8, sa[lds_slot(wm * 64 + i * 32 + l31, c)]);
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[j] = __builtin_bit_cast(bf16x8, sb[lds_slot(wn * 64 + j * 32 + l31, c)]);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA has landed ...
        __syncthreads();                                     // ... and so has everyone else's
    }
    store_tile_out<OUT_BF16>(acc, Cv, ldc, M, N, m0, n0, wm, wn, half, l31, ep);
}

// Bias and mask operands of the staged epilogues (band_store), fetched as BATCHES of independent loads.
// Left per element -- "if (ep.bias && n < N) x += ep.bias[n]" -- hipcc emits one conditional global_load + s_waitcnt
// vmcnt(0) per output element: 128 serialised L2 round trips (~12-25 us) at the end of every workgroup, which was half
// of the convolution kernels on the wide maps and 15-20 % of the mid-sized head products.  A lane's columns inside the
// wave's 64 are cw(j, g) = 32 j + 8 g + 4 (lane >> 5) + {0..3}, the same for every row band: the bias goes through LDS
// once per wave (band_store), the mask comes as one 8-byte load per column group and band.
// mask words of one row for the lane's 8 column groups: bit test "(bf16 & 0x7fff) != 0" per element.  vec = rows and
// columns of the mask allow one 8-byte load per group (ldmask % 4 == 0, 8-byte aligned base, N % 4 == 0).
template <class EP>
__device__ __forceinline__ void epi_load_mask(const EP& ep, bool vec, long long m, int nw, int N, int half,
                                              uint2 (&mk)[2][4]) {
    const unsigned short* row = ep.mask + (size_t)(m > 0 ? m : 0) * ep.ldmask;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = nw + j * 32 + 8 * g + 4 * half;
            if (vec) {
                mk[j][g] = *reinterpret_cast<const uint2*>(row + (n < N ? n : 0));
            } else {
                unsigned short e[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) e[q] = row[n + q < N ? n + q : 0];
                mk[j][g] = make_uint2((uint32_t)e[0] | ((uint32_t)e[1] << 16), (uint32_t)e[2] | ((uint32_t)e[3] << 16));
            }
        }
}
__device__ __forceinline__ bool epi_mask_zero(const uint2& w, int q) {
    const uint32_t d = q < 2 ? w.x : w.y;
    return (((q & 1) ? (d >> 16) : d) & 0x7fffu) == 0;
}

// Staged epilogue of a wave's (NI x 32) x 64 block of C held as TRANSPOSED accumulators (the kernels run their MFMAs
// with the operands swapped: lane & 31 = row, each group of 4 registers = 4 consecutive columns,
// col = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)).  rowmap(r) = row of C for row r of the wave's block, or -1 = not stored
// (rows past M; pixels past the image edge in the halo-tile convolution, whose rows are not consecutive in C).
// Per 32-row band the wave applies the fused epilogue (alpha, bias, ReLU, ReLU-backward mask, counter-based dropout),
// parks the band in its private LDS region as 8-byte (bf16) / 16-byte (fp32) pieces (padded rows, conflict-free) and
// streams it out with one 16-byte store per lane: full 128-byte (bf16) / 256-byte (fp32) row segments instead of
// 2-byte scatters.  Rows of C must be 16-byte aligned and ldc >= N rounded up to the 16-byte chunk.
template <bool OUT_BF16, int NI, class RowMap, class EP>
__device__ __forceinline__ void band_store(const f32x16 (&acc)[NI][2], void* __restrict__ Cv, int ldc, int N, int nw,
                                           int wave, int lane, const EP& ep, char* lds, RowMap rowmap) {
    constexpr int kEl = OUT_BF16 ? 2 : 4;
    constexpr int kRowBytes = 64 * kEl + (OUT_BF16 ? 8 : 16);       // padded: conflict-free b64 / b128 writes
    char* region = lds + wave * (32 * kRowBytes);
    const int half = lane >> 5, l31 = lane & 31;
    const float keep_scale = ep.drop_p > 0.0f ? 1.0f / (1.0f - ep.drop_p) : 1.0f;
    // the wave's 64 bias values: one load per lane, parked in LDS behind the staging regions (32 registers held
    // across the bands made the 256x256 kernel spill); each column group reads its 4 back with one ds_read_b128
    float* const bias_s = reinterpret_cast<float*>(lds + 8 * 32 * (64 * 4 + 16)) + wave * 64;
    {
        const int n = nw + lane;
        bias_s[lane] = ep.bias ? ep.bias[n < N ? n : N - 1] : 0.0f;       // columns >= N are never stored
    }
    const bool mask_vec = ep.mask && ep.ldmask % 4 == 0 && N % 4 == 0 && (((uintptr_t)ep.mask) & 7) == 0;
    unsigned amax = 0;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const long long m = rowmap(i * 32 + l31);
        uint2 mk[2][4];
        if (ep.mask) epi_load_mask(ep, mask_vec, m, nw, N, half, mk);
        int srow = ep.seg_row[0];
        uint32_t k0 = ep.seg_k0[0], k1 = ep.seg_k1[0];
        uint32_t lrow = 0;
        if (ep.drop_p > 0.0f) {
            #pragma unroll
            for (int sg = 1; sg < EP::kSegs; ++sg)
                if (ep.nseg > sg && m >= ep.seg_row[sg]) { srow = ep.seg_row[sg]; k0 = ep.seg_k0[sg]; k1 = ep.seg_k1[sg]; }
            lrow = (ep.row_ids && m >= 0) ? (uint32_t)ep.row_ids[m] : (uint32_t)((int)m - srow);
            if (EP::kDyn && ep.row_tab && m >= 0) { const uint4 rt = ep.row_tab[m]; lrow = rt.x; k0 = rt.y; k1 = rt.z; }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cw = j * 32 + 8 * g + 4 * half;            // column inside the wave's 64
                const float4 b4 = *reinterpret_cast<const float4*>(bias_s + cw);
                const float bq[4] = {b4.x, b4.y, b4.z, b4.w};
                float v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float x = acc[i][j][4 * g + q] * ep.alpha + bq[q];
                    if (ep.relu) x = fmaxf(x, 0.0f);
                    if (ep.mask && epi_mask_zero(mk[j][g], q)) x = 0.0f;
                    if (ep.drop_p > 0.0f) {
                        const uint32_t idx = lrow * (uint32_t)N + (uint32_t)(nw + cw + q);
                        x = odw_uniform(idx, k0, k1) >= ep.drop_p ? x * keep_scale : 0.0f;
                    }
                    v[q] = x;
                }
                if (OUT_BF16) {
                    uint2 pk;
                    pk.x = f2bf_pk(v[0], v[1]);
                    pk.y = f2bf_pk(v[2], v[3]);
                    *reinterpret_cast<uint2*>(region + l31 * kRowBytes + cw * 2) = pk;
                } else {
                    *reinterpret_cast<float4*>(region + l31 * kRowBytes + cw * 4) = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
        }
        // the wave's own LDS traffic is ordered: read the band back row-major, 16 bytes per lane
        constexpr int kLanesPerRow = 64 * kEl / 16;                  // 8 (bf16) or 16 (fp32)
        constexpr int kRowsPerPass = 64 / kLanesPerRow;
#pragma unroll
        for (int t = 0; t < 32 / kRowsPerPass; ++t) {
            const int r = t * kRowsPerPass + lane / kLanesPerRow, cchunk = lane % kLanesPerRow;
            const uint4 d = *reinterpret_cast<const uint4*>(region + r * kRowBytes + cchunk * 16);
            const long long gm = rowmap(i * 32 + r);
            const int gn = nw + cchunk * (16 / kEl);
            if (gm < 0 || gn >= N) continue;
            char* dst = reinterpret_cast<char*>(Cv) + ((size_t)gm * ldc + gn) * kEl;
            if (!OUT_BF16 && ep.accumulate) {
                const float4 o = *reinterpret_cast<const float4*>(dst);
                const float4 a = __builtin_bit_cast(float4, d);
                const float4 w = make_float4(o.x + a.x, o.y + a.y, o.z + a.z, o.w + a.w);
                *reinterpret_cast<float4*>(dst) = w;
                if (EP::kDyn && ep.absmax) amax = max(max(amax, max(epi_absbits(w.x), epi_absbits(w.y))), max(epi_absbits(w.z), epi_absbits(w.w)));
            } else {
                *reinterpret_cast<uint4*>(dst) = d;
                // (a 16-byte chunk that starts below N may reach past it when N % 4 != 0: those lanes hold values of columns
                // that do not exist -- the launch refuses absmax for such an N)
                if (!OUT_BF16 && EP::kDyn && ep.absmax)
                    amax = max(max(amax, max(d.x & 0x7fffffffu, d.y & 0x7fffffffu)), max(d.z & 0x7fffffffu, d.w & 0x7fffffffu));
            }
        }
    }
    if (!OUT_BF16 && EP::kDyn && ep.absmax) epi_absmax_commit(amax, ep.absmax);
}

// The same block without the 16-byte-alignment conditions (any N, any ldc): element stores straight from the
// transposed accumulators.  Only odd-shaped outputs take it (the 357-column predictor written in place).
template <bool OUT_BF16, int NI, class RowMap, class EP>
__device__ __forceinline__ void band_store_scalar(const f32x16 (&acc)[NI][2], void* __restrict__ Cv, int ldc, int N, int nw,
                                                  int lane, const EP& ep, RowMap rowmap) {
    const int half = lane >> 5, l31 = lane & 31;
    const float keep_scale = ep.drop_p > 0.0f ? 1.0f / (1.0f - ep.drop_p) : 1.0f;
    unsigned amax = 0;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const long long m = rowmap(i * 32 + l31);
        if (m < 0) continue;
        int srow = ep.seg_row[0];
        uint32_t k0 = ep.seg_k0[0], k1 = ep.seg_k1[0];
        uint32_t lrow = 0;
        if (ep.drop_p > 0.0f) {
            #pragma unroll
            for (int sg = 1; sg < EP::kSegs; ++sg)
                if (ep.nseg > sg && m >= ep.seg_row[sg]) { srow = ep.seg_row[sg]; k0 = ep.seg_k0[sg]; k1 = ep.seg_k1[sg]; }
            lrow = ep.row_ids ? (uint32_t)ep.row_ids[m] : (uint32_t)((int)m - srow);
            if (EP::kDyn && ep.row_tab) { const uint4 rt = ep.row_tab[m]; lrow = rt.x; k0 = rt.y; k1 = rt.z; }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = nw + j * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (n >= N) continue;
                float x = acc[i][j][r] * ep.alpha + (ep.bias ? ep.bias[n] : 0.0f);
                if (ep.relu) x = fmaxf(x, 0.0f);
                if (ep.mask && (ep.mask[(size_t)m * ep.ldmask + n] & 0x7fff) == 0) x = 0.0f;
                if (ep.drop_p > 0.0f)
                    x = odw_uniform(lrow * (uint32_t)N + (uint32_t)n, k0, k1) >= ep.drop_p ? x * keep_scale : 0.0f;
                if (OUT_BF16) {
                    reinterpret_cast<unsigned short*>(Cv)[(size_t)m * ldc + n] = f2bf(x);
                } else {
                    float* c = reinterpret_cast<float*>(Cv) + (size_t)m * ldc + n;
                    x = ep.accumulate ? *c + x : x;
                    *c = x;
                    amax = max(amax, epi_absbits(x));
                }
            }
    }
    if (!OUT_BF16 && EP::kDyn && ep.absmax) epi_absmax_commit(amax, ep.absmax);
}

// rows of C reachable with 16-byte vector stores for every 16-byte column chunk that starts below N
__device__ __forceinline__ bool band_store_ok(const void* Cv, int ldc, int N, int el) {
    const int chunk = 16 / el;
    return (((uintptr_t)Cv) & 15) == 0 && ((size_t)ldc * el) % 16 == 0 && ldc >= (N + chunk - 1) / chunk * chunk;
}

// ---- 256x128 tile, 3-stage LDS ring, counted vmcnt -------------------------------------------------
// The 128x128 kernel above is LATENCY bound: a tile's DMA is issued one K-step (~0.35 us of MFMA)
// before it is needed, HBM/L2 latency under load is 1-2 us, so every K-step ends in a vmcnt(0) stall.
// Here 8 waves (4 x 2, still 64x64 = 2x2 MFMA 32x32x16 per wave) own a 256x128 tile, the operand
// tiles cycle through THREE 48 KB LDS slots (144 of the CU's 160 KB, one workgroup per CU) and each
// wave keeps two tiles of DMA in flight: the wait before the barrier is s_waitcnt vmcnt(6) -- "all but
// my 6 newest loads have landed" -- never vmcnt(0) in the main loop, and the barrier is the raw
// s_barrier (a __syncthreads() would drain the DMA queue).  25 % less L2->LDS traffic per FLOP too.
// Tried and rejected on this tile (same inputs, tools/gemm_bench.py): splitting the 8 waves into two role
// groups half a K-step apart (one group issues DMA + reads all 16 fragments while the other runs 16
// back-to-back MFMAs, two barriers per K-step) -- 555 TF vs 912 TF: the memory half-step takes ~1500 cycles
// by itself, i.e. the kernel is bound by operand delivery (9.9 GB of L2->LDS traffic per fc6 GEMM, ~11 TB/s),
// not by MFMA issue; and a 256x256 tile with 128x64 per wave needs > 256 VGPRs at 2 waves/SIMD (spills).
constexpr int RM = 256, RN = 128;
constexpr int kRingThreads = 512;
constexpr int kRingStageChunks = (RM + RN) * kChunksPerRow;      // 3072 uint4 = 48 KB
constexpr int kRingStages = 3;

template <int ROWS_PER_WAVE>
__device__ __forceinline__ void dma_rows(const unsigned short* __restrict__ G, int ld, int nrows, int row0, int k0,
                                         uint4* __restrict__ tile, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < ROWS_PER_WAVE / 8; ++i) {
        const int rbase = wave * ROWS_PER_WAVE + i * 8;
        const int row = rbase + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        int g = row0 + row;
        g = g < nrows ? g : nrows - 1;
        const unsigned short* src = G + (size_t)g * ld + k0 + c * 8;
        __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(tile + rbase * kChunksPerRow), 16, 0, 0);
    }
}

template <bool OUT_BF16>
__global__ __launch_bounds__(kRingThreads, 2) void gemm_nt_bf16_ring_kernel(
    const unsigned short* __restrict__ A, int lda, const unsigned short* __restrict__ B, int ldb, int M, int N,
    int K, void* __restrict__ Cv, int ldc, Epilogue ep, int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) uint4 lds[];       // [stage][A 256 rows | B 128 rows]
    ODW_SPLITK_ENTER(RM, RN);
    int tm, tn;
    tile_coords<4>(blockIdx.x, tiles_m, tiles_n, tm, tn, ep.pm);
    const int m0 = tm * RM, n0 = tn * RN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;            // 4 x 2 waves, 64 x 64 each
    const int half = lane >> 5, l31 = lane & 31;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x16){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

    const int nk = (K + BK - 1) / BK;
    // prologue: tiles 0 and 1 in flight (6 DMA instructions per wave per tile: 4 for A, 2 for B)
    dma_rows<32>(A, lda, M, m0, 0, lds, wave, lane);
    dma_rows<16>(B, ldb, N, n0, 0, lds + RM * kChunksPerRow, wave, lane);
    if (nk > 1) {
        uint4* s1 = lds + kRingStageChunks;
        dma_rows<32>(A, lda, M, m0, BK, s1, wave, lane);
        dma_rows<16>(B, ldb, N, n0, BK, s1 + RM * kChunksPerRow, wave, lane);
    }
    for (int kt = 0; kt < nk; ++kt) {
        // tile kt has landed once at most the 6 loads of tile kt+1 are still outstanding
        if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();           // everyone's share of tile kt is in LDS; slot (kt+2)%3 is free
        if (kt + 2 < nk) {
            uint4* sn = lds + (size_t)((kt + 2) % kRingStages) * kRingStageChunks;
            dma_rows<32>(A, lda, M, m0, (kt + 2) * BK, sn, wave, lane);
            dma_rows<16>(B, ldb, N, n0, (kt + 2) * BK, sn + RM * kChunksPerRow, wave, lane);
        }
        const uint4* sa = lds + (size_t)(kt % kRingStages) * kRingStageChunks;
        const uint4* sb = sa + RM * kChunksPerRow;
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            const int c = kk * 2 + half;
            bf16x8 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = __builtin_bit_cast(bf16x8, sa[lds_slot(wm * 64 + i * 32 + l31, c)]);
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[j] = __builtin_bit_cast(bf16x8, sb[lds_slot(wn * 64 + j * 32 + l31, c)]);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)       // operands swapped: transposed accumulators for band_store
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        }
    }
    const int mw = m0 + wm * 64;
    auto rowmap = [&](int r) -> long long { return mw + r < M ? (long long)(mw + r) : -1ll; };
    if (band_store_ok(Cv, ldc, N, OUT_BF16 ? 2 : 4)) {           // workgroup-uniform
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                             // every wave is done with the operand tiles
        band_store<OUT_BF16, 2>(acc, Cv, ldc, N, n0 + wn * 64, wave, lane, ep, reinterpret_cast<char*>(lds), rowmap);
    } else {
        band_store_scalar<OUT_BF16, 2>(acc, Cv, ldc, N, n0 + wn * 64, lane, ep, rowmap);
    }
}

// ---- the first head Linear over CELL-MAJOR planes, clean + DropBlock outputs from ONE pass (round 4) ----------------
// ROIWeakRegHead evaluates fc6 on the pooled features and on their DropBlock view (weak_head.py:107-112); DropBlock
// zeroes whole cells of a ROI's 7 x 7 grid for every channel and rescales by a global factor (drop_block.py:38-50):
//     y_clean[r] =      sum_s  P_s[r],                P_s[r] = sum_c x[r][c][s] W[.][c][s]   (the cell's partial product)
//     y_drop[r]  = g * sum_s keep[r][s] P_s[r]
// Rounds 1-3 ran the two as one stacked product (M = 2P): every P_s was computed twice.  Here the reduction walks the
// cells in order (operands laid out k' = s * C + c: split.hip split_rows_cm_kernel, roi_pool.hip), the wave keeps the
// RUNNING sum A_s = P_0 + ... + P_s in its MFMA accumulators -- which is y_clean at the end -- and a second register
// set D that is touched only at the 49 cell boundaries:
//     sum_s keep_s (A_s - A_{s-1})  =  sum_s A_s (keep_s - keep_{s+1}),   keep_S = 0         (summation by parts)
// i.e. D += (keep_s - keep_{s+1}) * A after cell s, a coefficient in {-1, 0, 1} per ROI: 64 FMAs per wave and cell
// against 3 x 8 K tiles of 16 MFMAs.  Half the matrix-core work of the stacked pass, and the DropBlock half of the
// stacked operand is not read.  The two STORED planes [hi | mid] of each operand meet as the three products of
// precision.py's "bf16x2f" (hi.hi, hi.mid, mid.hi) through the K-tile map below -- no duplicated hi plane.
// Tile / ring / DMA as gemm_nt_bf16_ring_kernel (64 x 64 per wave: 64 + 64 accumulator registers).  keep == null: the
// plain product over the same layout (the sampled-row views of the contrastive loss), optionally split over cells.
struct CmArgs {
    int C, S;                   // K = S cells x C channels per plane; C % 64 == 0, S <= 64
    int a_mid, b_mid;           // element offset of the mid plane inside a row of A / B (the hi plane starts at 0)
    const float* keep;          // (M x S) DropBlock keep mask, or null
    const float* keep_sum;      // its sum (device scalar): g = M * S / sum
    int drop_row0;              // first row of C of the DropBlock half (>= M)
};

template <bool PAIR> struct CmEp { typedef Epilogue type; };
template <> struct CmEp<true> { typedef EpilogueS type; };      // (see EpilogueS)

template <bool PAIR, int SHARE>
__global__ __launch_bounds__(kRingThreads, 2) void gemm_nt_cm_kernel(
    const unsigned short* __restrict__ A, int lda, const unsigned short* __restrict__ B, int ldb, int M, int N,
    void* __restrict__ Cv, int ldc, typename CmEp<PAIR>::type ep, CmArgs cm, int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) uint4 lds[];       // SHARE: [3 A slots | 4 B slots]; else [stage][A | B]
    if (!PAIR && ep.m_dev) {          // rows that exist (device-resident count): the capacity-sized grid's other workgroups exit
        const int md_ = *ep.m_dev;
        M = md_ < M ? md_ : M;
        tiles_m = (M + RM - 1) / RM;
        if ((int)blockIdx.x >= tiles_m * tiles_n) return;
    }
    int cell_lo = 0, cell_hi = cm.S;
    if (ep.kchunk > 0) {                                              // split over cells (plain mode)
        cell_lo = blockIdx.y * ep.kchunk;
        cell_hi = cell_lo + ep.kchunk < cm.S ? cell_lo + ep.kchunk : cm.S;
        Cv = reinterpret_cast<char*>(Cv) + (long long)blockIdx.y * ep.split_stride;
    }
    int tm, tn;
    tile_coords<4>(blockIdx.x, tiles_m, tiles_n, tm, tn, ep.pm);
    const int m0 = tm * RM, n0 = tn * RN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;            // 4 x 2 waves, 64 x 64 each
    const int half = lane >> 5, l31 = lane & 31;
    const int mw = m0 + wm * 64;

    f32x16 acc[2][2], dacc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            acc[i][j] = (f32x16){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            dacc[i][j] = (f32x16){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        }
    // the lane's two ROIs (transposed accumulators: lane & 31 = row) and their keep masks as bit sets
    unsigned long long kbits[2] = {0ull, 0ull};
    if (PAIR) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int r = mw + i * 32 + l31;
            r = r < M ? r : M - 1;
            const float* kr = cm.keep + (size_t)r * cm.S;
            for (int s = 0; s < cm.S; ++s) kbits[i] |= (unsigned long long)(kr[s] != 0.0f) << s;
        }
        // the mask loads are complete HERE: the counted vmcnt waits of the K loop must only ever see operand DMA
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(kbits[0]), "+v"(kbits[1]) : : "memory");
    }

    const int ntc = cm.C / BK, per_cell = 3 * ntc;      // K tiles per plane product of a cell, per cell
    // the D += coef * A step at the end of cell `cell` (summation by parts, see above)
    auto cell_done = [&](int cell) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int cur = (int)((kbits[i] >> cell) & 1ull);
            const int nxt = cell + 1 < cell_hi ? (int)((kbits[i] >> (cell + 1)) & 1ull) : 0;
            const float coef = (float)(cur - nxt);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) dacc[i][j][r] = fmaf(coef, acc[i][j][r], dacc[i][j][r]);
        }
    };
    auto mfma_step = [&](const uint4* sa, const uint4* sb) {
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            const int c = kk * 2 + half;
            bf16x8 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = __builtin_bit_cast(bf16x8, sa[lds_slot(wm * 64 + i * 32 + l31, c)]);
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[j] = __builtin_bit_cast(bf16x8, sb[lds_slot(wn * 64 + j * 32 + l31, c)]);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)       // operands swapped: transposed accumulators for band_store
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        }
    };
    if (SHARE == 1) {
        // The three plane products of one 64-channel group g (columns g * 64 of both planes: cell * C + ct * 64 == g * 64)
        // read FOUR operand tiles, not six: steps (Ah, Bh), (Ah, Bm), (Am, Bh).  LDS = 3 A slots of 32 KB + 4 B slots of
        // 16 KB (160 KB); A tiles in the order Ah(0) Am(0) Ah(1) ... cycle through the A slots, B tiles Bh(0) Bm(0) Bh(1)
        // ... through the B slots.  Issue schedule (each tile three steps before its first use, into a slot whose last
        // reader finished before the barrier in front of the issue):
        //     top of step 3g    : Ah(g+1), Bh(g+1)         wait for (Ah, Bh)(g)   = all but the 6 newest loads
        //     top of step 3g + 1: Bm(g+1)                  wait for Bm(g)         = all but the 10 newest
        //     top of step 3g + 2: Am(g+1)                  wait for Am(g)         = all but the 8 newest
        // (4 DMA instructions per wave for an A tile, 2 for a B tile; past the end the last group is fetched again into
        // the free slots so that the counts stay constant.)  A third less L2 -> LDS traffic than the 48 KB-per-step ring.
        const int g_lo = cell_lo * ntc, g_hi = cell_hi * ntc;
        uint4* const a_slots = lds;
        uint4* const b_slots = lds + 3 * RM * kChunksPerRow;
        auto a_slot = [&](int t) { return a_slots + (size_t)(t % 3) * (RM * kChunksPerRow); };
        auto b_slot = [&](int t) { return b_slots + (size_t)(t & 3) * (RN * kChunksPerRow); };
        auto col = [&](int g) { return (g < g_hi ? g : g_hi - 1) * BK; };
        if (g_hi > g_lo) {
            dma_rows<32>(A, lda, M, m0, col(g_lo), a_slot(0), wave, lane);
            dma_rows<16>(B, ldb, N, n0, col(g_lo), b_slot(0), wave, lane);
            dma_rows<16>(B, ldb, N, n0, col(g_lo) + cm.b_mid, b_slot(1), wave, lane);
            dma_rows<32>(A, lda, M, m0, col(g_lo) + cm.a_mid, a_slot(1), wave, lane);
        }
        int in_cell = 0, cell = cell_lo;
        for (int g = g_lo, t = 0; g < g_hi; ++g, t += 2) {          // t = index of Ah(g) / Bh(g) in the tile sequences
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            dma_rows<32>(A, lda, M, m0, col(g + 1), a_slot(t + 2), wave, lane);
            dma_rows<16>(B, ldb, N, n0, col(g + 1), b_slot(t + 2), wave, lane);
            mfma_step(a_slot(t), b_slot(t));
            asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            dma_rows<16>(B, ldb, N, n0, col(g + 1) + cm.b_mid, b_slot(t + 3), wave, lane);
            mfma_step(a_slot(t), b_slot(t + 1));
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            dma_rows<32>(A, lda, M, m0, col(g + 1) + cm.a_mid, a_slot(t + 3), wave, lane);
            mfma_step(a_slot(t + 1), b_slot(t));
            if (++in_cell == ntc) {
                if (PAIR) cell_done(cell);
                in_cell = 0;
                ++cell;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the re-fetched tail tiles land in LDS the epilogue reuses
    } else {
    const int nk = (cell_hi - cell_lo) * per_cell;
    // K tile kt -> (cell, product p, channel tile): A reads planes [hi hi mid], B reads [hi mid hi]
    auto issue = [&](int kt, uint4* slot) {
        const int cell = cell_lo + kt / per_cell, rem = kt % per_cell, p = rem / ntc, ct = rem - p * ntc;
        const int base = cell * cm.C + ct * BK;
        dma_rows<32>(A, lda, M, m0, base + (p == 2 ? cm.a_mid : 0), slot, wave, lane);
        dma_rows<16>(B, ldb, N, n0, base + (p == 1 ? cm.b_mid : 0), slot + RM * kChunksPerRow, wave, lane);
    };
    if (nk > 0) issue(0, lds);
    if (nk > 1) issue(1, lds + kRingStageChunks);
    int in_cell = 0, cell = cell_lo;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + 2 < nk) issue(kt + 2, lds + (size_t)((kt + 2) % kRingStages) * kRingStageChunks);
        const uint4* sa = lds + (size_t)(kt % kRingStages) * kRingStageChunks;
        mfma_step(sa, sa + RM * kChunksPerRow);
        if (++in_cell == per_cell) {              // the cell is complete
            if (PAIR) cell_done(cell);
            in_cell = 0;
            ++cell;
        }
    }
    }
    auto rowmap = [&](int r) -> long long { return mw + r < M ? (long long)(mw + r) : -1ll; };
    auto rowmap_d = [&](int r) -> long long { return mw + r < M ? (long long)(cm.drop_row0 + mw + r) : -1ll; };
    typename CmEp<PAIR>::type epd = ep;
    if (PAIR) epd.alpha = ep.alpha * ((float)((double)M * cm.S) / *cm.keep_sum);
    if (band_store_ok(Cv, ldc, N, 4)) {           // workgroup-uniform
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                             // every wave is done with the operand tiles
        band_store<false, 2>(acc, Cv, ldc, N, n0 + wn * 64, wave, lane, ep, reinterpret_cast<char*>(lds), rowmap);
        if (PAIR) band_store<false, 2>(dacc, Cv, ldc, N, n0 + wn * 64, wave, lane, epd, reinterpret_cast<char*>(lds), rowmap_d);
    } else {
        band_store_scalar<false, 2>(acc, Cv, ldc, N, n0 + wn * 64, lane, ep, rowmap);
        if (PAIR) band_store_scalar<false, 2>(dacc, Cv, ldc, N, n0 + wn * 64, lane, epd, rowmap_d);
    }
}

// ---- 256x256 tile, 8 waves of 128x64 ---------------------------------------------------------
// The ring kernel is bound by operand delivery, not by MFMA issue (PMC: matrix cores busy 48 %, waves
// parked on vmcnt/barrier 38 %): a 64x64 wave tile reads 4 KB of LDS per 4 MFMAs and a 256x128 block
// pulls 48 KB from L2 per K-step of 1024 MFMA cycles.  Here each wave owns 128x64 (4x2 accumulators =
// 128 registers, still two waves per SIMD inside the 512-entry file) and the block 256x256: LDS
// reads per MFMA drop 25 %, L2 -> LDS bytes per FLOP drop 33 %, and one K-step is 2048 MFMA cycles per
// SIMD -- long enough that a plain double buffer (2 x 64 KB) hides the DMA of the next tile completely.
// Default since the end of round 1 (X == 7): the spare 32 KB of LDS hold a THIRD B slot, B is fetched two tiles
// ahead and the closing wait becomes vmcnt(4) -- the critical lookahead grows from 0.75 to 1.0 K step:
// fc6 forward 1.06 -> 1.12 PF, dgrad 0.88 -> 0.95, wgrad 0.92 -> 1.00, 8192^3 1.27 -> 1.33.  (Moving the four A pieces of
// the closing slice from its second half to right behind the barrier, +6 % of a step of lookahead for A: no change.)
// Measured against hipBLASLt (tools/gemm_vs_lib.py, bf16 in/out, no epilogue), two-slot form: stacked fc6 forward
// 1.09 vs 1.22 PF, fc6 dgrad 0.88 vs 1.05, fc6 wgrad 0.94 vs 1.25, 8192^3 1.26 vs 1.56 -- the library is 12-35 % ahead
// on the plain product.  Tried for that gap and rejected: the same 256x256 tile as FOUR waves of 128x128 with the 16 accumulators
// in AGPRs (one wave per SIMD, LDS reads per K step 192 -> 128 KB, two asm statements per K slice, reads ordered by
// need with counted lgkmcnt waits): bit-correct on the first run, but 1.04-1.08 PF, and 1.30 PF with the operand DMA
// removed where this 8-wave form reaches 1.55-1.6 (= the MFMA pipe at the ~1.6 GHz it holds under load).  With a
// single wave per SIMD nothing covers the barrier / fragment waits of a K step; what the library adds instead is
// operand prefetch through registers (its depth is not bounded by the 160 KB of LDS).  Next: register-staged
// prefetch of the B tile on top of this form.
constexpr int GM = 256, GN = 256;
constexpr int kBigThreads = 512;
constexpr int kBigStageChunks = (GM + GN) * kChunksPerRow;        // 4096 uint4 = 64 KB

// One K-slice (16 of K) of a wave's 128x64 block as ONE asm statement, so that the order inside is ours and not
// the machine scheduler's (left alone, hipcc emits ds_read -> s_waitcnt -> 2 MFMAs and chains MFMAs on one
// accumulator): wait for the current fragments, then 8 MFMAs on 8 different accumulators with the 6 ds_read_b128
// of the NEXT slice's fragments issued in their shadow.  SYNC = the slice that closes a K-step: once its own
// fragments are in registers the wave is done with the current LDS slot, so it also waits for its share of the
// next tile's DMA and joins the block barrier -- after which the prefetch below reads the OTHER slot and the
// slot just left can be refilled.  The MFMA pipe never drains around the barrier.
template <bool SYNC, int X = 0>
__device__ __forceinline__ void big_slice(f32x16 (&acc)[4][2], const bf16x8 (&ca)[4], const bf16x8 (&cb)[2],
                                          bf16x8 (&na)[4], bf16x8 (&nb)[2], unsigned addr_a, unsigned addr_b) {
#define ODW_BIG_BODY                                                                                   \
        "ds_read_b128 %8, %20\n\t"                                                                     \
        "v_mfma_f32_32x32x16_bf16 %0, %18, %14, %0\n\t"                                                \
        "ds_read_b128 %12, %21\n\t"                                                                    \
        "v_mfma_f32_32x32x16_bf16 %1, %19, %14, %1\n\t"                                                \
        "ds_read_b128 %9, %20 offset:4096\n\t"                                                         \
        "v_mfma_f32_32x32x16_bf16 %2, %18, %15, %2\n\t"                                                \
        "ds_read_b128 %13, %21 offset:4096\n\t"                                                        \
        "v_mfma_f32_32x32x16_bf16 %3, %19, %15, %3\n\t"                                                \
        "ds_read_b128 %10, %20 offset:8192\n\t"                                                        \
        "v_mfma_f32_32x32x16_bf16 %4, %18, %16, %4\n\t"                                                \
        "ds_read_b128 %11, %20 offset:12288\n\t"                                                       \
        "v_mfma_f32_32x32x16_bf16 %5, %19, %16, %5\n\t"                                                \
        "v_mfma_f32_32x32x16_bf16 %6, %18, %17, %6\n\t"                                                \
        "v_mfma_f32_32x32x16_bf16 %7, %19, %17, %7\n\t"
#define ODW_BIG_OPERANDS                                                                               \
        : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[2][0]), "+v"(acc[2][1]),  \
          "+v"(acc[3][0]), "+v"(acc[3][1]), "=&v"(na[0]), "=&v"(na[1]), "=&v"(na[2]), "=&v"(na[3]), "=&v"(nb[0]),  \
          "=&v"(nb[1])                                                                                 \
        : "v"(ca[0]), "v"(ca[1]), "v"(ca[2]), "v"(ca[3]), "v"(cb[0]), "v"(cb[1]), "v"(addr_a), "v"(addr_b)       \
        : "memory"
    if (SYNC && X == 2) {
        asm volatile("s_waitcnt lgkmcnt(0)\n\t"
                     "s_barrier\n\t" ODW_BIG_BODY ODW_BIG_OPERANDS);
    } else if (SYNC && X == 3) {
        asm volatile("s_waitcnt lgkmcnt(0)\n\t"
                     "s_waitcnt vmcnt(0)\n\t" ODW_BIG_BODY ODW_BIG_OPERANDS);
    } else if (SYNC) {
        asm volatile("s_waitcnt lgkmcnt(0)\n\t"
                     "s_waitcnt vmcnt(0)\n\t"
                     "s_barrier\n\t" ODW_BIG_BODY ODW_BIG_OPERANDS);
    } else {
        asm volatile("s_waitcnt lgkmcnt(0)\n\t" ODW_BIG_BODY ODW_BIG_OPERANDS);
    }
#undef ODW_BIG_BODY
#undef ODW_BIG_OPERANDS
}

// The same slice with four LDS-DMA pieces of one operand tile (4 KB of this wave's share) issued one per MFMA in
// its second half, instead of four back-to-back between slices: m0 = LDS destination of the piece, the source is
// sbase + voff[lane] (SGPR base advancing 128 B per K-step, constant per-lane row/chunk offsets).
template <bool SYNC, int VM = 0>
__device__ __forceinline__ void big_slice_dma(f32x16 (&acc)[4][2], const bf16x8 (&ca)[4], const bf16x8 (&cb)[2],
                                              bf16x8 (&na)[4], bf16x8 (&nb)[2], unsigned addr_a, unsigned addr_b,
                                              unsigned lds_dst, const unsigned (&voff)[4], unsigned long long sbase) {
#define ODW_BIG_BODY_DMA                                                                               \
        "ds_read_b128 %8, %20\n\t"                                                                     \
        "v_mfma_f32_32x32x16_bf16 %0, %18, %14, %0\n\t"                                                \
        "ds_read_b128 %12, %21\n\t"                                                                    \
        "v_mfma_f32_32x32x16_bf16 %1, %19, %14, %1\n\t"                                                \
        "ds_read_b128 %9, %20 offset:4096\n\t"                                                         \
        "v_mfma_f32_32x32x16_bf16 %2, %18, %15, %2\n\t"                                                \
        "ds_read_b128 %13, %21 offset:4096\n\t"                                                        \
        "s_mov_b32 m0, %22\n\t"                                                                        \
        "v_mfma_f32_32x32x16_bf16 %3, %19, %15, %3\n\t"                                                \
        "ds_read_b128 %10, %20 offset:8192\n\t"                                                        \
        "global_load_lds_dwordx4 %23, %27\n\t"                                                         \
        "v_mfma_f32_32x32x16_bf16 %4, %18, %16, %4\n\t"                                                \
        "ds_read_b128 %11, %20 offset:12288\n\t"                                                       \
        "s_add_u32 m0, %22, 0x400\n\t"                                                                 \
        "v_mfma_f32_32x32x16_bf16 %5, %19, %16, %5\n\t"                                                \
        "global_load_lds_dwordx4 %24, %27\n\t"                                                         \
        "s_add_u32 m0, %22, 0x800\n\t"                                                                 \
        "v_mfma_f32_32x32x16_bf16 %6, %18, %17, %6\n\t"                                                \
        "global_load_lds_dwordx4 %25, %27\n\t"                                                         \
        "s_add_u32 m0, %22, 0xc00\n\t"                                                                 \
        "v_mfma_f32_32x32x16_bf16 %7, %19, %17, %7\n\t"                                                \
        "global_load_lds_dwordx4 %26, %27\n\t"
#define ODW_BIG_OPERANDS_DMA                                                                           \
        : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[2][0]), "+v"(acc[2][1]),  \
          "+v"(acc[3][0]), "+v"(acc[3][1]), "=&v"(na[0]), "=&v"(na[1]), "=&v"(na[2]), "=&v"(na[3]), "=&v"(nb[0]),  \
          "=&v"(nb[1])                                                                                 \
        : "v"(ca[0]), "v"(ca[1]), "v"(ca[2]), "v"(ca[3]), "v"(cb[0]), "v"(cb[1]), "v"(addr_a), "v"(addr_b),      \
          "s"(lds_dst), "v"(voff[0]), "v"(voff[1]), "v"(voff[2]), "v"(voff[3]), "s"(sbase)                      \
        : "memory", "scc"       /* m0 is rewritten too: hipcc reserves it and reloads it before each use of its own */
    if (SYNC && VM == 4) {      // three-slot B ring: the four newest pieces (B of tile k+2) may stay in flight
        asm volatile("s_waitcnt lgkmcnt(0)\n\t"
                     "s_waitcnt vmcnt(4)\n\t"
                     "s_barrier\n\t" ODW_BIG_BODY_DMA ODW_BIG_OPERANDS_DMA);
    } else if (SYNC) {
        asm volatile("s_waitcnt lgkmcnt(0)\n\t"
                     "s_waitcnt vmcnt(0)\n\t"
                     "s_barrier\n\t" ODW_BIG_BODY_DMA ODW_BIG_OPERANDS_DMA);
    } else {
        asm volatile("s_waitcnt lgkmcnt(0)\n\t" ODW_BIG_BODY_DMA ODW_BIG_OPERANDS_DMA);
    }
#undef ODW_BIG_BODY_DMA
#undef ODW_BIG_OPERANDS_DMA
}

template <bool OUT_BF16, int X = 0>
__global__ __launch_bounds__(kBigThreads, 2) void gemm_nt_bf16_big_kernel(
    const unsigned short* __restrict__ A, int lda, const unsigned short* __restrict__ B, int ldb, int M, int N,
    int K, void* __restrict__ Cv, int ldc, Epilogue ep, int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) uint4 lds[];       // [slot][A 256 rows | B 256 rows]
    ODW_SPLITK_ENTER(GM, GN);
    int tm, tn;
    tile_coords<4>(blockIdx.x, tiles_m, tiles_n, tm, tn, ep.pm);
    const int m0 = tm * GM, n0 = tn * GN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;            // 2 x 4 waves, 128 x 64 each
    const int half = lane >> 5, l31 = lane & 31;

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x16){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

    // LDS byte address of this lane's fragment chunk for K-slice kk: fragment i of A sits 4096*i further on
    // (32 rows x 128 B), the B tile 32 KB after the A tile, slot 1 64 KB after slot 0
    const int row_a = wm * 128 + l31, row_b = wn * 64 + l31;
    unsigned off_a[4], off_b[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        off_a[kk] = (unsigned)lds_slot(row_a, 2 * kk + half) * 16u;
        off_b[kk] = (unsigned)lds_slot(row_b, 2 * kk + half) * 16u + (unsigned)(GM * kChunksPerRow * 16);
    }
    constexpr unsigned kSlotBytes = kBigStageChunks * 16;
    uint4* const slot0 = lds;
    uint4* const slot1 = lds + kBigStageChunks;

    const int nk = (K + BK - 1) / BK;
    // X == 7: LDS = [A0 | A1 | B0 | B1 | B2] (5 x 32 KB = the whole 160 KB): B is fetched TWO tiles ahead, so the wait
    // that closes a K step only concerns pieces issued at least one full step earlier (A: 1.0 step, B: 1.75; the
    // two-slot form waits for B pieces issued 0.75 step earlier)
    constexpr unsigned kHalf = (unsigned)(GM * kChunksPerRow * 16);      // 32 KB
    if (X == 7) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) off_b[kk] += kHalf;                // B region starts at 64 KB
    }
    // prologue: tile 0 -> slot 0, landed and visible; A half of tile 1 in flight
    dma_rows<32>(A, lda, M, m0, 0, slot0, wave, lane);
    dma_rows<32>(B, ldb, N, n0, 0, X == 7 ? lds + 2 * GM * kChunksPerRow : slot0 + GM * kChunksPerRow, wave, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (nk > 1) dma_rows<32>(A, lda, M, m0, BK, X == 7 ? lds + GM * kChunksPerRow : slot1, wave, lane);
    if (X == 7 && nk > 1) dma_rows<32>(B, ldb, N, n0, BK, lds + 3 * GM * kChunksPerRow, wave, lane);
    if (X == 1 && nk > 1) dma_rows<32>(B, ldb, N, n0, BK, slot1 + GM * kChunksPerRow, wave, lane);
    bf16x8 f0a[4], f0b[2], f1a[4], f1b[2];
    asm volatile("ds_read_b128 %0, %6\n\t"
                 "ds_read_b128 %4, %7\n\t"
                 "ds_read_b128 %1, %6 offset:4096\n\t"
                 "ds_read_b128 %5, %7 offset:4096\n\t"
                 "ds_read_b128 %2, %6 offset:8192\n\t"
                 "ds_read_b128 %3, %6 offset:12288\n\t"
                 : "=&v"(f0a[0]), "=&v"(f0a[1]), "=&v"(f0a[2]), "=&v"(f0a[3]), "=&v"(f0b[0]), "=&v"(f0b[1])
                 : "v"(off_a[0]), "v"(off_b[0])
                 : "memory");
    if (X == 0 || X == 6 || X == 7) {
        // per-lane source offsets of this wave's four pieces of an A / B tile (rows clamped at the matrix edge)
        unsigned voa[4], vob[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = wave * 32 + i * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((row >> 1) & 7);
            int ga = m0 + row, gb = n0 + row;
            ga = ga < M ? ga : M - 1;
            gb = gb < N ? gb : N - 1;
            voa[i] = (unsigned)ga * (unsigned)lda * 2u + (unsigned)c * 16u;
            vob[i] = (unsigned)gb * (unsigned)ldb * 2u + (unsigned)c * 16u;
        }
        const unsigned lds0 = (unsigned)(uintptr_t)(lds_void_t*)lds + (unsigned)wave * 4096u;
        if (X == 7) {
            unsigned bs = 0;                                     // B slot of tile kt
            for (int kt = 0; kt < nk; ++kt) {
                const unsigned cur_a = (kt & 1) ? kHalf : 0u, nxt_a = kHalf - cur_a;
                const unsigned b1 = bs == 2 ? 0u : bs + 1u, b2 = b1 == 2 ? 0u : b1 + 1u;
                const unsigned cur_b = bs * kHalf, nxt_b = b1 * kHalf, dst_b = b2 * kHalf;
                const int k2 = kt + 2 < nk ? kt + 2 : nk - 1;
                big_slice_dma<false>(acc, f0a, f0b, f1a, f1b, off_a[1] + cur_a, off_b[1] + cur_b, lds0 + 2 * kHalf + dst_b, vob,
                                     (unsigned long long)(uintptr_t)B + (unsigned long long)k2 * (BK * 2));
                big_slice<false>(acc, f1a, f1b, f0a, f0b, off_a[2] + cur_a, off_b[2] + cur_b);
                big_slice<false>(acc, f0a, f0b, f1a, f1b, off_a[3] + cur_a, off_b[3] + cur_b);
                big_slice_dma<true, 4>(acc, f1a, f1b, f0a, f0b, off_a[0] + nxt_a, off_b[0] + nxt_b, lds0 + cur_a, voa,
                                       (unsigned long long)(uintptr_t)A + (unsigned long long)k2 * (BK * 2));
                bs = b1;
            }
        } else
        for (int kt = 0; kt < nk; ++kt) {
            const unsigned cur = (kt & 1) ? kSlotBytes : 0u, nxt = kSlotBytes - cur;
            // (past the end the pieces re-fetch the last tile into a slot nobody reads: no branches in the loop)
            const int kb = kt + 1 < nk ? kt + 1 : nk - 1, ka = kt + 2 < nk ? kt + 2 : nk - 1;
            // B half of tile kt+1 -> the other slot, inside slice 0
            big_slice_dma<false>(acc, f0a, f0b, f1a, f1b, off_a[1] + cur, off_b[1] + cur,
                                 lds0 + nxt + (unsigned)(GM * kChunksPerRow * 16), vob,
                                 (unsigned long long)(uintptr_t)B + (unsigned long long)kb * (BK * 2));
            big_slice<false>(acc, f1a, f1b, f0a, f0b, off_a[2] + cur, off_b[2] + cur);
            big_slice<false>(acc, f0a, f0b, f1a, f1b, off_a[3] + cur, off_b[3] + cur);
            // closes the step; A half of tile kt+2 -> the slot just left, inside the same slice (after its barrier)
            big_slice_dma<true>(acc, f1a, f1b, f0a, f0b, off_a[0] + nxt, off_b[0] + nxt, lds0 + cur, voa,
                                (unsigned long long)(uintptr_t)A + (unsigned long long)ka * (BK * 2));
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else
    for (int kt = 0; kt < nk; ++kt) {
        const unsigned cur = (kt & 1) ? kSlotBytes : 0u, nxt = kSlotBytes - cur;
        uint4* const fill = (kt & 1) ? slot0 : slot1;          // slot of tile kt+1
        big_slice<false>(acc, f0a, f0b, f1a, f1b, off_a[1] + cur, off_b[1] + cur);
        if (X == 5) { if (kt + 1 < nk) dma_rows<32>(B, ldb, N, n0, 0, fill + GM * kChunksPerRow, wave, lane); }
        else if (X != 1 && X != 4 && kt + 1 < nk) dma_rows<32>(B, ldb, N, n0, (kt + 1) * BK, fill + GM * kChunksPerRow, wave, lane);
        big_slice<false>(acc, f1a, f1b, f0a, f0b, off_a[2] + cur, off_b[2] + cur);
        big_slice<false>(acc, f0a, f0b, f1a, f1b, off_a[3] + cur, off_b[3] + cur);
        // closes the step: tile kt+1 complete and visible, slot of tile kt free; prefetch slice 0 of tile kt+1
        big_slice<true, X>(acc, f1a, f1b, f0a, f0b, off_a[0] + nxt, off_b[0] + nxt);
        if (X == 5) { if (kt + 2 < nk) dma_rows<32>(A, lda, M, m0, 0, (kt & 1) ? slot1 : slot0, wave, lane); }
        else if (X != 4 && kt + 2 < nk) dma_rows<32>(A, lda, M, m0, (kt + 2) * BK, (kt & 1) ? slot1 : slot0, wave, lane);
        if (X == 1 && kt + 2 < nk) dma_rows<32>(B, ldb, N, n0, (kt + 2) * BK, ((kt & 1) ? slot1 : slot0) + GM * kChunksPerRow, wave, lane);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the last (unused) prefetch must not outlive its registers
    __builtin_amdgcn_s_barrier();                           // every wave is done with the operand tiles
    const int mw = m0 + wm * 128;
    band_store<OUT_BF16, 4>(acc, Cv, ldc, N, n0 + wn * 64, wave, lane, ep, reinterpret_cast<char*>(lds),
                            [&](int r) -> long long { return mw + r < M ? (long long)(mw + r) : -1ll; });
}

// ---- implicit-GEMM 3x3 convolution on the same tile ------------------------------------------
// Backbone convolutions (modeling/backbone/vgg16.py:58-83: 3x3, stride 1, padding = dilation in
// {1,2}) as C[m][co] = sum_{tap,ci} X[m + shift(tap)][ci] * Wk[co][tap*C + ci] with NHWC bf16
// activations: row m = (b,h,w) of the output, K = 9*C walks (tap, ci).  Nothing is materialised: the
// LDS-DMA source address of each 16-byte chunk (8 channels of one tap) is computed per lane, taps
// that fall into the zero padding (or past K) read a zero page.  sign = -1 mirrors the taps (input
// gradient: the same kernel on dZ with the [ci][tap][co] weight copy).  C must be a power of two >= 8.
struct ConvGeom {
    int H, W, C, logC, dil, sign;
    const unsigned short* zero;     // >= 16 bytes of zeros in global memory
};

__device__ __forceinline__ void dma_tile_conv(const unsigned short* __restrict__ X, const ConvGeom& g, int k0,
                                              uint4* __restrict__ tile, int wave, int lane, const int (&rm)[4],
                                              const int (&rh)[4], const int (&rw)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rbase = wave * 32 + i * 8;
        const int row = rbase + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        const int k = k0 + c * 8;
        const int tap = k >> g.logC, ci = k & (g.C - 1);
        const int ty = (tap * 11) >> 5;             // tap / 3 for tap in [0, 15]
        const int tx = tap - 3 * ty;
        const int dh = (ty - 1) * g.dil * g.sign, dw = (tx - 1) * g.dil * g.sign;
        const int y = rh[i] + dh, x = rw[i] + dw;
        const bool ok = rm[i] >= 0 && tap < 9 && (unsigned)y < (unsigned)g.H && (unsigned)x < (unsigned)g.W;
        const unsigned short* src = ok ? X + ((size_t)(rm[i] + dh * g.W + dw) << g.logC) + ci : g.zero;
        __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(tile + rbase * kChunksPerRow), 16, 0, 0);
    }
}

// The same operand DMA when a K step of 64 lies inside ONE tap (C >= 64): the tap decode, its shift and the channel
// offset are wave-uniform, and a lane's four rows carry a 9-bit "tap in bounds" mask and a byte offset computed once
// before the K loop -- ~7 VALU instructions per row and step instead of ~30.  Measured: 2-4 % per layer (831 -> 812 us
// for the 13 forward layers): the arithmetic was not the bound.  One workgroup alone on a CU runs 2.06 TF = one K step
// (2.1 MFLOP) per ~1 us, i.e. per loaded DMA latency: with 64 KB of operands in flight per CU (2 workgroups x 1 stage)
// the 128x128 tile is latency-bound, and a third stage only trades a co-resident workgroup for it (the 3-slot ring
// form measured the same); larger tiles do not fit the 76x76 layers' 184-tile grids.
__device__ __forceinline__ void dma_tile_conv_uniform(const unsigned short* __restrict__ X, const ConvGeom& g, int k0,
                                                      uint4* __restrict__ tile, int wave, const unsigned (&voff)[4],
                                                      const unsigned (&vmask)[4]) {
    const int tap = __builtin_amdgcn_readfirstlane(k0 >> g.logC);
    const int ci0 = k0 & (g.C - 1);
    const int ty = (tap * 11) >> 5, tx = tap - 3 * ty;
    const int dh = (ty - 1) * g.dil * g.sign, dw = (tx - 1) * g.dil * g.sign;
    const int delta = (((dh * g.W + dw) << g.logC) + ci0) * 2;          // bytes, wave-uniform (may be negative)
    const char* base = reinterpret_cast<const char*>(X) + delta;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const bool ok = (vmask[i] >> tap) & 1u;
        const void* src = ok ? static_cast<const void*>(base + voff[i]) : static_cast<const void*>(g.zero);
        __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(tile + (wave * 32 + i * 8) * kChunksPerRow), 16, 0, 0);
    }
}

template <bool OUT_BF16>
__global__ __launch_bounds__(kThreads, 2) void conv3x3_glds_kernel(
    const unsigned short* __restrict__ X, ConvGeom g, const unsigned short* __restrict__ B, int ldb, int M, int N,
    void* __restrict__ Cv, int ldc, Epilogue ep, int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) uint4 lds[];
    const int nblk = tiles_m * tiles_n;
    int tile;
    {
        const int b = blockIdx.x;
        const int q = nblk / 8, r = nblk % 8, xcd = b % 8, j = b / 8;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    // consecutive tiles walk N first: the (few) weight bands of one output row-block share its activations
    const int tm = tile / tiles_n, tn = tile % tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int half = lane >> 5, l31 = lane & 31;
    int K = 9 * g.C, kbase = 0;
    if (ep.kchunk > 0) {              // split-K: this workgroup's K range, fp32 partial into the workspace
        kbase = blockIdx.y * ep.kchunk;
        B += kbase;
        K = K - kbase < ep.kchunk ? K - kbase : ep.kchunk;
        Cv = reinterpret_cast<char*>(Cv) + (long long)blockIdx.y * ep.split_stride;
    }

    int rm[4], rh[4], rw[4];          // this lane's 4 staging rows: pixel index, y, x
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wave * 32 + i * 8 + (lane >> 3);
        const int hw = g.H * g.W;
        const int p = m % hw;
        rm[i] = m < M ? m : -1;
        rh[i] = p / g.W;
        rw[i] = p - rh[i] * g.W;
    }

    // C >= 64: every K step of 64 stays inside one tap (and K, kbase are multiples of 64)
    const bool uni = g.logC >= 6 && (K & (BK - 1)) == 0;
    unsigned voff[4], vmask[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = wave * 32 + i * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        voff[i] = (((unsigned)(rm[i] < 0 ? 0 : rm[i]) << g.logC) + (unsigned)c * 8u) * 2u;
        unsigned mk = 0;
        if (rm[i] >= 0) {
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int y = rh[i] + (t / 3 - 1) * g.dil * g.sign, x = rw[i] + (t % 3 - 1) * g.dil * g.sign;
                if ((unsigned)y < (unsigned)g.H && (unsigned)x < (unsigned)g.W) mk |= 1u << t;
            }
        }
        vmask[i] = mk;
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x16){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

    const int nk = (K + BK - 1) / BK;
    if (uni) dma_tile_conv_uniform(X, g, kbase, lds, wave, voff, vmask);
    else dma_tile_conv(X, g, kbase, lds, wave, lane, rm, rh, rw);
    dma_tile(B, ldb, N, n0, 0, lds + kTileChunks, wave, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int stage = kt & 1;
        uint4* sa = lds + (size_t)stage * 2 * kTileChunks;
        uint4* sb = sa + kTileChunks;
        if (kt + 1 < nk) {
            uint4* na = lds + (size_t)(stage ^ 1) * 2 * kTileChunks;
            if (uni) dma_tile_conv_uniform(X, g, kbase + (kt + 1) * BK, na, wave, voff, vmask);
            else dma_tile_conv(X, g, kbase + (kt + 1) * BK, na, wave, lane, rm, rh, rw);
            dma_tile(B, ldb, N, n0, (kt + 1) * BK, na + kTileChunks, wave, lane);
        }
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            const int c = kk * 2 + half;
            bf16x8 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = __builtin_bit_cast(bf16x8, sa[lds_slot(wm * 64 + i * 32 + l31, c)]);
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[j] = __builtin_bit_cast(bf16x8, sb[lds_slot(wn * 64 + j * 32 + l31, c)]);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    store_tile_out<OUT_BF16>(acc, Cv, ldc, M, N, m0, n0, wm, wn, half, l31, ep);
}

// ---- halo-tile implicit GEMM: the activation patch is fetched ONCE per channel block, not once per tap ---------------
// What bounds conv3x3_glds_kernel (profiles/r02/pmc_conv_*.txt: MFMA busy 19.5 %, no LDS conflicts, L2 hits 86-92 %) is
// the L2 -> LDS byte rate a CU sustains (~40 GB/s per CU in every DMA kernel of this file), and an implicit GEMM whose
// K walks (tap, channel) pulls each activation row through it nine times -- once per tap -- next to the weights.
// Here a workgroup owns a 16 x 16 SPATIAL tile of output pixels x 128 output channels.  Per 64-channel block it stages
// the (16 + 2 dil)^2 halo patch of the input in LDS once (41 / 51 KB, double buffered, padding and out-of-image pixels
// read a zero page), and the nine taps of that block are nine K steps whose A fragments are ds_read_b128 of the SAME
// patch at a wave-uniform row shift (dy * pitch + dx); only the 128 x 64 weight tile of each (block, tap) streams,
// through a three-slot ring with counted vmcnt (the ring GEMM's pipeline).  Bytes per K step of 256 x 128 x 64:
// 16 KB of weights + 1/9 of the patch = 20.6 KB against 48 KB (256x128 ring) / 64 KB (two 128x128 tiles).
// LDS rows are one pixel's 64 channels (128 B); chunk c of halo row r lives in slot c ^ ((r >> 1) & 7), so the 16 lanes
// of a ds_read_b128 group -- 16 horizontally adjacent pixels = 16 consecutive halo rows at ANY shift -- hit 16
// different 16-byte bank groups.  C (the channel count of X) must be a multiple of 64.
constexpr int HT = 16;                                        // tile side (pixels)
constexpr int kHaloThreads = 512;
constexpr int kHaloBStage = RN * kChunksPerRow;               // 1024 uint4 = 16 KB: one (block, tap) weight tile

// MFMA row r (0..31) of a 32-pixel fragment -> (tile row 0 / 1 of the pair, x 0..15).  ds_read_b128 is serviced in the
// lane groups {0-3, 12-15, 20-27} and {4-11, 16-19, 28-31} (not 0-15 / 16-31): the mapping gives each group 16
// horizontally adjacent pixels = 16 consecutive halo rows, which the slot swizzle makes conflict-free at any tap shift
// (the plain r -> (r >> 4, r & 15) mapping measured SQ_LDS_BANK_CONFLICT = 28 % of the LDS cycles).
__device__ __forceinline__ int halo_row_of(int r) { return r < 4 ? 0 : r < 12 ? 1 : r < 16 ? 0 : r < 20 ? 1 : r < 28 ? 0 : 1; }
__device__ __forceinline__ int halo_x_of(int r) { return r < 4 ? r : r < 12 ? r - 4 : r < 20 ? r - 8 : r < 28 ? r - 12 : r - 16; }

template <int DIL>
struct Halo {
    static constexpr int HWp = HT + 2 * DIL;                  // patch pitch (pixels)
    static constexpr int kRows = HWp * HWp;                   // 324 / 400 halo rows of 128 B
    static constexpr int NA = (kRows + 63) / 64;              // DMA instructions per wave per patch (8 rows each, 8 waves)
    static constexpr int kBufChunks = NA * 64 * kChunksPerRow;
    static constexpr size_t kLdsBytes = ((size_t)2 * kBufChunks + 3 * kHaloBStage) * sizeof(uint4);   // 144 / 160 KB
};

// N64 (round 4): a layer with exactly 64 output channels (conv1_2) -- the 8 waves take 32 pixels x 64 channels each
// (8 x 1 instead of 4 x 2 waves of 64 x 64): the 4 x 2 form ran its second channel half on clamped weight rows, i.e. half of
// its MFMAs for nothing (conv1_2 was the slowest forward layer of the body at 0.18 of the peak).
template <bool OUT_BF16, int DIL, int DBG = 0, bool N64 = false>
__global__ __launch_bounds__(kHaloThreads, 2) void conv3x3_halo_kernel(
    const unsigned short* __restrict__ X, ConvGeom g, const unsigned short* __restrict__ B, int ldb, int n_img, int N,
    void* __restrict__ Cv, int ldc, Epilogue ep, int tiles_y, int tiles_x, int tiles_n, int splits, int cb_per_split) {
    using HC = Halo<DIL>;
    extern __shared__ __attribute__((aligned(16))) uint4 lds[];     // [patch 0 | patch 1 | weight ring of 3]
    uint4* const bring = lds + 2 * HC::kBufChunks;
    // workgroup -> (spatial tile, channel tile, K slice).  Workgroup b runs on XCD b % 8 and each XCD owns a contiguous
    // chunk of the order below, in which the spatial tiles of ONE (channel tile, slice) are consecutive: the workgroups
    // resident on an XCD stream the same weight tiles through its L2 together.
    const int nsp = n_img * tiles_y * tiles_x;
    const int nblk = nsp * tiles_n * splits;
    int t;
    {
        const int b = blockIdx.x;
        const int q = nblk / 8, r = nblk % 8, xcd = b % 8, j = b / 8;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int combo = t / nsp, sp = t - combo * nsp;
    const int tn = combo % tiles_n, split = combo / tiles_n;
    const int img = sp / (tiles_y * tiles_x), rem = sp - img * (tiles_y * tiles_x);
    const int y0 = (rem / tiles_x) * HT, x0 = (rem % tiles_x) * HT, n0 = tn * RN;
    const int cb0 = split * cb_per_split;
    const int ncb_all = g.C >> 6;
    const int cb1 = cb0 + cb_per_split < ncb_all ? cb0 + cb_per_split : ncb_all;
    if (splits > 1) Cv = reinterpret_cast<char*>(Cv) + (long long)split * ep.split_stride;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int MI = N64 ? 1 : 2;                     // 32-pixel fragments per wave
    const int wm = N64 ? wave : wave >> 1, wn = N64 ? 0 : wave & 1;     // 4 x 2 waves of 64 pixels x 64 channels (N64: 8 x 1 of 32 x 64)
    const int half = lane >> 5, l31 = lane & 31;

    // this lane's share of a patch: byte offset of (pixel, 16-byte chunk) in X for channel block 0; ~0 = zero page
    unsigned voff[HC::NA];
#pragma unroll
    for (int i = 0; i < HC::NA; ++i) {
        const int hr = (i * 8 + wave) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((hr >> 1) & 7);
        const int hy = hr / HC::HWp, hx = hr - hy * HC::HWp;
        const int y = y0 - DIL + hy, x = x0 - DIL + hx;
        const bool ok = hr < HC::kRows && (unsigned)y < (unsigned)g.H && (unsigned)x < (unsigned)g.W;
        voff[i] = ok ? ((unsigned)((img * g.H + y) * g.W + x) * (unsigned)g.C) * 2u + (unsigned)c * 16u : 0xffffffffu;   // (C: any multiple of 64)
    }
    auto dma_patch = [&](int cb, uint4* buf) {
        const char* base = reinterpret_cast<const char*>(X) + (size_t)cb * 128;
#pragma unroll
        for (int i = 0; i < HC::NA; ++i) {
            const void* src = voff[i] != 0xffffffffu ? static_cast<const void*>(base + voff[i])
                                                     : static_cast<const void*>(g.zero);
            __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(buf + (i * 8 + wave) * 64), 16, 0, 0);
        }
    };
    // halo row of this lane's two fragment rows at the centre tap
    int hrb[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        hrb[i] = (wm * (2 * MI) + i * 2 + halo_row_of(l31) + DIL) * HC::HWp + halo_x_of(l31) + DIL;
    }

    f32x16 acc[MI][2];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x16){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

    const int nsteps = (cb1 - cb0) * 9;
    // prologue: patch of the first block, weight tiles of steps 0 and 1
    dma_patch(cb0, lds);
    dma_rows<16>(B, ldb, N, n0, cb0 * 64, bring, wave, lane);
    if (nsteps > 1) dma_rows<16>(B, ldb, N, n0, g.C + cb0 * 64, bring + kHaloBStage, wave, lane);
    int tap = 0, cb = cb0, slot = 0;                     // step t
    int tap2 = 2, cb2 = cb0;                             // step t + 2 (the weight tile issued during step t)
    for (int st = 0; st < nsteps; ++st) {
        // Issue order of a step: weight tile t+2 (2 pieces per wave), then -- at tap 0 -- the next block's patch (NA
        // pieces).  Tile t has landed when at most the pieces issued after it are outstanding.
        if (st + 2 >= nsteps) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if ((tap == 1 || tap == 2) && cb + 1 < cb1) {
            if (HC::NA == 6) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
        } else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (DBG != 1 && DBG != 5 && st + 2 < nsteps) {
            const int s2 = slot + 2 >= 3 ? slot - 1 : slot + 2;
            dma_rows<16>(B, ldb, N, n0, tap2 * g.C + cb2 * 64, bring + s2 * kHaloBStage, wave, lane);
        }
        if (DBG != 1 && DBG != 6 && tap == 0 && cb + 1 < cb1) dma_patch(cb + 1, lds + (((cb - cb0) + 1) & 1) * HC::kBufChunks);
        const uint4* sa = lds + ((cb - cb0) & 1) * HC::kBufChunks;
        const uint4* sb = bring + slot * kHaloBStage;
        const int ty_ = (tap * 11) >> 5, tx_ = tap - 3 * ty_;
        const int delta = ((ty_ - 1) * HC::HWp + (tx_ - 1)) * DIL * g.sign;
        int arow[MI], asw[MI];
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int hr = hrb[i] + delta;
            arow[i] = hr * kChunksPerRow;
            asw[i] = (hr >> 1) & 7;
        }
        if (DBG == 0 || DBG == 8) {
            // the step's 16 fragment reads and 16 MFMAs with the issue order given to the scheduler explicitly: left
            // alone hipcc emits 4 reads -> s_waitcnt lgkmcnt(0) -> 4 MFMAs per slice (1-2 % slower)
            bf16x8 ga[4][MI], gb[4][2];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int c = kk * 2 + half;
#pragma unroll
                for (int i = 0; i < MI; ++i) ga[kk][i] = __builtin_bit_cast(bf16x8, sa[arow[i] + (c ^ asw[i])]);
#pragma unroll
                for (int j = 0; j < 2; ++j) gb[kk][j] = __builtin_bit_cast(bf16x8, sb[lds_slot(wn * 64 + j * 32 + l31, c)]);
            }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gb[kk][j], ga[kk][i], acc[i][j], 0, 0, 0);
            // issue order for the scheduler: the first slice's reads, then one read of the NEXT slice behind each MFMA
            if (N64) {                  // 12 reads, 8 MFMAs
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
            } else {                    // 16 reads, 16 MFMAs
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
                for (int t = 0; t < 12; ++t) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            }
        } else
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            const int c = kk * 2 + half;
            bf16x8 fa[MI], fb[2];
            if (DBG == 2) {             // timing experiment: no LDS reads
#pragma unroll
                for (int i = 0; i < MI; ++i) fa[i] = __builtin_bit_cast(bf16x8, make_uint4(arow[i], c, asw[i], st));
#pragma unroll
                for (int j = 0; j < 2; ++j) fb[j] = __builtin_bit_cast(bf16x8, make_uint4(j, c, lane, st));
            } else {
#pragma unroll
            for (int i = 0; i < MI; ++i) fa[i] = __builtin_bit_cast(bf16x8, sa[arow[i] + (c ^ asw[i])]);
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[j] = __builtin_bit_cast(bf16x8, sb[lds_slot(wn * 64 + j * 32 + l31, c)]);
            }
            if (DBG == 3) {             // timing experiment: no MFMAs (the fragments are still consumed)
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const uint4 ua = __builtin_bit_cast(uint4, fa[i]), ub = __builtin_bit_cast(uint4, fb[j]);
                        acc[i][j][0] += __uint_as_float((ua.x ^ ub.y) & 0x3fffffffu);
                        acc[i][j][1] += __uint_as_float((ua.z ^ ub.w) & 0x3fffffffu);
                    }
            } else {
            // operands swapped: the accumulator holds the TRANSPOSED 32x32 tile (lane & 31 = pixel, 4 consecutive
            // registers = 4 consecutive channels), which is what the staged epilogue wants
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
            }
        }
        slot = slot == 2 ? 0 : slot + 1;
        if (++tap == 9) { tap = 0; ++cb; }
        if (++tap2 == 9) { tap2 = 0; ++cb2; }
    }

    // ---- epilogue: rows of the tile are pixels (ty, tx); the ones past the image edge are not stored.
    // Measured (ODW_HALO_DBG=4, profiles/r02/conv_halo_experiments.txt): with 2-byte stores straight from the
    // accumulators the epilogue was HALF of the kernel on the wide maps (conv1_2: 133 us with, 60 us without); it now
    // goes through LDS in 32-pixel bands and leaves as 16-byte vectors (band_store).
    if (DBG == 4 && acc[0][0][0] != 12345.0f) return;        // timing experiment: no stores
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                            // every wave is done with the operand tiles
    const int hw = g.H * g.W;
    band_store<OUT_BF16, MI>(acc, Cv, ldc, N, n0 + wn * 64, wave, lane, ep, reinterpret_cast<char*>(lds),
                            [&](int r) -> long long {
                                const int prow = wm * (32 * MI) + r;
                                const int y = y0 + wm * (2 * MI) + (r >> 5) * 2 + halo_row_of(r & 31), x = x0 + halo_x_of(r & 31);
                                if (DBG == 7) return (y < g.H && x < g.W) ? (long long)(prow + 256 * (blockIdx.x & 15)) : -1ll;
                                return (y < g.H && x < g.W) ? (long long)img * hw + (long long)y * g.W + x : -1ll;
                            });
}

// ---- round 5: the halo-tile convolution of the "bf16x2f" FORWARD on TWO stored planes ----------------------------------
// The forward of the timed mode is the sum of three bf16 plane products per layer, x_hi w_hi + x_hi w_mid + x_mid w_hi
// (od_wscl_amd/precision.py).  Rounds 3-4 ran them as THREE passes of the kernel above over an operand stored as the
// channel blocks [hi | hi | mid] against weights [hi | mid | hi]: the hi patch of every 64-channel block was staged
// twice, the hi weight tile of every (block, tap) streamed twice, 6 bytes per activation element stored, and a K step
// (one barrier, one weight tile, 16 fragment reads) fed 16 MFMAs per wave.
// Here the operand is stored as the two planes it consists of, [hi (C) | mid (C)] per pixel, and a K step covers 32
// CHANNELS of both: the patch row of a pixel in LDS is [hi 32 | mid 32] (128 B: the DMA gathers two 64-byte pieces),
// the weight tile row of (output channel, tap, block) is [hi 32 | mid 32] as well (packed that way by
// weight_prep_batch_kernel, T = -2), and the step's 16 fragment reads feed the 24 MFMAs of ALL THREE products:
//     chunk pairs  a0 a1 = x_hi, a2 a3 = x_mid;  b0 b1 = w_hi, b2 b3 = w_mid
//     acc += b0 a0 + b1 a1  (hi hi)  + b2 a0 + b3 a1  (x_hi w_mid)  + b0 a2 + b1 a3  (x_mid w_hi)
// Two thirds of the K steps (barriers, weight DMA pieces, patch stagings) of the three-pass form for the same MFMAs,
// 4 instead of 6 bytes per stored activation, and the hi plane -- the first C columns of a pixel row -- is what the
// single-plane backward reads (weight gradient operand, ReLU mask), in place.
// OUTM: 0 = fp32 rows (ldc floats; a pooled layer's pre-pool activation, the feature map, split-K partials),
//       1 = the next layer's operand itself: bf16 planes [hi (N) | mid (N)] per pixel, ldc bf16 elements per row --
//           the epilogue splits in registers (odw_planes.h: split2), no fp32 activation, no split_rows pass.
template <int NI, class RowMap>
__device__ __forceinline__ void band_store_planes2(const f32x16 (&acc)[NI][2], unsigned short* __restrict__ Cv, int ldc, int N,
                                                   int nw, int wave, int lane, const Epilogue& ep, char* lds, RowMap rowmap) {
    constexpr int kRowBytes = 256 + 16;                 // [hi 64 x 2 B | mid 64 x 2 B] + pad: conflict-free b64 writes
    char* region = lds + wave * (32 * kRowBytes);
    const int half = lane >> 5, l31 = lane & 31;
    float* const bias_s = reinterpret_cast<float*>(lds + 8 * 32 * kRowBytes) + wave * 64;
    {
        const int n = nw + lane;
        bias_s[lane] = ep.bias ? ep.bias[n < N ? n : N - 1] : 0.0f;
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cw = j * 32 + 8 * g + 4 * half;
                const float4 b4 = *reinterpret_cast<const float4*>(bias_s + cw);
                float v[4] = {acc[i][j][4 * g] + b4.x, acc[i][j][4 * g + 1] + b4.y, acc[i][j][4 * g + 2] + b4.z,
                              acc[i][j][4 * g + 3] + b4.w};
                if (ep.relu) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.0f);
                }
                unsigned h0, m0, h1, m1, lo_;
                odwpl::split2(v[0], v[1], false, h0, m0, lo_);
                odwpl::split2(v[2], v[3], false, h1, m1, lo_);
                *reinterpret_cast<uint2*>(region + l31 * kRowBytes + cw * 2) = make_uint2(h0, h1);
                *reinterpret_cast<uint2*>(region + l31 * kRowBytes + 128 + cw * 2) = make_uint2(m0, m1);
            }
        }
        // read the band back row-major: 16 lanes per pixel row (8 x 16 B of the hi plane, 8 x 16 B of the mid plane)
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int r = t * 4 + (lane >> 4), cchunk = lane & 15;
            const uint4 d = *reinterpret_cast<const uint4*>(region + r * kRowBytes + cchunk * 16);
            const long long gm = rowmap(i * 32 + r);
            const int gn = nw + (cchunk & 7) * 8;
            if (gm < 0 || gn >= N) continue;
            *reinterpret_cast<uint4*>(Cv + (size_t)gm * ldc + (cchunk >> 3) * (ldc >> 1) + gn) = d;      // mid plane: ldc / 2 further
        }
    }
}

// OUTM = 2: bias + ReLU, the 2 x 2 / 2 max pool AND the split in the epilogue -- a pooled layer whose pre-pool activation nobody
// reads again (the frozen conv1_2 / conv2_2 of VGG16: no backward) writes the next layer's operand at a quarter of the pixels
// and no fp32 activation at all (94.6 MB at 608^2 x 64, written and read back by the pooling pass before).  A wave's 32-pixel
// fragment is two tile rows x 16 pixels (halo_row_of / halo_x_of) = eight complete 2 x 2 windows: the fragment is staged in the
// wave's LDS region like the fp32 epilogue's, then lane = (window q, 8 channels) takes the maximum of its four pixels, splits it
// and stores 16 bytes of each plane.  poolmap(i, q) = pooled pixel row of C for window q of fragment i, or -1.
template <int NI, class PoolMap>
__device__ __forceinline__ void band_store_pool_planes2(const f32x16 (&acc)[NI][2], unsigned short* __restrict__ Cv, int ldc, int N,
                                                        int nw, int wave, int lane, const Epilogue& ep, char* lds, PoolMap poolmap) {
    constexpr int kRowBytes = 256 + 16;
    char* region = lds + wave * (32 * kRowBytes);
    const int half = lane >> 5, l31 = lane & 31;
    float* const bias_s = reinterpret_cast<float*>(lds + 8 * 32 * kRowBytes) + wave * 64;
    {
        const int n = nw + lane;
        bias_s[lane] = ep.bias ? ep.bias[n < N ? n : N - 1] : 0.0f;
    }
    // fragment row of pixel (tile row 0 / 1, x): the inverse of halo_row_of / halo_x_of
    const int q = lane >> 3, c8 = (lane & 7) * 8;
    const int xa = 2 * q, xb = 2 * q + 1;
    auto r_top = [](int x) { return x < 4 ? x : x < 8 ? x + 8 : x + 12; };
    auto r_bot = [](int x) { return x < 8 ? x + 4 : x < 12 ? x + 8 : x + 16; };
    const int rr[4] = {r_top(xa), r_top(xb), r_bot(xa), r_bot(xb)};
#pragma unroll
    for (int i = 0; i < NI; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cw = j * 32 + 8 * g + 4 * half;
                const float4 b4 = *reinterpret_cast<const float4*>(bias_s + cw);
                float v[4] = {acc[i][j][4 * g] + b4.x, acc[i][j][4 * g + 1] + b4.y, acc[i][j][4 * g + 2] + b4.z,
                              acc[i][j][4 * g + 3] + b4.w};
                if (ep.relu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.0f);
                }
                *reinterpret_cast<float4*>(region + l31 * kRowBytes + cw * 4) = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
        float m[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float4 a = *reinterpret_cast<const float4*>(region + rr[k] * kRowBytes + c8 * 4);
            const float4 b = *reinterpret_cast<const float4*>(region + rr[k] * kRowBytes + c8 * 4 + 16);
            const float e[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
            for (int u = 0; u < 8; ++u) m[u] = (k == 0 || e[u] > m[u]) ? e[u] : m[u];      // maxpool_f32_planes2_kernel's comparisons
        }
        unsigned h[4], md[4], lo_;
#pragma unroll
        for (int u = 0; u < 4; ++u) odwpl::split2(m[2 * u], m[2 * u + 1], false, h[u], md[u], lo_);
        const long long gm = poolmap(i, q);
        const int gn = nw + c8;
        if (gm >= 0 && gn < N) {
            unsigned short* dst = Cv + (size_t)gm * ldc + gn;
            *reinterpret_cast<uint4*>(dst) = make_uint4(h[0], h[1], h[2], h[3]);
            *reinterpret_cast<uint4*>(dst + (ldc >> 1)) = make_uint4(md[0], md[1], md[2], md[3]);
        }
    }
}

// PAIR (dilation 1, end of round 5): TWO K steps per workgroup barrier on a ring of four weight slots.  The barrier of a K step
// is where a one-workgroup-per-CU kernel loses its matrix pipe (all eight waves meet, then all eight start on LDS reads); with the
// tiles of steps 2m and 2m + 1 both landed at the barrier of pair m, the second step of a pair starts behind the first without
// one, and the tiles of the next pair (and, once per channel block, the next patch) are issued for a whole pair of steps.
// 2 x 48 KB of patches + 4 x 16 KB of weights = the 160 KB of a CU (dilation 2's patches leave no room for the fourth slot).
template <int OUTM, int DIL, bool N64 = false, bool PAIR = false>
__global__ __launch_bounds__(kHaloThreads, 2) void conv3x3_halo2_kernel(
    const unsigned short* __restrict__ X, int ldx, ConvGeom g, const unsigned short* __restrict__ B, int ldb, int n_img, int N,
    void* __restrict__ Cv, int ldc, Epilogue ep, int tiles_y, int tiles_x, int tiles_n, int splits, int cb_per_split) {
    using HC = Halo<DIL>;
    extern __shared__ __attribute__((aligned(16))) uint4 lds[];     // [patch 0 | patch 1 | weight ring of 3]
    uint4* const bring = lds + 2 * HC::kBufChunks;
    const int nsp = n_img * tiles_y * tiles_x;
    const int nblk = nsp * tiles_n * splits;
    int t;
    {
        const int b = blockIdx.x;
        const int q = nblk / 8, r = nblk % 8, xcd = b % 8, j = b / 8;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int combo = t / nsp, sp = t - combo * nsp;
    const int tn = combo % tiles_n, split = combo / tiles_n;
    const int img = sp / (tiles_y * tiles_x), rem = sp - img * (tiles_y * tiles_x);
    const int y0 = (rem / tiles_x) * HT, x0 = (rem % tiles_x) * HT, n0 = tn * RN;
    const int cb0 = split * cb_per_split;
    const int ncb_all = g.C >> 5;                         // blocks of 32 channels (g.C = channels of ONE plane)
    const int cb1 = cb0 + cb_per_split < ncb_all ? cb0 + cb_per_split : ncb_all;
    if (splits > 1) Cv = reinterpret_cast<char*>(Cv) + (long long)split * ep.split_stride;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int MI = N64 ? 1 : 2;
    const int wm = N64 ? wave : wave >> 1, wn = N64 ? 0 : wave & 1;
    const int half = lane >> 5, l31 = lane & 31;

    // this lane's share of a patch: byte offset of (pixel, 16-byte chunk) in X for channel block 0; ~0 = zero page.
    // LDS chunk c of a pixel row: c < 4 = channels 8c .. 8c+7 of the hi plane, c >= 4 = the same of the mid plane
    unsigned voff[HC::NA];
#pragma unroll
    for (int i = 0; i < HC::NA; ++i) {
        const int hr = (i * 8 + wave) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((hr >> 1) & 7);
        const int hy = hr / HC::HWp, hx = hr - hy * HC::HWp;
        const int y = y0 - DIL + hy, x = x0 - DIL + hx;
        const bool ok = hr < HC::kRows && (unsigned)y < (unsigned)g.H && (unsigned)x < (unsigned)g.W;
        // (offsets inside THIS image: 32 bits; the image's base is added as a 64-bit pointer below -- any batch size)
        voff[i] = ok ? ((unsigned)(y * g.W + x) * (unsigned)ldx) * 2u + (unsigned)(c & 3) * 16u + (c >= 4 ? (unsigned)g.C * 2u : 0u)
                     : 0xffffffffu;
    }
    const char* const ximg = reinterpret_cast<const char*>(X) + (size_t)img * g.H * g.W * (size_t)ldx * 2;
    auto dma_patch = [&](int cb, uint4* buf) {
        const char* base = ximg + (size_t)cb * 64;
#pragma unroll
        for (int i = 0; i < HC::NA; ++i) {
            const void* src = voff[i] != 0xffffffffu ? static_cast<const void*>(base + voff[i])
                                                     : static_cast<const void*>(g.zero);
            __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(buf + (i * 8 + wave) * 64), 16, 0, 0);
        }
    };
    int hrb[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) hrb[i] = (wm * (2 * MI) + i * 2 + halo_row_of(l31) + DIL) * HC::HWp + halo_x_of(l31) + DIL;

    f32x16 acc[MI][2];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x16){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

    const int ktap = 2 * g.C;                            // weight row: per tap the blocks [hi 32 | mid 32] of all channels
    const int nsteps = (cb1 - cb0) * 9;
    dma_patch(cb0, lds);
    dma_rows<16>(B, ldb, N, n0, cb0 * 64, bring, wave, lane);
    if (nsteps > 1) dma_rows<16>(B, ldb, N, n0, ktap + cb0 * 64, bring + kHaloBStage, wave, lane);
    // one K step = (tap, 32-channel block): 16 fragment reads + 24 MFMAs per wave out of patch `cbrel & 1` and weight slot `slot`
    auto k_step = [&](int tap, int cbrel, int slot) {
        const uint4* sa = lds + (cbrel & 1) * HC::kBufChunks;
        const uint4* sb = bring + slot * kHaloBStage;
        const int ty_ = (tap * 11) >> 5, tx_ = tap - 3 * ty_;
        const int delta = ((ty_ - 1) * HC::HWp + (tx_ - 1)) * DIL * g.sign;
        int arow[MI], asw[MI];
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int hr = hrb[i] + delta;
            arow[i] = hr * kChunksPerRow;
            asw[i] = (hr >> 1) & 7;
        }
        bf16x8 ga[4][MI], gb[4][2];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int c = kk * 2 + half;
#pragma unroll
            for (int i = 0; i < MI; ++i) ga[kk][i] = __builtin_bit_cast(bf16x8, sa[arow[i] + (c ^ asw[i])]);
#pragma unroll
            for (int j = 0; j < 2; ++j) gb[kk][j] = __builtin_bit_cast(bf16x8, sb[lds_slot(wn * 64 + j * 32 + l31, c)]);
        }
        // (ka, kb): hi hi over both 16-channel halves, x_hi w_mid, x_mid w_hi -- in the order the fragments arrive
        constexpr int KA[6] = {0, 1, 0, 1, 2, 3}, KB[6] = {0, 1, 2, 3, 0, 1};
#pragma unroll
        for (int p = 0; p < 6; ++p)
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gb[KB[p]][j], ga[KA[p]][i], acc[i][j], 0, 0, 0);
        if (N64) {                  // 12 reads, 12 MFMAs
            __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
#pragma unroll
            for (int u = 0; u < 9; ++u) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
        } else {                    // 16 reads, 24 MFMAs
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
            for (int u = 0; u < 12; ++u) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
        }
    };
    if (PAIR) {
        static_assert(!PAIR || (DIL == 1 && !N64), "the four-slot ring fits beside dilation 1's patches only");
        // tile of step s lives in slot s & 3; pair m = steps (2m, 2m + 1).  Issue order inside a pair: tile 2m + 2, tile 2m + 3
        // (2 pieces per wave each), then -- in the one pair of a channel block whose first step is its tap 0 or 1 -- the NEXT
        // block's patch (NA pieces): its buffer was last read by the previous block's tap 8, which lies in an earlier pair
        // either way (a block starts on an even or an odd step: 9 taps), and it is needed 8-9 steps later.  At the top of a
        // pair everything but a patch issued in the pair before must have landed.
        int tap_a = 0, cb_a = cb0;
        bool patch_prev = false;
        for (int st = 0; st < nsteps; st += 2) {
            if (patch_prev) {
                if (HC::NA == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
            } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            int tap_b = tap_a + 1, cb_b = cb_a;
            if (tap_b == 9) { tap_b = 0; ++cb_b; }
            int tap_c = tap_b + 1, cb_c = cb_b;
            if (tap_c == 9) { tap_c = 0; ++cb_c; }
            int tap_d = tap_c + 1, cb_d = cb_c;
            if (tap_d == 9) { tap_d = 0; ++cb_d; }
            if (st + 2 < nsteps)
                dma_rows<16>(B, ldb, N, n0, tap_c * ktap + cb_c * 64, bring + ((st + 2) & 3) * kHaloBStage, wave, lane);
            if (st + 3 < nsteps)
                dma_rows<16>(B, ldb, N, n0, tap_d * ktap + cb_d * 64, bring + ((st + 3) & 3) * kHaloBStage, wave, lane);
            patch_prev = tap_a <= 1 && cb_a + 1 < cb1;
            if (patch_prev) dma_patch(cb_a + 1, lds + (((cb_a - cb0) + 1) & 1) * HC::kBufChunks);
            k_step(tap_a, cb_a - cb0, st & 3);
            if (st + 1 < nsteps) k_step(tap_b, cb_b - cb0, (st + 1) & 3);
            tap_a = tap_c; cb_a = cb_c;
        }
    } else {
        int tap = 0, cb = cb0, slot = 0;
        int tap2 = 2, cb2 = cb0;
        for (int st = 0; st < nsteps; ++st) {
            if (st + 2 >= nsteps) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if ((tap == 1 || tap == 2) && cb + 1 < cb1) {
                if (HC::NA == 6) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
            } else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (st + 2 < nsteps) {
                const int s2 = slot + 2 >= 3 ? slot - 1 : slot + 2;
                dma_rows<16>(B, ldb, N, n0, tap2 * ktap + cb2 * 64, bring + s2 * kHaloBStage, wave, lane);
            }
            if (tap == 0 && cb + 1 < cb1) dma_patch(cb + 1, lds + (((cb - cb0) + 1) & 1) * HC::kBufChunks);
            k_step(tap, cb - cb0, slot);
            slot = slot == 2 ? 0 : slot + 1;
            if (++tap == 9) { tap = 0; ++cb; }
            if (++tap2 == 9) { tap2 = 0; ++cb2; }
        }
    }

    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const int hw = g.H * g.W;
    auto rowmap = [&](int r) -> long long {
        const int y = y0 + wm * (2 * MI) + (r >> 5) * 2 + halo_row_of(r & 31), x = x0 + halo_x_of(r & 31);
        return (y < g.H && x < g.W) ? (long long)img * hw + (long long)y * g.W + x : -1ll;
    };
    if (OUTM == 2) {
        const int hw2 = (g.H >> 1) * (g.W >> 1);
        band_store_pool_planes2<MI>(acc, reinterpret_cast<unsigned short*>(Cv), ldc, N, n0 + wn * 64, wave, lane, ep,
                                    reinterpret_cast<char*>(lds), [&](int i, int q) -> long long {
                                        const int y = y0 + wm * (2 * MI) + i * 2, x = x0 + 2 * q;       // top-left pixel of the window
                                        return (y + 1 < g.H && x + 1 < g.W)
                                                   ? (long long)img * hw2 + (long long)(y >> 1) * (g.W >> 1) + (x >> 1) : -1ll;
                                    });
    } else if (OUTM == 1)
        band_store_planes2<MI>(acc, reinterpret_cast<unsigned short*>(Cv), ldc, N, n0 + wn * 64, wave, lane, ep,
                               reinterpret_cast<char*>(lds), rowmap);
    else
        band_store<false, MI>(acc, Cv, ldc, N, n0 + wn * 64, wave, lane, ep, reinterpret_cast<char*>(lds), rowmap);
}

// ---- "TN" product on the ring pipeline: C[M,N] = sum_k A[k][m] * B[k][n], BOTH operands K-major ----------------------
// A weight gradient is dW = dZ^T X with dZ (rows x N_out) and X (rows x K_in) stored row-major: the reduction index is
// the ROW of both operands.  The NT kernels above want it contiguous, which costs a transposed copy of each operand
// per product (transpose_bf16_vec, the dZ^T of linear_bwd_prep) and, for a convolution, the 9x transposed im2col.
// gfx950 can read an MFMA fragment out of a K-major LDS tile instead: ds_read_b64_tr_b16 hands lane l the four
// K-consecutive values of column (l & 15) of a [4 k][16 columns] block whose four 32-byte rows the 16 lanes of the group
// address (tools/exp/tr_probe.hip prints the mapping); two of them make the 8 k of a 32x32x16 operand.
// Tile 256 (M) x 128 (N), 8 waves of 64x64, K steps of 64 rows through the same three-slot ring / counted vmcnt as
// gemm_nt_bf16_ring_kernel; LDS rows are K rows (512 B of A, 256 B of B); 16-byte chunk c of row k sits in slot
// c ^ 4 (k & 3): the 32 lanes one transposed read is serviced in touch 4 rows x 64 B = all 64 banks once.
// CONV: B is the implicit im2col of an NHWC activation: column block n0 = (tap, 128 input channels), row k = pixel
// k + shift(tap), out-of-image pixels (and rows past K, both operands) read the zero page.
typedef short v4s16 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4s16 lds_v4s16;

template <int PITCH>
__device__ __forceinline__ bf16x8 tn_frag(const unsigned char* tile, unsigned off) {
    struct { v4s16 lo, hi; } r;
    r.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s16*)(tile + off));
    r.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s16*)(tile + off + 4 * PITCH));
    return __builtin_bit_cast(bf16x8, r);
}

template <bool CONV>
__global__ __launch_bounds__(kRingThreads, 2) void gemm_tn_bf16_ring_kernel(
    const unsigned short* __restrict__ A, int lda, const unsigned short* __restrict__ B, int ldb, int M, int N, int K,
    void* __restrict__ Cv, int ldc, Epilogue ep, int tiles_m, int tiles_n, ConvGeom g) {
    extern __shared__ __attribute__((aligned(16))) uint4 lds[];       // [stage][A 64 x 256 | B 64 x 128]
    int kbase = 0, kend = K;
    if (ep.kchunk > 0) {
        kbase = blockIdx.y * ep.kchunk;
        kend = kbase + ep.kchunk < K ? kbase + ep.kchunk : K;
        Cv = reinterpret_cast<char*>(Cv) + (long long)blockIdx.y * ep.split_stride;
    }
    int tm, tn;
    tile_coords<4>(blockIdx.x, tiles_m, tiles_n, tm, tn, ep.pm);
    const int m0 = tm * RM, n0 = tn * RN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    // ---- DMA shares of this lane (columns fixed for the whole K walk; rows advance by 64 per step)
    int a_row[4], b_row[2];
    unsigned a_col[4], b_col[2];          // element offset of the lane's 16-byte chunk inside an operand row
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = wave * 4 + i;
        a_row[i] = 2 * q + (lane >> 5);
        const int chunk = (lane & 31) ^ (4 * (a_row[i] & 3));
        int col = m0 + chunk * 8;
        a_col[i] = (unsigned)(col + 8 <= M ? col : (M >= 8 ? M - 8 : 0));
    }
    int tap_dh = 0, tap_dw = 0;
    if (CONV) {
        const int tap = n0 >> g.logC;
        const int ty = (tap * 11) >> 5, tx = tap - 3 * ty;
        tap_dh = (ty - 1) * g.dil;
        tap_dw = (tx - 1) * g.dil;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q = wave * 2 + i;
        b_row[i] = 4 * q + (lane >> 4);
        const int chunk = (lane & 15) ^ (4 * (b_row[i] & 3));
        if (CONV) {
            b_col[i] = (unsigned)((n0 & (g.C - 1)) + chunk * 8);
        } else {
            const int col = n0 + chunk * 8;
            b_col[i] = (unsigned)(col + 8 <= N ? col : (N >= 8 ? N - 8 : 0));
        }
    }
    // CONV: (y, x) of this lane's two B rows, kept incrementally (rows advance by 64 pixels per stage: a division per
    // row and stage was ~200 VALU instructions per K step beside 16 MFMAs)
    int b_y[2] = {0, 0}, b_x[2] = {0, 0};
    if (CONV) {
        const int hw = g.H * g.W;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int p = (kbase + b_row[i]) % hw;
            b_y[i] = p / g.W;
            b_x[i] = p - b_y[i] * g.W;
        }
    }
    auto dma_stage = [&](int k0, uint4* st) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int gk = k0 + a_row[i];
            const void* src = gk < kend ? static_cast<const void*>(A + (size_t)gk * lda + a_col[i])
                                        : static_cast<const void*>(g.zero);
            __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(st + (wave * 4 + i) * 64), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int gk = k0 + b_row[i];
            const void* src = g.zero;
            if (gk < kend) {
                if (CONV) {
                    const int y = b_y[i] + tap_dh, x = b_x[i] + tap_dw;
                    if ((unsigned)y < (unsigned)g.H && (unsigned)x < (unsigned)g.W)
                        src = B + ((size_t)(gk + tap_dh * g.W + tap_dw) << g.logC) + b_col[i];
                } else {
                    src = B + (size_t)gk * ldb + b_col[i];
                }
            }
            __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(st + RM * 8 + (wave * 2 + i) * 64), 16, 0, 0);
            if (CONV) {                  // the stages are issued in K order: step this row's pixel by 64
                b_x[i] += BK;
                while (b_x[i] >= g.W) { b_x[i] -= g.W; ++b_y[i]; }
                while (b_y[i] >= g.H) b_y[i] -= g.H;
            }
        }
    };

    // ---- fragment addressing (bytes inside a stage): row of the K slice, swizzled chunk, half chunk
    const unsigned kq = 8u * (lane >> 5) + ((lane & 15) >> 2);
    const unsigned swz = 4u * ((lane >> 2) & 3);
    const unsigned sub = 8u * (lane & 1);
    const unsigned cbase = 2u * ((lane >> 4) & 1) + ((lane & 3) >> 1);
    unsigned offa[2], offb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        offa[i] = kq * 512u + (((unsigned)(wm * 8 + i * 4) + cbase) ^ swz) * 16u + sub;
        offb[i] = (unsigned)(RM * 8 * 16) + kq * 256u + (((unsigned)(wn * 8 + i * 4) + cbase) ^ swz) * 16u + sub;
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x16){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

    const int nk = (kend - kbase + BK - 1) / BK;
    dma_stage(kbase, lds);
    if (nk > 1) dma_stage(kbase + BK, lds + kRingStageChunks);
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + 2 < nk) dma_stage(kbase + (kt + 2) * BK, lds + (size_t)((kt + 2) % kRingStages) * kRingStageChunks);
        const unsigned char* st = reinterpret_cast<const unsigned char*>(lds + (size_t)(kt % kRingStages) * kRingStageChunks);
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            bf16x8 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = tn_frag<512>(st, offa[i] + kk * 16 * 512);
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[j] = tn_frag<256>(st, offb[j] + kk * 16 * 256);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        }
        // issue order: the first slice's 8 transposed reads, then two reads of the next slice behind each MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
        for (int t = 0; t < 12; ++t) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
    }
    const int mw = m0 + wm * 64;
    auto rowmap = [&](int r) -> long long { return mw + r < M ? (long long)(mw + r) : -1ll; };
    if (band_store_ok(Cv, ldc, N, 4)) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        band_store<false, 2>(acc, Cv, ldc, N, n0 + wn * 64, wave, lane, ep, reinterpret_cast<char*>(lds), rowmap);
    } else {
        band_store_scalar<false, 2>(acc, Cv, ldc, N, n0 + wn * 64, lane, ep, rowmap);
    }
}

// ---- layout helpers ---------------------------------------------------------------------
// out[c][r] = bf16(in[r][c]); in is fp32 or bf16 (IN_F32).  32x32 tiles through LDS.
// Device-resident row count of the transposes / backward prologues below (round 6): R / M of the launch is the CAPACITY the
// grid covers; *r_dev rows exist; the transposed output is zero padded up to r64(*r_dev) columns and starts *col_off_dev
// columns into `out` (the column block of a weight-gradient batch whose predecessors have device-resident widths too);
// src_rows: row r of the input is in[src_rows[r]] (a fused gather).  All three null = the static form.
constexpr unsigned kDynRowTiles = 16;      // 64-row tiles a dynamic launch puts in its grid (the rest: the kernels' tile loop)
struct DynRows {
    const int* r_dev;
    const int* col_off_dev;
    const int* src_rows;
};
#define ODW_DYNROWS_ENTER(ROWS, OUT_COLS, OUT_PTR, ROW0)                                    \
    if (dyn.r_dev) {                                                                       \
        const int rd_ = *dyn.r_dev;                                                        \
        ROWS = rd_ < ROWS ? rd_ : ROWS;                                                    \
        OUT_COLS = (ROWS + 63) / 64 * 64;                                                  \
        if ((ROW0) >= OUT_COLS) return;                                                    \
        if (dyn.col_off_dev) OUT_PTR += *dyn.col_off_dev;                                  \
    }
// (the dynamic launches cover the live rows with a BOUNDED grid: blockIdx.y walks the row tiles in steps of gridDim.y --
// a grid sized for the capacity cost ~0.35 us per thousand workgroups that exit at once, 10-20 us per call at 12 k rows)

template <bool IN_F32>
__global__ __launch_bounds__(256) void transpose_to_bf16_kernel(const void* __restrict__ in, int ld_in, int R,
                                                                int Cc, unsigned short* __restrict__ out,
                                                                int ld_out, int out_cols, DynRows dyn) {
    __shared__ unsigned short t[32][33];
    const int c0 = blockIdx.x * 32;
    ODW_DYNROWS_ENTER(R, out_cols, out, (int)blockIdx.y * 32);
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int r0 = blockIdx.y * 32; r0 < out_cols; r0 += gridDim.y * 32) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = r0 + ty + 8 * k, c = c0 + tx;
            unsigned short v = 0;
            if (r < R && c < Cc) {
                const size_t sr = dyn.src_rows ? (size_t)dyn.src_rows[r] : (size_t)r;
                v = IN_F32 ? f2bf(reinterpret_cast<const float*>(in)[sr * ld_in + c])
                           : reinterpret_cast<const unsigned short*>(in)[sr * ld_in + c];
            }
            t[ty + 8 * k][tx] = v;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = c0 + ty + 8 * k, r = r0 + tx;
            if (c < Cc && r < out_cols) out[(size_t)c * ld_out + r] = r < R ? t[tx][ty + 8 * k] : (unsigned short)0;
        }
        __syncthreads();
    }
}

// bf16 -> bf16 transpose with 16-byte accesses both ways: 64 x 64 tiles through LDS (rows padded to 72 elements);
// needs Cc, ld_in, ld_out, out_cols multiples of 8 and 16-byte-aligned pointers (every operand of the head's GEMMs).
__global__ __launch_bounds__(256) void transpose_bf16_vec_kernel(const unsigned short* __restrict__ in, int ld_in, int R,
                                                                 int Cc, unsigned short* __restrict__ out, int ld_out,
                                                                 int out_cols, DynRows dyn) {
    __shared__ __attribute__((aligned(16))) unsigned short tile[64][72];
    const int c0 = blockIdx.x * 64;
    ODW_DYNROWS_ENTER(R, out_cols, out, (int)blockIdx.y * 64);
    for (int r0 = blockIdx.y * 64; r0 < out_cols; r0 += gridDim.y * 64) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int item = threadIdx.x + 256 * k;
            const int rl = item >> 3, ch = item & 7;
            const int r = r0 + rl, c = c0 + ch * 8;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (r < R && c < Cc) v = *reinterpret_cast<const uint4*>(in + (dyn.src_rows ? (size_t)dyn.src_rows[r] : (size_t)r) * ld_in + c);
            *reinterpret_cast<uint4*>(&tile[rl][ch * 8]) = v;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int item = threadIdx.x + 256 * k;
            const int cl = item >> 3, rc = item & 7;
            const int c = c0 + cl, r = r0 + rc * 8;
            if (c < Cc && r < out_cols) {
                unsigned short v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = tile[rc * 8 + q][cl];
                uint4 o;
                o.x = (unsigned)v[0] | ((unsigned)v[1] << 16);
                o.y = (unsigned)v[2] | ((unsigned)v[3] << 16);
                o.z = (unsigned)v[4] | ((unsigned)v[5] << 16);
                o.w = (unsigned)v[6] | ((unsigned)v[7] << 16);
                *reinterpret_cast<uint4*>(out + (size_t)c * ld_out + r) = o;
            }
        }
        __syncthreads();
    }
}

__global__ void f32_to_bf16_kernel(const float* __restrict__ in, unsigned short* __restrict__ out, size_t n) {
    const size_t n4 = n / 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 v = reinterpret_cast<const float4*>(in)[i];
        ushort4 o;
        o.x = f2bf(v.x); o.y = f2bf(v.y); o.z = f2bf(v.z); o.w = f2bf(v.w);
        reinterpret_cast<ushort4*>(out)[i] = o;
    }
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = f2bf(in[i]);
}

// Backward prologue of a fused Linear: dZ = dY * [Y != 0] * scale (ReLU + dropout mask re-derived
// from the saved bf16 output; Y == nullptr -> dZ = dY), written row-major (ld_z, zero padded) AND
// transposed (N x ld_t, zero padded) for the dgrad / wgrad GEMMs, plus the bias gradient
// db[n] += sum_m dZ[m][n].  32x32 tiles through LDS; dY is fp32 or bf16.
template <bool DY_F32, bool Y_F32>
__global__ __launch_bounds__(256) void linear_bwd_prep_kernel(const void* __restrict__ dY, int ld_dy,
                                                              const void* __restrict__ Y_, int ld_y,
                                                              int M, int N, float scale,
                                                              unsigned short* __restrict__ dZ, int ld_z,
                                                              unsigned short* __restrict__ dZT, int ld_t, int t_cols,
                                                              float* __restrict__ db, DynRows dyn) {
    __shared__ float t[32][33];
    const unsigned short* Y = reinterpret_cast<const unsigned short*>(Y_);
    const unsigned int* Y32 = reinterpret_cast<const unsigned int*>(Y_);
    const int n0 = blockIdx.x * 32;
    ODW_DYNROWS_ENTER(M, t_cols, dZT, (int)blockIdx.y * 32);
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int m0 = blockIdx.y * 32; m0 < t_cols; m0 += gridDim.y * 32) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int m = m0 + ty + 8 * k, n = n0 + tx;
        float v = 0.0f;
        if (m < M && n < N) {
            v = DY_F32 ? reinterpret_cast<const float*>(dY)[(size_t)m * ld_dy + n]
                       : __uint_as_float((unsigned int)reinterpret_cast<const unsigned short*>(dY)[(size_t)m * ld_dy + n] << 16);
            if (Y_) {
                const size_t my = dyn.src_rows ? (size_t)dyn.src_rows[m] : (size_t)m;
                const bool on = Y_F32 ? (Y32[my * ld_y + n] & 0x7fffffffu) != 0 : (Y[my * ld_y + n] & 0x7fff) != 0;
                v = on ? v * scale : 0.0f;
            }
        }
        t[ty + 8 * k][tx] = v;
        if (m < M && n < ld_z) dZ[(size_t)m * ld_z + n] = f2bf(v);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int n = n0 + ty + 8 * k, m = m0 + tx;
        if (n < N && m < t_cols) dZT[(size_t)n * ld_t + m] = f2bf(t[tx][ty + 8 * k]);
    }
    if (db && ty == 0) {
        float sum = 0.0f;
#pragma unroll
        for (int r = 0; r < 32; ++r) sum += t[r][tx];
        if (n0 + tx < N) atomicAdd(db + n0 + tx, sum);
    }
    __syncthreads();
    }
}

// The same prologue with 16-byte accesses: 64 x 64 tiles, fp32 tile in LDS (column sums for the bias gradient in
// fp32 before any rounding, as above), dZ written as it is loaded, dZ^T as 8 transposed values per lane.
template <bool DY_F32, bool Y_F32>
__global__ __launch_bounds__(256) void linear_bwd_prep_vec_kernel(const void* __restrict__ dY, int ld_dy,
                                                                  const void* __restrict__ Y_, int ld_y,
                                                                  int M, int N, float scale,
                                                                  unsigned short* __restrict__ dZ, int ld_z,
                                                                  unsigned short* __restrict__ dZT, int ld_t, int t_cols,
                                                                  float* __restrict__ db, DynRows dyn) {
    __shared__ float t[64][65];
    const int n0 = blockIdx.x * 64;
    ODW_DYNROWS_ENTER(M, t_cols, dZT, (int)blockIdx.y * 64);
    for (int m0 = blockIdx.y * 64; m0 < t_cols; m0 += gridDim.y * 64) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int item = threadIdx.x + 256 * k;
        const int ml = item >> 3, ch = item & 7;
        const int m = m0 + ml, n = n0 + ch * 8;
        const size_t my = (dyn.src_rows && m < M) ? (size_t)dyn.src_rows[m] : (size_t)m;
        float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (m < M && n < N) {            // N % 8 == 0: a chunk is all-in or all-out
            if (DY_F32) {
                const float4 a = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(dY) + (size_t)m * ld_dy + n);
                const float4 b = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(dY) + (size_t)m * ld_dy + n + 4);
                v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
            } else {
                const uint4 a = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(dY) + (size_t)m * ld_dy + n);
                const unsigned w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
                for (int q = 0; q < 4; ++q) { v[2 * q] = __uint_as_float(w[q] << 16); v[2 * q + 1] = __uint_as_float(w[q] & 0xffff0000u); }
            }
            if (Y_ && Y_F32) {         // the saved fp32 output of a split-precision forward (precision "bf16x2f")
                const uint4 ya = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned int*>(Y_) + my * ld_y + n);
                const uint4 yb = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned int*>(Y_) + my * ld_y + n + 4);
                const unsigned w[8] = {ya.x, ya.y, ya.z, ya.w, yb.x, yb.y, yb.z, yb.w};
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = (w[q] & 0x7fffffffu) ? v[q] * scale : 0.0f;
            } else if (Y_) {
                const uint4 yv = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(Y_) + my * ld_y + n);
                const unsigned w[4] = {yv.x, yv.y, yv.z, yv.w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    v[2 * q] = (w[q] & 0x7fffu) ? v[2 * q] * scale : 0.0f;
                    v[2 * q + 1] = (w[q] & 0x7fff0000u) ? v[2 * q + 1] * scale : 0.0f;
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) t[ml][ch * 8 + q] = v[q];
        if (m < M && n < ld_z) {         // ld_z % 8 == 0 as well; beyond N the chunk is zero padding
            uint4 o;
            o.x = f2bf_pk(v[0], v[1]);
            o.y = f2bf_pk(v[2], v[3]);
            o.z = f2bf_pk(v[4], v[5]);
            o.w = f2bf_pk(v[6], v[7]);
            *reinterpret_cast<uint4*>(dZ + (size_t)m * ld_z + n) = o;
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int item = threadIdx.x + 256 * k;
        const int nl = item >> 3, mc = item & 7;
        const int n = n0 + nl, m = m0 + mc * 8;
        if (n < N && m < t_cols) {
            uint4 o;
            o.x = f2bf_pk(t[mc * 8 + 0][nl], t[mc * 8 + 1][nl]);
            o.y = f2bf_pk(t[mc * 8 + 2][nl], t[mc * 8 + 3][nl]);
            o.z = f2bf_pk(t[mc * 8 + 4][nl], t[mc * 8 + 5][nl]);
            o.w = f2bf_pk(t[mc * 8 + 6][nl], t[mc * 8 + 7][nl]);
            *reinterpret_cast<uint4*>(dZT + (size_t)n * ld_t + m) = o;
        }
    }
    if (db && threadIdx.x < 64) {
        float sum = 0.0f;
#pragma unroll 8
        for (int r = 0; r < 64; ++r) sum += t[r][threadIdx.x];
        if (n0 + (int)threadIdx.x < N) atomicAdd(db + n0 + threadIdx.x, sum);
    }
    __syncthreads();
    }
}

// Fused SGD with momentum over flat fp32 buffers (torch.optim.SGD semantics, solver/build.py:10-24:
// d = g + wd*p ; buf = mu*buf + d ; p -= lr*buf), optionally refreshing the bf16 shadow of p.
__global__ void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf,
                           unsigned short* __restrict__ shadow, size_t n, float lr, float wd, float mu,
                           float gscale, int first) {
    const size_t n4 = n / 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 pv = reinterpret_cast<float4*>(p)[i];
        float4 gv = reinterpret_cast<const float4*>(g)[i];
        float4 bv = first ? make_float4(0, 0, 0, 0) : reinterpret_cast<float4*>(buf)[i];
        float d;
        d = gv.x * gscale + wd * pv.x; bv.x = first ? d : mu * bv.x + d; pv.x -= lr * bv.x;
        d = gv.y * gscale + wd * pv.y; bv.y = first ? d : mu * bv.y + d; pv.y -= lr * bv.y;
        d = gv.z * gscale + wd * pv.z; bv.z = first ? d : mu * bv.z + d; pv.z -= lr * bv.z;
        d = gv.w * gscale + wd * pv.w; bv.w = first ? d : mu * bv.w + d; pv.w -= lr * bv.w;
        reinterpret_cast<float4*>(p)[i] = pv;
        reinterpret_cast<float4*>(buf)[i] = bv;
        if (shadow) {
            ushort4 o;
            o.x = f2bf(pv.x); o.y = f2bf(pv.y); o.z = f2bf(pv.z); o.w = f2bf(pv.w);
            reinterpret_cast<ushort4*>(shadow)[i] = o;
        }
    }
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float pv = p[i];
        float d = g[i] * gscale + wd * pv;
        float bv = first ? d : mu * buf[i] + d;
        pv -= lr * bv;
        p[i] = pv; buf[i] = bv;
        if (shadow) shadow[i] = f2bf(pv);
    }
}


// The streaming form of the same update (n % 4 == 0): nontemporal loads / stores and two float4 per lane in flight.
// 153 M parameters x 22 B: 606 us (5.55 TB/s) with the plain kernel on 8192 workgroups, 528 us (6.37 TB/s) with
// MODE 3 on 65536 (MODE bit 0 = nontemporal, bit 1 = two vectors in flight; ODW_SGD_MODE / ODW_SGD_GRID to compare).
template <int MODE>
__global__ __launch_bounds__(256) void sgd_kernel_x(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf,
                                                    unsigned short* __restrict__ shadow, size_t n4, float lr, float wd,
                                                    float mu, float gscale, int first) {
    constexpr bool NT = (MODE & 1) != 0;
    constexpr int U = (MODE & 2) ? 2 : 1;
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef unsigned short us4 __attribute__((ext_vector_type(4)));
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n4; i0 += stride * U) {
        f4 pv[U], gv[U], bv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t i = i0 + u * stride;
            if (i < n4) {
                pv[u] = NT ? __builtin_nontemporal_load(reinterpret_cast<f4*>(p) + i) : reinterpret_cast<f4*>(p)[i];
                gv[u] = NT ? __builtin_nontemporal_load(reinterpret_cast<const f4*>(g) + i) : reinterpret_cast<const f4*>(g)[i];
                bv[u] = first ? (f4)(0.0f) : (NT ? __builtin_nontemporal_load(reinterpret_cast<f4*>(buf) + i) : reinterpret_cast<f4*>(buf)[i]);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t i = i0 + u * stride;
            if (i < n4) {
                us4 o;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float d = gv[u][k] * gscale + wd * pv[u][k];
                    bv[u][k] = first ? d : mu * bv[u][k] + d;
                    pv[u][k] -= lr * bv[u][k];
                    o[k] = f2bf(pv[u][k]);
                }
                if (NT) {
                    __builtin_nontemporal_store(pv[u], reinterpret_cast<f4*>(p) + i);
                    __builtin_nontemporal_store(bv[u], reinterpret_cast<f4*>(buf) + i);
                    if (shadow) __builtin_nontemporal_store(o, reinterpret_cast<us4*>(shadow) + i);
                } else {
                    reinterpret_cast<f4*>(p)[i] = pv[u];
                    reinterpret_cast<f4*>(buf)[i] = bv[u];
                    if (shadow) reinterpret_cast<us4*>(shadow)[i] = o;
                }
            }
        }
    }
}

// Second pass of a split-K product: C = epilogue(sum_s partial[s]) with the full fused epilogue (bias, ReLU,
// dropout keys, bf16 / fp32, accumulate) -- 4 consecutive columns per thread, fixed summation order (deterministic).
template <bool OUT_BF16>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int S, long long stride_f,
                                                            int M, int N, int ldw, void* __restrict__ Cv, int ldc,
                                                            Epilogue ep) {
    const int n4 = ldw / 4;             // ldw = row stride of the partials, N rounded up to 4 (N itself may be odd)
    if (ep.m_dev) { const int md_ = *ep.m_dev; M = md_ < M ? md_ : M; }
    const long long total = (long long)M * n4;
    const float keep_scale = ep.drop_p > 0.0f ? 1.0f / (1.0f - ep.drop_p) : 1.0f;
    unsigned amax = 0;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int m = (int)(i / n4), n = (int)(i - (long long)m * n4) * 4;
        const float* p = ws + (size_t)m * ldw + n;
        float4 a = *reinterpret_cast<const float4*>(p);
        for (int sidx = 1; sidx < S; ++sidx) {
            const float4 b = *reinterpret_cast<const float4*>(p + (size_t)sidx * stride_f);
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        float v[4] = {a.x, a.y, a.z, a.w};
        int srow = ep.seg_row[0];
        uint32_t k0 = ep.seg_k0[0], k1 = ep.seg_k1[0];
        if (ep.drop_p > 0.0f) {
            #pragma unroll
            for (int sg = 1; sg < kMaxSeg; ++sg)
                if (ep.nseg > sg && m >= ep.seg_row[sg]) { srow = ep.seg_row[sg]; k0 = ep.seg_k0[sg]; k1 = ep.seg_k1[sg]; }
        }
        // bias and mask of the four columns as one batch of independent loads (clamped; per-element "if (ep.bias) x +=
        // ep.bias[n]" compiles to a conditional load + s_waitcnt per element)
        float bq[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        unsigned short mq[4] = {1, 1, 1, 1};
        if (ep.bias) {
#pragma unroll
            for (int q = 0; q < 4; ++q) bq[q] = ep.bias[n + q < N ? n + q : N - 1];
        }
        if (ep.mask) {
            const unsigned short* mr = ep.mask + (size_t)m * ep.ldmask;
#pragma unroll
            for (int q = 0; q < 4; ++q) mq[q] = mr[n + q < N ? n + q : N - 1];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (n + q >= N) break;
            float x = v[q] * ep.alpha + bq[q];
            if (ep.relu) x = fmaxf(x, 0.0f);
            if ((mq[q] & 0x7fff) == 0) x = 0.0f;
            if (ep.drop_p > 0.0f) {
                uint32_t lrow = ep.row_ids ? (uint32_t)ep.row_ids[m] : (uint32_t)(m - srow);
                if (ep.row_tab) { const uint4 rt = ep.row_tab[m]; lrow = rt.x; k0 = rt.y; k1 = rt.z; }
                const uint32_t idx = lrow * (uint32_t)N + (uint32_t)(n + q);
                x = odw_uniform(idx, k0, k1) >= ep.drop_p ? x * keep_scale : 0.0f;
            }
            v[q] = x;
        }
        if (OUT_BF16) {
            unsigned short* c = reinterpret_cast<unsigned short*>(Cv) + (size_t)m * ldc + n;
            if (n + 3 < N && ((((uintptr_t)c) & 7) == 0)) {
                *reinterpret_cast<uint2*>(c) = make_uint2(f2bf_pk(v[0], v[1]), f2bf_pk(v[2], v[3]));
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) if (n + q < N) c[q] = f2bf(v[q]);
            }
        } else {
            float* c = reinterpret_cast<float*>(Cv) + (size_t)m * ldc + n;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (n + q < N) {
                    const float x = ep.accumulate ? c[q] + v[q] : v[q];
                    c[q] = x;
                    amax = max(amax, epi_absbits(x));
                }
        }
    }
    if (!OUT_BF16 && ep.absmax) epi_absmax_commit(amax, ep.absmax);     // (after the loop: every lane of the wave is back)
}

// The same second pass when the consumer is the next convolution of the "bf16x2f" forward (conv3x3_halo2_kernel): the sum of
// the K slices + bias, ReLU, written as the two bf16 planes [hi (N) | mid (N)] per row (ldc bf16 elements per row).
__global__ __launch_bounds__(256) void splitk_reduce_planes2_kernel(const float* __restrict__ ws, int S, long long stride_f,
                                                                    int M, int N, unsigned short* __restrict__ C, int ldc,
                                                                    const float* __restrict__ bias, int relu) {
    const int n4 = N / 4;
    const long long total = (long long)M * n4;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int m = (int)(i / n4), n = (int)(i - (long long)m * n4) * 4;
        const float* p = ws + (size_t)m * N + n;
        float4 a = *reinterpret_cast<const float4*>(p);
        for (int sidx = 1; sidx < S; ++sidx) {
            const float4 b = *reinterpret_cast<const float4*>(p + (size_t)sidx * stride_f);
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        if (bias) {
            const float4 b = *reinterpret_cast<const float4*>(bias + n);
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        if (relu) { a.x = fmaxf(a.x, 0.0f); a.y = fmaxf(a.y, 0.0f); a.z = fmaxf(a.z, 0.0f); a.w = fmaxf(a.w, 0.0f); }
        unsigned h0, m0, h1, m1, lo_;
        odwpl::split2(a.x, a.y, false, h0, m0, lo_);
        odwpl::split2(a.z, a.w, false, h1, m1, lo_);
        unsigned short* row = C + (size_t)m * ldc + n;
        *reinterpret_cast<uint2*>(row) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(row + (ldc >> 1)) = make_uint2(m0, m1);
    }
}

// Second pass of a convolution's weight gradient: dw[co][ci][t] = sum_s partial[s][co][t * Cp + ci] -- the reduction
// over the K slices and the unpacking into torch's (Cout, Cin, 3, 3) layout in one pass (was: splitk_reduce into a
// packed fp32 matrix + wgrad_unpack_kernel).
__global__ __launch_bounds__(256) void wgrad_reduce_unpack_kernel(const float* __restrict__ ws, int S, long long stride_f,
                                                                  int Co, int Ci, int Cp, int ldw, float* __restrict__ dw,
                                                                  int accumulate, const float* __restrict__ bias_ws = nullptr,
                                                                  float* __restrict__ db = nullptr) {
    // (conv_wgrad_halo_kernel's bias workgroups: db[co] += sum over the splits, in split order; block 0 only)
    if (bias_ws && blockIdx.x == 0)
        for (int co = threadIdx.x; co < Co; co += 256) {
            float a = bias_ws[co];
            for (int sidx = 1; sidx < S; ++sidx) a += bias_ws[(size_t)sidx * Co + co];
            db[co] += a;
        }
    // a workgroup = (co, 64 input channels): thread (j = channel, g = tap group) sums taps g, g + 4, g + 8 over the
    // slices in slice order (256-byte coalesced reads, four slices of loads in flight), LDS [j][t], 576 floats out
    __shared__ float sm[64 * 9];
    const int chunks = (Ci + 63) / 64;
    const int j = threadIdx.x & 63, g = threadIdx.x >> 6;
    for (int b = blockIdx.x; b < Co * chunks; b += gridDim.x) {
        const int co = b / chunks, ci0 = (b - co * chunks) * 64;
        const int ci = ci0 + j;
        if (ci < Ci) {
            for (int t = g; t < 9; t += 4) {
                const float* p = ws + (size_t)co * ldw + t * Cp + ci;
                float a = p[0];
                int sidx = 1;
                for (; sidx + 3 < S; sidx += 4) {
                    const float x0 = p[(size_t)sidx * stride_f], x1 = p[(size_t)(sidx + 1) * stride_f];
                    const float x2 = p[(size_t)(sidx + 2) * stride_f], x3 = p[(size_t)(sidx + 3) * stride_f];
                    a += x0; a += x1; a += x2; a += x3;                  // fixed order
                }
                for (; sidx < S; ++sidx) a += p[(size_t)sidx * stride_f];
                sm[j * 9 + t] = a;
            }
        }
        __syncthreads();
        const int n = (Ci - ci0 < 64 ? Ci - ci0 : 64) * 9;
        float* out = dw + ((size_t)co * Ci + ci0) * 9;
        for (int k = threadIdx.x; k < n; k += 256) out[k] = accumulate ? out[k] + sm[k] : sm[k];
        __syncthreads();
    }
}

// Which kernel serves a product, and in how many K slices.  Variant 0 = register-staged 128x128 (any alignment),
// 1 = LDS-DMA 128x128 (two workgroups per CU), 2 = LDS-DMA 256x128 three-slot ring, 3 = 256x256 asm-scheduled.
// The DMA kernels need operand rows padded to a multiple of 64; the 256x256 kernel also 16-byte-aligned rows of C.
// Cost model (seconds): rounds of workgroups over 256 CUs x tile FLOPs / measured per-CU rate, plus -- for a
// split -- the reduction pass over the fp32 partials.  Big tiles are ~25 % faster per FLOP but lose when their
// grid leaves CUs idle (M = 2000: 128 tiles of 256x256); products with a small C and a long K (the conv weight
// gradients: 512 x 4608 x 5776 pixels = 72 tiles; the fc6 pass over the ~450 sampled rows) fill the chip only when
// K is split.  ODW_GEMM_VARIANT = reg | glds | ring | big and ODW_GEMM_SPLITK = n force a choice (tools/gemm_var.py).
struct Plan { int variant, splits, kchunk; };

__host__ Plan pick_plan(int M, int N, int K, int lda, int ldb, const void* C, int ldc, int c_is_bf16, bool allow_split) {
    const int k64 = (K + 63) / 64 * 64;
    const bool dma_ok = lda >= k64 && ldb >= k64 && K > 0;
    const int c_el = c_is_bf16 ? 2 : 4;
    // the 256x256 kernel forms its per-lane operand offsets in 32 bits (row * ld * 2 bytes): operands past 4 GiB -- the
    // R-50 head's fc6 in the bf16x3 mode is 4000 x 602112 bf16 -- go to the ring kernel, which addresses in 64 bits
    const bool fits32 = (unsigned long long)M * (unsigned long long)lda * 2ull < (1ull << 32) &&
                        (unsigned long long)N * (unsigned long long)ldb * 2ull < (1ull << 32);
    const bool big_ok = dma_ok && fits32 && N % 8 == 0 && ((size_t)ldc * c_el) % 16 == 0 && (((uintptr_t)C) & 15) == 0;
    const char* force = getenv("ODW_GEMM_VARIANT");
    const char* fsplit = getenv("ODW_GEMM_SPLITK");
    int only = -1;
    if (force) {
        if (force[0] == 'r' && force[1] == 'e') return {0, 1, 0};
        if (force[0] == 'g') return {dma_ok ? 1 : 0, 1, 0};
        if (force[0] == 'r' && force[1] == 'i') only = 2;
        if (force[0] == 'b') only = big_ok ? 3 : 1;
    }
    if (!dma_ok) return {0, 1, 0};
    const double kCuRate = 3.4e12;                       // FLOP/s of one CU inside the 256x128 ring kernel
    const int kSplits[] = {1, 2, 3, 4, 6, 8, 12, 16, 24, 32};
    struct V { int id, tm, tn, per_cu; double rate; bool splittable; };
    const V vs[] = {{1, BM, BN, 2, 0.97, false}, {2, RM, RN, 1, 1.0, true}, {3, GM, GN, 1, 1.25, true}};
    Plan best = {1, 1, 0};
    double best_t = 1e30;
    for (const V& v : vs) {
        if (v.id == 3 && !big_ok) continue;
        if (only >= 0 && v.id != only) continue;
        const long tiles = (long)((M + v.tm - 1) / v.tm) * ((N + v.tn - 1) / v.tn);
        for (int sp : kSplits) {
            if (sp > 1 && (!allow_split || !v.splittable)) break;
            if (fsplit && sp != atoi(fsplit) && !(sp == 1 && atoi(fsplit) <= 1)) continue;
            const int kc = ((K + sp - 1) / sp + 63) / 64 * 64;
            if (sp > 1 && (kc < 256 || (long)kc * (sp - 1) >= K)) break;
            const long rounds = (tiles * sp + 256L * v.per_cu - 1) / (256L * v.per_cu);
            // (the 256x256 kernel run UNSPLIT sustains 1.7, not 1.25: a round of fc6's input gradient, 256 tiles x K = 4096,
            // takes 90 us -- profiles/r05/gemm_smallm.txt; the sampled-row views, M = 750-900, went to the 128x128 kernel at
            // 213-235 us where this one takes 186.  Its split form is priced as before: the partials' traffic costs it more
            // than the term below says, and the 256x128 ring wins those shapes.)
            const double rate = (v.id == 3 && sp == 1) ? 1.7 : v.rate;
            double t = (double)rounds * v.per_cu * v.tm * v.tn * 2.0 * kc / (rate * kCuRate) + rounds * 2e-6;
            if (sp > 1) t += ((double)sp * M * N * 4.0 + (double)M * N * c_el) / 2.5e12 + 4e-6;
            if (t < best_t) { best_t = t; best = {v.id, sp, sp > 1 ? kc : 0}; }
        }
    }
    return best;
}

// Tail split of a 256x256-tiled product whose tile count leaves a thin last round: T = tiles_m * tiles_n workgroups on
// 256 CUs run ceil(T / 256) rounds, and e.g. fc6's input gradient (2000 x 25088: 8 x 98 = 784 tiles) spends a whole
// fourth round on 16 tiles.  When the remainder is a few whole tile COLUMNS, the product is run as two: the first
// (tiles_n - q) tile columns = a multiple of 256 workgroups, and the last q columns as their own small product, which
// the planner splits along K over the whole chip.  Returns the width of that tail (0 = no split).  Needs no dropout
// (its counter index is tied to the full row width).
int tail_columns(int M, int N, int K, int lda, int ldb, const void* C, int ldc, int c_is_bf16, float drop_p,
                 bool have_workspace) {
    static const bool off = getenv("ODW_GEMM_NO_TAIL") != nullptr;
    if (off || drop_p > 0.0f || !have_workspace) return 0;
    if (pick_plan(M, N, K, lda, ldb, C, ldc, c_is_bf16, true).splits > 1) return 0;
    if (pick_plan(M, N, K, lda, ldb, C, ldc, c_is_bf16, false).variant != 3) return 0;
    const long tm = (M + GM - 1) / GM, tn = (N + GN - 1) / GN, T = tm * tn;
    if (T <= 256) return 0;
    const long rem = T % 256;
    if (rem == 0 || rem > 96 || rem % tm != 0) return 0;
    const long q = rem / tm;
    if (q >= tn) return 0;
    return N - (int)(tn - q) * GN;
}

}  // namespace

// ---- cell-major plane GEMM (gemm_nt_cm_kernel) ---------------------------------------------------------------------
static int cm_cells_per_split(int M, int N, int S, bool allow) {
    // One workgroup per CU (160 KB of LDS): a launch of T tiles x s splits takes ceil(T s / 256) rounds of ceil(S / s) cells.
    // Pick the s with the fewest cells on the critical path (ties: fewer splits = fewer partials); e.g. the sampled-row views
    // of the contrastive loss, M ~ 450: 64 tiles -> 4 splits of 13 cells in ONE round (6 splits of 9 cells took two: 0.40 ms
    // against 0.29).
    const long tiles = (long)((M + RM - 1) / RM) * ((N + RN - 1) / RN);
    if (!allow || tiles >= 256) return 0;                      // one pass
    int best_s = 1;
    long best = (long)S;                                       // rounds(1) = 1
    for (int sp = 2; sp <= 16 && sp <= S; ++sp) {
        const int kc = (S + sp - 1) / sp;
        const int real = (S + kc - 1) / kc;                    // splits that actually get cells
        const long cost = ((tiles * real + 255) / 256) * kc + 1;      // (+1: the reduction pass is not free)
        if (cost < best) { best = cost; best_s = real; }
    }
    if (best_s <= 1) return 0;
    return (S + best_s - 1) / best_s;
}

ODW_EXPORT int64_t odw_gemm_nt_cm_workspace(int M, int N, int S) {
    const int kc = cm_cells_per_split(M, N, S, true);
    if (kc == 0) return 0;
    return (int64_t)((S + kc - 1) / kc) * M * ((N + 3) / 4 * 4) * 4;
}

ODW_EXPORT int64_t odw_gemm_nt_cm_pair_workspace(int M, int N, int S) {      // the pair form's partials hold both halves
    return 2 * odw_gemm_nt_cm_workspace(M, N, S);
}

struct CmDyn { const int* m_dev; int m_hint; const uint4* row_tab; };

static int gemm_nt_cm_launch(const void* A, int lda, int a_mid, const void* B, int ldb, int b_mid, int M, int N, int C, int S,
                             const float* keep, const float* keep_sum, int drop_row0, float* Cout, int ldc,
                             const float* bias, int relu, float drop_p, int nseg, const int* seg_rows,
                             const uint32_t* seg_keys, const int* row_ids, void* workspace, int64_t workspace_bytes,
                             void* stream_, const CmDyn* dyn);

ODW_EXPORT int odw_gemm_nt_cm(const void* A, int lda, int a_mid, const void* B, int ldb, int b_mid, int M, int N, int C, int S,
                              const float* keep, const float* keep_sum, int drop_row0, float* Cout, int ldc,
                              const float* bias, int relu, float drop_p, int nseg, const int* seg_rows,
                              const uint32_t* seg_keys, const int* row_ids, void* workspace, int64_t workspace_bytes,
                              void* stream_) {
    return gemm_nt_cm_launch(A, lda, a_mid, B, ldb, b_mid, M, N, C, S, keep, keep_sum, drop_row0, Cout, ldc, bias, relu, drop_p, nseg,
                             seg_rows, seg_keys, row_ids, workspace, workspace_bytes, stream_, nullptr);
}

// The plain (non-pair) product over cell-major planes with the number of rows on the device (the sampled-row views of the
// contrastive loss: their count is what loss_lists.hip computes).  M_cap rows of A / Cout exist as memory; *m_dev of them
// are computed.  The cell split is planned for M_hint, its partials sized for M_cap.
ODW_EXPORT int64_t odw_gemm_nt_cm_dyn_workspace(int M_cap, int M_hint, int N, int S) {
    const int kc = cm_cells_per_split(M_hint > 0 && M_hint < M_cap ? M_hint : M_cap, N, S, true);
    if (kc == 0) return 0;
    return (int64_t)((S + kc - 1) / kc) * M_cap * ((N + 3) / 4 * 4) * 4;
}

ODW_EXPORT int odw_gemm_nt_cm_dyn(const void* A, int lda, int a_mid, const void* B, int ldb, int b_mid, int M_cap, int N, int C,
                                  int S, float* Cout, int ldc, const float* bias, int relu, float drop_p, const void* row_tab,
                                  const int* m_dev, int M_hint, void* workspace, int64_t workspace_bytes, void* stream_) {
    ODW_REQUIRE(m_dev, "gemm_nt_cm_dyn: m_dev is null (use odw_gemm_nt_cm)");
    ODW_REQUIRE(drop_p == 0.0f || row_tab, "gemm_nt_cm_dyn: dropout needs the per-row draw table");
    CmDyn d;
    d.m_dev = m_dev; d.m_hint = M_hint > 0 && M_hint < M_cap ? M_hint : M_cap; d.row_tab = (const uint4*)row_tab;
    return gemm_nt_cm_launch(A, lda, a_mid, B, ldb, b_mid, M_cap, N, C, S, nullptr, nullptr, 0, Cout, ldc, bias, relu, drop_p, 0,
                             nullptr, nullptr, nullptr, workspace, workspace_bytes, stream_, &d);
}

static int gemm_nt_cm_launch(const void* A, int lda, int a_mid, const void* B, int ldb, int b_mid, int M, int N, int C, int S,
                             const float* keep, const float* keep_sum, int drop_row0, float* Cout, int ldc,
                             const float* bias, int relu, float drop_p, int nseg, const int* seg_rows,
                             const uint32_t* seg_keys, const int* row_ids, void* workspace, int64_t workspace_bytes,
                             void* stream_, const CmDyn* dyn) {
    hipStream_t stream = (hipStream_t)stream_;
    ODW_REQUIRE(M >= 0 && N >= 0 && C > 0 && C % BK == 0 && S >= 1 && S <= 64, "gemm_nt_cm: bad dims M=%d N=%d C=%d S=%d", M, N, C, S);
    if (M == 0 || N == 0) return ODW_OK;
    const long long K = (long long)C * S;
    ODW_REQUIRE(A && B && Cout, "gemm_nt_cm: null pointer");
    ODW_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && (((uintptr_t)A) & 15) == 0 && (((uintptr_t)B) & 15) == 0 && a_mid % 8 == 0 &&
                b_mid % 8 == 0 && a_mid >= K && b_mid >= K && (long long)a_mid + K <= lda && (long long)b_mid + K <= ldb,
                "gemm_nt_cm: planes [hi | mid] of %lld elements each must fit the rows (lda=%d a_mid=%d ldb=%d b_mid=%d)",
                K, lda, a_mid, ldb, b_mid);
    ODW_REQUIRE(drop_p >= 0.0f && drop_p < 1.0f && nseg >= 0 && nseg <= kMaxSeg, "gemm_nt_cm: bad dropout args");
    ODW_REQUIRE(!keep || (keep_sum && drop_row0 >= M && !row_ids), "gemm_nt_cm: the pair form needs keep_sum, drop_row0 >= M, no row_ids");
    Epilogue ep;
    ep.bias = bias; ep.relu = relu; ep.drop_p = drop_p; ep.nseg = nseg; ep.accumulate = 0; ep.alpha = 1.0f;
    ep.mask = nullptr; ep.ldmask = 0; ep.kchunk = 0; ep.split_stride = 0; ep.row_ids = row_ids; ep.pm = 0;
    for (int i = 0; i < kMaxSeg; ++i) {
        ep.seg_row[i] = (i < nseg && seg_rows) ? seg_rows[i] : 0;
        ep.seg_k0[i] = (i < nseg && seg_keys) ? seg_keys[2 * i] : 0;
        ep.seg_k1[i] = (i < nseg && seg_keys) ? seg_keys[2 * i + 1] : 0;
    }
    if (dyn) { ep.m_dev = dyn->m_dev; ep.row_tab = dyn->row_tab; }
    if (drop_p > 0.0f && !(dyn && dyn->row_tab))
        ODW_REQUIRE(nseg >= 1 && seg_rows && seg_keys && seg_rows[0] == 0, "gemm_nt_cm: dropout needs row segments starting at 0");
    if (keep && drop_p > 0.0f) ODW_REQUIRE(nseg == 2 && seg_rows[1] == drop_row0, "gemm_nt_cm: the pair form takes two dropout segments (clean rows, DropBlock rows)");
    CmArgs cm;
    cm.C = C; cm.S = S; cm.a_mid = a_mid; cm.b_mid = b_mid; cm.keep = keep; cm.keep_sum = keep_sum; cm.drop_row0 = drop_row0;
    const int tiles_m = (M + RM - 1) / RM, tiles_n = (N + RN - 1) / RN;
    const size_t ring_lds = (size_t)kRingStages * kRingStageChunks * sizeof(uint4);
    const size_t share_lds = (size_t)(3 * RM + 4 * RN) * kChunksPerRow * sizeof(uint4);      // 160 KB
    // Four operand tiles per 64-channel group on a 3 A + 4 B slot ring (ODW_CM_VARIANT=0: the six-tile ring, comparison).
    // Measured at P = 2000 x 4096 x (49 x 512), isolated launch: six-tile ring 1.30 ms, this form 1.08-1.09 ms; with every
    // operand L2-resident (all workgroups on the same tiles) 1.06-1.08 ms -- cache misses are not the bound; with the loop's
    // DMA removed 0.86 ms (MFMA floor at the ~1.6 GHz held under load: 0.77 ms).  Tried on top, no gain: 32-channel groups
    // with all four tiles in one 48 KB stage, one barrier per 24 MFMAs and each fragment read once per K slice (1.18 ms:
    // 64-byte DMA rows); two barriers per group with (Ah, Bh) + (Ah, Bm) as one block (1.08 ms).
    static const int share = getenv("ODW_CM_VARIANT") ? atoi(getenv("ODW_CM_VARIANT")) : 1;
    if (keep) {
#define ODW_CM_LAUNCH(PAIRV, SV, LDSB, GRID, OUT, EP)                                                                   \
        do {                                                                                                             \
            const hipError_t attr_ = odw_set_max_lds(reinterpret_cast<const void*>(gemm_nt_cm_kernel<PAIRV, SV>), \
                                                                (int)(LDSB)); \
            ODW_CHECK_HIP(attr_, "gemm_nt_cm attr");                                                                     \
            gemm_nt_cm_kernel<PAIRV, SV><<<GRID, kRingThreads, LDSB, stream>>>(                                          \
                (const unsigned short*)A, lda, (const unsigned short*)B, ldb, M, N, OUT, ldc_,                           \
                epilogue_narrow<CmEp<PAIRV>::type>(EP), cm, tiles_m, tiles_n);                                    \
        } while (0)
        const int ldc_ = ldc;
        // few ROIs: the cells are split over blockIdx.y; a split's partial clean sums land in rows [0, M) of its slice
        // of the workspace and its partial DropBlock sums (already scaled by g) in rows [M, 2M); one reduction pass adds
        // the slices and applies the epilogue of both halves (needs the halves adjacent in Cout: drop_row0 == M)
        const int ldw_p = (N + 3) / 4 * 4;
        int kcp = drop_row0 == M ? cm_cells_per_split(M, N, S, workspace != nullptr) : 0;
        if (kcp > 0 && (workspace_bytes < (int64_t)((S + kcp - 1) / kcp) * 2 * M * ldw_p * 4 || (((uintptr_t)workspace) & 15) != 0)) kcp = 0;
        if (kcp > 0) {
            const int splits = (S + kcp - 1) / kcp;
            Epilogue pe = ep;
            pe.bias = nullptr; pe.relu = 0; pe.drop_p = 0.0f; pe.nseg = 0;
            pe.kchunk = kcp; pe.split_stride = (long long)2 * M * ldw_p * 4;
            const int ldc_ = ldw_p;
            const dim3 grid((unsigned)(tiles_m * tiles_n), (unsigned)splits);
            ODW_CM_LAUNCH(true, 1, share_lds, grid, workspace, pe);
            ODW_CHECK_LAUNCH("gemm_nt_cm_kernel<pair, split>");
            const long long quads = (long long)2 * M * (ldw_p / 4);
            const int rblocks = (int)((quads + 255) / 256 < 4096 ? (quads + 255) / 256 : 4096);
            splitk_reduce_kernel<false><<<rblocks, 256, 0, stream>>>((const float*)workspace, splits, (long long)2 * M * ldw_p, 2 * M, N,
                                                                      ldw_p, Cout, ldc, ep);
            ODW_CHECK_LAUNCH("splitk_reduce_kernel");
            return ODW_OK;
        }
        if (share) ODW_CM_LAUNCH(true, 1, share_lds, tiles_m * tiles_n, Cout, ep);
        else ODW_CM_LAUNCH(true, 0, ring_lds, tiles_m * tiles_n, Cout, ep);
        ODW_CHECK_LAUNCH("gemm_nt_cm_kernel<pair>");
        return ODW_OK;
    }
    const int ldw = (N + 3) / 4 * 4;
    int kc = cm_cells_per_split(dyn ? dyn->m_hint : M, N, S, workspace != nullptr);
    if (kc > 0 && workspace_bytes < (int64_t)((S + kc - 1) / kc) * M * ldw * 4) kc = 0;
    if (kc > 0) {
        ODW_REQUIRE((((uintptr_t)workspace) & 15) == 0, "gemm_nt_cm: workspace must be 16-byte aligned");
        const int splits = (S + kc - 1) / kc;
        Epilogue pe = ep;
        pe.bias = nullptr; pe.relu = 0; pe.drop_p = 0.0f; pe.nseg = 0; pe.row_ids = nullptr;
        pe.kchunk = kc; pe.split_stride = (long long)M * ldw * 4;
        {
            const int ldc_ = ldw;
            const dim3 grid((unsigned)(tiles_m * tiles_n), (unsigned)splits);
            ODW_CM_LAUNCH(false, 1, share_lds, grid, workspace, pe);
        }
        ODW_CHECK_LAUNCH("gemm_nt_cm_kernel<split>");
        const long long quads = (long long)M * (ldw / 4);
        const int rblocks = (int)((quads + 255) / 256 < 4096 ? (quads + 255) / 256 : 4096);
        splitk_reduce_kernel<false><<<rblocks, 256, 0, stream>>>((const float*)workspace, splits, (long long)M * ldw, M, N, ldw,
                                                                  Cout, ldc, ep);
        ODW_CHECK_LAUNCH("splitk_reduce_kernel");
        return ODW_OK;
    }
    {
        const int ldc_ = ldc;
        ODW_CM_LAUNCH(false, 1, share_lds, tiles_m * tiles_n, Cout, ep);
    }
#undef ODW_CM_LAUNCH
    ODW_CHECK_LAUNCH("gemm_nt_cm_kernel");
    return ODW_OK;
}

ODW_EXPORT int odw_gemm_nt_bf16_variant(int M, int N, int K, int lda, int ldb, const void* C, int ldc,
                                        int c_is_bf16) {
    return pick_plan(M, N, K, lda, ldb, C, ldc, c_is_bf16, false).variant;
}

ODW_EXPORT int64_t odw_gemm_nt_bf16_workspace(int M, int N, int K, int lda, int ldb, const void* C, int ldc,
                                              int c_is_bf16, int* variant_out) {
    const int nt = tail_columns(M, N, K, lda, ldb, C, ldc, c_is_bf16, 0.0f, true);
    if (nt > 0) {       // the partials of the tail product (the main part runs unsplit)
        const Plan pt = pick_plan(M, nt, K, lda, ldb, C, ldc, c_is_bf16, true);
        if (variant_out) *variant_out = 3;
        return pt.splits > 1 ? (int64_t)pt.splits * M * ((nt + 3) / 4 * 4) * 4 : 0;
    }
    const Plan p = pick_plan(M, N, K, lda, ldb, C, ldc, c_is_bf16, true);
    if (variant_out) *variant_out = p.variant;
    return p.splits > 1 ? (int64_t)p.splits * M * ((N + 3) / 4 * 4) * 4 : 0;
}

ODW_EXPORT int odw_gemm_nt_bf16_ws(const void* A, int lda, const void* B, int ldb, int M, int N, int K, void* C,
                                   int ldc, int c_is_bf16, const float* bias, int relu, float alpha, float drop_p,
                                   int nseg, const int* seg_rows, const uint32_t* seg_keys, const int* row_ids,
                                   int accumulate, void* workspace, int64_t workspace_bytes, void* stream_);

ODW_EXPORT int odw_gemm_nt_bf16(const void* A, int lda, const void* B, int ldb, int M, int N, int K, void* C,
                                int ldc, int c_is_bf16, const float* bias, int relu, float alpha, float drop_p,
                                int nseg, const int* seg_rows, const uint32_t* seg_keys, int accumulate,
                                void* stream_) {
    return odw_gemm_nt_bf16_ws(A, lda, B, ldb, M, N, K, C, ldc, c_is_bf16, bias, relu, alpha, drop_p, nseg, seg_rows,
                               seg_keys, nullptr, accumulate, nullptr, 0, stream_);
}

// Device-resident extents of a launch (odw_gemm_nt_bf16_dyn): M / K of the call are CAPACITIES (grid, workspace, bounds);
// the kernels read the live values from m_dev / k_dev; the plan (kernel variant, K slices) is made for the hints.
struct DynExtent {
    const int* m_dev; int m_hint;
    const int* k_dev; int k_hint;
    const uint4* row_tab;
    unsigned* absmax;       // odw_gemm_nt_bf16_absmax: max |C| of the launch, for the consumer's fixed-point scale
};

static int gemm_nt_launch(const void* A, int lda, const void* B, int ldb, int M, int N, int K, void* C,
                          int ldc, int c_is_bf16, const float* bias, int relu, float alpha, float drop_p,
                          int nseg, const int* seg_rows, const uint32_t* seg_keys, const int* row_ids,
                          int accumulate, void* workspace, int64_t workspace_bytes, void* stream_, const DynExtent* dyn);

ODW_EXPORT int odw_gemm_nt_bf16_ws(const void* A, int lda, const void* B, int ldb, int M, int N, int K, void* C,
                                   int ldc, int c_is_bf16, const float* bias, int relu, float alpha, float drop_p,
                                   int nseg, const int* seg_rows, const uint32_t* seg_keys, const int* row_ids,
                                   int accumulate, void* workspace, int64_t workspace_bytes, void* stream_) {
    return gemm_nt_launch(A, lda, B, ldb, M, N, K, C, ldc, c_is_bf16, bias, relu, alpha, drop_p, nseg, seg_rows, seg_keys, row_ids,
                          accumulate, workspace, workspace_bytes, stream_, nullptr);
}

static int dyn_hint(int hint, int cap) { return hint > 0 ? (hint < cap ? hint : cap) : cap; }

// workspace of the dynamic form: the split-K partials of the plan made for the hints, over the CAPACITY rows
ODW_EXPORT int64_t odw_gemm_nt_bf16_dyn_workspace(int M_cap, int M_hint, int N, int K_cap, int K_hint, int lda, int ldb,
                                                  const void* C, int ldc, int c_is_bf16, int* variant_out) {
    // M_hint <= 0: the rows are NOT device-resident (only the reduction is) -- the tail-column split of the static form applies
    if (M_hint <= 0) {
        const int kh = dyn_hint(K_hint, K_cap);
        const int nt = tail_columns(M_cap, N, kh, lda, ldb, C, ldc, c_is_bf16, 0.0f, true);
        if (nt > 0) {
            const Plan pt = pick_plan(M_cap, nt, kh, lda, ldb, C, ldc, c_is_bf16, true);
            if (variant_out) *variant_out = 3;
            return pt.splits > 1 ? (int64_t)pt.splits * M_cap * ((nt + 3) / 4 * 4) * 4 : 0;
        }
    }
    const Plan p = pick_plan(dyn_hint(M_hint, M_cap), N, dyn_hint(K_hint, K_cap), lda, ldb, C, ldc, c_is_bf16, true);
    if (variant_out) *variant_out = p.variant;
    return p.splits > 1 ? (int64_t)p.splits * M_cap * ((N + 3) / 4 * 4) * 4 : 0;
}

// C[M x N] (+)= epilogue(alpha A B^T) where the number of rows M and / or the reduction length K live on the DEVICE
// (loss_lists.hip writes them): the launch covers M_cap rows and K_cap columns of the operands, which must exist as memory
// (rows >= *m_dev may hold anything finite or not: they are never read into a stored result; columns of A / B in
// [*k_dev, r64(*k_dev)) must be zero).  *k_dev must be a multiple of 8.  row_tab (M_cap x uint4, device): per-row dropout
// draw (logical row, key0, key1, -) -- the stacked views' segment boundaries are device values too.  The hints size nothing:
// they pick the kernel variant and the number of K slices (performance only).
ODW_EXPORT int odw_gemm_nt_bf16_dyn(const void* A, int lda, const void* B, int ldb, int M_cap, int N, int K_cap, void* C,
                                    int ldc, int c_is_bf16, const float* bias, int relu, float alpha, float drop_p,
                                    const void* row_tab, const int* m_dev, int M_hint, const int* k_dev, int K_hint,
                                    int accumulate, void* workspace, int64_t workspace_bytes, void* stream_) {
    ODW_REQUIRE(m_dev || k_dev, "gemm_nt_bf16_dyn: neither extent is device-resident (use odw_gemm_nt_bf16_ws)");
    ODW_REQUIRE(drop_p == 0.0f || row_tab, "gemm_nt_bf16_dyn: dropout needs the per-row draw table");
    DynExtent d;
    d.m_dev = m_dev; d.m_hint = dyn_hint(M_hint, M_cap); d.k_dev = k_dev; d.k_hint = dyn_hint(K_hint, K_cap);
    d.row_tab = (const uint4*)row_tab; d.absmax = nullptr;
    return gemm_nt_launch(A, lda, B, ldb, M_cap, N, K_cap, C, ldc, c_is_bf16, bias, relu, alpha, drop_p, 0, nullptr, nullptr,
                          nullptr, accumulate, workspace, workspace_bytes, stream_, &d);
}

// C[M x N] (fp32) = alpha A B^T, and *absmax = max(*absmax, bit pattern of max |C[m][n]|) -- what odwfx::absmax_kernel would
// find by re-reading C (odw_fixed.h).  The caller zeroes the word and hands it to the scatter that consumes C
// (odw_roi_pool_stack_backward_scaled): fc6's input gradient, 200 MB, is no longer read twice.  Plans, workspace and variant as
// odw_gemm_nt_bf16_workspace / _ws say for the same shape.
ODW_EXPORT int odw_gemm_nt_bf16_absmax(const void* A, int lda, const void* B, int ldb, int M, int N, int K, float* C, int ldc,
                                       float alpha, void* absmax, void* workspace, int64_t workspace_bytes, void* stream_) {
    ODW_REQUIRE(absmax && (((uintptr_t)absmax) & 3) == 0, "gemm_nt_bf16_absmax: the 4-byte word the maximum lands in");
    ODW_REQUIRE(N % 4 == 0, "gemm_nt_bf16_absmax: N=%d must be a multiple of 4 (the staged store's 16-byte chunks end at N)", N);
    DynExtent d;
    d.m_dev = nullptr; d.m_hint = M; d.k_dev = nullptr; d.k_hint = K; d.row_tab = nullptr; d.absmax = (unsigned*)absmax;
    return gemm_nt_launch(A, lda, B, ldb, M, N, K, C, ldc, 0, nullptr, 0, alpha, 0.0f, 0, nullptr, nullptr, nullptr, 0, workspace,
                          workspace_bytes, stream_, &d);
}

static int gemm_nt_launch(const void* A, int lda, const void* B, int ldb, int M, int N, int K, void* C,
                          int ldc, int c_is_bf16, const float* bias, int relu, float alpha, float drop_p,
                          int nseg, const int* seg_rows, const uint32_t* seg_keys, const int* row_ids,
                          int accumulate, void* workspace, int64_t workspace_bytes, void* stream_, const DynExtent* dyn) {
    hipStream_t stream = (hipStream_t)stream_;
    ODW_REQUIRE(M >= 0 && N >= 0 && K >= 0, "gemm_nt_bf16: bad dims M=%d N=%d K=%d", M, N, K);
    if (M == 0 || N == 0) return ODW_OK;
    ODW_REQUIRE(A && B && C, "gemm_nt_bf16: null pointer");
    ODW_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && (((uintptr_t)A) & 15) == 0 && (((uintptr_t)B) & 15) == 0,
                "gemm_nt_bf16: A/B rows must be 16-byte aligned (lda=%d ldb=%d)", lda, ldb);
    ODW_REQUIRE(((K + 7) / 8) * 8 <= lda && ((K + 7) / 8) * 8 <= ldb,
                "gemm_nt_bf16: K=%d rounded up to 8 must fit in lda=%d / ldb=%d (zero padded)", K, lda, ldb);
    ODW_REQUIRE(drop_p >= 0.0f && drop_p < 1.0f && nseg >= 0 && nseg <= kMaxSeg, "gemm_nt_bf16: bad dropout args");
    ODW_REQUIRE(!(accumulate && c_is_bf16), "gemm_nt_bf16: accumulate needs an fp32 C");
    // (device-resident extents: with the ROWS on the device the tile count is unknown -- no tail split; with only the reduction
    // on the device -- the weight-gradient batches -- it applies as in the static form, planned for the hinted K)
    const bool tail_ok = !dyn || (!dyn->m_dev && !dyn->row_tab);
    if (const int nt = tail_ok ? tail_columns(M, N, dyn ? dyn->k_hint : K, lda, ldb, C, ldc, c_is_bf16, drop_p, workspace != nullptr) : 0) {
        const int n1 = N - nt;
        const int rc = gemm_nt_launch(A, lda, B, ldb, M, n1, K, C, ldc, c_is_bf16, bias, relu, alpha, 0.0f, 0, nullptr,
                                      nullptr, nullptr, accumulate, nullptr, 0, stream_, dyn);
        if (rc != ODW_OK) return rc;
        return gemm_nt_launch(A, lda, reinterpret_cast<const unsigned short*>(B) + (size_t)n1 * ldb, ldb, M, nt, K,
                              reinterpret_cast<char*>(C) + (size_t)n1 * (c_is_bf16 ? 2 : 4), ldc, c_is_bf16,
                              bias ? bias + n1 : nullptr, relu, alpha, 0.0f, 0, nullptr, nullptr, nullptr, accumulate,
                              workspace, workspace_bytes, stream_, dyn);
    }
    Epilogue ep;
    ep.bias = bias; ep.relu = relu; ep.drop_p = drop_p; ep.nseg = nseg; ep.accumulate = accumulate; ep.alpha = alpha;
    ep.mask = nullptr; ep.ldmask = 0; ep.kchunk = 0; ep.split_stride = 0; ep.row_ids = row_ids;
    { const char* e = getenv("ODW_GEMM_PM"); ep.pm = e ? atoi(e) : 0; }
    for (int i = 0; i < kMaxSeg; ++i) {
        ep.seg_row[i] = (i < nseg && seg_rows) ? seg_rows[i] : 0;
        ep.seg_k0[i] = (i < nseg && seg_keys) ? seg_keys[2 * i] : 0;
        ep.seg_k1[i] = (i < nseg && seg_keys) ? seg_keys[2 * i + 1] : 0;
    }
    if (dyn) { ep.m_dev = dyn->m_dev; ep.k_dev = dyn->k_dev; ep.row_tab = dyn->row_tab; ep.absmax = dyn->absmax; }
    if (drop_p > 0.0f && !(dyn && dyn->row_tab))
        ODW_REQUIRE(nseg >= 1 && seg_rows && seg_keys && seg_rows[0] == 0, "gemm_nt_bf16: dropout needs row segments starting at 0");
    const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
    const size_t lds_bytes = (size_t)2 * 2 * kTileChunks * sizeof(uint4);   // 64 KB
    // (device-resident extents: the plan is made for the HINTS -- what the extents are expected to be --, every grid and
    // the workspace for the capacities M / K of the call)
    const int Mp = dyn ? dyn->m_hint : M, Kp = dyn ? dyn->k_hint : K;
    Plan plan = pick_plan(Mp, N, Kp, lda, ldb, C, ldc, c_is_bf16, workspace != nullptr);
    const int ldw = (N + 3) / 4 * 4;        // row stride of the fp32 partials
    if (plan.splits > 1 && workspace_bytes < (int64_t)plan.splits * M * ldw * 4)
        plan = pick_plan(Mp, N, Kp, lda, ldb, C, ldc, c_is_bf16, false);
    if (plan.splits > 1) {
        // partial products (plain, fp32) into the workspace, then one reduction pass with the fused epilogue
        ODW_REQUIRE((((uintptr_t)workspace) & 15) == 0, "gemm_nt_bf16: workspace must be 16-byte aligned");
        Epilogue pe = ep;
        pe.bias = nullptr; pe.relu = 0; pe.drop_p = 0.0f; pe.nseg = 0; pe.accumulate = 0; pe.alpha = 1.0f; pe.row_ids = nullptr;
        pe.absmax = nullptr;          // (the partials are not what is stored: the reduction pass takes the maximum)
        pe.kchunk = plan.kchunk; pe.split_stride = (long long)M * ldw * 4;
        const dim3 grid_r((unsigned)(((M + RM - 1) / RM) * ((N + RN - 1) / RN)), (unsigned)plan.splits);
        const dim3 grid_b((unsigned)(((M + GM - 1) / GM) * ((N + GN - 1) / GN)), (unsigned)plan.splits);
        if (plan.variant == 3) {
            const size_t big_lds = (size_t)5 * GM * kChunksPerRow * sizeof(uint4);      // three-slot B ring
            ODW_CHECK_HIP(odw_set_max_lds(reinterpret_cast<const void*>(gemm_nt_bf16_big_kernel<false, 7>),
                                              (int)big_lds), "big attr");
            gemm_nt_bf16_big_kernel<false, 7><<<grid_b, kBigThreads, big_lds, stream>>>(
                (const unsigned short*)A, lda, (const unsigned short*)B, ldb, M, N, K, workspace, ldw, pe,
                (M + GM - 1) / GM, (N + GN - 1) / GN);
        } else {
            const size_t ring_lds = (size_t)kRingStages * kRingStageChunks * sizeof(uint4);
            ODW_CHECK_HIP(odw_set_max_lds(reinterpret_cast<const void*>(gemm_nt_bf16_ring_kernel<false>),
                                              (int)ring_lds), "ring attr");
            gemm_nt_bf16_ring_kernel<false><<<grid_r, kRingThreads, ring_lds, stream>>>(
                (const unsigned short*)A, lda, (const unsigned short*)B, ldb, M, N, K, workspace, ldw, pe,
                (M + RM - 1) / RM, (N + RN - 1) / RN);
        }
        ODW_CHECK_HIP(hipGetLastError(), "gemm_nt_bf16 split-K launch");
        const long long quads = (long long)M * (ldw / 4);
        const int rblocks = (int)((quads + 255) / 256 < 4096 ? (quads + 255) / 256 : 4096);
        if (c_is_bf16)
            splitk_reduce_kernel<true><<<rblocks, 256, 0, stream>>>((const float*)workspace, plan.splits,
                                                                     (long long)M * ldw, M, N, ldw, C, ldc, ep);
        else
            splitk_reduce_kernel<false><<<rblocks, 256, 0, stream>>>((const float*)workspace, plan.splits,
                                                                      (long long)M * ldw, M, N, ldw, C, ldc, ep);
        ODW_CHECK_LAUNCH("splitk_reduce_kernel");
        return ODW_OK;
    }
    const int variant = plan.variant;
    const bool use_big = variant == 3, use_ring = variant == 2, use_glds = variant >= 1;
    const int rtiles_m = (M + RM - 1) / RM, rtiles_n = (N + RN - 1) / RN;
    const int btiles_m = (M + GM - 1) / GM, btiles_n = (N + GN - 1) / GN;
    if (use_big) {
        const size_t big_lds = (size_t)2 * kBigStageChunks * sizeof(uint4);   // 128 KB
        static const int ring3 = !getenv("ODW_GEMM_EXP") || atoi(getenv("ODW_GEMM_EXP")) == 7;
        if (ring3) {                      // the default: three-slot B ring, 160 KB of LDS (ODW_GEMM_EXP=0: two slots)
            const size_t lds7 = (size_t)5 * GM * kChunksPerRow * sizeof(uint4);
            if (c_is_bf16) {
                ODW_CHECK_HIP(odw_set_max_lds(reinterpret_cast<const void*>(gemm_nt_bf16_big_kernel<true, 7>),
                                                  (int)lds7), "big7 attr");
                gemm_nt_bf16_big_kernel<true, 7><<<btiles_m * btiles_n, kBigThreads, lds7, stream>>>(
                    (const unsigned short*)A, lda, (const unsigned short*)B, ldb, M, N, K, C, ldc, ep, btiles_m, btiles_n);
            } else {
                ODW_CHECK_HIP(odw_set_max_lds(reinterpret_cast<const void*>(gemm_nt_bf16_big_kernel<false, 7>),
                                                  (int)lds7), "big7 attr");
                gemm_nt_bf16_big_kernel<false, 7><<<btiles_m * btiles_n, kBigThreads, lds7, stream>>>(
                    (const unsigned short*)A, lda, (const unsigned short*)B, ldb, M, N, K, C, ldc, ep, btiles_m, btiles_n);
            }
            ODW_CHECK_HIP(hipGetLastError(), "gemm_nt_bf16 big7 launch");
            return 0;
        }
        if (c_is_bf16) {
            ODW_CHECK_HIP(odw_set_max_lds(reinterpret_cast<const void*>(gemm_nt_bf16_big_kernel<true>),
                                              (int)big_lds), "big attr");
            gemm_nt_bf16_big_kernel<true><<<btiles_m * btiles_n, kBigThreads, big_lds, stream>>>(
                (const unsigned short*)A, lda, (const unsigned short*)B, ldb, M, N, K, C, ldc, ep, btiles_m, btiles_n);
        } else {
#ifdef ODW_EXPERIMENTS      // x = 1..6: timing experiments that knowingly produce WRONG results (experiment builds only)
            const char* xe = getenv("ODW_GEMM_EXP");
            const int x = xe ? atoi(xe) : 0;
#else
            const int x = 0;
#endif
#define ODW_BIG_X(XV)                                                                                            \
            do {                                                                                                 \
                ODW_CHECK_HIP(odw_set_max_lds(reinterpret_cast<const void*>(gemm_nt_bf16_big_kernel<false, XV>), \
                                                  (int)big_lds), "big attr"); \
                gemm_nt_bf16_big_kernel<false, XV><<<btiles_m * btiles_n, kBigThreads, big_lds, stream>>>(       \
                    (const unsigned short*)A, lda, (const unsigned short*)B, ldb, M, N, K, C, ldc, ep, btiles_m, btiles_n); \
            } while (0)
            if (x == 1) ODW_BIG_X(1); else if (x == 2) ODW_BIG_X(2); else if (x == 3) ODW_BIG_X(3);
            else if (x == 4) ODW_BIG_X(4); else if (x == 5) ODW_BIG_X(5); else if (x == 6) ODW_BIG_X(6); else ODW_BIG_X(0);
#undef ODW_BIG_X
        }
        ODW_CHECK_HIP(hipGetLastError(), "gemm_nt_bf16 big launch");
        return 0;
    }
#define ODW_LAUNCH_GEMM(KERNEL, OUTBF)                                                                          \
    do {                                                                                                        \
        ODW_CHECK_HIP(odw_set_max_lds(reinterpret_cast<const void*>(KERNEL<OUTBF>),                         \
                                          (int)lds_bytes), "gemm attr"); \
        KERNEL<OUTBF><<<tiles_m * tiles_n, kThreads, lds_bytes, stream>>>(                                      \
            (const unsigned short*)A, lda, (const unsigned short*)B, ldb, M, N, K, C, ldc, ep, tiles_m, tiles_n); \
    } while (0)
    if (use_ring) {
        const size_t ring_lds = (size_t)kRingStages * kRingStageChunks * sizeof(uint4);   // 144 KB
        if (c_is_bf16) {
            ODW_CHECK_HIP(odw_set_max_lds(reinterpret_cast<const void*>(gemm_nt_bf16_ring_kernel<true>),
                                              (int)ring_lds), "ring attr");
            gemm_nt_bf16_ring_kernel<true><<<rtiles_m * rtiles_n, kRingThreads, ring_lds, stream>>>(
                (const unsigned short*)A, lda, (const unsigned short*)B, ldb, M, N, K, C, ldc, ep, rtiles_m, rtiles_n);
        } else {
            ODW_CHECK_HIP(odw_set_max_lds(reinterpret_cast<const void*>(gemm_nt_bf16_ring_kernel<false>),
                                              (int)ring_lds), "ring attr");
            gemm_nt_bf16_ring_kernel<false><<<rtiles_m * rtiles_n, kRingThreads, ring_lds, stream>>>(
                (const unsigned short*)A, lda, (const unsigned short*)B, ldb, M, N, K, C, ldc, ep, rtiles_m, rtiles_n);
        }
    } else if (use_glds) {
        if (c_is_bf16) ODW_LAUNCH_GEMM(gemm_nt_bf16_glds_kernel, true);
        else ODW_LAUNCH_GEMM(gemm_nt_bf16_glds_kernel, false);
    } else {
        if (c_is_bf16) ODW_LAUNCH_GEMM(gemm_nt_bf16_kernel, true);
        else ODW_LAUNCH_GEMM(gemm_nt_bf16_kernel, false);
    }
#undef ODW_LAUNCH_GEMM
    ODW_CHECK_LAUNCH("gemm_nt_bf16_kernel");
    return ODW_OK;
}

// ---- convolution weight gradient: dW = dZ^T im2col(X) as ONE call (partial products + reduce-and-unpack) ----------
// dzt (Co x ld, K-contiguous bf16) and colt ((9 Cp) x ld) are the operands odw_linear_bwd_prep / odw_im2col_t_bf16 write;
// dw = the parameter gradient in torch's (Co, Ci, 3, 3) fp32 layout.  workspace: odw_conv_wgrad_workspace bytes.
ODW_EXPORT int64_t odw_conv_wgrad_workspace(int Co, int Cp, int K, int lda, int ldb) {
    const int N = 9 * Cp, ldw = (N + 3) / 4 * 4;
    const Plan p = pick_plan(Co, N, K, lda, ldb, nullptr, ldw, 0, true);
    return (int64_t)(p.splits > 1 ? p.splits : 1) * Co * ldw * 4;
}

ODW_EXPORT int odw_conv_wgrad_nt(const void* dzt, int lda, const void* colt, int ldb, int Co, int Ci, int Cp, int K,
                                 float* dw, int accumulate, void* workspace, int64_t workspace_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    ODW_REQUIRE(Co > 0 && Ci > 0 && Cp >= Ci && K > 0 && dzt && colt && dw && workspace, "conv_wgrad_nt: bad arguments");
    ODW_REQUIRE((((uintptr_t)workspace) & 15) == 0, "conv_wgrad_nt: workspace must be 16-byte aligned");
    const int N = 9 * Cp, ldw = (N + 3) / 4 * 4;
    Plan plan = pick_plan(Co, N, K, lda, ldb, workspace, ldw, 0, true);
    const int S = plan.splits > 1 ? plan.splits : 1;
    ODW_REQUIRE(workspace_bytes >= (int64_t)S * Co * ldw * 4, "conv_wgrad_nt: workspace of %lld bytes, need %lld",
                (long long)workspace_bytes, (long long)S * Co * ldw * 4);
    if (S > 1) {
        // the partial products, exactly as odw_gemm_nt_bf16_ws launches them
        Epilogue pe;
        pe.bias = nullptr; pe.relu = 0; pe.drop_p = 0.0f; pe.nseg = 0; pe.accumulate = 0; pe.alpha = 1.0f;
        pe.mask = nullptr; pe.ldmask = 0; pe.pm = 0; pe.row_ids = nullptr;
        for (int i = 0; i < kMaxSeg; ++i) { pe.seg_row[i] = 0; pe.seg_k0[i] = 0; pe.seg_k1[i] = 0; }
        pe.kchunk = plan.kchunk; pe.split_stride = (long long)Co * ldw * 4;
        if (plan.variant == 3) {
            const size_t big_lds = (size_t)5 * GM * kChunksPerRow * sizeof(uint4);
            const dim3 grid((unsigned)(((Co + GM - 1) / GM) * ((N + GN - 1) / GN)), (unsigned)S);
            ODW_CHECK_HIP(odw_set_max_lds(reinterpret_cast<const void*>(gemm_nt_bf16_big_kernel<false, 7>),
                                              (int)big_lds), "big attr");
            gemm_nt_bf16_big_kernel<false, 7><<<grid, kBigThreads, big_lds, stream>>>(
                (const unsigned short*)dzt, lda, (const unsigned short*)colt, ldb, Co, N, K, workspace, ldw, pe,
                (Co + GM - 1) / GM, (N + GN - 1) / GN);
        } else {
            const size_t ring_lds = (size_t)kRingStages * kRingStageChunks * sizeof(uint4);
            const dim3 grid((unsigned)(((Co + RM - 1) / RM) * ((N + RN - 1) / RN)), (unsigned)S);
            ODW_CHECK_HIP(odw_set_max_lds(reinterpret_cast<const void*>(gemm_nt_bf16_ring_kernel<false>),
                                              (int)ring_lds), "ring attr");
            gemm_nt_bf16_ring_kernel<false><<<grid, kRingThreads, ring_lds, stream>>>(
                (const unsigned short*)dzt, lda, (const unsigned short*)colt, ldb, Co, N, K, workspace, ldw, pe,
                (Co + RM - 1) / RM, (N + RN - 1) / RN);
        }
        ODW_CHECK_HIP(hipGetLastError(), "conv_wgrad_nt partial launch");
    } else {
        const int rc = odw_gemm_nt_bf16_ws(dzt, lda, colt, ldb, Co, N, K, workspace, ldw, 0, nullptr, 0, 1.0f, 0.0f, 0, nullptr,
                                           nullptr, nullptr, 0, nullptr, 0, stream_);
        if (rc != ODW_OK) return rc;
    }
    const long long units = (long long)Co * ((Ci + 63) / 64);
    const int rblocks = (int)(units < 16384 ? units : 16384);
    wgrad_reduce_unpack_kernel<<<rblocks, 256, 0, stream>>>((const float*)workspace, S, (long long)Co * ldw, Co, Ci, Cp, ldw,
                                                            dw, accumulate);
    ODW_CHECK_LAUNCH("wgrad_reduce_unpack_kernel");
    return ODW_OK;
}

// ---- convolution weight gradient, output-stationary "halo" form (round 3) ---------------------------------------------
// dW[co][tap][ci] = sum over pixels p of dZ[p][co] X[p + shift(tap)][ci].  gemm_tn_bf16_ring_kernel above treats the nine
// taps as nine column blocks of one long GEMM: every 64-pixel K step moves 48 KB L2 -> LDS for 4.2 MFLOP, X nine times
// over (87 FLOP / B: the kernel sat at 18 % MFMA busy, bound by the CU's 64 B/clk of operand delivery).  Here a workgroup
// OWNS a (64 co) x (9 taps x 64 ci) block of dW in its accumulators -- 9 x 16 registers per lane -- and walks 16 x 16
// (dilation 2: 8 x 16) spatial tiles: per tile ONE dZ tile (pixels x 64 co) and ONE halo patch of X ((rows + 2 dil) x
// (16 + 2 dil) pixels x 64 ci) reach LDS, and all nine taps read the SAME patch at shifted pixel rows: 73 KB per
// 18.9 MFLOP = 258 FLOP / B.  Both operands stay K-major (pixel rows of 128 B, 16-byte chunk c of row r in slot
// c ^ 4 ((r >> 1) & 1): any four consecutive rows cover all 64 banks) and are read with ds_read_b64_tr_b16 like the
// ring form; a tile row of 16 pixels is one MFMA K block, and a tap's fragment is the patch at row
// (y + dy dil) PW + dx dil -- 16 consecutive patch pixels.  Waves: (co half) x (ci half) x (K half: the upper / lower tile
// rows); the two K halves are summed through LDS at the end.  K is split over workgroups by ranges of spatial tiles
// (fp32 partials + wgrad_reduce_unpack_kernel, as before).  Double-buffered DMA: tile t + 1 lands while t is computed.
struct WhGeom {
    int B, H, W, Cp, ld_dz, ldx, tiles_x, tiles_y, tiles_total, tiles_per_split, co_tiles, out_tiles, splits, chunk;
    long long split_stride;             // floats between the partial sums of consecutive splits
    int ldw;                            // 9 * Cp
    const unsigned short* zero;
    float* bias_ws;                     // [split][Co] partial column sums of dZ (the bias gradient), or null
    int Co;
};

// tools/exp/wgrad_timeline.hip compiles this file with ODW_WH_TIMELINE: lane 0 of every wave stamps wall_clock64() (100 MHz)
// at its phase boundaries of the first tiles: 4 t + {0: tile landed (after the barrier), 1: first reads + next DMA issued,
// 2: K loop done}; 30: before the K-half reduction, 31: exit
#ifdef ODW_WH_TIMELINE
__device__ long long g_wh_tl[1024 * 8 * 32];
#define WH_T(i) do { const int e_ = 4 * (t - t_begin) + (i); if (lane == 0 && e_ < 30) g_wh_tl[(wg * 8 + wave) * 32 + e_] = wall_clock64(); } while (0)
#define WH_TE(e) do { if (lane == 0) g_wh_tl[(wg * 8 + wave) * 32 + (e)] = wall_clock64(); } while (0)
#else
#define WH_T(i) do { } while (0)
#define WH_TE(e) do { } while (0)
#endif

template <int TR, int DIL>
struct WhCfg {
    static constexpr int PW = 16 + 2 * DIL, PH = TR + 2 * DIL;
    static constexpr int kARows = TR * 16;
    static constexpr int kPRows = (PH * PW + 7) / 8 * 8;
    static constexpr int kStageBytes = (kARows + kPRows) * 128;
    static constexpr int kLds = 2 * kStageBytes;
};

template <int TR, int DIL>
__global__ __launch_bounds__(512, 1) void conv_wgrad_halo_kernel(const unsigned short* __restrict__ dz,
                                                                 const unsigned short* __restrict__ X,
                                                                 float* __restrict__ ws, WhGeom g) {
    using C = WhCfg<TR, DIL>;
    extern __shared__ __attribute__((aligned(16))) unsigned char wh_lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave & 1, wi = (wave >> 1) & 1, wk = wave >> 2;
    // workgroup id -> (split, output tile): consecutive ids land on consecutive XCDs, so XCD k takes the k-th CONTIGUOUS
    // chunk of the split-major order -- its ~32 co-resident workgroups walk the same spatial tiles (one split, all co
    // blocks x a few ci blocks) and share them in ITS L2.  With (tile, split) as blockIdx every XCD pulled every dZ / X
    // tile of every split: 47 % L2 misses, 94 MB of fabric traffic per launch for 12 MB of operands, and the kernel ran
    // at the pace of those misses (4.1 us per tile against 2.4 us of MFMA time, whatever the LDS side did).
    const int wg = (int)(blockIdx.x % 8) * g.chunk + (int)(blockIdx.x / 8);
    const int per_split = g.out_tiles + (g.bias_ws ? g.co_tiles : 0);
    if (wg >= per_split * g.splits) return;
    const int split = wg / per_split, ot = wg - split * per_split;
    const bool bias_wg = ot >= g.out_tiles;          // the last co_tiles workgroups of a split: column sums of dZ only
    const int co0 = (bias_wg ? ot - g.out_tiles : ot % g.co_tiles) * 64, ci0 = bias_wg ? 0 : (ot / g.co_tiles) * 64;
    const int t_begin = split * g.tiles_per_split;
    const int t_end = t_begin + g.tiles_per_split < g.tiles_total ? t_begin + g.tiles_per_split : g.tiles_total;

    // ---- DMA: 8 rows x 8 chunks per wave instruction; lane -> (row, physical chunk); the swizzle is on the source chunk.
    // A wave issues NA + NP instructions per tile; everything about them that does not depend on the tile is computed
    // ONCE here (element offset inside the image, position inside the tile / patch): the first version recomputed rows,
    // a division by PW, bounds and 64-bit addresses per instruction and spent ~200 cycles on each -- 1 us per tile,
    // serial with 2.4 us of MFMAs (the same instructions issued by half the waves cost 7 us more per launch).
    constexpr int NA = C::kARows / 64, NP = (C::kPRows / 8 + 7) / 8;
    const int drow = lane >> 3, dpc = lane & 7;
    int a_off[NA], p_off[NP], p_yx[NP];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int r = (i * 8 + wave) * 8 + drow;
        a_off[i] = ((r >> 4) * g.W + (r & 15)) * g.ld_dz + (dpc ^ (4 * ((r >> 1) & 1))) * 8;
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int r = (i * 8 + wave) * 8 + drow;
        const int py = r / C::PW, px = r - py * C::PW;
        p_off[i] = ((py - DIL) * g.W + (px - DIL)) * g.ldx + (dpc ^ (4 * ((r >> 1) & 1))) * 8;
        p_yx[i] = r < C::PH * C::PW ? ((py - DIL) & 0xffff) | ((px - DIL) << 16) : 0x7fff7fff;      // rows past the patch: never valid
    }
    auto issue = [&](int t, unsigned char* stage) {
        const int per_img = g.tiles_x * g.tiles_y;
        const int img = t / per_img, tt = t - img * per_img;
        const int y0 = (tt / g.tiles_x) * TR, x0 = (tt % g.tiles_x) * 16;
        const size_t pix0 = (size_t)img * g.H * g.W + (size_t)y0 * g.W + x0;
        const unsigned short* abase = dz + pix0 * g.ld_dz + co0;
        const unsigned short* pbase = X + pix0 * g.ldx + ci0;
#pragma unroll
        for (int i = 0; i < NA; ++i) {                        // dZ tile: pixel rows (ty, tx), 64 co
            const int rbase = (i * 8 + wave) * 8;
            const int y = y0 + (rbase >> 4), x = x0 + (rbase & 15) + drow;
            const void* src = (y < g.H && x < g.W) ? static_cast<const void*>(abase + a_off[i]) : static_cast<const void*>(g.zero);
            __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(stage + rbase * 128), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NP; ++i) {                        // halo patch of X: (PH x PW) pixel rows, 64 ci
            const int j = i * 8 + wave;
            if (j < C::kPRows / 8 && !bias_wg) {
                const int y = y0 + (short)(p_yx[i] & 0xffff), x = x0 + (p_yx[i] >> 16);
                const void* src = ((unsigned)y < (unsigned)g.H && (unsigned)x < (unsigned)g.W)
                                      ? static_cast<const void*>(pbase + p_off[i]) : static_cast<const void*>(g.zero);
                __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(stage + (C::kARows + j * 8) * 128), 16, 0, 0);
            }
        }
    };

    // ---- fragment addressing (tn_frag: lane supplies row kq (+4) of the 16-pixel K block, 8-byte piece `sub` of chunk cb)
    const unsigned kq = 8u * (lane >> 5) + ((lane & 15) >> 2);
    const unsigned cb = 2u * ((lane >> 4) & 1) + ((lane & 3) >> 1);
    const unsigned sub = 8u * (lane & 1);
    const unsigned offa = kq * 128u + (((unsigned)(wc * 4) + cb) ^ (4u * ((kq >> 1) & 1))) * 16u + sub;
    const unsigned cbi = (unsigned)(wi * 4) + cb;

    if (bias_wg) {
        // ---- the bias gradient db[co] = sum over pixels of dZ[p][co], on the matrix pipe: dZ^T x ones.  Every column of the
        // 32 x 32 result holds the same 32 sums; wave = (co half) x (quarter of the tile rows).  One summation order
        // (MFMA K order, tiles in order, row quarters and splits added in index order): identical from run to run.
        // (Was two launches per layer -- colsum_bf16_part / _finish -- that re-read dZ: 18 launches, ~0.1 ms per step.)
        f32x16 accb = (f32x16){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        const bf16x8 ones = __builtin_bit_cast(bf16x8, make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u));
        const int rq = wave >> 1;                          // rows [rq TR / 4, (rq + 1) TR / 4)
        int bstage = 0;
        if (t_begin < t_end) issue(t_begin, wh_lds);
        for (int t = t_begin; t < t_end; ++t) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (t + 1 < t_end) issue(t + 1, wh_lds + (bstage ^ 1) * C::kStageBytes);
            const unsigned char* sa = wh_lds + bstage * C::kStageBytes;
#pragma unroll
            for (int kb = 0; kb < TR / 4; ++kb)
                accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tn_frag<128>(sa, offa + (unsigned)(rq * (TR / 4) + kb) * 16u * 128u),
                                                               ones, accb, 0, 0, 0);
            bstage ^= 1;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        float* red = reinterpret_cast<float*>(wh_lds);       // [row quarter][64 co]
        if ((lane & 31) == 0) {
#pragma unroll
            for (int k = 0; k < 16; ++k) red[rq * 64 + wc * 32 + (k & 3) + 8 * (k >> 2) + 4 * (lane >> 5)] = accb[k];
        }
        __syncthreads();
        if (tid < 64) g.bias_ws[(size_t)split * g.Co + co0 + tid] = ((red[tid] + red[64 + tid]) + red[128 + tid]) + red[192 + tid];
        return;
    }

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = (f32x16){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

    int stage = 0;
    if (t_begin < t_end) issue(t_begin, wh_lds);
    for (int t = t_begin; t < t_end; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // tile t has landed (this wave's pieces) ...
        __builtin_amdgcn_s_barrier();                         // ... everyone's; and everyone is done with the other stage
        asm volatile("" ::: "memory");
        WH_T(0);
        const unsigned char* sa = wh_lds + stage * C::kStageBytes;
        const unsigned char* sp = sa + C::kARows * 128;
        // A tap's fragment at tile row y is the patch at pixel row y + dy DIL: the fragments of one patch row serve
        // three tile rows, so the wave keeps a rolling window of 2 DIL + 1 patch rows x 3 column shifts in registers
        // and reads THREE new fragments per K block instead of nine (ds_read_b64_tr_b16 runs at half the plain b64
        // rate: with nine the kernel was LDS-bound at 5.4 us per tile against 2.4 us of MFMA time).
        // Nine fragment reads per K block, straight from the patch.  (A rolling window of three patch rows x three column
        // shifts in registers -- three new fragments per K block -- ran the same 37 us: the K loop is MFMA-bound either
        // way.  It cost 50 more registers, and at 241 two waves per SIMD leave no room for a wave of the optimiser's
        // side-stream kernel, which shares the chip with the backbone's backward: the two kernels took turns on every CU,
        // 58 / 118 us per launch inside the step against 26-39 us alone.  At 190 (2 x 192 + 64 <= 512) they co-reside.)
        const int y_first = wk * (TR / 2);
        if (t + 1 < t_end) issue(t + 1, wh_lds + (stage ^ 1) * C::kStageBytes);
        WH_T(1);
#pragma unroll 1
        for (int kb = 0; kb < TR / 2; ++kb) {
            const int y = y_first + kb;
            const bf16x8 fa = tn_frag<128>(sa, offa + (unsigned)y * 16u * 128u);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const unsigned row = (unsigned)((y + (tap / 3) * DIL) * C::PW + (tap % 3) * DIL) + kq;
                const bf16x8 fb = tn_frag<128>(sp, row * 128u + ((cbi ^ (4u * ((row >> 1) & 1))) * 16u) + sub);
                acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[tap], 0, 0, 0);
            }
        }
        WH_T(2);
        stage ^= 1;
    }
    WH_TE(30);

    // ---- the two K halves summed through LDS, three taps per round (12 KB per wave pair), then the partial sums out:
    // accumulator register k of lane (half, n) is dW[co0 + 32 wc + crow(k)][tap][ci0 + 32 wi + n]: 128-byte row segments
    float* red = reinterpret_cast<float*>(wh_lds) + (size_t)(wave & 3) * (3 * 16 * 64);
    const int half = lane >> 5, l31 = lane & 31;
    float* out = ws + (size_t)split * g.split_stride + (size_t)(co0 + wc * 32 + 4 * half) * g.ldw + ci0 + wi * 32 + l31;
#pragma unroll
    for (int round = 0; round < 3; ++round) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (wk == 1) {
#pragma unroll
            for (int tt = 0; tt < 3; ++tt)
#pragma unroll
                for (int k = 0; k < 16; ++k) red[(tt * 16 + k) * 64 + lane] = acc[round * 3 + tt][k];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (wk == 0) {
#pragma unroll
            for (int tt = 0; tt < 3; ++tt) {
                const int tap = round * 3 + tt;
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const float v = acc[tap][k] + red[(tt * 16 + k) * 64 + lane];
                    out[(size_t)((k & 3) + 8 * (k >> 2)) * g.ldw + (size_t)tap * g.Cp] = v;
                }
            }
        }
    }
    WH_TE(31);
}

// ---- convolution weight gradient WITHOUT the transposed operands: dW = dZ^T im2col(X) on gemm_tn_bf16_ring_kernel --------
// dz (n_pix x ld_dz, the masked output gradient, NHWC bf16) and X (n_pix x Cp, the layer input, NHWC bf16) are read as
// they are: no dZ^T, no 9x transposed im2col.  Needs Cp a power of two >= 128 (one tap per 128-column tile).
namespace {
struct TnPlan { int tiles_m, tiles_n, splits, kchunk; };
TnPlan conv_wgrad_tn_plan(int Co, int Cp, int K) {
    TnPlan p;
    p.tiles_m = (Co + RM - 1) / RM;
    p.tiles_n = 9 * Cp / RN;
    int sp = 256 / (p.tiles_m * p.tiles_n);
    sp = sp < 1 ? 1 : (sp > 32 ? 32 : sp);
    const char* f = getenv("ODW_GEMM_SPLITK");
    if (f && atoi(f) > 0) sp = atoi(f);
    p.kchunk = ((K + sp - 1) / sp + 63) / 64 * 64;
    if (p.kchunk < 256) p.kchunk = 256;
    p.splits = (K + p.kchunk - 1) / p.kchunk;
    return p;
}
}  // namespace

// the halo form's split of the spatial tiles (see conv_wgrad_halo_kernel): ~256 workgroups, one per CU
struct WhPlan { int tr, tiles_x, tiles_y, tiles_total, tiles_per_split, splits; };
static WhPlan conv_wgrad_halo_plan(int Co, int Cp, int B, int H, int W, int dilation) {
    WhPlan p;
    p.tr = dilation == 1 ? 16 : 8;
    p.tiles_x = (W + 15) / 16;
    p.tiles_y = (H + p.tr - 1) / p.tr;
    p.tiles_total = B * p.tiles_x * p.tiles_y;
    const int out_tiles = (Co / 64) * (Cp / 64);
    int sp = (ODW_NUM_CU + out_tiles - 1) / out_tiles;
    sp = sp < 1 ? 1 : (sp > p.tiles_total ? p.tiles_total : sp);
    p.tiles_per_split = (p.tiles_total + sp - 1) / sp;
    p.splits = (p.tiles_total + p.tiles_per_split - 1) / p.tiles_per_split;
    return p;
}
static bool conv_wgrad_halo_ok(int Co, int Cp, int dilation) {
    static const bool off = getenv("ODW_WGRAD_HALO") && atoi(getenv("ODW_WGRAD_HALO")) == 0;
    return !off && Co % 64 == 0 && Cp % 64 == 0 && (dilation == 1 || dilation == 2);
}

// H, W unknown here: the halo form never needs more than the ring form's plan times 2 (asserted at launch)
ODW_EXPORT int64_t odw_conv_wgrad_tn_workspace(int Co, int Cp, int n_pix) {
    if (Co <= 0 || Cp < 128 || n_pix <= 0) return 0;
    const TnPlan p = conv_wgrad_tn_plan(Co, Cp, n_pix);
    int64_t splits = p.splits;
    if (conv_wgrad_halo_ok(Co, Cp, 1)) {
        const int out_tiles = (Co / 64) * (Cp / 64);
        const int64_t sp = (ODW_NUM_CU + out_tiles - 1) / out_tiles;
        splits = sp > splits ? sp : splits;
    }
    return splits * Co * 9 * Cp * 4;
}

ODW_EXPORT int odw_colsum_bf16_ws(const void* X, int ld, int M, int N, float* out, void* workspace, int64_t workspace_bytes,
                                  void* stream_);
ODW_EXPORT int64_t odw_colsum_workspace(int M, int N);
static int conv_wgrad_tn_impl(const void* dz, int ld_dz, const void* X, int ldx, int n_pix, int H, int W, int Cp, int dilation,
                              int Co, int Ci, float* dw, float* db, int accumulate, const void* zero_page, void* workspace,
                              int64_t workspace_bytes, void* stream_);

ODW_EXPORT int odw_conv_wgrad_tn(const void* dz, int ld_dz, const void* X, int n_pix, int H, int W, int Cp, int dilation,
                                 int Co, int Ci, float* dw, int accumulate, const void* zero_page, void* workspace,
                                 int64_t workspace_bytes, void* stream_) {
    return conv_wgrad_tn_impl(dz, ld_dz, X, Cp, n_pix, H, W, Cp, dilation, Co, Ci, dw, nullptr, accumulate, zero_page, workspace,
                              workspace_bytes, stream_);
}

// The same with the layer's BIAS gradient: db[co] += sum over the pixels of dz[p][co] (added onto what db holds, one
// summation order).  The halo form computes it in extra workgroups of the same launch (dZ^T x ones on the matrix
// pipe); otherwise the two-launch column sum runs beside the ring form.  Workspace: odw_conv_wgrad_tn_bias_workspace.
ODW_EXPORT int64_t odw_conv_wgrad_tn_bias_workspace(int Co, int Cp, int n_pix) {
    const int64_t a = odw_align_up(odw_conv_wgrad_tn_workspace(Co, Cp, n_pix), 256);
    if (a == 0) return 0;
    int64_t splits = odw_conv_wgrad_tn_workspace(Co, Cp, n_pix) / ((int64_t)Co * 9 * Cp * 4);
    int64_t b = splits * Co * 4, c = odw_colsum_workspace(n_pix, Co);
    return a + odw_align_up(b > c ? b : c, 256);
}
ODW_EXPORT int odw_conv_wgrad_tn_bias(const void* dz, int ld_dz, const void* X, int ldx, int n_pix, int H, int W, int Cp,
                                      int dilation, int Co, int Ci, float* dw, float* db, int accumulate, const void* zero_page,
                                      void* workspace, int64_t workspace_bytes, void* stream_) {
    ODW_REQUIRE(db, "conv_wgrad_tn_bias: null bias gradient");
    ODW_REQUIRE(ldx >= Cp && ldx % 8 == 0, "conv_wgrad_tn_bias: ldx=%d (row stride of X in elements: >= Cp, a multiple of 8)", ldx);
    ODW_REQUIRE(workspace_bytes >= odw_conv_wgrad_tn_bias_workspace(Co, Cp, n_pix), "conv_wgrad_tn_bias: workspace of %lld bytes, need %lld",
                (long long)workspace_bytes, (long long)odw_conv_wgrad_tn_bias_workspace(Co, Cp, n_pix));
    return conv_wgrad_tn_impl(dz, ld_dz, X, ldx, n_pix, H, W, Cp, dilation, Co, Ci, dw, db, accumulate, zero_page, workspace,
                              workspace_bytes, stream_);
}

static int conv_wgrad_tn_impl(const void* dz, int ld_dz, const void* X, int ldx, int n_pix, int H, int W, int Cp, int dilation,
                              int Co, int Ci, float* dw, float* db, int accumulate, const void* zero_page, void* workspace,
                              int64_t workspace_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    // (with db: the tail of the workspace, behind the weight-gradient partials, holds the bias partials / column-sum scratch)
    const int64_t main_bytes = db ? odw_align_up(odw_conv_wgrad_tn_workspace(Co, Cp, n_pix), 256) : workspace_bytes;
    void* tail = db ? (void*)((char*)workspace + main_bytes) : nullptr;
    const int64_t tail_bytes = db ? workspace_bytes - main_bytes : 0;
    if (db) workspace_bytes = main_bytes;
    ODW_REQUIRE(n_pix > 0 && H > 0 && W > 0 && n_pix % (H * W) == 0 && Co > 0 && Ci > 0 && Cp >= Ci, "conv_wgrad_tn: bad dims");
    ODW_REQUIRE(Cp >= 128 && (Cp & (Cp - 1)) == 0, "conv_wgrad_tn: Cp=%d must be a power of two >= 128", Cp);
    ODW_REQUIRE(Co % 8 == 0 && ld_dz % 8 == 0 && ld_dz >= Co, "conv_wgrad_tn: Co=%d / ld_dz=%d must be multiples of 8", Co, ld_dz);
    ODW_REQUIRE(dilation >= 1 && dilation <= 4, "conv_wgrad_tn: dilation %d", dilation);
    ODW_REQUIRE(dz && X && dw && zero_page && workspace, "conv_wgrad_tn: null pointer");
    ODW_REQUIRE((((uintptr_t)dz) & 15) == 0 && (((uintptr_t)X) & 15) == 0 && (((uintptr_t)zero_page) & 15) == 0 &&
                    (((uintptr_t)workspace) & 15) == 0, "conv_wgrad_tn: 16-byte alignment");
    const int N = 9 * Cp;
    if (conv_wgrad_halo_ok(Co, Cp, dilation)) {
        const WhPlan hp = conv_wgrad_halo_plan(Co, Cp, n_pix / (H * W), H, W, dilation);
        if (workspace_bytes >= (int64_t)hp.splits * Co * N * 4) {
            WhGeom hg;
            hg.B = n_pix / (H * W); hg.H = H; hg.W = W; hg.Cp = Cp; hg.ld_dz = ld_dz; hg.ldx = ldx;
            hg.tiles_x = hp.tiles_x; hg.tiles_y = hp.tiles_y; hg.tiles_total = hp.tiles_total;
            hg.tiles_per_split = hp.tiles_per_split; hg.co_tiles = Co / 64; hg.out_tiles = (Co / 64) * (Cp / 64);
            hg.splits = hp.splits;
            hg.bias_ws = db ? (float*)tail : nullptr; hg.Co = Co;
            hg.chunk = ((hg.out_tiles + (db ? hg.co_tiles : 0)) * hp.splits + 7) / 8;
            hg.split_stride = (long long)Co * N; hg.ldw = N; hg.zero = (const unsigned short*)zero_page;
            const dim3 grid((unsigned)(8 * hg.chunk));
            if (dilation == 1) {
                const hipError_t a1 = odw_set_max_lds(reinterpret_cast<const void*>(conv_wgrad_halo_kernel<16, 1>),
                                                                 WhCfg<16, 1>::kLds);
                ODW_CHECK_HIP(a1, "wgrad halo attr");
                conv_wgrad_halo_kernel<16, 1><<<grid, 512, WhCfg<16, 1>::kLds, stream>>>(
                    (const unsigned short*)dz, (const unsigned short*)X, (float*)workspace, hg);
            } else {
                const hipError_t a2 = odw_set_max_lds(reinterpret_cast<const void*>(conv_wgrad_halo_kernel<8, 2>),
                                                                 WhCfg<8, 2>::kLds);
                ODW_CHECK_HIP(a2, "wgrad halo attr");
                conv_wgrad_halo_kernel<8, 2><<<grid, 512, WhCfg<8, 2>::kLds, stream>>>(
                    (const unsigned short*)dz, (const unsigned short*)X, (float*)workspace, hg);
            }
            ODW_CHECK_HIP(hipGetLastError(), "conv_wgrad_halo launch");
            const long long units = (long long)Co * ((Ci + 63) / 64);
            const int rblocks = (int)(units < 16384 ? units : 16384);
            wgrad_reduce_unpack_kernel<<<rblocks, 256, 0, stream>>>((const float*)workspace, hp.splits, (long long)Co * N, Co, Ci,
                                                                    Cp, N, dw, accumulate, db ? (const float*)tail : nullptr, db);
            ODW_CHECK_LAUNCH("wgrad_reduce_unpack_kernel");
            return ODW_OK;
        }
    }
    ODW_REQUIRE(ldx == Cp, "conv_wgrad_tn: the ring form reads X with row stride Cp (got %d, Cp = %d)", ldx, Cp);
    const TnPlan plan = conv_wgrad_tn_plan(Co, Cp, n_pix);
    ODW_REQUIRE(workspace_bytes >= (int64_t)plan.splits * Co * N * 4, "conv_wgrad_tn: workspace of %lld bytes, need %lld",
                (long long)workspace_bytes, (long long)plan.splits * Co * N * 4);
    ConvGeom g;
    g.H = H; g.W = W; g.C = Cp; g.dil = dilation; g.sign = 1; g.zero = (const unsigned short*)zero_page;
    g.logC = 0;
    while ((1 << g.logC) < Cp) ++g.logC;
    Epilogue pe;
    pe.bias = nullptr; pe.relu = 0; pe.drop_p = 0.0f; pe.nseg = 0; pe.accumulate = 0; pe.alpha = 1.0f;
    pe.mask = nullptr; pe.ldmask = 0; pe.pm = 0; pe.row_ids = nullptr;
    for (int i = 0; i < kMaxSeg; ++i) { pe.seg_row[i] = 0; pe.seg_k0[i] = 0; pe.seg_k1[i] = 0; }
    pe.kchunk = plan.kchunk; pe.split_stride = (long long)Co * N * 4;
    const size_t ring_lds = (size_t)kRingStages * kRingStageChunks * sizeof(uint4);
    ODW_CHECK_HIP(odw_set_max_lds(reinterpret_cast<const void*>(gemm_tn_bf16_ring_kernel<true>),
                                      (int)ring_lds), "tn attr");
    gemm_tn_bf16_ring_kernel<true><<<dim3((unsigned)(plan.tiles_m * plan.tiles_n), (unsigned)plan.splits), kRingThreads,
                                     ring_lds, stream>>>((const unsigned short*)dz, ld_dz, (const unsigned short*)X, Cp, Co, N,
                                                         n_pix, workspace, N, pe, plan.tiles_m, plan.tiles_n, g);
    ODW_CHECK_HIP(hipGetLastError(), "conv_wgrad_tn launch");
    const long long units = (long long)Co * ((Ci + 63) / 64);
    const int rblocks = (int)(units < 16384 ? units : 16384);
    wgrad_reduce_unpack_kernel<<<rblocks, 256, 0, stream>>>((const float*)workspace, plan.splits, (long long)Co * N, Co, Ci, Cp,
                                                            N, dw, accumulate);
    ODW_CHECK_LAUNCH("wgrad_reduce_unpack_kernel");
    if (db) return odw_colsum_bf16_ws(dz, ld_dz, n_pix, Co, db, tail, tail_bytes, stream_);
    return ODW_OK;
}

ODW_EXPORT int odw_transpose_to_bf16_part(const void* in, int in_is_f32, int ld_in, int R, int Cc, void* out, int ld_out,
                                          int out_cols, void* stream_);

ODW_EXPORT int odw_transpose_to_bf16(const void* in, int in_is_f32, int ld_in, int R, int Cc, void* out, int ld_out,
                                     void* stream_) {
    return odw_transpose_to_bf16_part(in, in_is_f32, ld_in, R, Cc, out, ld_out, ld_out, stream_);
}

static int transpose_to_bf16_launch(const void* in, int in_is_f32, int ld_in, int R, int Cc, void* out, int ld_out,
                                    int out_cols, void* stream_, DynRows dyn) {
    hipStream_t stream = (hipStream_t)stream_;
    ODW_REQUIRE(R >= 0 && Cc >= 0 && ld_in >= Cc && out_cols >= R && ld_out >= out_cols, "transpose_to_bf16: bad dims");
    if (R == 0 || Cc == 0) return ODW_OK;
    ODW_REQUIRE(in && out, "transpose_to_bf16: null pointer");
    if (!in_is_f32 && Cc % 8 == 0 && ld_in % 8 == 0 && ld_out % 8 == 0 && out_cols % 8 == 0 &&
        (((uintptr_t)in) & 15) == 0 && (((uintptr_t)out) & 15) == 0) {
        dim3 vgrid((Cc + 63) / 64, (out_cols + 63) / 64);
        if (dyn.r_dev && vgrid.y > kDynRowTiles) vgrid.y = kDynRowTiles;
        transpose_bf16_vec_kernel<<<vgrid, 256, 0, stream>>>((const unsigned short*)in, ld_in, R, Cc, (unsigned short*)out,
                                                             ld_out, out_cols, dyn);
        ODW_CHECK_LAUNCH("transpose_bf16_vec_kernel");
        return ODW_OK;
    }
    dim3 grid((Cc + 31) / 32, (out_cols + 31) / 32);   // covers the zero padding up to out_cols
    if (dyn.r_dev && grid.y > 2 * kDynRowTiles) grid.y = 2 * kDynRowTiles;
    if (in_is_f32)
        transpose_to_bf16_kernel<true><<<grid, 256, 0, stream>>>(in, ld_in, R, Cc, (unsigned short*)out, ld_out, out_cols, dyn);
    else
        transpose_to_bf16_kernel<false><<<grid, 256, 0, stream>>>(in, ld_in, R, Cc, (unsigned short*)out, ld_out, out_cols, dyn);
    ODW_CHECK_LAUNCH("transpose_to_bf16_kernel");
    return ODW_OK;
}

ODW_EXPORT int odw_transpose_to_bf16_part(const void* in, int in_is_f32, int ld_in, int R, int Cc, void* out, int ld_out,
                                          int out_cols, void* stream_) {
    return transpose_to_bf16_launch(in, in_is_f32, ld_in, R, Cc, out, ld_out, out_cols, stream_, DynRows{nullptr, nullptr, nullptr});
}

// out[c][*col_off_dev + r] = bf16(in[src_rows ? src_rows[r] : r][c]) for r < *r_dev, zeros up to r64(*r_dev): the X^T / dZ^T
// column block of an evaluation whose row count (and whose predecessors' widths in the batch) live on the device.
// R_cap rows bound the launch; `out` must hold *col_off_dev + r64(*r_dev) <= ld_out columns (the caller's capacity).
ODW_EXPORT int odw_transpose_to_bf16_dyn(const void* in, int in_is_f32, int ld_in, int R_cap, int Cc, void* out, int ld_out,
                                         const int* r_dev, const int* col_off_dev, const int* src_rows, void* stream_) {
    ODW_REQUIRE(r_dev, "transpose_to_bf16_dyn: r_dev is null");
    const int cols = (R_cap + 63) / 64 * 64;
    ODW_REQUIRE(ld_out >= cols, "transpose_to_bf16_dyn: ld_out=%d < r64(R_cap)=%d", ld_out, cols);
    return transpose_to_bf16_launch(in, in_is_f32, ld_in, R_cap, Cc, out, ld_out, cols, stream_, DynRows{r_dev, col_off_dev, src_rows});
}

ODW_EXPORT int odw_f32_to_bf16(const float* in, void* out, int64_t n, void* stream_) {
    ODW_REQUIRE(n >= 0, "f32_to_bf16: bad n");
    if (n == 0) return ODW_OK;
    ODW_REQUIRE(in && out && (((uintptr_t)in) & 15) == 0 && (((uintptr_t)out) & 7) == 0, "f32_to_bf16: pointers");
    size_t g = ((size_t)n / 4 + 255) / 256;
    f32_to_bf16_kernel<<<(int)(g < 1 ? 1 : (g > 4096 ? 4096 : g)), 256, 0, (hipStream_t)stream_>>>(in, (unsigned short*)out, (size_t)n);
    ODW_CHECK_LAUNCH("f32_to_bf16_kernel");
    return ODW_OK;
}

ODW_EXPORT int odw_linear_bwd_prep_part(const void* dY, int dy_is_f32, int ld_dy, const void* Y, int ld_y, int M, int N,
                                        float scale, void* dZ, int ld_z, void* dZT, int ld_t, int t_cols, float* db,
                                        void* stream_);

ODW_EXPORT int odw_linear_bwd_prep(const void* dY, int dy_is_f32, int ld_dy, const void* Y, int ld_y, int M, int N,
                                   float scale, void* dZ, int ld_z, void* dZT, int ld_t, float* db, void* stream_) {
    return odw_linear_bwd_prep_part(dY, dy_is_f32, ld_dy, Y, ld_y, M, N, scale, dZ, ld_z, dZT, ld_t, ld_t, db, stream_);
}

static int linear_bwd_prep_launch(const void* dY, int dy_is_f32, int ld_dy, const void* Y, int ld_y, int M, int N,
                                  float scale, void* dZ, int ld_z, void* dZT, int ld_t, int t_cols, float* db,
                                  void* stream_, DynRows dyn);

ODW_EXPORT int odw_linear_bwd_prep_part(const void* dY, int dy_is_f32, int ld_dy, const void* Y, int ld_y, int M, int N,
                                        float scale, void* dZ, int ld_z, void* dZT, int ld_t, int t_cols, float* db,
                                        void* stream_) {
    return linear_bwd_prep_launch(dY, dy_is_f32, ld_dy, Y, ld_y, M, N, scale, dZ, ld_z, dZT, ld_t, t_cols, db, stream_,
                                  DynRows{nullptr, nullptr, nullptr});
}

// The backward prologue of a Linear whose row count lives on the device: M_cap bounds the launch, *m_dev rows exist; dZ^T
// goes to column block [*tcol_off_dev, + r64(*m_dev)) of dZT (zero padded); y_rows: row m of the mask source is Y[y_rows[m]]
// (the saved output of a LARGER evaluation these rows are re-attached from -- no gathered copy of it).
ODW_EXPORT int odw_linear_bwd_prep_dyn(const void* dY, int dy_is_f32, int ld_dy, const void* Y, int ld_y, int M_cap, int N,
                                       float scale, void* dZ, int ld_z, void* dZT, int ld_t, float* db, const int* m_dev,
                                       const int* tcol_off_dev, const int* y_rows, void* stream_) {
    ODW_REQUIRE(m_dev, "linear_bwd_prep_dyn: m_dev is null");
    const int cols = (M_cap + 63) / 64 * 64;
    ODW_REQUIRE(ld_t >= cols, "linear_bwd_prep_dyn: ld_t=%d < r64(M_cap)=%d", ld_t, cols);
    return linear_bwd_prep_launch(dY, dy_is_f32, ld_dy, Y, ld_y, M_cap, N, scale, dZ, ld_z, dZT, ld_t, cols, db, stream_,
                                  DynRows{m_dev, tcol_off_dev, y_rows});
}

static int linear_bwd_prep_launch(const void* dY, int dy_is_f32, int ld_dy, const void* Y, int ld_y, int M, int N,
                                  float scale, void* dZ, int ld_z, void* dZT, int ld_t, int t_cols, float* db,
                                  void* stream_, DynRows dyn) {
    hipStream_t stream = (hipStream_t)stream_;
    ODW_REQUIRE(M >= 0 && N >= 0 && ld_z >= N && t_cols >= M && ld_t >= t_cols && ld_dy >= N, "linear_bwd_prep: bad dims");
    if (M == 0 || N == 0) return ODW_OK;
    ODW_REQUIRE(dY && dZ && dZT, "linear_bwd_prep: null pointer");
    // dy_is_f32: bit 0 = dY is fp32, bit 1 = Y (the saved output the mask is re-derived from) is fp32
    const bool dyf = (dy_is_f32 & 1) != 0, yf = (dy_is_f32 & 2) != 0;
    const bool vec = N % 8 == 0 && ld_dy % 8 == 0 && ld_z % 8 == 0 && ld_t % 8 == 0 && t_cols % 8 == 0 && (!Y || ld_y % 8 == 0) &&
                     (((uintptr_t)dY) & 15) == 0 && (((uintptr_t)Y) & 15) == 0 && (((uintptr_t)dZ) & 15) == 0 &&
                     (((uintptr_t)dZT) & 15) == 0;
#define ODW_PREP_LAUNCH(KERNEL, GRID)                                                                                  \
    do {                                                                                                               \
        if (dyf && yf) KERNEL<true, true><<<GRID, 256, 0, stream>>>(dY, ld_dy, Y, ld_y, M, N, scale, (unsigned short*)dZ, ld_z, (unsigned short*)dZT, ld_t, t_cols, db, dyn); \
        else if (dyf) KERNEL<true, false><<<GRID, 256, 0, stream>>>(dY, ld_dy, Y, ld_y, M, N, scale, (unsigned short*)dZ, ld_z, (unsigned short*)dZT, ld_t, t_cols, db, dyn); \
        else if (yf) KERNEL<false, true><<<GRID, 256, 0, stream>>>(dY, ld_dy, Y, ld_y, M, N, scale, (unsigned short*)dZ, ld_z, (unsigned short*)dZT, ld_t, t_cols, db, dyn); \
        else KERNEL<false, false><<<GRID, 256, 0, stream>>>(dY, ld_dy, Y, ld_y, M, N, scale, (unsigned short*)dZ, ld_z, (unsigned short*)dZT, ld_t, t_cols, db, dyn); \
    } while (0)
    if (vec) {
        dim3 vgrid((ld_z + 63) / 64, (t_cols + 63) / 64);
        if (dyn.r_dev && vgrid.y > kDynRowTiles) vgrid.y = kDynRowTiles;
        ODW_PREP_LAUNCH(linear_bwd_prep_vec_kernel, vgrid);
        ODW_CHECK_LAUNCH("linear_bwd_prep_vec_kernel");
        return ODW_OK;
    }
    dim3 grid((ld_z + 31) / 32, (t_cols + 31) / 32);     // covers the zero padding of both outputs
    if (dyn.r_dev && grid.y > 2 * kDynRowTiles) grid.y = 2 * kDynRowTiles;
    ODW_PREP_LAUNCH(linear_bwd_prep_kernel, grid);
#undef ODW_PREP_LAUNCH
    ODW_CHECK_LAUNCH("linear_bwd_prep_kernel");
    return ODW_OK;
}

ODW_EXPORT int odw_sgd_momentum_paced(float* p, const float* g, float* buf, void* shadow_bf16, int64_t n, float lr, float wd,
                                      float momentum, float grad_scale, int first_step, int max_workgroups,
                                      void* stream_);

ODW_EXPORT int odw_sgd_momentum(float* p, const float* g, float* buf, void* shadow_bf16, int64_t n, float lr, float wd,
                                float momentum, float grad_scale, int first_step, void* stream_) {
    return odw_sgd_momentum_paced(p, g, buf, shadow_bf16, n, lr, wd, momentum, grad_scale, first_step, 0, stream_);
}

ODW_EXPORT int odw_sgd_momentum_paced(float* p, const float* g, float* buf, void* shadow_bf16, int64_t n, float lr, float wd,
                                      float momentum, float grad_scale, int first_step, int max_workgroups,
                                      void* stream_) {
    ODW_REQUIRE(n >= 0, "sgd_momentum: bad n");
    if (n == 0) return ODW_OK;
    ODW_REQUIRE(p && g && buf, "sgd_momentum: null pointer");
    ODW_REQUIRE((((uintptr_t)p) & 15) == 0 && (((uintptr_t)g) & 15) == 0 && (((uintptr_t)buf) & 15) == 0 &&
                    (((uintptr_t)shadow_bf16) & 7) == 0, "sgd_momentum: buffers must be 16-byte aligned");
    size_t blocks = ((size_t)n / 4 + 255) / 256;
    static const int mode = getenv("ODW_SGD_MODE") ? atoi(getenv("ODW_SGD_MODE")) : 3;
    static const int env_cap = getenv("ODW_SGD_GRID") ? atoi(getenv("ODW_SGD_GRID")) : 65536;
    const int cap = max_workgroups > 0 ? max_workgroups : env_cap;
    const int grid = (int)(blocks < 1 ? 1 : (blocks > (size_t)cap ? (size_t)cap : blocks));
    if (mode && n % 4 == 0) {
        hipStream_t st = (hipStream_t)stream_;
        unsigned short* sh = (unsigned short*)shadow_bf16;
        if (mode == 1) sgd_kernel_x<1><<<grid, 256, 0, st>>>(p, g, buf, sh, (size_t)n / 4, lr, wd, momentum, grad_scale, first_step);
        else if (mode == 2) sgd_kernel_x<2><<<grid, 256, 0, st>>>(p, g, buf, sh, (size_t)n / 4, lr, wd, momentum, grad_scale, first_step);
        else sgd_kernel_x<3><<<grid, 256, 0, st>>>(p, g, buf, sh, (size_t)n / 4, lr, wd, momentum, grad_scale, first_step);
        ODW_CHECK_LAUNCH("sgd_kernel_x");
        return ODW_OK;
    }
    sgd_kernel<<<grid, 256, 0, (hipStream_t)stream_>>>(
        p, g, buf, (unsigned short*)shadow_bf16, (size_t)n, lr, wd, momentum, grad_scale, first_step);
    ODW_CHECK_LAUNCH("sgd_kernel");
    return ODW_OK;
}

namespace {
// K slices of a convolution (1 = unsplit).  The deep layers of the backbone (76x76 pixels: M = 5776, N = 512,
// K = 4608) are 184 tiles of 128x128: one workgroup on 184 of the 256 CUs, each alone on its CU with nothing to
// overlap its barrier phases with.  Two K slices make 368 workgroups, all resident at two per CU: 72 -> 60 us per
// layer including the reduction pass over the fp32 partials (which applies bias / ReLU / the ReLU-backward mask).
// Measured and rejected: 3-4 slices (552+ workgroups = a second round), and the same on the 256x128 three-slot ring
// pipeline (62 us at 2 slices, 86 at 3: the per-lane tap/bounds arithmetic of the operand DMA, not the depth of
// the pipeline, is what the small layers pay for).
int conv_splits(int n_pix, int N, int C) {
    const char* f = getenv("ODW_CONV_SPLITK");
    if (f) return atoi(f);
    const long t128 = (long)((n_pix + BM - 1) / BM) * ((N + BN - 1) / BN);
    if (t128 > 256 || N % 4 != 0 || 9 * C < 4096) return 1;
    return 2;
}
}  // namespace

ODW_EXPORT int64_t odw_conv3x3_workspace(int n_pix, int C, int N) {
    const int sp = conv_splits(n_pix, N, C);
    return sp > 1 ? (int64_t)sp * n_pix * N * 4 : 0;
}

namespace {
// The halo-tile kernel serves every layer whose channel count is a multiple of 64 (all of VGG16 past the stem, the 3x3
// convolutions of the ResNet bodies, every layer of the split-precision modes).  K slices: its grid is (images x 16x16
// tiles) x (N / 128); the 76x76 layers make 25 x 4 = 100 workgroups, so two slices of the channel blocks fill 200 CUs.
struct HaloPlan { bool use; int tiles_y, tiles_x, tiles_n, splits, cb_per_split; };
HaloPlan halo_plan(int n_pix, int H, int W, int C, int N, int dilation) {
    HaloPlan p = {false, 0, 0, 0, 1, 0};
    const char* he = getenv("ODW_CONV_HALO");              // ODW_CONV_HALO=0: the 128x128 kernel (comparison runs)
    if ((he && atoi(he) == 0) || C < 64 || C % 64 != 0 || (dilation != 1 && dilation != 2)) return p;
    if ((unsigned long long)n_pix * (unsigned long long)C * 2ull >= (1ull << 32)) return p;
    p.use = true;
    p.tiles_y = (H + HT - 1) / HT; p.tiles_x = (W + HT - 1) / HT; p.tiles_n = (N + RN - 1) / RN;
    const int ncb = C / 64;
    const long tiles = (long)(n_pix / (H * W)) * p.tiles_y * p.tiles_x * p.tiles_n;
    int sp = (int)(256 / (tiles > 0 ? tiles : 1));
    if (sp > 4) sp = 4;
    if (sp > ncb) sp = ncb;
    if (sp < 1 || N % 4 != 0) sp = 1;
    const char* f = getenv("ODW_CONV_SPLITK");
    if (f && N % 4 == 0) { sp = atoi(f); if (sp > ncb) sp = ncb; if (sp < 1) sp = 1; }
    p.cb_per_split = (ncb + sp - 1) / sp;
    p.splits = (ncb + p.cb_per_split - 1) / p.cb_per_split;
    return p;
}
}  // namespace

// Workspace of the convolution launcher for a given geometry (bytes; 0 = none needed).  Covers whichever kernel the
// launcher will pick (halo-tile or 128x128), so a caller that passes this many bytes never falls back.
ODW_EXPORT int64_t odw_conv3x3_workspace_hw(int n_pix, int H, int W, int C, int N, int dilation) {
    if (H <= 0 || W <= 0 || n_pix <= 0) return 0;
    const HaloPlan hp = halo_plan(n_pix, H, W, C, N, dilation);
    if (hp.use) return hp.splits > 1 ? (int64_t)hp.splits * n_pix * N * 4 : 0;
    return odw_conv3x3_workspace(n_pix, C, N);
}

ODW_EXPORT int odw_conv3x3_nhwc_bf16_ws(const void* X, int n_pix, int H, int W, int C, int dilation, int mirror,
                                        const void* Wk, int ldw, int N, void* Y, int ldy, int y_is_bf16,
                                        const float* bias, int relu, const void* mask, int ldmask,
                                        const void* zero_page, void* workspace, int64_t workspace_bytes, void* stream_);

ODW_EXPORT int odw_conv3x3_nhwc_bf16(const void* X, int n_pix, int H, int W, int C, int dilation, int mirror,
                                     const void* Wk, int ldw, int N, void* Y, int ldy, int y_is_bf16,
                                     const float* bias, int relu, const void* mask, int ldmask, const void* zero_page,
                                     void* stream_) {
    return odw_conv3x3_nhwc_bf16_ws(X, n_pix, H, W, C, dilation, mirror, Wk, ldw, N, Y, ldy, y_is_bf16, bias, relu, mask,
                                    ldmask, zero_page, nullptr, 0, stream_);
}

ODW_EXPORT int odw_conv3x3_nhwc_bf16_ws(const void* X, int n_pix, int H, int W, int C, int dilation, int mirror,
                                        const void* Wk, int ldw, int N, void* Y, int ldy, int y_is_bf16,
                                        const float* bias, int relu, const void* mask, int ldmask,
                                        const void* zero_page, void* workspace, int64_t workspace_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    ODW_REQUIRE(n_pix >= 0 && H > 0 && W > 0 && N > 0 && n_pix % (H * W) == 0, "conv3x3: bad dims");
    ODW_REQUIRE(dilation >= 1 && dilation <= 4, "conv3x3: dilation %d", dilation);
    if (n_pix == 0) return ODW_OK;
    ODW_REQUIRE(X && Wk && Y && zero_page, "conv3x3: null pointer");
    const int K = 9 * C, k64 = (K + 63) / 64 * 64;
    ODW_REQUIRE(ldw >= k64 && ldw % 8 == 0, "conv3x3: weight rows must be zero padded to %d (ldw=%d)", k64, ldw);
    ODW_REQUIRE((((uintptr_t)X) & 15) == 0 && (((uintptr_t)Wk) & 15) == 0 && (((uintptr_t)zero_page) & 15) == 0,
                "conv3x3: 16-byte alignment");
    ConvGeom g;
    g.H = H; g.W = W; g.C = C; g.dil = dilation; g.sign = mirror ? -1 : 1; g.zero = (const unsigned short*)zero_page;
    g.logC = 0;
    while ((1 << g.logC) < C) ++g.logC;
    Epilogue ep;
    ep.bias = bias; ep.relu = relu; ep.drop_p = 0.0f; ep.nseg = 0; ep.accumulate = 0; ep.alpha = 1.0f;
    ep.mask = (const unsigned short*)mask; ep.ldmask = ldmask; ep.pm = 0; ep.kchunk = 0; ep.split_stride = 0; ep.row_ids = nullptr;
    for (int i = 0; i < kMaxSeg; ++i) { ep.seg_row[i] = 0; ep.seg_k0[i] = 0; ep.seg_k1[i] = 0; }
    HaloPlan hp = halo_plan(n_pix, H, W, C, N, dilation);
    // its epilogue stores 16-byte vectors: rows of Y must be 16-byte aligned (every layer of the bodies is)
    if (hp.use && ((((uintptr_t)Y) & 15) != 0 || ((size_t)ldy * (y_is_bf16 ? 2 : 4)) % 16 != 0 || N % 8 != 0)) hp.use = false;
    // the halo-tile kernel takes any multiple of 64 channels (three plane blocks of a split-precision operand); the
    // 128x128 kernel decodes (tap, channel) with shifts
    ODW_REQUIRE(C >= 8 && (hp.use || (C & (C - 1)) == 0), "conv3x3: channel count %d must be a power of two >= 8 (or a multiple "
                "of 64 with 16-byte aligned output rows)", C);
    if (hp.use) {
        if (hp.splits > 1 && (!workspace || workspace_bytes < (int64_t)hp.splits * n_pix * N * 4)) {
            hp.splits = 1; hp.cb_per_split = C / 64;           // no room for partials: one slice
        }
        const int n_img = n_pix / (H * W);
        const unsigned grid = (unsigned)(n_img * hp.tiles_y * hp.tiles_x * hp.tiles_n * hp.splits);
        Epilogue pe = ep;
        void* out = Y;
        int ldo = ldy;
        bool out_bf16 = y_is_bf16 != 0;
        if (hp.splits > 1) {
            ODW_REQUIRE((((uintptr_t)workspace) & 15) == 0, "conv3x3: workspace must be 16-byte aligned");
            pe.bias = nullptr; pe.relu = 0; pe.mask = nullptr; pe.ldmask = 0;
            pe.split_stride = (long long)n_pix * N * 4;
            out = workspace; ldo = N; out_bf16 = false;
        }
#define ODW_LAUNCH_HALO(OUTBF, D)                                                                                  \
        do {                                                                                                       \
            ODW_CHECK_HIP(odw_set_max_lds(reinterpret_cast<const void*>(conv3x3_halo_kernel<OUTBF, D>),        \
                                              (int)Halo<D>::kLdsBytes), \
                          "halo attr");                                                                            \
            conv3x3_halo_kernel<OUTBF, D><<<grid, kHaloThreads, Halo<D>::kLdsBytes, stream>>>(                      \
                (const unsigned short*)X, g, (const unsigned short*)Wk, ldw, n_img, N, out, ldo, pe, hp.tiles_y,    \
                hp.tiles_x, hp.tiles_n, hp.splits, hp.cb_per_split);                                               \
        } while (0)
#ifdef ODW_EXPERIMENTS      // timing experiments that knowingly produce WRONG results: compiled only into experiment builds
        const char* de = getenv("ODW_HALO_DBG");      // timing experiments (wrong results): 1 no DMA, 2 no LDS reads,
        const int dbg = de ? atoi(de) : 0;            // 3 no MFMA, 4 no stores, 5 no weight DMA, 6 no patch DMA
        if (dbg > 0 && out_bf16 && dilation == 1) {
#define ODW_LAUNCH_HALO_DBG(X_)                                                                                    \
            do {                                                                                                   \
                ODW_CHECK_HIP(odw_set_max_lds(reinterpret_cast<const void*>(conv3x3_halo_kernel<true, 1, X_>), \
                                                  (int)Halo<1>::kLdsBytes), \
                              "halo attr");                                                                        \
                conv3x3_halo_kernel<true, 1, X_><<<grid, kHaloThreads, Halo<1>::kLdsBytes, stream>>>(              \
                    (const unsigned short*)X, g, (const unsigned short*)Wk, ldw, n_img, N, out, ldo, pe, hp.tiles_y, \
                    hp.tiles_x, hp.tiles_n, hp.splits, hp.cb_per_split);                                           \
            } while (0)
            if (dbg == 1) ODW_LAUNCH_HALO_DBG(1); else if (dbg == 2) ODW_LAUNCH_HALO_DBG(2); else if (dbg == 3) ODW_LAUNCH_HALO_DBG(3);
            else if (dbg == 4) ODW_LAUNCH_HALO_DBG(4); else if (dbg == 5) ODW_LAUNCH_HALO_DBG(5); else if (dbg == 6) ODW_LAUNCH_HALO_DBG(6); else if (dbg == 7) ODW_LAUNCH_HALO_DBG(7); else ODW_LAUNCH_HALO_DBG(8);
#undef ODW_LAUNCH_HALO_DBG
        } else
#endif
        if (N == 64 && hp.tiles_n == 1 && dilation == 1 && !getenv("ODW_CONV_NO_N64")) {
#define ODW_LAUNCH_HALO64(OUTBF)                                                                                   \
            do {                                                                                                   \
                ODW_CHECK_HIP(odw_set_max_lds(reinterpret_cast<const void*>(conv3x3_halo_kernel<OUTBF, 1, 0, true>), \
                                                  (int)Halo<1>::kLdsBytes), \
                              "halo attr");                                                                        \
                conv3x3_halo_kernel<OUTBF, 1, 0, true><<<grid, kHaloThreads, Halo<1>::kLdsBytes, stream>>>(        \
                    (const unsigned short*)X, g, (const unsigned short*)Wk, ldw, n_img, N, out, ldo, pe, hp.tiles_y, \
                    hp.tiles_x, hp.tiles_n, hp.splits, hp.cb_per_split);                                           \
            } while (0)
            if (out_bf16) ODW_LAUNCH_HALO64(true); else ODW_LAUNCH_HALO64(false);
#undef ODW_LAUNCH_HALO64
        } else
        if (dilation == 1) { if (out_bf16) ODW_LAUNCH_HALO(true, 1); else ODW_LAUNCH_HALO(false, 1); }
        else { if (out_bf16) ODW_LAUNCH_HALO(true, 2); else ODW_LAUNCH_HALO(false, 2); }
#undef ODW_LAUNCH_HALO
        ODW_CHECK_HIP(hipGetLastError(), "conv3x3 halo launch");
        if (hp.splits > 1) {
            const long long quads = (long long)n_pix * (N / 4);
            const int rblocks = (int)((quads + 255) / 256 < 4096 ? (quads + 255) / 256 : 4096);
            if (y_is_bf16)
                splitk_reduce_kernel<true><<<rblocks, 256, 0, stream>>>((const float*)workspace, hp.splits,
                                                                         (long long)n_pix * N, n_pix, N, N, Y, ldy, ep);
            else
                splitk_reduce_kernel<false><<<rblocks, 256, 0, stream>>>((const float*)workspace, hp.splits,
                                                                          (long long)n_pix * N, n_pix, N, N, Y, ldy, ep);
        }
        ODW_CHECK_LAUNCH("conv3x3_halo_kernel");
        return ODW_OK;
    }
    const int sp = workspace ? conv_splits(n_pix, N, C) : 1;
    if (sp > 1 && workspace_bytes >= (int64_t)sp * n_pix * N * 4) {
        ODW_REQUIRE((((uintptr_t)workspace) & 15) == 0, "conv3x3: workspace must be 16-byte aligned");
        Epilogue pe = ep;
        pe.bias = nullptr; pe.relu = 0; pe.mask = nullptr; pe.ldmask = 0;
        pe.kchunk = ((K + sp - 1) / sp + 63) / 64 * 64;
        pe.split_stride = (long long)n_pix * N * 4;
        const int tm_ = (n_pix + BM - 1) / BM, tn_ = (N + BN - 1) / BN;
        const size_t lds_b = (size_t)2 * 2 * kTileChunks * sizeof(uint4);
        ODW_CHECK_HIP(odw_set_max_lds(reinterpret_cast<const void*>(conv3x3_glds_kernel<false>),
                                          (int)lds_b), "conv attr");
        conv3x3_glds_kernel<false><<<dim3((unsigned)(tm_ * tn_), (unsigned)sp), kThreads, lds_b, stream>>>(
            (const unsigned short*)X, g, (const unsigned short*)Wk, ldw, n_pix, N, workspace, N, pe, tm_, tn_);
        ODW_CHECK_HIP(hipGetLastError(), "conv3x3 split launch");
        const long long quads = (long long)n_pix * (N / 4);
        const int rblocks = (int)((quads + 255) / 256 < 4096 ? (quads + 255) / 256 : 4096);
        if (y_is_bf16)
            splitk_reduce_kernel<true><<<rblocks, 256, 0, stream>>>((const float*)workspace, sp, (long long)n_pix * N,
                                                                     n_pix, N, N, Y, ldy, ep);
        else
            splitk_reduce_kernel<false><<<rblocks, 256, 0, stream>>>((const float*)workspace, sp, (long long)n_pix * N,
                                                                      n_pix, N, N, Y, ldy, ep);
        ODW_CHECK_LAUNCH("splitk_reduce_kernel");
        return ODW_OK;
    }
    const int tiles_m = (n_pix + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
    const size_t lds_bytes = (size_t)2 * 2 * kTileChunks * sizeof(uint4);
    if (y_is_bf16) {
        ODW_CHECK_HIP(odw_set_max_lds(reinterpret_cast<const void*>(conv3x3_glds_kernel<true>),
                                          (int)lds_bytes), "conv attr");
        conv3x3_glds_kernel<true><<<tiles_m * tiles_n, kThreads, lds_bytes, stream>>>(
            (const unsigned short*)X, g, (const unsigned short*)Wk, ldw, n_pix, N, Y, ldy, ep, tiles_m, tiles_n);
    } else {
        ODW_CHECK_HIP(odw_set_max_lds(reinterpret_cast<const void*>(conv3x3_glds_kernel<false>),
                                          (int)lds_bytes), "conv attr");
        conv3x3_glds_kernel<false><<<tiles_m * tiles_n, kThreads, lds_bytes, stream>>>(
            (const unsigned short*)X, g, (const unsigned short*)Wk, ldw, n_pix, N, Y, ldy, ep, tiles_m, tiles_n);
    }
    ODW_CHECK_LAUNCH("conv3x3_glds_kernel");
    return ODW_OK;
}

// ---- the two-plane forward convolution of the "bf16x2f" mode (conv3x3_halo2_kernel) ---------------------------------------
namespace {
HaloPlan halo2_plan(int n_pix, int H, int W, int C, int N) {
    HaloPlan p = {true, (H + HT - 1) / HT, (W + HT - 1) / HT, (N + RN - 1) / RN, 1, 0};
    const int ncb = C / 32;
    const long tiles = (long)(n_pix / (H * W)) * p.tiles_y * p.tiles_x * p.tiles_n;
    int sp = (int)(256 / (tiles > 0 ? tiles : 1));
    if (sp > 4) sp = 4;
    if (sp > ncb) sp = ncb;
    if (sp < 1) sp = 1;
    const char* f = getenv("ODW_CONV_SPLITK");
    if (f) { sp = atoi(f); if (sp > ncb) sp = ncb; if (sp < 1) sp = 1; }
    p.cb_per_split = (ncb + sp - 1) / sp;
    p.splits = (ncb + p.cb_per_split - 1) / p.cb_per_split;
    return p;
}
}  // namespace

ODW_EXPORT int64_t odw_conv3x3_planes2_workspace(int n_pix, int H, int W, int C, int N) {
    if (H <= 0 || W <= 0 || n_pix <= 0 || C < 32 || C % 32 != 0) return 0;
    const HaloPlan hp = halo2_plan(n_pix, H, W, C, N);
    return hp.splits > 1 ? (int64_t)hp.splits * n_pix * N * 4 : 0;
}

ODW_EXPORT int odw_conv3x3_planes2_ws(const void* X, int ldx, int n_pix, int H, int W, int C, int dilation, const void* Wk, int ldw,
                                      int N, void* Y, int ldy, int y_planes, const float* bias, int relu, const void* zero_page,
                                      void* workspace, int64_t workspace_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    ODW_REQUIRE(n_pix >= 0 && H > 0 && W > 0 && N > 0 && n_pix % (H * W) == 0, "conv3x3_planes2: bad dims");
    ODW_REQUIRE(dilation == 1 || dilation == 2, "conv3x3_planes2: dilation %d", dilation);
    if (n_pix == 0) return ODW_OK;
    ODW_REQUIRE(X && Wk && Y && zero_page, "conv3x3_planes2: null pointer");
    ODW_REQUIRE(C >= 32 && C % 32 == 0 && ldx >= 2 * C && ldx % 8 == 0, "conv3x3_planes2: C=%d channels per plane (multiple of 32), "
                "row stride %d >= 2 C", C, ldx);
    ODW_REQUIRE(N % 64 == 0, "conv3x3_planes2: N=%d output channels must be a multiple of 64", N);
    ODW_REQUIRE(ldw >= 18 * C && ldw % 8 == 0, "conv3x3_planes2: weight rows hold 9 x 2 x %d values (ldw=%d)", C, ldw);
    ODW_REQUIRE((((uintptr_t)X) & 15) == 0 && (((uintptr_t)Wk) & 15) == 0 && (((uintptr_t)zero_page) & 15) == 0 &&
                (((uintptr_t)Y) & 15) == 0, "conv3x3_planes2: 16-byte alignment");
    ODW_REQUIRE(y_planes ? (ldy >= 2 * N && ldy % 16 == 0) : (ldy >= N && ldy % 4 == 0), "conv3x3_planes2: output row stride %d", ldy);
    ODW_REQUIRE(y_planes >= 0 && y_planes <= 2 && (y_planes != 2 || (H % 2 == 0 && W % 2 == 0)),
                "conv3x3_planes2: y_planes = 0 (fp32) | 1 (planes) | 2 (2 x 2 max pool + planes: even H and W)");
    ODW_REQUIRE((unsigned long long)H * W * (unsigned long long)ldx * 2ull < (1ull << 32), "conv3x3_planes2: one image of the operand "
                "must stay below 4 GB (32-bit offsets inside an image)");
    ConvGeom g;
    g.H = H; g.W = W; g.C = C; g.dil = dilation; g.sign = 1; g.zero = (const unsigned short*)zero_page; g.logC = 0;
    Epilogue ep;
    ep.bias = bias; ep.relu = relu; ep.drop_p = 0.0f; ep.nseg = 0; ep.accumulate = 0; ep.alpha = 1.0f;
    ep.mask = nullptr; ep.ldmask = 0; ep.pm = 0; ep.kchunk = 0; ep.split_stride = 0; ep.row_ids = nullptr;
    for (int i = 0; i < kMaxSeg; ++i) { ep.seg_row[i] = 0; ep.seg_k0[i] = 0; ep.seg_k1[i] = 0; }
    HaloPlan hp = halo2_plan(n_pix, H, W, C, N);
    if (hp.splits > 1 && (!workspace || workspace_bytes < (int64_t)hp.splits * n_pix * N * 4 || (((uintptr_t)workspace) & 15) != 0)) {
        hp.splits = 1; hp.cb_per_split = C / 32;
    }
    if (y_planes == 2) { hp.splits = 1; hp.cb_per_split = C / 32; }      // (the pooled epilogue takes whole sums: no K slices)
    const int n_img = n_pix / (H * W);
    const unsigned grid = (unsigned)(n_img * hp.tiles_y * hp.tiles_x * hp.tiles_n * hp.splits);
    Epilogue pe = ep;
    void* out = Y;
    int ldo = ldy;
    int outm = y_planes;
    if (hp.splits > 1) {
        pe.bias = nullptr; pe.relu = 0;
        pe.split_stride = (long long)n_pix * N * 4;
        out = workspace; ldo = N; outm = 0;
    }
#define ODW_LAUNCH_HALO2(OM, D, N64V)                                                                              \
    do {                                                                                                           \
        ODW_CHECK_HIP(odw_set_max_lds(reinterpret_cast<const void*>(conv3x3_halo2_kernel<OM, D, N64V>),            \
                                      (int)Halo<D>::kLdsBytes), "halo2 attr");                                     \
        conv3x3_halo2_kernel<OM, D, N64V><<<grid, kHaloThreads, Halo<D>::kLdsBytes, stream>>>(                      \
            (const unsigned short*)X, ldx, g, (const unsigned short*)Wk, ldw, n_img, N, out, ldo, pe, hp.tiles_y,   \
            hp.tiles_x, hp.tiles_n, hp.splits, hp.cb_per_split);                                                   \
    } while (0)
    if (N == 64 && dilation == 1) {
        if (outm == 2) ODW_LAUNCH_HALO2(2, 1, true); else if (outm) ODW_LAUNCH_HALO2(1, 1, true); else ODW_LAUNCH_HALO2(0, 1, true);
    } else if (dilation == 1) {
        // two K steps per barrier on a four-slot weight ring (the kernel's PAIR form; ODW_CONV_PAIRSTEP=0: one step per barrier)
        const char* ps = getenv("ODW_CONV_PAIRSTEP");        // (read per launch: the tests run both forms in one process)
        const bool pairstep = !(ps && atoi(ps) == 0);
        if (pairstep) {
#define ODW_LAUNCH_HALO2P(OM)                                                                                      \
    do {                                                                                                           \
        constexpr size_t kLdsP = Halo<1>::kLdsBytes + (size_t)kHaloBStage * sizeof(uint4);                         \
        static_assert(kLdsP <= (size_t)ODW_LDS_BYTES, "the four-slot ring must fit the CU's LDS");                 \
        ODW_CHECK_HIP(odw_set_max_lds(reinterpret_cast<const void*>(conv3x3_halo2_kernel<OM, 1, false, true>),     \
                                      (int)kLdsP), "halo2 pair attr");                                             \
        conv3x3_halo2_kernel<OM, 1, false, true><<<grid, kHaloThreads, kLdsP, stream>>>(                            \
            (const unsigned short*)X, ldx, g, (const unsigned short*)Wk, ldw, n_img, N, out, ldo, pe, hp.tiles_y,   \
            hp.tiles_x, hp.tiles_n, hp.splits, hp.cb_per_split);                                                   \
    } while (0)
            if (outm == 2) ODW_LAUNCH_HALO2P(2); else if (outm) ODW_LAUNCH_HALO2P(1); else ODW_LAUNCH_HALO2P(0);
#undef ODW_LAUNCH_HALO2P
        } else {
            if (outm == 2) ODW_LAUNCH_HALO2(2, 1, false); else if (outm) ODW_LAUNCH_HALO2(1, 1, false); else ODW_LAUNCH_HALO2(0, 1, false);
        }
    } else {
        if (outm == 2) ODW_LAUNCH_HALO2(2, 2, false); else if (outm) ODW_LAUNCH_HALO2(1, 2, false); else ODW_LAUNCH_HALO2(0, 2, false);
    }
#undef ODW_LAUNCH_HALO2
    ODW_CHECK_HIP(hipGetLastError(), "conv3x3_planes2 launch");
    if (hp.splits > 1) {
        const long long quads = (long long)n_pix * (N / 4);
        const int rblocks = (int)((quads + 255) / 256 < 4096 ? (quads + 255) / 256 : 4096);
        if (y_planes)
            splitk_reduce_planes2_kernel<<<rblocks, 256, 0, stream>>>((const float*)workspace, hp.splits, (long long)n_pix * N,
                                                                       n_pix, N, (unsigned short*)Y, ldy, bias, relu);
        else
            splitk_reduce_kernel<false><<<rblocks, 256, 0, stream>>>((const float*)workspace, hp.splits, (long long)n_pix * N,
                                                                      n_pix, N, N, Y, ldy, ep);
    }
    ODW_CHECK_LAUNCH("conv3x3_halo2_kernel");
    return ODW_OK;
}
