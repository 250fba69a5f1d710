// discover.hip -- the data-dependent selection logic of the OD-WSCL loss, on the device.
//
// Reference: roi_heads/weak_head/loss.py:281-345 (loop 1 "IoU sampling" and loop 2 "object
// discovery") + the pseudo-GT bookkeeping of od_layer (pseudo_label_generator.py:143-166).
// The reference runs these as Python loops of tiny kernels with a host sync at every
// argmax / nonzero / unique / NMS (hundreds per step).  Here one workgroup per image walks the
// same (branch, class) sequence with all sets held as P-bit masks in LDS and emits index lists;
// the host reads two small count vectors per step.
//
//   discover_iou  : per (branch i, positive class c): top = first argmax of the branch's score
//                   column; pgt_index[c] = union over branches of { r : IoU+1(r, top) >= thres }
//                   (utils/utils.py:22-26 cal_iou) -> sorted row lists (== .unique()).
//   discover_sim  : per (i, c) in the reference's order: similarity row of the top proposal,
//                   threshold = mean similarity to the class bank (loss.py:320, Q2), multi-class
//                   filter (Q3: bool >= float), torchvision-semantics NMS ordered by class score
//                   (easy_nms, utils/utils.py:28-33, Q10), fallback to top, set difference against
//                   pgt_index (loss.py:336-338) and its update; plus od_layer's pseudo-GT lists
//                   with the row-zeroing quirk (Q5).
// Quirks are reproduced, not fixed.  fp32 comparisons use the same operations as the oracle;
// dot products are accumulated in k order (the reference: BLAS order) -- see the decision
// margins recorded with the golden vectors.
#include "odw_common.h"

namespace {

constexpr int kThreads = 1024;
constexpr int kD = 128;

struct ArgMax { float v; int i; };

__device__ __forceinline__ ArgMax better(ArgMax a, ArgMax b) {   // first maximum (torch.argmax)
    if (b.v > a.v || (b.v == a.v && b.i < a.i)) return b;
    return a;
}

// block-wide first-argmax of f(r), r in [0,P).  red: LDS scratch of kThreads ArgMax.
template <typename F>
__device__ int block_argmax(int P, F f, ArgMax* red) {
    ArgMax m = {-__builtin_inff(), 0x7fffffff};
    for (int r = threadIdx.x; r < P; r += kThreads) {
        ArgMax c = {f(r), r};
        m = better(m, c);
    }
    red[threadIdx.x] = m;
    __syncthreads();
    for (int off = kThreads / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] = better(red[threadIdx.x], red[threadIdx.x + off]);
        __syncthreads();
    }
    int out = red[0].i;
    __syncthreads();
    return out == 0x7fffffff ? 0 : out;
}

__device__ __forceinline__ float iou_plus1(const float4 p, const float4 q) {   // boxlist_ops.py:127-160
    float ap = (p.z - p.x + 1) * (p.w - p.y + 1);
    float aq = (q.z - q.x + 1) * (q.w - q.y + 1);
    float w = fminf(p.z, q.z) - fmaxf(p.x, q.x) + 1;
    float h = fminf(p.w, q.w) - fmaxf(p.y, q.y) + 1;
    w = w < 0 ? 0 : w;
    h = h < 0 ? 0 : h;
    float inter = w * h;
    return inter / (ap + aq - inter);
}

__device__ __forceinline__ bool tv_overlap(const float4 a, const float4 b, float thr) {   // torchvision nms
    float aa = (a.z - a.x) * (a.w - a.y);
    float ab = (b.z - b.x) * (b.w - b.y);
    float w = fmaxf(0.0f, fminf(a.z, b.z) - fmaxf(a.x, b.x));
    float h = fmaxf(0.0f, fminf(a.w, b.w) - fmaxf(a.y, b.y));
    float inter = w * h;
    return inter / (aa + ab - inter) > thr;
}


// <X[r], v> for one row handled by a whole wave: lanes read the 512-byte row coalesced (2 floats each),
// butterfly-reduce.  v is in LDS.  All 64 lanes return the sum.
__device__ __forceinline__ float wave_dot_loaded(const float2 x, const float* __restrict__ v, int lane) {
    float d = x.x * v[2 * lane] + x.y * v[2 * lane + 1];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) d += __shfl_xor(d, off);
    return d;
}
__device__ __forceinline__ float wave_row_dot(const float* __restrict__ row, const float* __restrict__ v, int lane) {
    return wave_dot_loaded(reinterpret_cast<const float2*>(row)[lane], v, lane);
}

// sorted list of the set bits of mask[0..W) (32-bit words) -> out, returns count.  scan: LDS int[W+1]
__device__ int mask_to_list(const unsigned int* mask, int W, int* out, int* scan) {
    for (int w = threadIdx.x; w < W; w += kThreads) scan[w + 1] = __popc(mask[w]);
    __syncthreads();
    if (threadIdx.x == 0) {
        scan[0] = 0;
        for (int w = 0; w < W; ++w) scan[w + 1] += scan[w];
    }
    __syncthreads();
    for (int w = threadIdx.x; w < W; w += kThreads) {
        unsigned int bits = mask[w];
        int o = scan[w];
        while (bits) {
            int b = __builtin_ctz(bits);
            bits &= bits - 1;
            out[o++] = w * 32 + b;
        }
    }
    int n = scan[W];
    __syncthreads();
    return n;
}

// ------------------------------------------------------------------------------- kernel A
// one workgroup per (image, positive class).  src: 3 score matrices (sumP x C); tops [img][3][maxpos];
// masks uint32 [img][maxpos][W32]; rows int32 [img][maxpos][pstride]; counts [img][maxpos]
__global__ __launch_bounds__(kThreads) void discover_iou_kernel(
    const float* __restrict__ s0, const float* __restrict__ s1, const float* __restrict__ s2, int C,
    const float* __restrict__ boxes, const int* __restrict__ img_off, const int* __restrict__ pos_cls,
    const int* __restrict__ n_pos, int maxpos, float thres, int W32, int pstride, int* __restrict__ tops,
    unsigned int* __restrict__ masks, int* __restrict__ rows, int* __restrict__ counts) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    ArgMax* red = reinterpret_cast<ArgMax*>(smem);
    unsigned int* mask = reinterpret_cast<unsigned int*>(smem + kThreads * sizeof(ArgMax));
    int* scan = reinterpret_cast<int*>(mask + W32);
    const int img = blockIdx.x;
    const int base = img_off[img], P = img_off[img + 1] - base;
    const float* src[3] = {s0, s1, s2};
    const float4* bx = reinterpret_cast<const float4*>(boxes) + base;
    // blockIdx.y = the positive class (its three branches OR into ONE mask; the classes share nothing): an image with three labels
    // walked 9 arg-max + IoU rounds in one workgroup
    {
        const int ci = blockIdx.y;
        if (ci >= n_pos[img]) return;
        const int c = pos_cls[img * maxpos + ci];
        for (int w = threadIdx.x; w < W32; w += kThreads) mask[w] = 0;
        __syncthreads();
        for (int i = 0; i < 3; ++i) {
            const float* s = src[i] + (size_t)base * C + (c + 1);
            const int top = block_argmax(P, [&](int r) { return s[(size_t)r * C]; }, red);
            if (threadIdx.x == 0) tops[(img * 3 + i) * maxpos + ci] = top;
            const float4 tb = bx[top];
            for (int r = threadIdx.x; r < P; r += kThreads)
                if (iou_plus1(bx[r], tb) >= thres) atomicOr(&mask[r >> 5], 1u << (r & 31));
            __syncthreads();
        }
        unsigned int* gm = masks + ((size_t)img * maxpos + ci) * W32;
        for (int w = threadIdx.x; w < W32; w += kThreads) gm[w] = mask[w];
        int n = mask_to_list(mask, W32, rows + ((size_t)img * maxpos + ci) * pstride, scan);
        if (threadIdx.x == 0) counts[img * maxpos + ci] = n;
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------- kernel B
struct SimArgs {
    const float* E;            // (sumP,128) unit embeddings
    const float* src[3];       // score matrices (sumP,C)
    const float* boxes;        // (sumP,4)
    const int* img_off;        // [n_img+1]
    const int* pos_cls;        // [n_img][maxpos]
    const int* n_pos;          // [n_img]
    const int* tops;           // [n_img][3][maxpos]   (kernel A)
    unsigned int* masks;       // [n_img][maxpos][W32] pgt_index bit sets, updated in place
    const float* bank;         // (sum bank rows,128): class banks, class-major
    const int* bank_off;       // [C-1] first row of class c's bank
    const int* bank_cnt;       // [C-1]
    int C, maxpos, W32, pstride;
    float nms_thr;
    int* inst_idx;             // [n_img][3][maxpos][pstride]  NMS survivors, descending score
    int* inst_cnt;             // [n_img][3][maxpos]
    int* fresh_idx;            // [n_img][3][maxpos][pstride]  survivors not yet in pgt_index, ascending
    int* fresh_cnt;            // [n_img][3][maxpos]
    int* gt_idx;               // [n_img][3][maxpos*pstride]   od_layer pseudo-GT proposals (class-major)
    int* gt_cls;               // same shape: class id (c+1)
    float* gt_score;           // same shape: score read from the row-zeroed clone (Q5)
    int* gt_cnt;               // [n_img][3]
};

__device__ __forceinline__ bool before(float sa, int ia, float sb, int ib) {
    return (sa > sb) || (sa == sb && ia < ib);
}

// Phase 1 of object discovery, one workgroup per (image, refinement branch, positive class): similarity threshold,
// candidate set, score sort and greedy NMS -> the instance list of loss.py:311-333.  These 3 x n_pos chains per
// image are independent of each other; only the bookkeeping that follows (fresh = instances not yet in pgt_index,
// which accumulates over the branches, and od_layer's pseudo-GT) is ordered, and it is cheap: phase 2 below.
__global__ __launch_bounds__(kThreads) void discover_sim_kernel(SimArgs a, int ppow2, int sbox_off) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // LDS carve
    ArgMax* red = reinterpret_cast<ArgMax*>(smem);                       // kThreads * 8
    float* etop = reinterpret_cast<float*>(red + kThreads);               // 128
    float* fred = etop + kD;                                              // kThreads (float reductions)
    float* ks = fred + kThreads;                                          // ppow2 sort keys
    int* ki = reinterpret_cast<int*>(ks + ppow2);                         // ppow2 sort ids / candidate list
    unsigned char* close = reinterpret_cast<unsigned char*>(ki + ppow2);  // ppow2 flags (close / alive)
    float4* sbox = reinterpret_cast<float4*>(smem + sbox_off);           // ppow2 candidate boxes, sorted order
    __shared__ int s_n;
    __shared__ unsigned long long s_mask[64];
    __shared__ float4 s_kbox[64];
    __shared__ float s_thr;

    const int img = blockIdx.x;
    const int base = a.img_off[img], P = a.img_off[img + 1] - base;
    const int npos = a.n_pos[img];
    const float4* bx = reinterpret_cast<const float4*>(a.boxes) + base;
    const float* E = a.E + (size_t)base * kD;

    {
        const int i = blockIdx.y / a.maxpos, ci = blockIdx.y - i * a.maxpos;
        if (ci >= npos) return;
        const float* S = a.src[i] + (size_t)base * a.C;
        {
            const int c = a.pos_cls[img * a.maxpos + ci];
            const int top = a.tops[(img * 3 + i) * a.maxpos + ci];
            const size_t slot = ((size_t)(img * 3 + i) * a.maxpos + ci);
            if (threadIdx.x < kD) etop[threadIdx.x] = E[(size_t)top * kD + threadIdx.x];
            __syncthreads();
            // ---- threshold: mean_j <e_top, bank_c[j]>   (loss.py:320)
            // (the dot loops keep four rows of loads in flight per wave: one row per iteration is pure L2 latency;
            //  per-row arithmetic and the order of the running sums are unchanged)
            {
                const int nb = a.bank_cnt[c];
                const float* bk = a.bank + (size_t)a.bank_off[c] * kD;
                const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
                constexpr int kW = kThreads / 64;
                float acc = 0.0f;
                int j = wave;
                for (; j + 3 * kW < nb; j += 4 * kW) {
                    float2 x[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) x[u] = reinterpret_cast<const float2*>(bk + (size_t)(j + u * kW) * kD)[lane];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float d = wave_dot_loaded(x[u], etop, lane);
                        if (lane == 0) acc += d;
                    }
                }
                for (; j < nb; j += kW) {
                    const float d = wave_row_dot(bk + (size_t)j * kD, etop, lane);
                    if (lane == 0) acc += d;
                }
                fred[threadIdx.x] = acc;
                __syncthreads();
                for (int off = kThreads / 2; off > 0; off >>= 1) {
                    if ((int)threadIdx.x < off) fred[threadIdx.x] += fred[threadIdx.x + off];
                    __syncthreads();
                }
                if (threadIdx.x == 0) s_thr = fred[0] / (float)nb;
                __syncthreads();
            }
            const float thr = s_thr;
            // ---- close = sim_mat[top] >= thr
            {
                const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
                constexpr int kW = kThreads / 64;
                int r = wave;
                for (; r + 7 * kW < P; r += 8 * kW) {          // eight rows of loads in flight per wave
                    float2 x[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) x[u] = reinterpret_cast<const float2*>(E + (size_t)(r + u * kW) * kD)[lane];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const float d = wave_dot_loaded(x[u], etop, lane);
                        if (lane == 0) close[r + u * kW] = d >= thr ? 1 : 0;
                    }
                }
                for (; r + 3 * kW < P; r += 4 * kW) {
                    float2 x[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) x[u] = reinterpret_cast<const float2*>(E + (size_t)(r + u * kW) * kD)[lane];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float d = wave_dot_loaded(x[u], etop, lane);
                        if (lane == 0) close[r + u * kW] = d >= thr ? 1 : 0;
                    }
                }
                for (; r < P; r += kW) {
                    const float d = wave_row_dot(E + (size_t)r * kD, etop, lane);
                    if (lane == 0) close[r] = d >= thr ? 1 : 0;
                }
            }
            __syncthreads();
            // ---- Q3: for every other positive class, close = (float(close) >= sim_mat[neg_top])
            if (npos > 1) {
                for (int cj = 0; cj < npos; ++cj) {
                    if (cj == ci) continue;
                    const int ntop = a.tops[(img * 3 + i) * a.maxpos + cj];
                    if (threadIdx.x < kD) etop[threadIdx.x] = E[(size_t)ntop * kD + threadIdx.x];
                    __syncthreads();
                    {
                        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
                        constexpr int kW = kThreads / 64;
                        int r = wave;
                        for (; r + 3 * kW < P; r += 4 * kW) {
                            float2 x[4];
#pragma unroll
                            for (int u = 0; u < 4; ++u) x[u] = reinterpret_cast<const float2*>(E + (size_t)(r + u * kW) * kD)[lane];
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                const float d = wave_dot_loaded(x[u], etop, lane);
                                if (lane == 0) close[r + u * kW] = ((close[r + u * kW] ? 1.0f : 0.0f) >= d) ? 1 : 0;
                            }
                        }
                        for (; r < P; r += kW) {
                            const float d = wave_row_dot(E + (size_t)r * kD, etop, lane);
                            if (lane == 0) close[r] = ((close[r] ? 1.0f : 0.0f) >= d) ? 1 : 0;
                        }
                    }
                    __syncthreads();
                }
            }
            // ---- candidates sorted by class score descending, ties by index ascending: the candidates are compacted
            // first (their order is irrelevant: (score, index) is a total order) and the bitonic network runs over the
            // next power of two above their NUMBER, not above P (was 66 barrier-separated passes over 2048 slots)
            if (threadIdx.x == 0) s_n = 0;
            __syncthreads();
            for (int r = threadIdx.x; r < P; r += kThreads) {
                if (close[r]) {
                    const int pos = atomicAdd(&s_n, 1);
                    ks[pos] = S[(size_t)r * a.C + c + 1];
                    ki[pos] = r;
                }
            }
            __syncthreads();
            const int n = s_n;
            int npow = 2;
            while (npow < n) npow <<= 1;
            for (int r = n + threadIdx.x; r < npow; r += kThreads) {
                ks[r] = -__builtin_inff();
                ki[r] = 0x40000000 + r;               // padding sorts last
            }
            __syncthreads();
            for (int k = 2; k <= npow; k <<= 1) {
                for (int j = k >> 1; j > 0; j >>= 1) {
                    // one compare-exchange PAIR per thread and round: pair q -> t = q with a 0 inserted at bit log2(j)
                    for (int q = threadIdx.x; q < (npow >> 1); q += kThreads) {
                        const int t = ((q & ~(j - 1)) << 1) | (q & (j - 1)), p = t | j;
                        const bool up = ((t & k) == 0);
                        const float sa = ks[t], sb = ks[p];
                        const int ia = ki[t], ib = ki[p];
                        const bool a_first = before(sa, ia, sb, ib);
                        if (up ? !a_first : a_first) { ks[t] = sb; ks[p] = sa; ki[t] = ib; ki[p] = ia; }
                    }
                    __syncthreads();
                }
            }
            // ---- greedy NMS in sorted order (torchvision: IoU without +1, suppress when > thr)
            for (int t = threadIdx.x; t < n; t += kThreads) {
                close[t] = 1;                                                   // alive flags by sorted position
                sbox[t] = bx[ki[t]];                                            // boxes in LDS: the greedy chain below
            }                                                                   // must not pay a global load per step
            __syncthreads();
            int* inst = a.inst_idx + slot * a.pstride;
            int n_inst = 0;
            // Greedy NMS in windows of 64 sorted positions (was: one barrier-separated round per KEPT box, ~100 rounds):
            //   1. every pair of the window's alive positions is tested in parallel -> 64 suppression masks in LDS;
            //   2. one wave resolves the window greedily in registers: position i is kept iff no earlier kept position
            //      of the window (or of an earlier window: phase 3) suppresses it -- one readlane + mask step per kept position;
            //   3. everyone suppresses the positions behind the window against the boxes the window kept.
            // The kept set is exactly the sequential greedy one: j dies iff an earlier KEPT i overlaps it.
            for (int k0 = 0; k0 < n; k0 += 64) {
                const int wn = n - k0 < 64 ? n - k0 : 64;
                if (threadIdx.x < 64) s_mask[threadIdx.x] = 0ull;
                __syncthreads();
                for (int pidx = threadIdx.x; pidx < 64 * 64; pidx += kThreads) {
                    const int i = pidx >> 6, j = pidx & 63;
                    if (j > i && j < wn && close[k0 + i] && close[k0 + j] && tv_overlap(sbox[k0 + i], sbox[k0 + j], a.nms_thr))
                        atomicOr(&s_mask[i], 1ull << j);
                }
                __syncthreads();
                if (threadIdx.x < 64) {                     // wave 0
                    const int l = threadIdx.x;
                    const unsigned long long m = s_mask[l];
                    const unsigned mlo = (unsigned)m, mhi = (unsigned)(m >> 32);
                    const unsigned long long alive = __ballot(l < wn && close[k0 + (l < wn ? l : 0)]);
                    unsigned long long kept = 0ull, rem = alive;
                    while (rem) {                           // wave-uniform: one step per KEPT position of the window
                        const int i = __builtin_amdgcn_readfirstlane(__ffsll((long long)rem) - 1);
                        const unsigned long long mi = (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)mlo, i) |
                                                      ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)mhi, i) << 32);
                        kept |= 1ull << i;
                        rem &= ~(mi | (1ull << i));
                    }
                    const bool mine = (kept >> l) & 1ull;
                    if (l < wn) close[k0 + l] = mine ? 1 : 0;
                    if (mine) {
                        const int pos = __popcll(kept & ((1ull << l) - 1ull));
                        inst[n_inst + pos] = ki[k0 + l];
                        s_kbox[pos] = sbox[k0 + l];
                    }
                    if (l == 0) s_n = __popcll(kept);
                }
                __syncthreads();
                const int kw = s_n;
                n_inst += kw;
                if (kw > 0) {
                    for (int t = k0 + 64 + threadIdx.x; t < n; t += kThreads) {
                        if (!close[t]) continue;
                        const float4 bt = sbox[t];
                        for (int q = 0; q < kw; ++q)
                            if (tv_overlap(s_kbox[q], bt, a.nms_thr)) { close[t] = 0; break; }
                    }
                }
                __syncthreads();
            }
            if (n_inst == 0) {                    // "avoid none" (loss.py:333)
                if (threadIdx.x == 0) inst[0] = top;
                n_inst = 1;
            }
            if (threadIdx.x == 0) a.inst_cnt[slot] = n_inst;
        }
    }
}

// Phase 2, one workgroup per image, branches and classes in the reference's order.
__global__ __launch_bounds__(kThreads) void discover_finish_kernel(SimArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    ArgMax* red = reinterpret_cast<ArgMax*>(smem);                       // kThreads * 8
    unsigned int* fresh_mask = reinterpret_cast<unsigned int*>(red + kThreads);   // W32
    unsigned int* zeroed = fresh_mask + a.W32;                            // W32 (od_layer zeroed rows)
    int* scan = reinterpret_cast<int*>(zeroed + a.W32);                   // W32 + 1
    const int img = blockIdx.x;
    const int base = a.img_off[img], P = a.img_off[img + 1] - base;
    const int npos = a.n_pos[img];
    // blockIdx.y < maxpos: the fresh lists of positive class ci over the three branches (the class's pgt_index accumulates over
    // them; classes share nothing); blockIdx.y >= maxpos: od_layer's pseudo-GT of branch y - maxpos (classes in order: each reads
    // the rows earlier classes zeroed; it needs the instance lists only, not the fresh ones).  One workgroup walked all
    // 3 x npos + 3 x npos rounds in sequence before (52 us at two labels).
    const bool fresh_part = (int)blockIdx.y < a.maxpos;
    if (fresh_part && (int)blockIdx.y >= npos) return;
    const int i_lo = fresh_part ? 0 : (int)blockIdx.y - a.maxpos, i_hi = fresh_part ? 3 : i_lo + 1;
    for (int i = i_lo; i < i_hi; ++i) {
        const float* S = a.src[i] + (size_t)base * a.C;
        for (int ci = fresh_part ? (int)blockIdx.y : npos; ci < (fresh_part ? (int)blockIdx.y + 1 : npos); ++ci) {
            const int top = a.tops[(img * 3 + i) * a.maxpos + ci];
            const size_t slot = ((size_t)(img * 3 + i) * a.maxpos + ci);
            const int* inst = a.inst_idx + slot * a.pstride;
            int n_inst = a.inst_cnt[slot];
            // ---- fresh = survivors not in pgt_index (ascending) ; fallback top ; pgt_index |= fresh
            unsigned int* pm = a.masks + ((size_t)img * a.maxpos + ci) * a.W32;
            for (int w = threadIdx.x; w < a.W32; w += kThreads) fresh_mask[w] = 0;
            __syncthreads();
            for (int t = threadIdx.x; t < n_inst; t += kThreads) {
                int r = inst[t];
                if (!((pm[r >> 5] >> (r & 31)) & 1u)) atomicOr(&fresh_mask[r >> 5], 1u << (r & 31));
            }
            __syncthreads();
            int* fresh = a.fresh_idx + slot * a.pstride;
            int n_fresh = mask_to_list(fresh_mask, a.W32, fresh, scan);
            if (n_fresh == 0) {
                if (threadIdx.x == 0) fresh[0] = top;
                n_fresh = 1;
                if (threadIdx.x == 0) atomicOr(&pm[top >> 5], 1u << (top & 31));
            } else {
                for (int w = threadIdx.x; w < a.W32; w += kThreads) pm[w] |= fresh_mask[w];
            }
            if (threadIdx.x == 0) a.fresh_cnt[slot] = n_fresh;
            __syncthreads();
        }
        if (fresh_part) continue;
        // ---- od_layer pseudo-GT for branch i (pseudo_label_generator.py:143-166): classes ascending,
        // scores read from a clone whose rows are zeroed at each earlier class's (mutated) argmax
        for (int w = threadIdx.x; w < a.W32; w += kThreads) zeroed[w] = 0;
        __syncthreads();
        int g = 0;
        const size_t gbase = (size_t)(img * 3 + i) * a.maxpos * a.pstride;
        for (int ci = 0; ci < npos; ++ci) {
            const int c = a.pos_cls[img * a.maxpos + ci];
            const size_t slot = ((size_t)(img * 3 + i) * a.maxpos + ci);
            const int n_inst = a.inst_cnt[slot];
            const int* inst = a.inst_idx + slot * a.pstride;
            const int t_c = block_argmax(P, [&](int r) {
                return ((zeroed[r >> 5] >> (r & 31)) & 1u) ? 0.0f : S[(size_t)r * a.C + c + 1]; }, red);
            for (int t = threadIdx.x; t < n_inst; t += kThreads) {
                const int r = inst[t];
                a.gt_idx[gbase + g + t] = r;
                a.gt_cls[gbase + g + t] = c + 1;
                a.gt_score[gbase + g + t] = ((zeroed[r >> 5] >> (r & 31)) & 1u) ? 0.0f : S[(size_t)r * a.C + c + 1];
            }
            g += n_inst;
            __syncthreads();
            if (threadIdx.x == 0) zeroed[t_c >> 5] |= 1u << (t_c & 31);
            __syncthreads();
        }
        if (threadIdx.x == 0) a.gt_cnt[img * 3 + i] = g;
        __syncthreads();
    }
}

int pow2_at_least(int n) { int p = 1; while (p < n) p <<= 1; return p; }

}  // namespace

ODW_EXPORT int odw_discover_iou(const float* s0, const float* s1, const float* s2, int C, const float* boxes,
                                const int* img_off, int n_img, int max_p, const int* pos_cls, const int* n_pos,
                                int maxpos, float thres, int* tops, uint32_t* masks, int* rows, int pstride,
                                int* counts, void* stream_) {
    ODW_REQUIRE(n_img >= 0 && C >= 2 && max_p >= 1 && maxpos >= 1 && pstride >= max_p, "discover_iou: bad dims");
    if (n_img == 0) return ODW_OK;
    ODW_REQUIRE(s0 && s1 && s2 && boxes && img_off && pos_cls && n_pos && tops && masks && rows && counts,
                "discover_iou: null pointer");
    ODW_REQUIRE((((uintptr_t)boxes) & 15) == 0, "discover_iou: boxes must be 16-byte aligned");
    const int W32 = (max_p + 31) / 32;
    size_t lds = kThreads * sizeof(ArgMax) + (size_t)W32 * 4 + (size_t)(W32 + 1) * 4;
    discover_iou_kernel<<<dim3(n_img, maxpos), kThreads, lds, (hipStream_t)stream_>>>(s0, s1, s2, C, boxes, img_off, pos_cls, n_pos,
                                                                      maxpos, thres, W32, pstride, tops, masks, rows,
                                                                      counts);
    ODW_CHECK_LAUNCH("discover_iou_kernel");
    return ODW_OK;
}

ODW_EXPORT int odw_discover_sim(const float* E, const float* s0, const float* s1, const float* s2, int C,
                                const float* boxes, const int* img_off, int n_img, int max_p, const int* pos_cls,
                                const int* n_pos, int maxpos, const int* tops, uint32_t* masks, const float* bank,
                                const int* bank_off, const int* bank_cnt, float nms_thr, int pstride, int* inst_idx,
                                int* inst_cnt, int* fresh_idx, int* fresh_cnt, int* gt_idx, int* gt_cls,
                                float* gt_score, int* gt_cnt, void* stream_) {
    ODW_REQUIRE(n_img >= 0 && C >= 2 && max_p >= 1 && max_p <= 8192 && maxpos >= 1 && pstride >= max_p,
                "discover_sim: bad dims (P=%d, at most 8192 proposals per image)", max_p);
    if (n_img == 0) return ODW_OK;
    ODW_REQUIRE(E && s0 && s1 && s2 && boxes && img_off && pos_cls && n_pos && tops && masks && bank && bank_off &&
                    bank_cnt && inst_idx && inst_cnt && fresh_idx && fresh_cnt && gt_idx && gt_cls && gt_score && gt_cnt,
                "discover_sim: null pointer");
    ODW_REQUIRE((((uintptr_t)boxes) & 15) == 0, "discover_sim: boxes must be 16-byte aligned");
    SimArgs a;
    a.E = E; a.src[0] = s0; a.src[1] = s1; a.src[2] = s2; a.boxes = boxes; a.img_off = img_off; a.pos_cls = pos_cls;
    a.n_pos = n_pos; a.tops = tops; a.masks = masks; a.bank = bank; a.bank_off = bank_off; a.bank_cnt = bank_cnt;
    a.C = C; a.maxpos = maxpos; a.W32 = (max_p + 31) / 32; a.pstride = pstride; a.nms_thr = nms_thr;
    a.inst_idx = inst_idx; a.inst_cnt = inst_cnt; a.fresh_idx = fresh_idx; a.fresh_cnt = fresh_cnt;
    a.gt_idx = gt_idx; a.gt_cls = gt_cls; a.gt_score = gt_score; a.gt_cnt = gt_cnt;
    const int ppow2 = pow2_at_least(max_p);
    size_t lds = kThreads * sizeof(ArgMax) + kD * 4 + kThreads * 4 + (size_t)ppow2 * 4 + (size_t)ppow2 * 4 +
                 (size_t)ppow2 + 64;
    const int sbox_off = (int)((lds + 15) / 16 * 16);
    lds = (size_t)sbox_off + (size_t)ppow2 * 16;
    ODW_CHECK_HIP(odw_set_max_lds(reinterpret_cast<const void*>(discover_sim_kernel),
                                      (int)lds), "discover_sim attr");
    discover_sim_kernel<<<dim3(n_img, 3 * maxpos), kThreads, lds, (hipStream_t)stream_>>>(a, ppow2, sbox_off);
    ODW_CHECK_LAUNCH("discover_sim_kernel");
    const size_t lds2 = kThreads * sizeof(ArgMax) + (size_t)a.W32 * 4 * 2 + (size_t)(a.W32 + 1) * 4;
    discover_finish_kernel<<<dim3(n_img, maxpos + 3), kThreads, lds2, (hipStream_t)stream_>>>(a);
    ODW_CHECK_LAUNCH("discover_finish_kernel");
    return ODW_OK;
}
