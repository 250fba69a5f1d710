// loss_lists.hip -- the index lists of the OD-WSCL loss, assembled ON THE DEVICE (round 6).
//
// Reference: roi_heads/weak_head/loss.py:281-347.  Its two Python loops append to lists (pgt_collection,
// pgt_update, instance_diff) whose lengths depend on the scores of the step; rounds 2-5 kept the selection itself on
// the device (discover.hip) but read the counts back TWICE per step and built every gather list with numpy
// (loss_fused.py), so the host could never run ahead of the GPU: any hiccup of the launching thread -- a pre-empted
// core, a page fault -- landed in the step time (BENCH_r05: three 14-19 ms steps among 9 ms ones).  The two kernels
// here turn the counts into everything the rest of the step consumes, as DEVICE-resident lists with DEVICE-resident
// lengths; the launches that follow are sized for the capacities and read their extents from `scal` (gemm_bf16.hip:
// m_dev / k_dev; head_aux.hip: the grouped views; contrastive.hip: supcon's N).  The host reads nothing back.
//
// Orders are the reference's (and loss_fused.py's host assembly, kept as the test-side restatement):
//   groups g = (image, positive class) in loop-1 order;  entries = their IoU-sampled rows, group after group;
//   views   = per group [k drop rows | k noise rows]  (loss.py:292-305);
//   bank[c] = per group of class c, in loop-1 order: [sampled proposal rows | the group's 2k view rows]   (Q2);
//   SupCon features = class-major: bank[c] then the discoveries of (image, branch) in loop-2 order (sim_loss.py:55-58);
//   SupCon weights  = APPEND order (Q1): per group 3 x its rows' scores, then the discoveries in loop-2 order.
#include "odw_common.h"

namespace {

constexpr int kThreads = 1024;
constexpr int kGW = 16;          // ints per row of the host's group table

__device__ __forceinline__ int r64i(int n) { return (n + 63) / 64 * 64; }

// group table row (host): 0 img, 1 ci, 2 cls (0-based foreground id), 3 base (first proposal row of the image),
//                         4..5 k6 drop, 6..7 k7 drop, 8..9 k6 noise, 10..11 k7 noise   (dropout keys of the views' fc6 / fc7)
struct ListsA {
    const int* grp;            // [G][kGW]
    const int* cls_order;      // [G]     group indices sorted by (class, loop-1 order): the bank layout
    const int* counts;         // [n_img][maxpos]            (discover_iou)
    const int* rows;           // [n_img][maxpos][pstride]   (discover_iou)
    int G, maxpos, pstride, sum_p, n_cls1, e_cap;
    int* scal;                 // [16] 0 E1, 1 V = 2 E1, 2 r64(V), 3 bank rows (3 E1), 4 overflow flag
    int* e0;                   // [G + 1]
    int* roi_index;            // [e_cap]
    int* bank_index;           // [3 e_cap]  rows of the virtual table [sim_feature (sum_p rows); view embeddings]
    int* bank_off;             // [n_cls1]
    int* bank_cnt;             // [n_cls1]
    uint4* row_tab6;           // [2 e_cap]  dropout draw of view row m in fc6: (logical row, key0, key1, -)
    uint4* row_tab7;           // [2 e_cap]  ... in fc7
};

__global__ __launch_bounds__(kThreads) void loss_lists_a_kernel(ListsA a) {
    __shared__ int s_e0[257], s_bpos[256];
    if (threadIdx.x == 0) {
        int e = 0, over = 0;
        for (int g = 0; g < a.G; ++g) {
            const int* row = a.grp + g * kGW;
            int k = a.counts[row[0] * a.maxpos + row[1]];
            if (e + k > a.e_cap) { k = a.e_cap - e; over = 1; }      // never write past the buffers; the flag makes it loud
            s_e0[g] = e;
            e += k;
        }
        s_e0[a.G] = e;
        for (int c = 0; c < a.n_cls1; ++c) { a.bank_off[c] = 0; a.bank_cnt[c] = 0; }
        int pos = 0;
        for (int j = 0; j < a.G; ++j) {
            const int g = a.cls_order[j], c = a.grp[g * kGW + 2], k = s_e0[g + 1] - s_e0[g];
            if (a.bank_cnt[c] == 0) a.bank_off[c] = pos;
            s_bpos[g] = pos;
            a.bank_cnt[c] += 3 * k;
            pos += 3 * k;
        }
        a.scal[0] = e; a.scal[1] = 2 * e; a.scal[2] = r64i(2 * e); a.scal[3] = 3 * e; a.scal[4] = over;
    }
    __syncthreads();
    for (int g = threadIdx.x; g <= a.G; g += kThreads) a.e0[g] = s_e0[g];
    for (int g = 0; g < a.G; ++g) {
        const int* row = a.grp + g * kGW;
        const int e0 = s_e0[g], k = s_e0[g + 1] - e0, base = row[3], bpos = s_bpos[g];
        const int* rows = a.rows + ((size_t)row[0] * a.maxpos + row[1]) * a.pstride;
        for (int r = threadIdx.x; r < k; r += kThreads) {
            const int p = base + rows[r];
            a.roi_index[e0 + r] = p;
            a.bank_index[bpos + r] = p;
        }
        const uint32_t k6d0 = row[4], k6d1 = row[5], k7d0 = row[6], k7d1 = row[7];
        const uint32_t k6n0 = row[8], k6n1 = row[9], k7n0 = row[10], k7n1 = row[11];
        for (int j = threadIdx.x; j < 2 * k; j += kThreads) {
            a.bank_index[bpos + k + j] = a.sum_p + 2 * e0 + j;
            const bool noise = j >= k;
            const uint32_t lrow = (uint32_t)(noise ? j - k : j);
            a.row_tab6[2 * e0 + j] = make_uint4(lrow, noise ? k6n0 : k6d0, noise ? k6n1 : k6d1, 0u);
            a.row_tab7[2 * e0 + j] = make_uint4(lrow, noise ? k7n0 : k7d0, noise ? k7n1 : k7d1, 0u);
        }
    }
}

struct ListsB {
    const int* grp;            // [G][kGW]
    const int* cls_order;      // [G]
    const int* img_off;        // [n_img + 1]
    const int* n_pos;          // [n_img]
    const int* pos_cls;        // [n_img][maxpos]
    const int* scal_a;         // lists_a's scalars
    const int* e0;             // [G + 1]
    const int* roi_index;      // [E1]
    const int* bank_index;     // [3 E1]
    const int* bank_off;       // [n_cls1]
    const int* bank_cnt;       // [n_cls1]
    const int* fresh_idx;      // [n_img][3][maxpos][pstride]   (discover_sim)
    const int* fresh_cnt;      // [n_img][3][maxpos]
    const int* gt_cnt;         // [n_img][3]
    const float* final_score;  // (sum_p, fs_cols)
    const float* colstat;      // flat; weight denominator of (img, c) = colstat[img * cs_ld + cs_off + c + 1]
    int G, n_img, maxpos, pstride, sum_p, fs_cols, cs_ld, cs_off, n_cap, a_cap, e_cap, p64, gt_max;
    int* scal;                 // [16] 0 N, 1 A, 2 E = A + E1, 3 overflow, 4 P64 + r64(V), 5 P64 + r64(V) + r64(A), 6 r64(V),
                               //      7 r64(V) + r64(A), 8 r64(A), 9 pseudo-GT overflow, 10 r64(N)
    int* feat_index;           // [n_cap]  < a_cap: row of the re-attached clean rows' table; else a_cap + view row
    int* labels;               // [n_cap]
    float* weights;            // [n_cap]
    int* act_rows;             // [a_cap]  ascending unique proposal rows the features reference
    int* roi_index_all;        // [a_cap + e_cap]  = [act_rows | roi_index]  (the pooling node's side-buffer entries)
    int* sticky;               // optional [2], never cleared by the kernels: [0] |= a capacity overflowed, [1] |= pseudo-GT overflow
                               // (the host reads the step scalars back only now and then; an overflow must not slip between reads)
};

// one workgroup; the bit set of referenced proposal rows lives in dynamic LDS: W32 words + W32 + 1 prefix counts
__global__ __launch_bounds__(kThreads) void loss_lists_b_kernel(ListsB b) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int W32 = (b.sum_p + 31) / 32;
    unsigned int* mask = reinterpret_cast<unsigned int*>(smem);
    int* scan = reinterpret_cast<int*>(mask + W32);
    const int E1 = b.scal_a[0], V = b.scal_a[1];
    for (int w = threadIdx.x; w < W32; w += kThreads) mask[w] = 0;
    __syncthreads();
    // ---- pass 1: features, class-major.  Positions are computed by walking the (few) segments serially per thread --
    // every thread walks the same segment table, a segment's elements are spread over the threads.
    int pos = 0;
    for (int j = 0; j < b.G; ) {
        const int c = b.grp[b.cls_order[j] * kGW + 2];
        // bank of class c (proposal rows and view rows, in bank order)
        const int boff = b.bank_off[c], bcnt = b.bank_cnt[c];
        for (int t = threadIdx.x; t < bcnt; t += kThreads) {
            if (pos + t >= b.n_cap) break;
            const int ix = b.bank_index[boff + t];
            b.feat_index[pos + t] = ix;                  // (proposal rows are re-mapped to their rank below)
            b.labels[pos + t] = c;
            if (ix < b.sum_p) atomicOr(&mask[ix >> 5], 1u << (ix & 31));
        }
        pos += bcnt;
        // the discoveries of class c: images in order, branches in order (loop 2)
        int j2 = j;
        while (j2 < b.G && b.grp[b.cls_order[j2] * kGW + 2] == c) {
            const int* row = b.grp + b.cls_order[j2] * kGW;
            const int img = row[0], ci = row[1], base = row[3];
            for (int i = 0; i < 3; ++i) {
                const size_t slot = ((size_t)(img * 3 + i) * b.maxpos + ci);
                const int n = b.fresh_cnt[slot];
                const int* fr = b.fresh_idx + slot * b.pstride;
                for (int t = threadIdx.x; t < n; t += kThreads) {
                    if (pos + t >= b.n_cap) break;
                    const int ix = base + fr[t];
                    b.feat_index[pos + t] = ix;
                    b.labels[pos + t] = c;
                    atomicOr(&mask[ix >> 5], 1u << (ix & 31));
                }
                pos += n;
            }
            ++j2;
        }
        j = j2;
    }
    const int N = pos;
    __syncthreads();
    // ---- the referenced proposal rows, ascending (== numpy.unique) and the rank of each
    for (int w = threadIdx.x; w < W32; w += kThreads) scan[w + 1] = __popc(mask[w]);
    __syncthreads();
    if (threadIdx.x == 0) {
        scan[0] = 0;
        for (int w = 0; w < W32; ++w) scan[w + 1] += scan[w];
    }
    __syncthreads();
    const int A = scan[W32];
    for (int w = threadIdx.x; w < W32; w += kThreads) {
        unsigned int bits = mask[w];
        int o = scan[w];
        while (bits) {
            const int bit = __builtin_ctz(bits);
            bits &= bits - 1;
            if (o < b.a_cap) { b.act_rows[o] = w * 32 + bit; b.roi_index_all[o] = w * 32 + bit; }
            ++o;
        }
    }
    const int Nc = N < b.n_cap ? N : b.n_cap;
    for (int t = threadIdx.x; t < Nc; t += kThreads) {
        const int ix = b.feat_index[t];
        if (ix < b.sum_p) b.feat_index[t] = scan[ix >> 5] + __popc(mask[ix >> 5] & ((1u << (ix & 31)) - 1u));
        else b.feat_index[t] = b.a_cap + (ix - b.sum_p);
    }
    const int Ac = A < b.a_cap ? A : b.a_cap;
    for (int t = threadIdx.x; t < E1; t += kThreads) b.roi_index_all[Ac + t] = b.roi_index[t];
    // ---- pass 2: weights, APPEND order (Q1, Q12): loop 1 -- per group its k rows three times (orig / drop / noise) --, then
    // loop 2 -- images, branches, classes
    pos = 0;
    for (int g = 0; g < b.G; ++g) {
        const int* row = b.grp + g * kGW;
        const int img = row[0], c = row[2], e0 = b.e0[g], k = b.e0[g + 1] - e0;
        const float den = b.colstat[(size_t)img * b.cs_ld + b.cs_off + c + 1];
        for (int t = threadIdx.x; t < 3 * k; t += kThreads) {
            if (pos + t >= b.n_cap) break;
            const int r = t < k ? t : (t < 2 * k ? t - k : t - 2 * k);
            b.weights[pos + t] = b.final_score[(size_t)b.roi_index[e0 + r] * b.fs_cols + c + 1] / den;
        }
        pos += 3 * k;
    }
    for (int img = 0; img < b.n_img; ++img) {
        const int base = b.img_off[img], npos = b.n_pos[img];
        for (int i = 0; i < 3; ++i)
            for (int ci = 0; ci < npos; ++ci) {
                const int c = b.pos_cls[img * b.maxpos + ci];
                const size_t slot = ((size_t)(img * 3 + i) * b.maxpos + ci);
                const int n = b.fresh_cnt[slot];
                const int* fr = b.fresh_idx + slot * b.pstride;
                const float den = b.colstat[(size_t)img * b.cs_ld + b.cs_off + c + 1];
                for (int t = threadIdx.x; t < n; t += kThreads) {
                    if (pos + t >= b.n_cap) break;
                    b.weights[pos + t] = b.final_score[(size_t)(base + fr[t]) * b.fs_cols + c + 1] / den;
                }
                pos += n;
            }
    }
    if (threadIdx.x == 0) {
        int gt_over = 0;
        for (int t = 0; t < b.n_img * 3; ++t) gt_over |= b.gt_cnt[t] > b.gt_max ? 1 : 0;
        const int over = (N > b.n_cap || A > b.a_cap || b.scal_a[4]) ? 1 : 0;
        b.scal[0] = Nc; b.scal[1] = Ac; b.scal[2] = Ac + E1; b.scal[3] = over;
        b.scal[4] = b.p64 + r64i(V); b.scal[5] = b.p64 + r64i(V) + r64i(Ac);
        b.scal[6] = r64i(V); b.scal[7] = r64i(V) + r64i(Ac); b.scal[8] = r64i(Ac); b.scal[9] = gt_over; b.scal[10] = r64i(Nc);
        if (b.sticky) {
            if (over) b.sticky[0] = 1;
            if (gt_over) b.sticky[1] = 1;
        }
    }
}

// rows [0, *n_dev) of out = rows of one of two tables: index < split -> t0[index], else t1[index - split]   (width D floats)
__global__ __launch_bounds__(256) void gather_rows2_kernel(const float* __restrict__ t0, const float* __restrict__ t1, int split,
                                                           const int* __restrict__ index, const int* __restrict__ n_dev,
                                                           int n_cap, int D, float* __restrict__ out) {
    const int n = *n_dev < n_cap ? *n_dev : n_cap;
    const int d4 = D / 4;
    const long long total = (long long)n * d4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i / d4), q = (int)(i - (long long)r * d4);
        const int ix = index[r];
        const float4* src = reinterpret_cast<const float4*>(ix < split ? t0 + (size_t)ix * D : t1 + (size_t)(ix - split) * D);
        reinterpret_cast<float4*>(out + (size_t)r * D)[q] = src[q];
    }
}

// the transposed operation: d(t0)[index] += scale * g, d(t1)[index - split] += scale * g   (fp32 atomics; a row is
// referenced by at most a handful of features; `scale` = a device scalar, e.g. the incoming gradient of the loss)
__global__ __launch_bounds__(256) void scatter_rows2_kernel(const float* __restrict__ g, const int* __restrict__ index,
                                                            const int* __restrict__ n_dev, int n_cap, int D, int split,
                                                            const float* __restrict__ scale, float alpha,
                                                            float* __restrict__ d0, float* __restrict__ d1) {
    const int n = *n_dev < n_cap ? *n_dev : n_cap;
    const float sc = (scale ? *scale : 1.0f) * alpha;
    const long long total = (long long)n * D;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i / D), d = (int)(i - (long long)r * D);
        const int ix = index[r];
        float* dst = ix < split ? d0 + (size_t)ix * D : d1 + (size_t)(ix - split) * D;
        atomicAdd(dst + d, g[i] * sc);
    }
}

// rows of a row-major table gathered by a device list of device length (any element size that is a multiple of 16 bytes per row)
__global__ __launch_bounds__(256) void gather_rows_bytes_kernel(const uint4* __restrict__ src, long long ld16,
                                                                const int* __restrict__ index, const int* __restrict__ n_dev,
                                                                int n_cap, int w16, uint4* __restrict__ out, long long ldo16) {
    const int n = *n_dev < n_cap ? *n_dev : n_cap;
    const long long total = (long long)n * w16;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i / w16), q = (int)(i - (long long)r * w16);
        out[(long long)r * ldo16 + q] = src[(long long)index[r] * ld16 + q];
    }
}

__global__ __launch_bounds__(256) void zero_rows_dyn_kernel(uint4* __restrict__ p, long long ld16, int w16, const int* __restrict__ n_dev,
                                                            int n_cap) {
    const int n = *n_dev < n_cap ? *n_dev : n_cap;
    const long long total = (long long)n * w16;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i / w16), q = (int)(i - (long long)r * w16);
        p[(long long)r * ld16 + q] = make_uint4(0, 0, 0, 0);
    }
}

int grid_for(long long items, int per_block = 256, int cap = 4096) {
    long long b = (items + per_block - 1) / per_block;
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace

ODW_EXPORT int odw_loss_lists_a(const int* grp, const int* cls_order, int G, const int* counts, const int* rows, int maxpos,
                                int pstride, int sum_p, int n_cls1, int e_cap, int* scal, int* e0, int* roi_index,
                                int* bank_index, int* bank_off, int* bank_cnt, void* row_tab6, void* row_tab7, void* stream_) {
    ODW_REQUIRE(G >= 1 && G <= 256 && maxpos >= 1 && pstride >= 1 && sum_p >= 1 && n_cls1 >= 1 && e_cap >= 1,
                "loss_lists_a: bad dims (G=%d: at most 256 (image, class) groups per step)", G);
    ODW_REQUIRE(grp && cls_order && counts && rows && scal && e0 && roi_index && bank_index && bank_off && bank_cnt && row_tab6 &&
                    row_tab7, "loss_lists_a: null pointer");
    ODW_REQUIRE((((uintptr_t)row_tab6) & 15) == 0 && (((uintptr_t)row_tab7) & 15) == 0, "loss_lists_a: row tables must be 16-byte aligned");
    ListsA a;
    a.grp = grp; a.cls_order = cls_order; a.counts = counts; a.rows = rows; a.G = G; a.maxpos = maxpos; a.pstride = pstride;
    a.sum_p = sum_p; a.n_cls1 = n_cls1; a.e_cap = e_cap; a.scal = scal; a.e0 = e0; a.roi_index = roi_index; a.bank_index = bank_index;
    a.bank_off = bank_off; a.bank_cnt = bank_cnt; a.row_tab6 = (uint4*)row_tab6; a.row_tab7 = (uint4*)row_tab7;
    loss_lists_a_kernel<<<1, kThreads, 0, (hipStream_t)stream_>>>(a);
    ODW_CHECK_LAUNCH("loss_lists_a_kernel");
    return ODW_OK;
}

ODW_EXPORT int odw_loss_lists_b(const int* grp, const int* cls_order, int G, const int* img_off, const int* n_pos,
                                const int* pos_cls, int n_img, int maxpos, int pstride, int sum_p, const int* scal_a, const int* e0,
                                const int* roi_index, const int* bank_index, const int* bank_off, const int* bank_cnt,
                                const int* fresh_idx, const int* fresh_cnt, const int* gt_cnt, int gt_max, const float* final_score,
                                int fs_cols, const float* colstat, int cs_ld, int cs_off, int n_cap, int a_cap, int e_cap, int p64,
                                int* scal, int* feat_index, int* labels, float* weights, int* act_rows, int* roi_index_all,
                                int* sticky, void* stream_) {
    ODW_REQUIRE(G >= 1 && G <= 256 && n_img >= 1 && maxpos >= 1 && pstride >= 1 && sum_p >= 1 && n_cap >= 1 && a_cap >= 1 && e_cap >= 1,
                "loss_lists_b: bad dims");
    ODW_REQUIRE(grp && cls_order && img_off && n_pos && pos_cls && scal_a && e0 && roi_index && bank_index && bank_off && bank_cnt &&
                    fresh_idx && fresh_cnt && gt_cnt && final_score && colstat && scal && feat_index && labels && weights &&
                    act_rows && roi_index_all, "loss_lists_b: null pointer");
    ListsB b;
    b.grp = grp; b.cls_order = cls_order; b.img_off = img_off; b.n_pos = n_pos; b.pos_cls = pos_cls; b.scal_a = scal_a; b.e0 = e0;
    b.roi_index = roi_index; b.bank_index = bank_index; b.bank_off = bank_off; b.bank_cnt = bank_cnt; b.fresh_idx = fresh_idx;
    b.fresh_cnt = fresh_cnt; b.gt_cnt = gt_cnt; b.final_score = final_score; b.colstat = colstat; b.G = G; b.n_img = n_img;
    b.maxpos = maxpos; b.pstride = pstride; b.sum_p = sum_p; b.fs_cols = fs_cols; b.cs_ld = cs_ld; b.cs_off = cs_off;
    b.n_cap = n_cap; b.a_cap = a_cap; b.e_cap = e_cap; b.p64 = p64; b.gt_max = gt_max; b.scal = scal; b.feat_index = feat_index;
    b.labels = labels; b.weights = weights; b.act_rows = act_rows; b.roi_index_all = roi_index_all; b.sticky = sticky;
    const int W32 = (sum_p + 31) / 32;
    const size_t lds = (size_t)W32 * 4 + (size_t)(W32 + 1) * 4;
    ODW_REQUIRE(lds <= (size_t)ODW_LDS_BYTES - 4096, "loss_lists_b: %d proposals in the batch (the bit set must fit LDS)", sum_p);
    ODW_CHECK_HIP(odw_set_max_lds(reinterpret_cast<const void*>(loss_lists_b_kernel), (int)lds), "loss_lists_b attr");
    loss_lists_b_kernel<<<1, kThreads, lds, (hipStream_t)stream_>>>(b);
    ODW_CHECK_LAUNCH("loss_lists_b_kernel");
    return ODW_OK;
}

// out[r] = (index[r] < split ? t0[index[r]] : t1[index[r] - split]) for r < *n_dev: fp32 rows of D values (D % 4 == 0)
ODW_EXPORT int odw_gather_rows2_dyn(const float* t0, const float* t1, int split, const int* index, const int* n_dev, int n_cap,
                                    int D, float* out, void* stream_) {
    ODW_REQUIRE(n_cap >= 1 && D >= 4 && D % 4 == 0 && t0 && t1 && index && n_dev && out, "gather_rows2_dyn: bad arguments");
    ODW_REQUIRE(((((uintptr_t)t0) | ((uintptr_t)t1) | ((uintptr_t)out)) & 15) == 0, "gather_rows2_dyn: 16-byte alignment");
    gather_rows2_kernel<<<grid_for((long long)n_cap * (D / 4)), 256, 0, (hipStream_t)stream_>>>(t0, t1, split, index, n_dev, n_cap, D, out);
    ODW_CHECK_LAUNCH("gather_rows2_kernel");
    return ODW_OK;
}

ODW_EXPORT int odw_scatter_rows2_dyn(const float* g, const int* index, const int* n_dev, int n_cap, int D, int split,
                                     const float* scale, float alpha, float* d0, float* d1, void* stream_) {
    ODW_REQUIRE(n_cap >= 1 && D >= 1 && g && index && n_dev && d0 && d1, "scatter_rows2_dyn: bad arguments");
    scatter_rows2_kernel<<<grid_for((long long)n_cap * D), 256, 0, (hipStream_t)stream_>>>(g, index, n_dev, n_cap, D, split, scale, alpha, d0, d1);
    ODW_CHECK_LAUNCH("scatter_rows2_kernel");
    return ODW_OK;
}

// out[r][0 : row_bytes) = src[index[r]][0 : row_bytes) for r < *n_dev; row_bytes, ld_src_bytes, ld_out_bytes multiples of 16
ODW_EXPORT int odw_gather_rows_dyn(const void* src, int64_t ld_src_bytes, const int* index, const int* n_dev, int n_cap,
                                   int64_t row_bytes, void* out, int64_t ld_out_bytes, void* stream_) {
    ODW_REQUIRE(n_cap >= 1 && row_bytes >= 16 && row_bytes % 16 == 0 && ld_src_bytes % 16 == 0 && ld_out_bytes % 16 == 0 && src &&
                    index && n_dev && out && ((((uintptr_t)src) | ((uintptr_t)out)) & 15) == 0, "gather_rows_dyn: bad arguments");
    gather_rows_bytes_kernel<<<grid_for((long long)n_cap * (row_bytes / 16), 256, 16384), 256, 0, (hipStream_t)stream_>>>(
        (const uint4*)src, ld_src_bytes / 16, index, n_dev, n_cap, (int)(row_bytes / 16), (uint4*)out, ld_out_bytes / 16);
    ODW_CHECK_LAUNCH("gather_rows_bytes_kernel");
    return ODW_OK;
}

// rows [0, *n_dev) of p zeroed (row_bytes, ld_bytes multiples of 16)
ODW_EXPORT int odw_zero_rows_dyn(void* p, int64_t ld_bytes, int64_t row_bytes, const int* n_dev, int n_cap, void* stream_) {
    ODW_REQUIRE(n_cap >= 1 && row_bytes >= 16 && row_bytes % 16 == 0 && ld_bytes % 16 == 0 && p && n_dev && (((uintptr_t)p) & 15) == 0,
                "zero_rows_dyn: bad arguments");
    zero_rows_dyn_kernel<<<grid_for((long long)n_cap * (row_bytes / 16)), 256, 0, (hipStream_t)stream_>>>(
        (uint4*)p, ld_bytes / 16, (int)(row_bytes / 16), n_dev, n_cap);
    ODW_CHECK_LAUNCH("zero_rows_dyn_kernel");
    return ODW_OK;
}
