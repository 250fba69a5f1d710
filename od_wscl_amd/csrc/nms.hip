// nms.hip -- non-maximum suppression + pairwise box IoU for gfx950.
//
// Three semantics behind one entry point (include/odwscl.h):
//   TV     torchvision.ops.nms, the one the reference's hot path calls
//          (structures/boxlist_ops.py:9,31-32,56-57 <- utils/utils.py:28-33 <-
//          roi_heads/weak_head/loss.py:332)
//   WT_GE  wetectron _C.nms CPU rule  (csrc/cpu/nms_cpu.cpp:6-65, ovr >= thr)
//   WT_GT  wetectron _C.nms CUDA rule (csrc/cuda/nms.cu:13-131,  ovr >  thr)
//
// Everything stays on the device (the reference copies the bitmask to the
// host and reduces there, nms.cu:100-125):
//   1. one workgroup bitonic-sorts (score desc, index asc) in LDS  (n <= 8192)
//   2. wave64-native bitmask: a 64-lane wave owns 64 sorted rows, each lane
//      builds one 64-bit word per column block -- one u64 = one wavefront
//   3. one wave walks the 64-row blocks: the in-block chain runs on the
//      diagonal words held in registers (v_readlane broadcast), the
//      off-diagonal ORs are independent coalesced row loads
//   4. kept flags are compacted by the same workgroup (LDS scan) into the
//      order the semantics require.
#include "odw_common.h"

namespace {

constexpr int kSortThreads = 1024;

struct Ws {
    float* sboxes;           // (npad,4) boxes in sorted order
    int* order;              // (npad) original index of sorted position
    unsigned long long* mask;  // (n, nblk)
};

__host__ __device__ inline int nms_nblk(int n) { return (n + 63) / 64; }

__device__ __forceinline__ bool before(float sa, int ia, float sb, int ib) {
    return (sa > sb) || (sa == sb && ia < ib);
}

__global__ __launch_bounds__(kSortThreads) void nms_sort_kernel(const float* __restrict__ boxes,
                                                               const float* __restrict__ scores, int n,
                                                               int npad, float* __restrict__ sboxes,
                                                               int* __restrict__ order) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* ks = reinterpret_cast<float*>(smem);
    int* ki = reinterpret_cast<int*>(smem + (size_t)npad * 4);
    for (int i = threadIdx.x; i < npad; i += blockDim.x) {
        // padding sorts last: -inf score, index beyond n
        ks[i] = i < n ? scores[i] : -__builtin_inff();
        ki[i] = i;
    }
    __syncthreads();
    for (int k = 2; k <= npad; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < npad; i += blockDim.x) {
                int p = i ^ j;
                if (p > i) {
                    bool up = ((i & k) == 0);
                    float sa = ks[i], sb = ks[p];
                    int ia = ki[i], ib = ki[p];
                    bool a_first = before(sa, ia, sb, ib);
                    if (up ? !a_first : a_first) {
                        ks[i] = sb; ks[p] = sa;
                        ki[i] = ib; ki[p] = ia;
                    }
                }
            }
            __syncthreads();
        }
    }
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        int o = ki[i];
        order[i] = o;
        reinterpret_cast<float4*>(sboxes)[i] = reinterpret_cast<const float4*>(boxes)[o];
    }
}

template <int MODE>
__device__ __forceinline__ bool overlaps(const float4 a, const float4 b, float thr) {
    if (MODE == ODW_NMS_TV) {
        float aa = (a.z - a.x) * (a.w - a.y);
        float ab = (b.z - b.x) * (b.w - b.y);
        float w = fmaxf(0.0f, fminf(a.z, b.z) - fmaxf(a.x, b.x));
        float h = fmaxf(0.0f, fminf(a.w, b.w) - fmaxf(a.y, b.y));
        float inter = w * h;
        return inter / (aa + ab - inter) > thr;
    } else {
        float aa = (a.z - a.x + 1) * (a.w - a.y + 1);
        float ab = (b.z - b.x + 1) * (b.w - b.y + 1);
        float w = fmaxf(0.0f, fminf(a.z, b.z) - fmaxf(a.x, b.x) + 1);
        float h = fmaxf(0.0f, fminf(a.w, b.w) - fmaxf(a.y, b.y) + 1);
        float inter = w * h;
        float ovr = inter / (aa + ab - inter);
        return MODE == ODW_NMS_WT_GE ? (ovr >= thr) : (ovr > thr);
    }
}

// grid (col_blk, row_blk), 64 threads.  Only col_blk >= row_blk is needed.
template <int MODE>
__global__ __launch_bounds__(64) void nms_mask_kernel(const float* __restrict__ sboxes, int n, float thr,
                                                      unsigned long long* __restrict__ mask) {
    const int cb = blockIdx.x, rb = blockIdx.y;
    if (cb < rb) return;
    __shared__ float4 cbox[64];
    const int nblk = nms_nblk(n);
    const int col = cb * 64 + threadIdx.x;
    cbox[threadIdx.x] = col < n ? reinterpret_cast<const float4*>(sboxes)[col] : make_float4(0, 0, 0, 0);
    __syncthreads();
    const int row = rb * 64 + threadIdx.x;
    if (row >= n) return;
    const float4 me = reinterpret_cast<const float4*>(sboxes)[row];
    const int ncol = min(64, n - cb * 64);
    unsigned long long bits = 0;
    const int start = (cb == rb) ? threadIdx.x + 1 : 0;
    for (int j = start; j < ncol; ++j)
        if (overlaps<MODE>(me, cbox[j], thr)) bits |= 1ull << j;
    mask[(size_t)row * nblk + cb] = bits;
}

// One workgroup.  Wave 0 runs the greedy chain; then everyone compacts.
__global__ __launch_bounds__(kSortThreads) void nms_reduce_kernel(const unsigned long long* __restrict__ mask,
                                                                 const int* __restrict__ order, int n,
                                                                 int mode, long long* __restrict__ keep,
                                                                 int* __restrict__ n_keep) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // keptpos[i] = 1 when sorted position i survives
    unsigned char* kept = smem;                                  // n bytes (sorted positions)
    int* scan = reinterpret_cast<int*>(smem + odw_align_up(n, 16));  // per-thread counts
    const int nblk = nms_nblk(n);
    for (int i = threadIdx.x; i < n; i += blockDim.x) kept[i] = 0;
    __syncthreads();
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        // removed-words owned by this lane: word w = lane + 64*s, s < 2 (n <= 8192 -> nblk <= 128)
        unsigned long long rem0 = 0, rem1 = 0;
        for (int kb = 0; kb < nblk; ++kb) {
            const int rows = min(64, n - kb * 64);
            // diagonal word of row kb*64+lane
            unsigned long long diag = lane < rows ? mask[(size_t)(kb * 64 + lane) * nblk + kb] : 0ull;
            unsigned long long mine = (kb < 64) ? rem0 : rem1;
            // broadcast the removed word of block kb from its owner lane
            unsigned int lo = __shfl((unsigned int)(mine & 0xffffffffu), kb & 63);
            unsigned int hi = __shfl((unsigned int)(mine >> 32), kb & 63);
            unsigned long long dead = ((unsigned long long)hi << 32) | lo;
            unsigned long long keptbits = 0;
            for (int r = 0; r < rows; ++r) {
                unsigned int dlo = __shfl((unsigned int)(diag & 0xffffffffu), r);
                unsigned int dhi = __shfl((unsigned int)(diag >> 32), r);
                if (!((dead >> r) & 1ull)) {
                    keptbits |= 1ull << r;
                    dead |= ((unsigned long long)dhi << 32) | dlo;
                }
            }
            if (lane < rows && ((keptbits >> lane) & 1ull)) kept[kb * 64 + lane] = 1;
            // OR the kept rows' off-diagonal words into the removed set (independent loads)
            unsigned long long acc0 = 0, acc1 = 0;
            unsigned long long kb_bits = keptbits;
            while (kb_bits) {
                int r = __builtin_ctzll(kb_bits);
                kb_bits &= kb_bits - 1;
                const unsigned long long* rowp = mask + (size_t)(kb * 64 + r) * nblk;
                if (lane > kb && lane < nblk) acc0 |= rowp[lane];
                if (lane + 64 > kb && lane + 64 < nblk) acc1 |= rowp[lane + 64];
            }
            rem0 |= acc0;
            rem1 |= acc1;
        }
    }
    __syncthreads();
    // compaction.  TV: sorted order.  WT: ascending original index.
    unsigned char* flag = kept;
    unsigned char* oflag = smem + odw_align_up(n, 16) + kSortThreads * 4;  // n bytes
    if (mode != ODW_NMS_TV) {
        for (int i = threadIdx.x; i < n; i += blockDim.x) oflag[i] = 0;
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += blockDim.x)
            if (kept[i]) oflag[order[i]] = 1;
        __syncthreads();
        flag = oflag;
    }
    const int per = (n + kSortThreads - 1) / kSortThreads;
    const int lo = threadIdx.x * per, hi = min(n, lo + per);
    int cnt = 0;
    for (int i = lo; i < hi; ++i) cnt += flag[i];
    scan[threadIdx.x] = cnt;
    __syncthreads();
    // Hillis-Steele inclusive scan over 1024 counts
    for (int off = 1; off < kSortThreads; off <<= 1) {
        int v = threadIdx.x >= off ? scan[threadIdx.x - off] : 0;
        __syncthreads();
        scan[threadIdx.x] += v;
        __syncthreads();
    }
    int base = scan[threadIdx.x] - cnt;
    for (int i = lo; i < hi; ++i)
        if (flag[i]) keep[base++] = (mode == ODW_NMS_TV) ? (long long)order[i] : (long long)i;
    if (threadIdx.x == kSortThreads - 1) n_keep[0] = scan[threadIdx.x];
}

__global__ void box_iou_kernel(const float* __restrict__ a, int N, const float* __restrict__ b, int M,
                               float* __restrict__ iou) {
    const size_t total = (size_t)N * M;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (size_t)gridDim.x * blockDim.x) {
        int i = (int)(t / M), j = (int)(t - (size_t)i * M);
        float4 p = reinterpret_cast<const float4*>(a)[i];
        float4 q = reinterpret_cast<const float4*>(b)[j];
        float ap = (p.z - p.x + 1) * (p.w - p.y + 1);
        float aq = (q.z - q.x + 1) * (q.w - q.y + 1);
        float w = fminf(p.z, q.z) - fmaxf(p.x, q.x) + 1;
        float h = fminf(p.w, q.w) - fmaxf(p.y, q.y) + 1;
        w = w < 0 ? 0 : w;
        h = h < 0 ? 0 : h;
        float inter = w * h;
        iou[t] = inter / (ap + aq - inter);
    }
}

int next_pow2(int n) { int p = 1; while (p < n) p <<= 1; return p; }

Ws carve(void* ws, int n) {
    Ws w;
    unsigned char* p = (unsigned char*)ws;
    int npad = next_pow2(n);
    w.sboxes = (float*)p; p += odw_align_up((int64_t)npad * 16, 256);
    w.order = (int*)p;    p += odw_align_up((int64_t)npad * 4, 256);
    w.mask = (unsigned long long*)p;
    return w;
}

}  // namespace

ODW_EXPORT int64_t odw_nms_workspace(int n) {
    if (n < 1) n = 1;
    int npad = next_pow2(n);
    return odw_align_up((int64_t)npad * 16, 256) + odw_align_up((int64_t)npad * 4, 256) +
           odw_align_up((int64_t)n * nms_nblk(n) * 8, 256);
}

ODW_EXPORT int odw_nms(const float* boxes, const float* scores, int n, float thr, int mode, int64_t* keep,
                       int32_t* n_keep, void* workspace, int64_t workspace_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    ODW_REQUIRE(n >= 0 && n <= ODW_NMS_MAX_N, "nms: n=%d out of range (max %d)", n, ODW_NMS_MAX_N);
    ODW_REQUIRE(mode >= 0 && mode <= 2, "nms: bad mode %d", mode);
    ODW_REQUIRE(n_keep, "nms: null n_keep");
    if (n == 0) {  // nms_cpu.cpp:13-15
        ODW_CHECK_HIP(hipMemsetAsync(n_keep, 0, 4, stream), "nms memset");
        return ODW_OK;
    }
    ODW_REQUIRE(boxes && scores && keep, "nms: null pointer");
    ODW_REQUIRE((((uintptr_t)boxes) & 15) == 0, "nms: boxes must be 16-byte aligned");
    if (!workspace || workspace_bytes < odw_nms_workspace(n)) {
        odw_set_error("nms: workspace %lld < %lld bytes", (long long)workspace_bytes,
                      (long long)odw_nms_workspace(n));
        return ODW_EWORKSPACE;
    }
    Ws w = carve(workspace, n);
    const int npad = next_pow2(n);
    const int nblk = nms_nblk(n);
    nms_sort_kernel<<<1, kSortThreads, (size_t)npad * 8, stream>>>(boxes, scores, n, npad, w.sboxes, w.order);
    ODW_CHECK_LAUNCH("nms_sort_kernel");
    dim3 grid(nblk, nblk);
    switch (mode) {
        case ODW_NMS_TV: nms_mask_kernel<ODW_NMS_TV><<<grid, 64, 0, stream>>>(w.sboxes, n, thr, w.mask); break;
        case ODW_NMS_WT_GE: nms_mask_kernel<ODW_NMS_WT_GE><<<grid, 64, 0, stream>>>(w.sboxes, n, thr, w.mask); break;
        default: nms_mask_kernel<ODW_NMS_WT_GT><<<grid, 64, 0, stream>>>(w.sboxes, n, thr, w.mask); break;
    }
    ODW_CHECK_LAUNCH("nms_mask_kernel");
    size_t lds = (size_t)odw_align_up(n, 16) * 2 + kSortThreads * 4;
    nms_reduce_kernel<<<1, kSortThreads, lds, stream>>>(w.mask, w.order, n, mode, (long long*)keep, n_keep);
    ODW_CHECK_LAUNCH("nms_reduce_kernel");
    return ODW_OK;
}

ODW_EXPORT int odw_box_iou(const float* a, int N, const float* b, int M, float* iou, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    ODW_REQUIRE(N >= 0 && M >= 0, "box_iou: bad dims");
    if (N == 0 || M == 0) return ODW_OK;
    ODW_REQUIRE(a && b && iou, "box_iou: null pointer");
    ODW_REQUIRE((((uintptr_t)a) & 15) == 0 && (((uintptr_t)b) & 15) == 0, "box_iou: boxes must be 16-byte aligned");
    size_t total = (size_t)N * M;
    int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    box_iou_kernel<<<grid, 256, 0, stream>>>(a, N, b, M, iou);
    ODW_CHECK_LAUNCH("box_iou_kernel");
    return ODW_OK;
}
