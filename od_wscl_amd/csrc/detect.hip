// detect.hip -- the inference tail of the ROI head on the device: box decoding, clipping, per-class score
// threshold, per-class NMS, in ONE launch.
//
// Reference: PostProcessor.forward / filter_results (wetectron/modeling/roi_heads/box_head/inference.py:41-90,
// 216-258) reached from ROIWeakRegHead.testing_forward (roi_heads/weak_head/weak_head.py:124-145); BoxCoder.decode
// (modeling/box_coder.py:52-95); BoxList.clip_to_image (structures/bounding_box.py:218-229, TO_REMOVE = 1);
// boxlist_nms -> torchvision.ops.nms (structures/boxlist_ops.py:13-36: IoU without +1, suppress when > thr).
// The reference loops over the classes in Python with a nonzero() host synchronisation and an NMS launch each;
// here workgroup (image, class) decodes its class's boxes straight into LDS, sorts the candidates (score >
// thresh) by descending score (ties: lower proposal index first), runs the greedy chain on the LDS-resident boxes
// and writes the survivors in kept order.  The final "keep the best max_det over all classes" (kthvalue, ties
// kept) needs the counts on the host anyway and stays there.
#include "odw_common.h"

namespace {

constexpr int kThreads = 512;

struct DetArgs {
    const float* prob;      // (sumP, C)
    const float* reg;       // (sumP, 4*C) or (sumP, 4) class-agnostic, or null (boxes are used as they are)
    const float* boxes;     // (sumP, 4) xyxy -- or (sumP, C, 4) already decoded per class (per_class)
    const int* img_off;     // (n_img + 1)
    const float* img_wh;    // (n_img, 2)
    int C, ld_reg, cls_agnostic, per_class;
    float wx, wy, ww, wh, xform_clip, score_thresh, nms_thr;
    int pstride;            // output slots per (image, class)
    float* out_boxes;       // (n_img, C-1, pstride, 4)
    float* out_scores;      // (n_img, C-1, pstride)
    int* out_index;         // (n_img, C-1, pstride) proposal index inside the image
    int* out_count;         // (n_img, C-1)
};

__device__ __forceinline__ bool before(float sa, int ia, float sb, int ib) { return (sa > sb) || (sa == sb && ia < ib); }

__device__ __forceinline__ bool tv_overlap(const float4 a, const float4 b, float thr) {   // torchvision nms
    float aa = (a.z - a.x) * (a.w - a.y);
    float ab = (b.z - b.x) * (b.w - b.y);
    float w = fmaxf(0.0f, fminf(a.z, b.z) - fmaxf(a.x, b.x));
    float h = fmaxf(0.0f, fminf(a.w, b.w) - fmaxf(a.y, b.y));
    float inter = w * h;
    return inter / (aa + ab - inter) > thr;
}

__global__ __launch_bounds__(kThreads) void detect_classes_kernel(DetArgs a, int ppow2) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float4* sbox = reinterpret_cast<float4*>(smem);                 // ppow2 boxes (by proposal, then by sorted position)
    float4* sbox2 = sbox + ppow2;                                   // sorted copy
    float* ks = reinterpret_cast<float*>(sbox2 + ppow2);            // sort keys
    int* ki = reinterpret_cast<int*>(ks + ppow2);                   // sort ids
    unsigned char* alive = reinterpret_cast<unsigned char*>(ki + ppow2);
    __shared__ int s_n, s_k;
    __shared__ unsigned long long s_mask[64];
    __shared__ float4 s_kbox[64];

    const int img = blockIdx.x, j = blockIdx.y + 1;                 // class 0 = background is skipped
    const int base = a.img_off[img], P = a.img_off[img + 1] - base;
    const float iw = a.per_class ? 0.0f : a.img_wh[2 * img], ih = a.per_class ? 0.0f : a.img_wh[2 * img + 1];
    int cnt_local = 0;
    for (int r = threadIdx.x; r < ppow2; r += kThreads) {
        float s = -__builtin_inff();
        int id = 0x40000000 + r;
        if (r < P) {
            const float sc = a.prob[(size_t)(base + r) * a.C + j];
            const float4 b = reinterpret_cast<const float4*>(a.boxes)[a.per_class ? (size_t)(base + r) * a.C + j
                                                                                   : (size_t)(base + r)];
            float4 o = b;
            if (a.per_class) {          // filter_results on a merged boxlist (engine/bbox_aug.py:70-75): boxes as given
                sbox[r] = o;
                if (sc > a.score_thresh) { s = sc; id = r; ++cnt_local; }
                ks[r] = s;
                ki[r] = id;
                continue;
            }
            if (a.reg) {
                const float* rc = a.reg + (size_t)(base + r) * a.ld_reg + (a.cls_agnostic ? 0 : 4 * j);
                const float w = b.z - b.x + 1.0f, h = b.w - b.y + 1.0f;
                const float cx = b.x + 0.5f * w, cy = b.y + 0.5f * h;
                const float dx = rc[0] / a.wx, dy = rc[1] / a.wy;
                const float dw = fminf(rc[2] / a.ww, a.xform_clip), dh = fminf(rc[3] / a.wh, a.xform_clip);
                const float pcx = dx * w + cx, pcy = dy * h + cy;
                const float pw = expf(dw) * w, ph = expf(dh) * h;
                o.x = pcx - 0.5f * pw;
                o.y = pcy - 0.5f * ph;
                o.z = pcx + 0.5f * pw - 1.0f;
                o.w = pcy + 0.5f * ph - 1.0f;
            }
            o.x = fminf(fmaxf(o.x, 0.0f), iw - 1.0f);
            o.y = fminf(fmaxf(o.y, 0.0f), ih - 1.0f);
            o.z = fminf(fmaxf(o.z, 0.0f), iw - 1.0f);
            o.w = fminf(fmaxf(o.w, 0.0f), ih - 1.0f);
            sbox[r] = o;
            if (sc > a.score_thresh) { s = sc; id = r; ++cnt_local; }
        }
        ks[r] = s;
        ki[r] = id;
    }
    // number of candidates
    __shared__ int red[kThreads];
    red[threadIdx.x] = cnt_local;
    __syncthreads();
    for (int off = kThreads / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    const int n = red[0];
    // bitonic sort: descending score, ties by ascending proposal index; non-candidates last
    for (int k = 2; k <= ppow2; k <<= 1) {
        for (int jj = k >> 1; jj > 0; jj >>= 1) {
            // one compare-exchange pair per thread and round: pair q -> t = q with a 0 inserted at bit log2(jj)
            for (int q = threadIdx.x; q < (ppow2 >> 1); q += kThreads) {
                const int t = ((q & ~(jj - 1)) << 1) | (q & (jj - 1)), p = t | jj;
                const bool up = ((t & k) == 0);
                const float sa = ks[t], sb = ks[p];
                const int ia = ki[t], ib = ki[p];
                const bool a_first = before(sa, ia, sb, ib);
                if (up ? !a_first : a_first) { ks[t] = sb; ks[p] = sa; ki[t] = ib; ki[p] = ia; }
            }
            __syncthreads();
        }
    }
    for (int t = threadIdx.x; t < n; t += kThreads) {
        alive[t] = 1;
        sbox2[t] = sbox[ki[t]];
    }
    __syncthreads();
    const size_t slot = ((size_t)img * (a.C - 1) + (j - 1)) * a.pstride;
    int kept = 0;
    // Greedy NMS in windows of 64 sorted positions (the scheme of csrc/discover.hip): all pairs of a window tested in
    // parallel into 64 suppression masks, one wave resolving the window in registers (one readlane + mask step per kept
    // position), everyone suppressing the tail against the boxes the window kept.  Same kept set and order as the
    // sequential greedy rule; was one barrier pair + a serial scan by thread 0 per kept box.
    for (int k0 = 0; k0 < n; k0 += 64) {
        const int wn = n - k0 < 64 ? n - k0 : 64;
        if (threadIdx.x < 64) s_mask[threadIdx.x] = 0ull;
        __syncthreads();
        for (int pidx = threadIdx.x; pidx < 64 * 64; pidx += kThreads) {
            const int i = pidx >> 6, jx = pidx & 63;
            if (jx > i && jx < wn && alive[k0 + i] && alive[k0 + jx] && tv_overlap(sbox2[k0 + i], sbox2[k0 + jx], a.nms_thr))
                atomicOr(&s_mask[i], 1ull << jx);
        }
        __syncthreads();
        if (threadIdx.x < 64) {                     // wave 0
            const int l = threadIdx.x;
            const unsigned long long m = s_mask[l];
            const unsigned mlo = (unsigned)m, mhi = (unsigned)(m >> 32);
            unsigned long long rem = __ballot(l < wn && alive[k0 + (l < wn ? l : 0)]), keptw = 0ull;
            while (rem) {                           // wave-uniform
                const int i = __builtin_amdgcn_readfirstlane(__ffsll((long long)rem) - 1);
                const unsigned long long mi = (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)mlo, i) |
                                              ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)mhi, i) << 32);
                keptw |= 1ull << i;
                rem &= ~(mi | (1ull << i));
            }
            const bool mine = (keptw >> l) & 1ull;
            if (l < wn) alive[k0 + l] = mine ? 1 : 0;
            if (mine) {
                const int pos = __popcll(keptw & ((1ull << l) - 1ull));
                const float4 bk = sbox2[k0 + l];
                reinterpret_cast<float4*>(a.out_boxes)[slot + kept + pos] = bk;
                a.out_scores[slot + kept + pos] = ks[k0 + l];
                a.out_index[slot + kept + pos] = ki[k0 + l];
                s_kbox[pos] = bk;
            }
            if (l == 0) s_k = __popcll(keptw);
        }
        __syncthreads();
        const int kw = s_k;
        kept += kw;
        if (kw > 0) {
            for (int t = k0 + 64 + threadIdx.x; t < n; t += kThreads) {
                if (!alive[t]) continue;
                const float4 bt = sbox2[t];
                for (int q = 0; q < kw; ++q)
                    if (tv_overlap(s_kbox[q], bt, a.nms_thr)) { alive[t] = 0; break; }
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) a.out_count[img * (a.C - 1) + (j - 1)] = kept;
    (void)s_n;
}

// PostProcessor.forward with bbox_aug_enabled (inference.py:57-90): every proposal decoded for every class and clipped,
// nothing filtered -- the (sumP, C, 4) boxlist the test-time augmentation merges.  One thread per (proposal, class).
__global__ __launch_bounds__(256) void detect_decode_kernel(DetArgs a, int n_img, int sum_p, float* __restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= sum_p * a.C) return;
    const int r = t / a.C, j = t - r * a.C;
    int img = 0;
    while (img + 1 < n_img && r >= a.img_off[img + 1]) ++img;
    const float iw = a.img_wh[2 * img], ih = a.img_wh[2 * img + 1];
    const float4 b = reinterpret_cast<const float4*>(a.boxes)[r];
    float4 o = b;
    if (a.reg) {
        const float* rc = a.reg + (size_t)r * a.ld_reg + (a.cls_agnostic ? 0 : 4 * j);
        const float w = b.z - b.x + 1.0f, h = b.w - b.y + 1.0f;
        const float cx = b.x + 0.5f * w, cy = b.y + 0.5f * h;
        const float dx = rc[0] / a.wx, dy = rc[1] / a.wy;
        const float dw = fminf(rc[2] / a.ww, a.xform_clip), dh = fminf(rc[3] / a.wh, a.xform_clip);
        const float pcx = dx * w + cx, pcy = dy * h + cy;
        const float pw = expf(dw) * w, ph = expf(dh) * h;
        o.x = pcx - 0.5f * pw;
        o.y = pcy - 0.5f * ph;
        o.z = pcx + 0.5f * pw - 1.0f;
        o.w = pcy + 0.5f * ph - 1.0f;
    }
    o.x = fminf(fmaxf(o.x, 0.0f), iw - 1.0f);
    o.y = fminf(fmaxf(o.y, 0.0f), ih - 1.0f);
    o.z = fminf(fmaxf(o.z, 0.0f), iw - 1.0f);
    o.w = fminf(fmaxf(o.w, 0.0f), ih - 1.0f);
    reinterpret_cast<float4*>(out)[t] = o;
}

int pow2_at_least(int n) { int p = 1; while (p < n) p <<= 1; return p; }

int launch_detect(DetArgs a, int n_img, int max_p, void* stream_) {
    const int ppow2 = pow2_at_least(max_p);
    const size_t lds = (size_t)ppow2 * (16 + 16 + 4 + 4 + 1) + 64;
    ODW_CHECK_HIP(odw_set_max_lds(reinterpret_cast<const void*>(detect_classes_kernel),
                                      (int)lds), "detect attr");
    detect_classes_kernel<<<dim3((unsigned)n_img, (unsigned)(a.C - 1)), kThreads, lds, (hipStream_t)stream_>>>(a, ppow2);
    ODW_CHECK_LAUNCH("detect_classes_kernel");
    return ODW_OK;
}

}  // namespace

ODW_EXPORT int odw_detect_postprocess(const float* prob, int C, const float* reg, int ld_reg, int cls_agnostic,
                                      const float* boxes, const int* img_off, const float* img_wh, int n_img,
                                      int max_p, float wx, float wy, float ww, float wh, float xform_clip,
                                      float score_thresh, float nms_thr, int pstride, float* out_boxes,
                                      float* out_scores, int* out_index, int* out_count, void* stream_) {
    ODW_REQUIRE(n_img >= 0 && C >= 2 && max_p >= 1 && max_p <= 4096 && pstride >= max_p,
                "detect_postprocess: bad dims (at most 4096 proposals per image, got %d)", max_p);
    if (n_img == 0) return ODW_OK;
    ODW_REQUIRE(prob && boxes && img_off && img_wh && out_boxes && out_scores && out_index && out_count,
                "detect_postprocess: null pointer");
    ODW_REQUIRE(nms_thr > 0.0f, "detect_postprocess: nms threshold must be > 0 (boxlist_ops.py:25-26 returns the input otherwise)");
    ODW_REQUIRE((((uintptr_t)boxes) & 15) == 0 && (((uintptr_t)out_boxes) & 15) == 0, "detect_postprocess: 16-byte alignment");
    DetArgs a;
    a.prob = prob; a.reg = reg; a.boxes = boxes; a.img_off = img_off; a.img_wh = img_wh; a.C = C; a.ld_reg = ld_reg;
    a.cls_agnostic = cls_agnostic; a.per_class = 0; a.wx = wx; a.wy = wy; a.ww = ww; a.wh = wh; a.xform_clip = xform_clip;
    a.score_thresh = score_thresh; a.nms_thr = nms_thr; a.pstride = pstride; a.out_boxes = out_boxes;
    a.out_scores = out_scores; a.out_index = out_index; a.out_count = out_count;
    return launch_detect(a, n_img, max_p, stream_);
}

ODW_EXPORT int odw_detect_decode(const float* reg, int ld_reg, int cls_agnostic, const float* boxes, const int* img_off,
                                 const float* img_wh, int n_img, int sum_p, int C, float wx, float wy, float ww, float wh,
                                 float xform_clip, float* out_boxes, void* stream_) {
    ODW_REQUIRE(n_img >= 1 && C >= 2 && sum_p >= 0, "detect_decode: bad dims");
    if (sum_p == 0) return ODW_OK;
    ODW_REQUIRE(boxes && img_off && img_wh && out_boxes, "detect_decode: null pointer");
    ODW_REQUIRE((((uintptr_t)boxes) & 15) == 0 && (((uintptr_t)out_boxes) & 15) == 0, "detect_decode: 16-byte alignment");
    DetArgs a = {};
    a.reg = reg; a.boxes = boxes; a.img_off = img_off; a.img_wh = img_wh; a.C = C; a.ld_reg = ld_reg;
    a.cls_agnostic = cls_agnostic; a.wx = wx; a.wy = wy; a.ww = ww; a.wh = wh; a.xform_clip = xform_clip;
    const long long total = (long long)sum_p * C;
    detect_decode_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream_>>>(a, n_img, sum_p, out_boxes);
    ODW_CHECK_LAUNCH("detect_decode_kernel");
    return ODW_OK;
}

ODW_EXPORT int odw_detect_filter(const float* prob, int C, const float* boxes_pc, const int* img_off, int n_img, int max_p,
                                 float score_thresh, float nms_thr, int pstride, float* out_boxes, float* out_scores,
                                 int* out_index, int* out_count, void* stream_) {
    ODW_REQUIRE(n_img >= 0 && C >= 2 && max_p >= 1 && max_p <= 4096 && pstride >= max_p,
                "detect_filter: bad dims (at most 4096 boxes per image and class, got %d)", max_p);
    if (n_img == 0) return ODW_OK;
    ODW_REQUIRE(prob && boxes_pc && img_off && out_boxes && out_scores && out_index && out_count, "detect_filter: null pointer");
    ODW_REQUIRE(nms_thr > 0.0f, "detect_filter: nms threshold must be > 0");
    ODW_REQUIRE((((uintptr_t)boxes_pc) & 15) == 0 && (((uintptr_t)out_boxes) & 15) == 0, "detect_filter: 16-byte alignment");
    DetArgs a = {};
    a.prob = prob; a.boxes = boxes_pc; a.img_off = img_off; a.img_wh = nullptr; a.C = C; a.per_class = 1;
    a.score_thresh = score_thresh; a.nms_thr = nms_thr; a.pstride = pstride; a.out_boxes = out_boxes;
    a.out_scores = out_scores; a.out_index = out_index; a.out_count = out_count;
    return launch_detect(a, n_img, max_p, stream_);
}
