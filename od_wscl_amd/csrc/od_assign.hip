// od_assign.hip -- pseudo-label assignment of the object-discovery layer, fused:
// IoU(P,G) -> row max / FIRST argmax -> labels, loss weights, background test, box-regression
// targets.  Reference: roi_heads/weak_head/pseudo_label_generator.py:171-190 (the IoU matrix is
// copied to the host and reduced with numpy there, :176-177) + modeling/box_coder.py:22-50.
// Eight lanes per proposal; the G pseudo-GT boxes (a handful) sit in LDS.
#include "odw_common.h"

namespace {

constexpr int kMaxGT = 2048;
constexpr int kLanes = 8;             // lanes per proposal in od_assign_kernel (a power of two <= 64)

// INDEXED: the pseudo-GT are given as int32 indices into `boxes` (gt_boxes = the index list) with int32 classes --
// the form the discovery kernels emit -- instead of gathered boxes + int64 classes.
template <bool INDEXED>
__global__ __launch_bounds__(256) void od_assign_kernel(const float* __restrict__ boxes, int P,
                                                        const void* __restrict__ gt_boxes,
                                                        const void* __restrict__ gt_classes,
                                                        const float* __restrict__ gt_scores, int G,
                                                        float fg_thresh, float wx, float wy, float ww, float wh,
                                                        long long* __restrict__ labels,
                                                        float* __restrict__ weights,
                                                        float* __restrict__ targets,
                                                        const int* __restrict__ g_dev = nullptr) {
    extern __shared__ __attribute__((aligned(16))) float sh[];   // G x 5: box + area
    if (g_dev) {                    // the count lives on the device (written by the discovery kernels): G = its capacity
        const int g = *g_dev;
        if (g < 1) {                // an image without a pseudo-GT box (no positive label): every proposal is background
            const int i = blockIdx.x * blockDim.x + threadIdx.x;         // with weight 0, like od_layer's early return
            if (i < P) {                                                 // (pseudo_label_generator.py:167-170)
                labels[i] = 0;
                weights[i] = 0.0f;
                reinterpret_cast<float4*>(targets)[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            }
            return;
        }
        G = g < G ? g : G;
    }
    for (int j = threadIdx.x; j < G; j += blockDim.x) {
        const float4 q = INDEXED ? reinterpret_cast<const float4*>(boxes)[reinterpret_cast<const int*>(gt_boxes)[j]]
                                 : reinterpret_cast<const float4*>(gt_boxes)[j];
        sh[5 * j + 0] = q.x; sh[5 * j + 1] = q.y; sh[5 * j + 2] = q.z; sh[5 * j + 3] = q.w;
        sh[5 * j + 4] = (q.z - q.x + 1) * (q.w - q.y + 1);
    }
    __syncthreads();
    // kLanes lanes per proposal, lane l taking the pseudo-GT boxes l, l + kLanes, ...: with one thread per proposal the launch
    // was 8 workgroups walking ~10^2 IoUs (a division each) one after the other -- 22 us of VALU latency on 3 % of the chip,
    // three times per step.  The lanes' (IoU, index) pairs are merged "larger IoU, then smaller index": the FIRST maximum, as the
    // sequential scan (numpy argmax) finds it.
    const int gt_ = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = gt_ / kLanes, l = gt_ % kLanes;
    const bool live = i < P;
    const float4 p = reinterpret_cast<const float4*>(boxes)[live ? i : 0];
    const float ap = (p.z - p.x + 1) * (p.w - p.y + 1);
    float best = -1.0f;
    int bj = 0;
    for (int j = l; j < G; j += kLanes) {
        float w = fminf(p.z, sh[5 * j + 2]) - fmaxf(p.x, sh[5 * j + 0]) + 1;
        float h = fminf(p.w, sh[5 * j + 3]) - fmaxf(p.y, sh[5 * j + 1]) + 1;
        w = w < 0 ? 0 : w;
        h = h < 0 ? 0 : h;
        const float inter = w * h;
        const float iou = inter / (ap + sh[5 * j + 4] - inter);   // boxlist_ops.py:154-159
        if (iou > best) { best = iou; bj = j; }                    // first maximum of this lane's subsequence
    }
#pragma unroll
    for (int off = 1; off < kLanes; off <<= 1) {
        const float ob = __shfl_xor(best, off, 64);
        const int oj = __shfl_xor(bj, off, 64);
        // (a lane with no box holds (-1, 0): never larger than a real IoU >= 0, and G >= 1 gives lane 0 a box)
        if (ob > best || (ob == best && oj < bj)) { best = ob; bj = oj; }
    }
    if (!live || l != 0) return;
    const long long cls = INDEXED ? (long long)reinterpret_cast<const int*>(gt_classes)[bj]
                                  : reinterpret_cast<const long long*>(gt_classes)[bj];
    labels[i] = best <= fg_thresh ? 0 : cls;                       // bg test is <= (:183)
    weights[i] = gt_scores[bj];
    // BoxCoder.encode(gt[bj], proposal i)
    const float ew = p.z - p.x + 1, eh = p.w - p.y + 1;
    const float ex = p.x + 0.5f * ew, ey = p.y + 0.5f * eh;
    const float gw = sh[5 * bj + 2] - sh[5 * bj + 0] + 1, gh = sh[5 * bj + 3] - sh[5 * bj + 1] + 1;
    const float gx = sh[5 * bj + 0] + 0.5f * gw, gy = sh[5 * bj + 1] + 0.5f * gh;
    float4 t;
    t.x = wx * (gx - ex) / ew;
    t.y = wy * (gy - ey) / eh;
    t.z = ww * logf(gw / ew);
    t.w = wh * logf(gh / eh);
    reinterpret_cast<float4*>(targets)[i] = t;
}

}  // namespace

ODW_EXPORT int odw_od_assign(const float* boxes, int P, const float* gt_boxes, const int64_t* gt_classes,
                             const float* gt_scores, int G, float fg_thresh, float wx, float wy, float ww,
                             float wh, int64_t* labels, float* weights, float* targets, void* stream_) {
    ODW_REQUIRE(P >= 0 && G >= 1 && G <= kMaxGT, "od_assign: P=%d G=%d (1..%d pseudo-GT boxes)", P, G, kMaxGT);
    if (P == 0) return ODW_OK;
    ODW_REQUIRE(boxes && gt_boxes && gt_classes && gt_scores && labels && weights && targets,
                "od_assign: null pointer");
    ODW_REQUIRE((((uintptr_t)boxes) & 15) == 0 && (((uintptr_t)gt_boxes) & 15) == 0 &&
                (((uintptr_t)targets) & 15) == 0, "od_assign: boxes/targets must be 16-byte aligned");
    od_assign_kernel<false><<<(P * kLanes + 255) / 256, 256, (size_t)G * 5 * 4, (hipStream_t)stream_>>>(
        boxes, P, gt_boxes, gt_classes, gt_scores, G, fg_thresh, wx, wy, ww, wh,
        (long long*)labels, weights, targets);
    ODW_CHECK_LAUNCH("od_assign_kernel");
    return ODW_OK;
}

ODW_EXPORT int odw_od_assign_indexed(const float* boxes, int P, const int* gt_index, const int* gt_classes,
                                     const float* gt_scores, int G, float fg_thresh, float wx, float wy, float ww,
                                     float wh, int64_t* labels, float* weights, float* targets, void* stream_) {
    ODW_REQUIRE(P >= 0 && G >= 1 && G <= kMaxGT, "od_assign_indexed: P=%d G=%d (1..%d pseudo-GT boxes)", P, G, kMaxGT);
    if (P == 0) return ODW_OK;
    ODW_REQUIRE(boxes && gt_index && gt_classes && gt_scores && labels && weights && targets,
                "od_assign_indexed: null pointer");
    ODW_REQUIRE((((uintptr_t)boxes) & 15) == 0 && (((uintptr_t)targets) & 15) == 0,
                "od_assign_indexed: boxes/targets must be 16-byte aligned");
    od_assign_kernel<true><<<(P * kLanes + 255) / 256, 256, (size_t)G * 5 * 4, (hipStream_t)stream_>>>(
        boxes, P, gt_index, gt_classes, gt_scores, G, fg_thresh, wx, wy, ww, wh, (long long*)labels, weights, targets);
    ODW_CHECK_LAUNCH("od_assign_kernel");
    return ODW_OK;
}

// The same with the number of pseudo-GT boxes read from device memory (n_gt_dev, written by odw_discover_sim) so that
// the launch does not wait for a host read of the discovery result; g_cap = capacity of the three lists (<= 2048).
ODW_EXPORT int odw_od_assign_indexed_dev(const float* boxes, int P, const int* gt_index, const int* gt_classes,
                                         const float* gt_scores, const int* n_gt_dev, int g_cap, float fg_thresh,
                                         float wx, float wy, float ww, float wh, int64_t* labels, float* weights,
                                         float* targets, void* stream_) {
    ODW_REQUIRE(P >= 0 && g_cap >= 1, "od_assign_indexed_dev: P=%d capacity=%d", P, g_cap);
    if (P == 0) return ODW_OK;
    if (g_cap > kMaxGT) g_cap = kMaxGT;
    ODW_REQUIRE(boxes && gt_index && gt_classes && gt_scores && n_gt_dev && labels && weights && targets,
                "od_assign_indexed_dev: null pointer");
    ODW_REQUIRE((((uintptr_t)boxes) & 15) == 0 && (((uintptr_t)targets) & 15) == 0,
                "od_assign_indexed_dev: boxes/targets must be 16-byte aligned");
    od_assign_kernel<true><<<(P * kLanes + 255) / 256, 256, (size_t)g_cap * 5 * 4, (hipStream_t)stream_>>>(
        boxes, P, gt_index, gt_classes, gt_scores, g_cap, fg_thresh, wx, wy, ww, wh, (long long*)labels, weights, targets,
        n_gt_dev);
    ODW_CHECK_LAUNCH("od_assign_kernel");
    return ODW_OK;
}
