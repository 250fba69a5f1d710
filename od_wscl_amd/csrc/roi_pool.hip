// roi_pool.hip -- ROIPool forward/backward for gfx950.
//
// Behaviour: wetectron/csrc/cuda/ROIPool_cuda.cu:17-108 (max over the bin,
// first maximum in row-major scan order, int32 argmax = h*W+w or -1, empty bin
// -> 0).  Structure is not the reference's thread-per-output gather (which is
// L1-transaction bound on CDNA: every wave-load touches ~10 cache lines):
//
//   * "plane-resident" kernels: one workgroup owns CG channel planes of one
//     image, pulls them into LDS ONCE with coalesced 16-byte loads (a 76x76
//     fp32 plane is 23 KB; 160 KB of LDS holds the plane of a 1600 px image),
//     then streams over every ROI of that image computing the CG*PH*PW
//     outputs from LDS.  HBM traffic = feature map once + outputs once, the
//     window gather never leaves the CU.
//   * backward = the same residency with LDS float atomics into the plane,
//     written back once: no global atomics, no zero-fill pass.
//   * bin boundaries (the roundf/floorf/ceilf chain of ROIPool_cuda.cu:30-56)
//     are computed once per ROI by a prologue kernel into an int table.
//   * planes that do not fit in LDS fall back to direct global kernels.
#include "odw_common.h"
#include <float.h>

namespace {

constexpr int kPlaneThreads = 1024;

// tab row: [batch, hs[PH], he[PH], ws[PW], we[PW]] -- already offset + clipped
__global__ void roi_bins_kernel(const float* __restrict__ rois, float scale, int R, int PH, int PW,
                                int H, int W, int* __restrict__ tab) {
    int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= R) return;
    const float* roi = rois + (size_t)n * 5;
    int* t = tab + (size_t)n * (1 + 2 * PH + 2 * PW);
    int sw = (int)roundf(roi[1] * scale);
    int sh = (int)roundf(roi[2] * scale);
    int ew = (int)roundf(roi[3] * scale);
    int eh = (int)roundf(roi[4] * scale);
    int rw = max(ew - sw + 1, 1);
    int rh = max(eh - sh + 1, 1);
    float bin_h = (float)rh / (float)PH;
    float bin_w = (float)rw / (float)PW;
    t[0] = (int)roi[0];
    for (int ph = 0; ph < PH; ++ph) {
        int hs = (int)floorf((float)ph * bin_h);
        int he = (int)ceilf((float)(ph + 1) * bin_h);
        t[1 + ph] = min(max(hs + sh, 0), H);
        t[1 + PH + ph] = min(max(he + sh, 0), H);
    }
    for (int pw = 0; pw < PW; ++pw) {
        int ws = (int)floorf((float)pw * bin_w);
        int we = (int)ceilf((float)(pw + 1) * bin_w);
        t[1 + 2 * PH + pw] = min(max(ws + sw, 0), W);
        t[1 + 2 * PH + PW + pw] = min(max(we + sw, 0), W);
    }
}

// cooperative copy of `count` floats global -> LDS (16-byte path when aligned)
__device__ __forceinline__ void copy_to_lds(float* dst, const float* __restrict__ src, int count) {
    if ((((uintptr_t)src) & 15) == 0 && (count & 3) == 0) {
        const float4* s4 = reinterpret_cast<const float4*>(src);
        float4* d4 = reinterpret_cast<float4*>(dst);
        for (int i = threadIdx.x; i < count / 4; i += blockDim.x) d4[i] = s4[i];
    } else {
        for (int i = threadIdx.x; i < count; i += blockDim.x) dst[i] = src[i];
    }
}

template <int CG>
__global__ __launch_bounds__(kPlaneThreads) void roi_pool_fwd_plane(
    const float* __restrict__ feat, const int* __restrict__ tab, int C, int H, int W, int R, int PH,
    int PW, float* __restrict__ out, int* __restrict__ argmax) {
    extern __shared__ __attribute__((aligned(16))) float plane[];
    const int groups = (C + CG - 1) / CG;
    const int b = blockIdx.x / groups;
    const int c0 = (blockIdx.x % groups) * CG;
    const int nc = min(CG, C - c0);
    const int HW = H * W;
    copy_to_lds(plane, feat + ((size_t)b * C + c0) * HW, nc * HW);
    __syncthreads();

    const int nb = PH * PW;
    const int per_roi = nc * nb;
    const int ts = 1 + 2 * PH + 2 * PW;
    // flat work index i = n*per_roi + r, advanced incrementally (no division in the loop)
    int n = threadIdx.x / per_roi;
    int r = threadIdx.x % per_roi;
    const int dn = kPlaneThreads / per_roi, dr = kPlaneThreads % per_roi;
    for (; n < R; n += dn, r += dr) {
        if (r >= per_roi) { r -= per_roi; ++n; if (n >= R) break; }
        const int* t = tab + (size_t)n * ts;
        if (t[0] != b) continue;
        const int cl = r / nb, bin = r - cl * nb;
        const int ph = bin / PW, pw = bin - ph * PW;
        const int hs = t[1 + ph], he = t[1 + PH + ph];
        const int ws = t[1 + 2 * PH + pw], we = t[1 + 2 * PH + PW + pw];
        const bool empty = (he <= hs) || (we <= ws);
        float best = empty ? 0.0f : -FLT_MAX;
        int besti = -1;
        const float* p = plane + cl * HW;
        for (int h = hs; h < he; ++h) {
            const float* row = p + h * W;
            for (int w = ws; w < we; ++w) {
                float v = row[w];
                if (v > best) { best = v; besti = h * W + w; }
            }
        }
        const size_t o = ((size_t)n * C + c0 + cl) * nb + bin;
        out[o] = best;
        argmax[o] = besti;
    }
}

// fallback: plane too large for LDS.  One thread per output, direct global reads.
__global__ void roi_pool_fwd_direct(const float* __restrict__ feat, const int* __restrict__ tab, int C,
                                    int H, int W, int R, int PH, int PW, float* __restrict__ out,
                                    int* __restrict__ argmax) {
    const int nb = PH * PW;
    const int ts = 1 + 2 * PH + 2 * PW;
    const size_t total = (size_t)R * C * nb;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        int bin = (int)(i % nb);
        int c = (int)((i / nb) % C);
        int n = (int)(i / nb / C);
        const int* t = tab + (size_t)n * ts;
        int ph = bin / PW, pw = bin - ph * PW;
        int hs = t[1 + ph], he = t[1 + PH + ph];
        int ws = t[1 + 2 * PH + pw], we = t[1 + 2 * PH + PW + pw];
        bool empty = (he <= hs) || (we <= ws);
        float best = empty ? 0.0f : -FLT_MAX;
        int besti = -1;
        const float* p = feat + ((size_t)t[0] * C + c) * H * W;
        for (int h = hs; h < he; ++h)
            for (int w = ws; w < we; ++w) {
                float v = p[h * W + w];
                if (v > best) { best = v; besti = h * W + w; }
            }
        out[i] = best;
        argmax[i] = besti;
    }
}

template <int CG>
__global__ __launch_bounds__(kPlaneThreads) void roi_pool_bwd_plane(
    const float* __restrict__ grad_out, const int* __restrict__ argmax, const float* __restrict__ rois,
    int C, int H, int W, int R, int nb, float* __restrict__ grad_in) {
    extern __shared__ __attribute__((aligned(16))) float acc[];
    const int groups = (C + CG - 1) / CG;
    const int b = blockIdx.x / groups;
    const int c0 = (blockIdx.x % groups) * CG;
    const int nc = min(CG, C - c0);
    const int HW = H * W;
    for (int i = threadIdx.x; i < nc * HW; i += blockDim.x) acc[i] = 0.0f;
    __syncthreads();

    const int per_roi = nc * nb;
    int n = threadIdx.x / per_roi;
    int r = threadIdx.x % per_roi;
    const int dn = kPlaneThreads / per_roi, dr = kPlaneThreads % per_roi;
    for (; n < R; n += dn, r += dr) {
        if (r >= per_roi) { r -= per_roi; ++n; if (n >= R) break; }
        if ((int)rois[(size_t)n * 5] != b) continue;
        const int cl = r / nb, bin = r - cl * nb;
        const size_t o = ((size_t)n * C + c0 + cl) * nb + bin;
        const int a = argmax[o];
        if (a >= 0) atomicAdd(&acc[cl * HW + a], grad_out[o]);
    }
    __syncthreads();
    float* dst = grad_in + ((size_t)b * C + c0) * HW;
    for (int i = threadIdx.x; i < nc * HW; i += blockDim.x) dst[i] = acc[i];
}

__global__ void roi_pool_bwd_direct(const float* __restrict__ grad_out, const int* __restrict__ argmax,
                                    const float* __restrict__ rois, int C, int H, int W, int R, int nb,
                                    float* __restrict__ grad_in) {
    const size_t total = (size_t)R * C * nb;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        int c = (int)((i / nb) % C);
        int n = (int)(i / nb / C);
        int a = argmax[i];
        if (a >= 0) {
            int b = (int)rois[(size_t)n * 5];
            atomicAdd(grad_in + ((size_t)b * C + c) * H * W + a, grad_out[i]);
        }
    }
}

// channels per workgroup: as many as fit while still giving every CU a workgroup
int pick_cg(int B, int C, int HW) {
    const int cands[3] = {4, 2, 1};
    int fit = 0;
    for (int k = 0; k < 3; ++k) {
        int cg = cands[k];
        if ((int64_t)cg * HW * 4 > ODW_LDS_BYTES) continue;
        if (!fit) fit = cg;
        if ((int64_t)B * ((C + cg - 1) / cg) >= ODW_NUM_CU) return cg;
    }
    return fit ? 1 : 0;  // small problems: most parallelism; 0 = does not fit
}

template <typename K>
hipError_t allow_lds(K kernel, size_t bytes) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

}  // namespace

ODW_EXPORT int64_t odw_roi_pool_workspace(int R, int PH, int PW) {
    return odw_align_up((int64_t)(R > 0 ? R : 1) * (1 + 2 * PH + 2 * PW) * 4, 256);
}

ODW_EXPORT int odw_roi_pool_forward(const float* feat, const float* rois, float spatial_scale, int B,
                                    int C, int H, int W, int R, int PH, int PW, float* out,
                                    int32_t* argmax, void* workspace, int64_t workspace_bytes,
                                    void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    ODW_REQUIRE(B >= 0 && C > 0 && H > 0 && W > 0 && PH > 0 && PW > 0 && R >= 0,
                "roi_pool_forward: bad dims B=%d C=%d H=%d W=%d R=%d PH=%d PW=%d", B, C, H, W, R, PH, PW);
    if (R == 0 || B == 0) return ODW_OK;  // ROIPool_cuda.cu:132-135
    ODW_REQUIRE(feat && rois && out && argmax, "roi_pool_forward: null pointer");
    ODW_REQUIRE(PH * PW <= kPlaneThreads / 4, "roi_pool_forward: pooled size %dx%d too large", PH, PW);
    if (workspace_bytes < odw_roi_pool_workspace(R, PH, PW) || !workspace) {
        odw_set_error("roi_pool_forward: workspace %lld < %lld bytes", (long long)workspace_bytes,
                      (long long)odw_roi_pool_workspace(R, PH, PW));
        return ODW_EWORKSPACE;
    }
    int* tab = (int*)workspace;
    roi_bins_kernel<<<(R + 255) / 256, 256, 0, stream>>>(rois, spatial_scale, R, PH, PW, H, W, tab);
    ODW_CHECK_LAUNCH("roi_bins_kernel");

    const int HW = H * W;
    const int cg = pick_cg(B, C, HW);
    if (cg == 0) {
        size_t total = (size_t)R * C * PH * PW;
        int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
        roi_pool_fwd_direct<<<grid, 256, 0, stream>>>(feat, tab, C, H, W, R, PH, PW, out, argmax);
        ODW_CHECK_LAUNCH("roi_pool_fwd_direct");
        return ODW_OK;
    }
    const int grid = B * ((C + cg - 1) / cg);
    const size_t lds = (size_t)cg * HW * 4;
    switch (cg) {
        case 4:
            ODW_CHECK_HIP(allow_lds(roi_pool_fwd_plane<4>, lds), "roi_pool_fwd_plane<4> attr");
            roi_pool_fwd_plane<4><<<grid, kPlaneThreads, lds, stream>>>(feat, tab, C, H, W, R, PH, PW, out, argmax);
            break;
        case 2:
            ODW_CHECK_HIP(allow_lds(roi_pool_fwd_plane<2>, lds), "roi_pool_fwd_plane<2> attr");
            roi_pool_fwd_plane<2><<<grid, kPlaneThreads, lds, stream>>>(feat, tab, C, H, W, R, PH, PW, out, argmax);
            break;
        default:
            ODW_CHECK_HIP(allow_lds(roi_pool_fwd_plane<1>, lds), "roi_pool_fwd_plane<1> attr");
            roi_pool_fwd_plane<1><<<grid, kPlaneThreads, lds, stream>>>(feat, tab, C, H, W, R, PH, PW, out, argmax);
            break;
    }
    ODW_CHECK_LAUNCH("roi_pool_fwd_plane");
    return ODW_OK;
}

ODW_EXPORT int odw_roi_pool_backward(const float* grad_out, const int32_t* argmax, const float* rois,
                                     int B, int C, int H, int W, int R, int PH, int PW, float* grad_in,
                                     void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    ODW_REQUIRE(B >= 0 && C > 0 && H > 0 && W > 0 && PH > 0 && PW > 0 && R >= 0,
                "roi_pool_backward: bad dims");
    if (B == 0) return ODW_OK;
    ODW_REQUIRE(grad_in, "roi_pool_backward: null grad_in");
    const size_t in_bytes = (size_t)B * C * H * W * 4;
    if (R == 0) {  // ROIPool_cuda.cu:172,180-183: zeros
        ODW_CHECK_HIP(hipMemsetAsync(grad_in, 0, in_bytes, stream), "roi_pool_backward memset");
        return ODW_OK;
    }
    ODW_REQUIRE(grad_out && argmax && rois, "roi_pool_backward: null pointer");
    ODW_REQUIRE(PH * PW <= kPlaneThreads / 4, "roi_pool_backward: pooled size too large");
    const int HW = H * W, nb = PH * PW;
    const int cg = pick_cg(B, C, HW);
    if (cg == 0) {
        ODW_CHECK_HIP(hipMemsetAsync(grad_in, 0, in_bytes, stream), "roi_pool_backward memset");
        size_t total = (size_t)R * C * nb;
        int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
        roi_pool_bwd_direct<<<grid, 256, 0, stream>>>(grad_out, argmax, rois, C, H, W, R, nb, grad_in);
        ODW_CHECK_LAUNCH("roi_pool_bwd_direct");
        return ODW_OK;
    }
    const int grid = B * ((C + cg - 1) / cg);
    const size_t lds = (size_t)cg * HW * 4;
    switch (cg) {
        case 4:
            ODW_CHECK_HIP(allow_lds(roi_pool_bwd_plane<4>, lds), "roi_pool_bwd_plane<4> attr");
            roi_pool_bwd_plane<4><<<grid, kPlaneThreads, lds, stream>>>(grad_out, argmax, rois, C, H, W, R, nb, grad_in);
            break;
        case 2:
            ODW_CHECK_HIP(allow_lds(roi_pool_bwd_plane<2>, lds), "roi_pool_bwd_plane<2> attr");
            roi_pool_bwd_plane<2><<<grid, kPlaneThreads, lds, stream>>>(grad_out, argmax, rois, C, H, W, R, nb, grad_in);
            break;
        default:
            ODW_CHECK_HIP(allow_lds(roi_pool_bwd_plane<1>, lds), "roi_pool_bwd_plane<1> attr");
            roi_pool_bwd_plane<1><<<grid, kPlaneThreads, lds, stream>>>(grad_out, argmax, rois, C, H, W, R, nb, grad_in);
            break;
    }
    ODW_CHECK_LAUNCH("roi_pool_bwd_plane");
    return ODW_OK;
}
