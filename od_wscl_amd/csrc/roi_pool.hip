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
#include "odw_fixed.h"
#include "odw_planes.h"
#include <float.h>
#include <stdlib.h>

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

// ---- deterministic backward --------------------------------------------------------------------------------
// ROIPool's backward is a pure scatter-add of grad_out through argmax (csrc/cuda/ROIPool_cuda.cu:80-108 does it with
// float atomicAdd: the summation order, hence the rounding, changes from run to run).  Integer addition is
// associative: every gradient is converted to fixed point with ONE power-of-two scale for the launch (2^40 / the
// power of two above max|grad_out|, found by an order-independent atomicMax pre-pass), accumulated with 64-bit LDS
// integer atomics (at most R*PH*PW < 2^18 terms of magnitude <= 2^40 per cell: no overflow), and converted back.
// The result is bit-identical from run to run and is the correctly rounded sum to ~2^-40 of the largest gradient.
__global__ __launch_bounds__(256) void absmax_kernel(const float4* __restrict__ x, size_t n4, const float* __restrict__ tail,
                                                     int ntail, unsigned* __restrict__ out) {
    unsigned m = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = x[i];
        m = max(m, __float_as_uint(fabsf(v.x)));        // non-negative floats order like their bit patterns (NaN on top)
        m = max(m, __float_as_uint(fabsf(v.y)));
        m = max(m, __float_as_uint(fabsf(v.z)));
        m = max(m, __float_as_uint(fabsf(v.w)));
    }
    if (blockIdx.x == 0 && (int)threadIdx.x < ntail) m = max(m, __float_as_uint(fabsf(tail[threadIdx.x])));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, off, 64));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}

__global__ __launch_bounds__(kPlaneThreads) void roi_pool_bwd_plane_det(
    const float* __restrict__ grad_out, const int* __restrict__ argmax, const float* __restrict__ rois,
    const unsigned* __restrict__ absmax_bits, int C, int H, int W, int R, int nb, int cells_per_pass,
    float* __restrict__ grad_in) {
    extern __shared__ __attribute__((aligned(16))) long long iacc[];
    const int b = blockIdx.x / C;
    const int c = blockIdx.x % C;
    const int HW = H * W;
    float* dst = grad_in + ((size_t)b * C + c) * HW;
    const odwfx::Scale sc = odwfx::scale_of(*absmax_bits);
    if (sc.state != 1) {                     // all-zero gradient, or inf / nan somewhere (no finite scale exists: the
        const float fill = sc.state == 0 ? 0.0f : __uint_as_float(0x7fc00000u);      // plane becomes NaN, like a sum would)
        for (int i = threadIdx.x; i < HW; i += blockDim.x) dst[i] = fill;
        return;
    }
    const float scale = sc.to_fixed, inv = sc.to_float;
    for (int lo = 0; lo < HW; lo += cells_per_pass) {
        const int hi = min(HW, lo + cells_per_pass);
        for (int i = threadIdx.x; i < hi - lo; i += blockDim.x) iacc[i] = 0;
        __syncthreads();
        int n = threadIdx.x / nb, r = threadIdx.x % nb;
        const int dn = kPlaneThreads / nb, dr = kPlaneThreads % nb;
        for (; n < R; n += dn, r += dr) {
            if (r >= nb) { r -= nb; ++n; if (n >= R) break; }
            if ((int)rois[(size_t)n * 5] != b) continue;
            const size_t o = ((size_t)n * C + c) * nb + r;
            const int a = argmax[o];
            if (a >= lo && a < hi)
                atomicAdd(reinterpret_cast<unsigned long long*>(&iacc[a - lo]),
                          (unsigned long long)__float2ll_rn(grad_out[o] * scale));
        }
        __syncthreads();
        for (int i = threadIdx.x; i < hi - lo; i += blockDim.x) dst[lo + i] = (float)iacc[i] * inv;
        __syncthreads();
    }
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


// ---- ROIPool fused with the operand staging of the first head GEMM -------------------------------------------
// Same plane-resident pooling, but the results leave the CU in the form the consumer wants: the bf16 (2R x C*nb)
// stacked operand of fc6 (row n = pooled features of ROI n, row R+n = their DropBlock view ((x*keep)*numel)/sum,
// head_aux.hip) and a 16-bit argmax -- 300 MB of writes instead of 400 MB fp32/int32 here plus 400 MB for the
// separate stacking pass.  blockIdx.y splits the ROI list so that 128 channel groups x 4 chunks fill the chip.
__device__ __forceinline__ unsigned short rp_f2bf(float f) {
    unsigned int u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float rp_bf2f(unsigned int h) { return __uint_as_float(h << 16); }

constexpr int kStackThreads = 512;

// The feature map of this path holds bf16 values (it comes from the bf16 backbone), so value and position fit ONE
// 32-bit key: (order-preserving image of the 16 value bits) << 16 | (0xFFFF - cell).  The window scan is then one
// LDS read + one v_max_u32 per cell -- max value AND first row-major position of it -- instead of read, compare,
// two selects and a branch.  0 = "no cell" (every real key is >= 0x00800000).
__device__ __forceinline__ unsigned rp_key(float v, int cell) {
    unsigned b = __float_as_uint(v) >> 16;
    b = b == 0x8000u ? 0u : b;          // -0.0 == +0.0 in the reference's scan (first cell wins)
    const unsigned ord = (b & 0x8000u) ? (~b & 0xFFFFu) : (b | 0x8000u);
    return (ord << 16) | (0xFFFFu - (unsigned)cell);
}

// One workgroup = one channel plane (+ an ROI chunk).  Next to the keys T0 the plane's LDS holds three more
// levels of a row-wise range-max table (T1/T2/T3[i] = max of the 2/4/8 keys starting at i): the maximum of a bin row
// [ws, we) is max(Tk[ws], Tk[we - 2^k]) -- two LDS reads per bin row instead of one per cell, and no divergent
// inner loop.  (Measured before: a straight per-cell scan, fp32 compare or key max, 2-byte or packed 8-byte stores:
// 600-710 us for P = 2000 on 76x76x512 -- ~1 G cell visits at 15 % of the LDS read rate; the scan, not the
// output traffic, bounds ROI pooling.  A thread per (ROI, bin column) walking its 7 bins with the column set-up
// hoisted: 437 us against 370 -- the rows of a strip diverge more across a wave than single bins do.
// Later experiments (ODW_RPS_ROWSTRIP): with the stores suppressed the kernel takes 187 us, table build alone 14 us --
// the 98-byte output segments of one plane are half of the time; XCD-contiguous planes recover 10 % of it.)
template <int PW_T>
__global__ __launch_bounds__(kPlaneThreads) void roi_pool_stack_fwd_plane(
    const float* __restrict__ feat, const int* __restrict__ tab, int C, int H, int W, int R, int PH, int PW_rt,
    const float* __restrict__ keep, const float* __restrict__ keep_sum, unsigned short* __restrict__ X, int ld,
    unsigned short* __restrict__ argmax, int rowstrip) {
    extern __shared__ __attribute__((aligned(16))) unsigned keys[];
    const int PW = PW_T > 0 ? PW_T : PW_rt;
    // workgroup i runs on XCD i % 8: give each XCD a CONTIGUOUS range of planes, so that the two 98-byte segments
    // neighbouring planes write into one 128-byte line of X meet in the same L2 and leave it as one full line
    // (round-robin planes: every line is written back partially by two L2s; stores were half of this kernel's time)
    int plane = blockIdx.x;
    if ((gridDim.x & 7) == 0 && !(rowstrip & 4)) plane = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    rowstrip &= 3;
    const int b = plane / C;
    const int c = plane % C;
    const int HW = H * W;
    unsigned* T0 = keys;
    unsigned* T1 = keys + HW;
    unsigned* T2 = keys + 2 * HW;
    unsigned* T3 = keys + 3 * HW;
    {
        const float* src = feat + ((size_t)b * C + c) * HW;
        for (int i = threadIdx.x; i < HW; i += blockDim.x) T0[i] = rp_key(src[i], i);
        __syncthreads();
        for (int i = threadIdx.x; i < HW; i += blockDim.x) T1[i] = max(T0[i], T0[min(i + 1, HW - 1)]);
        __syncthreads();
        for (int i = threadIdx.x; i < HW; i += blockDim.x) T2[i] = max(T1[i], T1[min(i + 2, HW - 1)]);
        __syncthreads();
        for (int i = threadIdx.x; i < HW; i += blockDim.x) T3[i] = max(T2[i], T2[min(i + 4, HW - 1)]);
        __syncthreads();
    }
    const int nb = PH * PW;
    const int ts = 1 + 2 * PH + 2 * PW;
    const int n_lo = (int)((long long)R * blockIdx.y / gridDim.y), n_hi = (int)((long long)R * (blockIdx.y + 1) / gridDim.y);
    const float numel = (float)((double)R * nb);
    const float sum = keep ? *keep_sum : 1.0f;
    if (PW_T == 7 && rowstrip) {
        // One thread = one bin ROW (ROI n, ph) = 7 bins that share the row range [hs, he): one pass over the rows with
        // 14 LDS reads each, the loop and address set-up paid once per 7 bins; trip counts differ across a wave exactly
        // as in the per-bin form (a column strip, 7 different row ranges in sequence, measured slower: 437 us).
        const int items = rowstrip == 3 ? 0 : (n_hi - n_lo) * PH;
        for (int it = threadIdx.x; it < items; it += blockDim.x) {
            const int n = n_lo + it / PH, ph = it - (it / PH) * PH;
            const int* t = tab + (size_t)n * ts;
            if (t[0] != b) continue;
            const int hs = t[1 + ph], he = t[1 + PH + ph];
            const unsigned* tb[7];
            int o0[7], o1[7];
            bool wide = false;
#pragma unroll
            for (int pw = 0; pw < 7; ++pw) {
                const int ws = t[1 + 2 * PH + pw], we = t[1 + 2 * PH + 7 + pw];
                const int bw = we - ws;
                const int lvl = bw >= 8 ? 3 : (bw >= 4 ? 2 : (bw >= 2 ? 1 : 0));
                tb[pw] = bw > 0 ? keys + lvl * HW : nullptr;
                o0[pw] = ws;
                o1[pw] = we - (1 << lvl);
                wide |= bw > 16;
            }
            unsigned best[7] = {0, 0, 0, 0, 0, 0, 0};
            if (!wide) {
                for (int h = hs; h < he; ++h) {
                    const int r = h * W;
#pragma unroll
                    for (int pw = 0; pw < 7; ++pw)
                        if (tb[pw]) {
                            unsigned m = max(tb[pw][r + o0[pw]], tb[pw][r + o1[pw]]);
                            // 8 < width <= 16: the two 8-wide windows at the ends cover the bin
                            best[pw] = max(best[pw], m);
                        }
                }
            } else {
                for (int h = hs; h < he; ++h) {
                    const unsigned* row = T3 + h * W;
#pragma unroll
                    for (int pw = 0; pw < 7; ++pw)
                        if (tb[pw]) {
                            unsigned m = max(tb[pw][h * W + o0[pw]], tb[pw][h * W + o1[pw]]);
                            for (int w = o0[pw] + 8; w < o1[pw]; w += 8) m = max(m, row[w]);
                            best[pw] = max(best[pw], m);
                        }
                }
            }
            const size_t col0 = (size_t)c * 49 + ph * 7;
            unsigned short* xr = X + (size_t)n * ld + col0;
            unsigned short* ar = argmax + (size_t)n * C * 49 + col0;
            const float* kr = keep ? keep + (size_t)n * 49 + ph * 7 : nullptr;
#pragma unroll
            for (int pw = 0; pw < 7; ++pw) {
                unsigned short vbits = 0, a = 0xFFFF;
                if (best[pw]) {
                    const unsigned ord = best[pw] >> 16;
                    vbits = (unsigned short)((ord & 0x8000u) ? (ord & 0x7FFFu) : (~ord & 0xFFFFu));
                    a = (unsigned short)(0xFFFFu - (best[pw] & 0xFFFFu));
                }
                if (rowstrip == 2 && best[pw] != 0xdeadbeefu) continue;
                xr[pw] = vbits;
                if (keep) xr[(size_t)R * ld + pw] = rp_f2bf(((rp_bf2f(vbits) * kr[pw]) * numel) / sum);
                ar[pw] = a;
            }
        }
        return;
    }
    int n = n_lo + threadIdx.x / nb;
    int bin = threadIdx.x % nb;
    const int dn = blockDim.x / nb, dr = blockDim.x % nb;
    for (; n < n_hi; n += dn, bin += dr) {
        if (bin >= nb) { bin -= nb; ++n; if (n >= n_hi) break; }
        const int* t = tab + (size_t)n * ts;
        if (t[0] != b) continue;
        const int ph = bin / PW, pw = bin - ph * PW;
        const int hs = t[1 + ph], he = t[1 + PH + ph];
        const int ws = t[1 + 2 * PH + pw], we = t[1 + 2 * PH + PW + pw];
        const int bw = we - ws;
        unsigned best = 0;
        if (bw > 0) {
            if (bw >= 8) {
                for (int h = hs; h < he; ++h) {
                    const unsigned* row = T3 + h * W;
                    for (int w = ws; w + 8 <= we; w += 8) best = max(best, row[w]);
                    best = max(best, row[we - 8]);
                }
            } else {
                const unsigned* tbl = bw >= 4 ? T2 : (bw >= 2 ? T1 : T0);
                const int span = bw >= 4 ? 4 : (bw >= 2 ? 2 : 1);
                for (int h = hs; h < he; ++h) best = max(best, max(tbl[h * W + ws], tbl[h * W + we - span]));
            }
        }
        unsigned short vbits = 0, a = 0xFFFF;            // empty bin -> 0, no argmax (ROIPool_cuda.cu:47-56)
        if (best) {
            const unsigned ord = best >> 16;
            vbits = (unsigned short)((ord & 0x8000u) ? (ord & 0x7FFFu) : (~ord & 0xFFFFu));
            a = (unsigned short)(0xFFFFu - (best & 0xFFFFu));
        }
        const size_t col = (size_t)c * nb + bin;
        X[(size_t)n * ld + col] = vbits;
        if (keep) X[(size_t)(R + n) * ld + col] = rp_f2bf(((rp_bf2f(vbits) * keep[(size_t)n * nb + bin]) * numel) / sum);
        argmax[(size_t)n * C * nb + col] = a;
    }
}

// gradient of the stacked operand (both halves) + the parked gradients of the sampled-row views (E rows of fp32,
// extra_roi[e] = ROI they belong to) scattered through the argmax into the feature planes, NCHW fp32 out
template <int CG, bool DX_F32>
__global__ __launch_bounds__(kPlaneThreads) void roi_pool_stack_bwd_plane(
    const void* __restrict__ dXv, int ld, const unsigned short* __restrict__ argmax, const float* __restrict__ rois,
    const float* __restrict__ keep, const float* __restrict__ keep_sum, const float* __restrict__ extra,
    const int* __restrict__ extra_roi, int E, int skip_clean, int C, int H, int W, int R, int nb,
    float* __restrict__ grad_in) {
    extern __shared__ __attribute__((aligned(16))) float acc[];
    const int groups = (C + CG - 1) / CG;
    const int b = blockIdx.x / groups;
    const int c0 = (blockIdx.x % groups) * CG;
    const int nc = min(CG, C - c0);
    const int HW = H * W;
    for (int i = threadIdx.x; i < nc * HW; i += blockDim.x) acc[i] = 0.0f;
    __syncthreads();
    const float numel = (float)((double)R * nb);
    const float sum = keep ? *keep_sum : 1.0f;
    const int per_roi = nc * nb;
    int n = threadIdx.x / per_roi;
    int r = threadIdx.x % per_roi;
    const int dn = kPlaneThreads / per_roi, dr = kPlaneThreads % per_roi;
    for (; n < R + E; n += dn, r += dr) {
        if (r >= per_roi) { r -= per_roi; ++n; if (n >= R + E) break; }
        const int roi = n < R ? n : extra_roi[n - R];
        if ((int)rois[(size_t)roi * 5] != b) continue;
        const int cl = r / nb, bin = r - cl * nb;
        const size_t col = (size_t)(c0 + cl) * nb + bin;
        float kp = 1.0f;
        if (n < R && keep) {
            kp = keep[(size_t)n * nb + bin];
            if (skip_clean && kp == 0.0f) continue;     // a dropped cell of the only differentiated half: g == 0 exactly
        }
        const unsigned short a = argmax[(size_t)roi * C * nb + col];
        if (a == 0xFFFF) continue;
        float g;
        if (n >= R) {
            g = extra[(size_t)(n - R) * C * nb + col];
        } else if (DX_F32) {
            const float* dX = reinterpret_cast<const float*>(dXv);
            g = skip_clean ? 0.0f : dX[(size_t)n * ld + col];
            if (keep) g += ((dX[(size_t)(R + n) * ld + col] * kp) * numel) / sum;
        } else {
            const unsigned short* dX = reinterpret_cast<const unsigned short*>(dXv);
            g = skip_clean ? 0.0f : rp_bf2f(dX[(size_t)n * ld + col]);
            if (keep) g += ((rp_bf2f(dX[(size_t)(R + n) * ld + col]) * kp) * numel) / sum;
        }
        atomicAdd(&acc[cl * HW + a], g);
    }
    __syncthreads();
    float* dst = grad_in + ((size_t)b * C + c0) * HW;
    for (int i = threadIdx.x; i < nc * HW; i += blockDim.x) dst[i] = acc[i];
}

// The same backward with fixed-point accumulation (odw_fixed.h): 64-bit integer LDS atomics run 10x the rate of
// ds_add_f32 on gfx950 (2.0 vs 0.2 T updates/s) and make the result independent of the order of arrival.
template <bool DX_F32>
__global__ __launch_bounds__(kPlaneThreads) void roi_pool_stack_bwd_plane_fx(
    const void* __restrict__ dXv, int ld, const unsigned short* __restrict__ argmax, const float* __restrict__ rois,
    const float* __restrict__ keep, const float* __restrict__ keep_sum, const float* __restrict__ extra,
    const int* __restrict__ extra_roi, int E, int skip_clean, const unsigned* __restrict__ absmax_bits, int C, int H,
    int W, int R, int nb, float* __restrict__ grad_in, const int* __restrict__ e_dev) {
    extern __shared__ __attribute__((aligned(16))) long long iacc[];
    if (e_dev) { const int ed = *e_dev; E = ed < E ? (ed > 0 ? ed : 0) : E; }      // side-buffer entries that exist (device-resident count)
    // workgroup i runs on XCD i % 8: hand each XCD a CONTIGUOUS range of planes -- neighbouring planes read the two
    // halves of the same 128-byte lines of dX / argmax (49 x 2 bytes per ROI and plane), which then meet in one L2
    int plane = blockIdx.x;
    if ((gridDim.x & 7) == 0) plane = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int b = plane / C;
    const int c = plane % C;
    const int HW = H * W;
    float* dst = grad_in + ((size_t)b * C + c) * HW;
    const float numel = (float)((double)R * nb);
    const float sum = keep ? *keep_sum : 1.0f;
    odwfx::Scale sc = odwfx::scale_of(*absmax_bits);
    if (sc.state != 1) {                     // all-zero gradient, or inf / nan somewhere: no finite scale exists
        const float fill = sc.state == 0 ? 0.0f : __uint_as_float(0x7fc00000u);
        for (int i = threadIdx.x; i < HW; i += blockDim.x) dst[i] = fill;
        return;
    }
    // a term is dX_clean + ((dX_aug * keep) * numel) / sum: bounded by amax * (1 + numel / sum)
    {
        int k;
        frexpf(1.0f + (keep ? numel / sum : 0.0f), &k);
        sc.to_fixed = ldexpf(sc.to_fixed, -k);
        sc.to_float = ldexpf(sc.to_float, k);
    }
    for (int i = threadIdx.x; i < HW; i += blockDim.x) iacc[i] = 0;
    __syncthreads();
    // Four (ROI, bin) items per thread and round (eight measured the same: 205 us, from 250), their loads issued as one batch before any of the adds: one item per
    // iteration is a chain of four dependent 2-/4-byte loads (ROI image, keep, argmax, gradient) per LDS atomic, i.e.
    // pure L2 latency.  Items that do not contribute (other image, dropped cell, empty bin) add nothing; integer adds
    // commute, so the result is the same bits as the one-item loop's.
    const int total = (R + E) * nb;
    constexpr int kU = 4;
    const bool one_image = (int)gridDim.x == C;      // a single image: every ROI belongs to it
    for (int base = threadIdx.x; base < total; base += kU * kPlaneThreads) {
        int nn[kU], roi[kU];
        size_t col[kU];
        bool ok[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int i = base + u * kPlaneThreads;
            ok[u] = i < total;
            const int ii = ok[u] ? i : 0;
            nn[u] = ii / nb;
            col[u] = (size_t)c * nb + (ii - nn[u] * nb);
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) roi[u] = nn[u] < R ? nn[u] : extra_roi[nn[u] - R];
        float img[kU], kp[kU], g0[kU], g1[kU];
        unsigned short am[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            img[u] = one_image ? (float)b : rois[(size_t)roi[u] * 5];
            am[u] = argmax[(size_t)roi[u] * C * nb + col[u]];
            kp[u] = (nn[u] < R && keep) ? keep[(size_t)nn[u] * nb + (col[u] - (size_t)c * nb)] : 1.0f;
            g0[u] = 0.0f;
            g1[u] = 0.0f;
            if (nn[u] >= R) {
                g0[u] = extra[(size_t)(nn[u] - R) * C * nb + col[u]];
            } else if (DX_F32) {
                const float* dX = reinterpret_cast<const float*>(dXv);
                if (!skip_clean) g0[u] = dX[(size_t)nn[u] * ld + col[u]];
                if (keep) g1[u] = dX[(size_t)(R + nn[u]) * ld + col[u]];
            } else {
                const unsigned short* dX = reinterpret_cast<const unsigned short*>(dXv);
                if (!skip_clean) g0[u] = rp_bf2f(dX[(size_t)nn[u] * ld + col[u]]);
                if (keep) g1[u] = rp_bf2f(dX[(size_t)(R + nn[u]) * ld + col[u]]);
            }
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            if (!ok[u] || (int)img[u] != b || am[u] == 0xFFFF) continue;
            if (nn[u] < R && keep && skip_clean && kp[u] == 0.0f) continue;
            float g = g0[u];
            if (nn[u] < R && keep) g += ((g1[u] * kp[u]) * numel) / sum;
            odwfx::add(&iacc[am[u]], g, sc.to_fixed);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < HW; i += blockDim.x) dst[i] = (float)iacc[i] * sc.to_float;
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

// ---- the same (ROI, 64-channel) pooling on an fp32 NHWC map, results written as bf16 PLANES --------------------
// Precision mode "bf16x2f" (split-precision forward): the backbone's feature map is fp32 and the first head GEMM reads
// its operand as planes [hi hi mid] along K.  This kernel pools straight from the fp32 map and writes the planes of
// both halves of the stacked operand itself (clean rows, DropBlock rows), the 16-bit argmax, and -- optionally -- the
// fp32 pooled values (the sampled-row views of the contrastive loss read those rows exactly).  It replaces the
// operator-form ROIPool (fp32 out + int32 argmax, 400 MB), stack_clean_aug (400 MB in, 400 MB out) and split_rows
// (400 MB in, 600 MB out).  Values and positions no longer fit one 32-bit key: a thread keeps (ordinal, cell) pairs
// and takes a cell only on a strictly greater ordinal -- cells arrive in row-major order, so the first maximum wins
// like the reference's scan (ROIPool_cuda.cu:62-70).
// pre-pass: order-preserving u32 image of the map; -0.0 folded onto +0.0 (they compare equal in the reference's `>`),
// NaN -> 0 (never greater than anything, like a NaN in that scan)
__global__ __launch_bounds__(256) void nhwc_ord_f32_kernel(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = in[i];
        unsigned d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            unsigned u = d[q];
            if (u == 0x80000000u) u = 0;
            const bool nan = (u & 0x7fffffffu) > 0x7f800000u;
            u ^= (u >> 31) ? 0xFFFFFFFFu : 0x80000000u;
            d[q] = nan ? 0u : u;
        }
        out[i] = make_uint4(d[0], d[1], d[2], d[3]);
    }
}

constexpr unsigned kOrdLowest = 0x00800000u;      // ordinal of -FLT_MAX: the reference's initial maxval (only `>` it wins)

template <int FLY>
__global__ __launch_bounds__(512) void roi_pool_stack_fwd_nhwc_f32(const unsigned* __restrict__ feat,
                                                                   const int* __restrict__ tab, int C, int H, int W, int R,
                                                                   const float* __restrict__ keep,
                                                                   const float* __restrict__ keep_sum, odwpl::Pattern pat,
                                                                   unsigned short* __restrict__ X, long long ld, int block,
                                                                   float* __restrict__ pooled,
                                                                   unsigned short* __restrict__ argmax,
                                                                   unsigned short* __restrict__ X_cm, long long ld_cm,
                                                                   long long cm_mid, int slice_fast) {
    __shared__ __attribute__((aligned(16))) float s_val[64 * 49];
    __shared__ __attribute__((aligned(16))) unsigned short s_arg[64 * 49];
    __shared__ float s_keep[49];
    __shared__ int s_tab[29];
    // workgroup -> (ROI, 64-channel slice).  A one-dimensional grid with the slice as the FAST index: workgroup b runs on XCD
    // b % 8, so with C / 64 a multiple of 8 every XCD pools ONE eighth of the channels for all ROIs and its 4 MB L2 holds that
    // slice of the map (1.5 MB at 76 x 76 x 512) instead of competing for all of it (slice_fast = 0: the ROI as the fast index,
    // rounds 2-4).
    const int nsl = C / 64;
    const int n = slice_fast ? (int)(blockIdx.x / nsl) : (int)(blockIdx.x % R);
    const int c0 = (slice_fast ? (int)(blockIdx.x % nsl) : (int)(blockIdx.x / R)) * 64;
    if (threadIdx.x < 29) s_tab[threadIdx.x] = tab[(size_t)n * 29 + threadIdx.x];
    if (threadIdx.x >= 64 && threadIdx.x < 64 + 49) s_keep[threadIdx.x - 64] = keep ? keep[(size_t)n * 49 + threadIdx.x - 64] : 0.0f;
    __syncthreads();
    const int bin = threadIdx.x >> 3, cg = threadIdx.x & 7;
    if (bin < 49) {
        const int ph = bin / 7, pw = bin - ph * 7;
        const int b = s_tab[0], hs = s_tab[1 + ph], he = s_tab[8 + ph], ws = s_tab[15 + pw], we = s_tab[22 + pw];
        unsigned bv[8], bp[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { bv[q] = kOrdLowest; bp[q] = 0xFFFFu; }
        const unsigned* base = feat + ((size_t)b * H * W) * C + c0 + cg * 8;
        // the window as ONE sequence of cells, FLY cells (2 FLY 16-byte loads) in flight; past the end the last cell is
        // repeated -- an equal ordinal never replaces the held one -- so the loads carry no branch
        const int nw = we - ws, ncell = (he > hs && nw > 0) ? (he - hs) * nw : 0;
        int h = hs, w = ws;
        for (int i = 0; i < ncell; i += FLY) {
            int cell[FLY];
#pragma unroll
            for (int u = 0; u < FLY; ++u) {
                cell[u] = h * W + w;
                if (i + u + 1 < ncell) { if (++w == we) { w = ws; ++h; } }
            }
            uint4 va[FLY], vb[FLY];
#pragma unroll
            for (int u = 0; u < FLY; ++u) {
                va[u] = *reinterpret_cast<const uint4*>(base + (size_t)cell[u] * C);
                vb[u] = *reinterpret_cast<const uint4*>(base + (size_t)cell[u] * C + 4);
            }
#pragma unroll
            for (int u = 0; u < FLY; ++u) {
                const unsigned d[8] = {va[u].x, va[u].y, va[u].z, va[u].w, vb[u].x, vb[u].y, vb[u].z, vb[u].w};
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const bool up = d[q] > bv[q];
                    bv[q] = up ? d[q] : bv[q];
                    bp[q] = up ? (unsigned)cell[u] : bp[q];
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            // empty bin: 0 and "no cell" (ROIPool_cuda.cu:60); otherwise the ordinal decoded (a bin whose every cell is
            // <= -FLT_MAX keeps the initial -FLT_MAX and no cell, like the reference)
            const unsigned bits = (bv[q] & 0x80000000u) ? (bv[q] ^ 0x80000000u) : ~bv[q];
            s_val[(cg * 8 + q) * 49 + bin] = ncell ? __uint_as_float(bits) : 0.0f;
            s_arg[(cg * 8 + q) * 49 + bin] = (unsigned short)bp[q];
        }
    }
    __syncthreads();
    // 64 channels x 49 bins = 3136 values = 392 vectors of 8, contiguous in the output row
    if (threadIdx.x < 392) {
        const int e0 = threadIdx.x * 8;
        const size_t col = (size_t)c0 * 49 + e0;
        const float4 a = *reinterpret_cast<const float4*>(s_val + e0), bq = *reinterpret_cast<const float4*>(s_val + e0 + 4);
        const float v[8] = {a.x, a.y, a.z, a.w, bq.x, bq.y, bq.z, bq.w};
        bool need_lo = false;
        for (int t = 0; t < pat.T; ++t) need_lo |= pat.p[t] == 2;
        *reinterpret_cast<uint4*>(argmax + (size_t)n * C * 49 + col) = *reinterpret_cast<const uint4*>(s_arg + e0);
        if (pooled) {
            *reinterpret_cast<float4*>(pooled + (size_t)n * C * 49 + col) = a;
            *reinterpret_cast<float4*>(pooled + (size_t)n * C * 49 + col + 4) = bq;
        }
        const float numel = (float)((double)R * 49), sum = keep ? *keep_sum : 1.0f;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if (half == 1 && !keep) break;
            float x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = half == 0 ? v[j] : ((v[j] * s_keep[(e0 + j) % 49]) * numel) / sum;
            unsigned hi[4], mid[4], lo[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) odwpl::split2(x[2 * j], x[2 * j + 1], need_lo, hi[j], mid[j], lo[j]);
            unsigned short* dst = X + (size_t)(half == 0 ? n : R + n) * ld + col;
            for (int t = 0; t < pat.T; ++t) {
                const int p = pat.p[t];
                const uint4 o = p == 0 ? make_uint4(hi[0], hi[1], hi[2], hi[3])
                              : (p == 1 ? make_uint4(mid[0], mid[1], mid[2], mid[3])
                              : (p == 2 ? make_uint4(lo[0], lo[1], lo[2], lo[3]) : make_uint4(0, 0, 0, 0)));
                *reinterpret_cast<uint4*>(dst + (size_t)t * block) = o;
            }
        }
        // the clean rows once more as the two CELL-MAJOR planes [hi | mid] (k' = bin * C + c) the shared clean + DropBlock
        // fc6 forward reads (gemm_bf16.hip: gemm_nt_cm_kernel): 8 lanes = 64 channels of one bin = one 128-byte line
        if (X_cm) {
            const int cbin = threadIdx.x >> 3, cg2 = threadIdx.x & 7;
            unsigned hi[4], mid[4], lo[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                odwpl::split2(s_val[(cg2 * 8 + 2 * j) * 49 + cbin], s_val[(cg2 * 8 + 2 * j + 1) * 49 + cbin], false, hi[j], mid[j], lo[j]);
            unsigned short* dst = X_cm + (size_t)n * ld_cm + (size_t)cbin * C + c0 + cg2 * 8;
            *reinterpret_cast<uint4*>(dst) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
            *reinterpret_cast<uint4*>(dst + cm_mid) = make_uint4(mid[0], mid[1], mid[2], mid[3]);
        }
    }
}

// ---- the OPERATOR form on the same (ROI, 64-channel) decomposition (round 4) -------------------------------------
// _C.roi_pool_forward (csrc/ROIPool.h:11-24: NCHW fp32 in, (R, C, PH, PW) fp32 + int32 argmax out) ran on the
// plane-resident kernel: a workgroup owns 1-4 channel planes and every ROI's output for them -- 196-byte output segments
// (one (ROI, channel) block of 7 x 7) and a per-cell LDS scan: 590 us at P = 2000 on 76 x 76 x 512, 0.09 of the HBM
// roofline on its 413 MB.  For a fixed ROI the outputs of 64 consecutive channels are ONE contiguous 12.5 KB block of
// both arrays, so the decomposition of the fused kernel above fits the operator's layout exactly: a pre-pass turns the
// NCHW map into the NHWC ordinal image (one tiled transpose, 12 MB), a workgroup pools (ROI n, channels c0 .. c0 + 63)
// with 16-byte channel-contiguous loads, and the block leaves as full lines.  Same scan order (row-major cells, strictly
// greater wins) => the same values and first-maximum positions, bit for bit.
__global__ __launch_bounds__(256) void nchw_to_nhwc_ord_kernel(const float* __restrict__ in, int C, int HW,
                                                               unsigned* __restrict__ out) {
    __shared__ unsigned tile[32][33];
    const int b = blockIdx.z, c0 = blockIdx.y * 32, p0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;             // 32 x 8
    const float* src = in + (size_t)b * C * HW;
    unsigned* dst = out + (size_t)b * HW * C;
#pragma unroll
    for (int j = 0; j < 32; j += 8) {
        const int c = c0 + ty + j, p = p0 + tx;
        unsigned u = 0;
        if (c < C && p < HW) {
            u = __float_as_uint(src[(size_t)c * HW + p]);
            if (u == 0x80000000u) u = 0;                                   // -0.0 == +0.0 in the reference's `>`
            const bool nan = (u & 0x7fffffffu) > 0x7f800000u;
            u ^= (u >> 31) ? 0xFFFFFFFFu : 0x80000000u;
            u = nan ? 0u : u;
        }
        tile[ty + j][tx] = u;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 32; j += 8) {
        const int p = p0 + ty + j, c = c0 + tx;
        if (p < HW && c < C) dst[(size_t)p * C + c] = tile[tx][ty + j];
    }
}

template <int FLY>
__global__ __launch_bounds__(512) void roi_pool_fwd_nhwc_op(const unsigned* __restrict__ feat, const int* __restrict__ tab,
                                                            int C, int H, int W, int PH, int PW, float* __restrict__ out,
                                                            int* __restrict__ argmax) {
    __shared__ __attribute__((aligned(16))) float s_val[64 * 64];
    __shared__ __attribute__((aligned(16))) int s_arg[64 * 64];
    __shared__ int s_tab[1 + 4 * 64];
    const int n = blockIdx.x, c0 = blockIdx.y * 64;
    const int nb = PH * PW, tl = 1 + 2 * PH + 2 * PW;
    for (int i = threadIdx.x; i < tl; i += blockDim.x) s_tab[i] = tab[(size_t)n * tl + i];
    __syncthreads();
    const int bin = threadIdx.x >> 3, cg = threadIdx.x & 7;
    if (bin < nb && c0 + cg * 8 < C) {
        const int ph = bin / PW, pw = bin - ph * PW;
        const int b = s_tab[0], hs = s_tab[1 + ph], he = s_tab[1 + PH + ph], ws = s_tab[1 + 2 * PH + pw], we = s_tab[1 + 2 * PH + PW + pw];
        unsigned bv[8];
        int bp[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { bv[q] = kOrdLowest; bp[q] = -1; }
        const unsigned* base = feat + ((size_t)b * H * W) * C + c0 + cg * 8;
        const int nw = we - ws, ncell = (he > hs && nw > 0) ? (he - hs) * nw : 0;
        int h = hs, w = ws;
        for (int i = 0; i < ncell; i += FLY) {
            int cell[FLY];
#pragma unroll
            for (int u = 0; u < FLY; ++u) {
                cell[u] = h * W + w;
                if (i + u + 1 < ncell) { if (++w == we) { w = ws; ++h; } }
            }
            uint4 va[FLY], vb[FLY];
#pragma unroll
            for (int u = 0; u < FLY; ++u) {
                va[u] = *reinterpret_cast<const uint4*>(base + (size_t)cell[u] * C);
                vb[u] = *reinterpret_cast<const uint4*>(base + (size_t)cell[u] * C + 4);
            }
#pragma unroll
            for (int u = 0; u < FLY; ++u) {
                const unsigned d[8] = {va[u].x, va[u].y, va[u].z, va[u].w, vb[u].x, vb[u].y, vb[u].z, vb[u].w};
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const bool up = d[q] > bv[q];
                    bv[q] = up ? d[q] : bv[q];
                    bp[q] = up ? cell[u] : bp[q];
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const unsigned bits = (bv[q] & 0x80000000u) ? (bv[q] ^ 0x80000000u) : ~bv[q];
            s_val[(cg * 8 + q) * nb + bin] = ncell ? __uint_as_float(bits) : 0.0f;        // empty bin: 0 (ROIPool_cuda.cu:60)
            s_arg[(cg * 8 + q) * nb + bin] = bp[q];
        }
    }
    __syncthreads();
    // channels c0 .. c0 + 63 of ROI n: ONE contiguous block of both arrays (C % 8 == 0: a multiple of 8 values, 32-byte aligned)
    const int nch = C - c0 < 64 ? C - c0 : 64;
    const int count4 = nch * nb / 4;
    const size_t off = ((size_t)n * C + c0) * nb;
    for (int i = threadIdx.x; i < count4; i += blockDim.x) {
        reinterpret_cast<float4*>(out + off)[i] = reinterpret_cast<const float4*>(s_val)[i];
        reinterpret_cast<int4*>(argmax + off)[i] = reinterpret_cast<const int4*>(s_arg)[i];
    }
}

template <typename K>
hipError_t allow_lds(K kernel, size_t bytes) {
    return odw_set_max_lds(reinterpret_cast<const void*>(kernel),
                               (int)bytes);
}

}  // namespace

ODW_EXPORT int64_t odw_roi_pool_workspace(int R, int PH, int PW) {
    return odw_align_up((int64_t)(R > 0 ? R : 1) * (1 + 2 * PH + 2 * PW) * 4, 256);
}

// Workspace of the (ROI, 64-channel) form of odw_roi_pool_forward: the bin table + the NHWC ordinal image of the map.
// (odw_roi_pool_workspace(R, PH, PW) -- the table alone -- still works: the plane-resident kernels run then.)
ODW_EXPORT int64_t odw_roi_pool_forward_workspace(int B, int C, int H, int W, int R, int PH, int PW) {
    return odw_roi_pool_workspace(R, PH, PW) + odw_align_up((int64_t)(B > 0 ? B : 1) * C * H * W * 4, 256);
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
    static const bool no_nhwc = getenv("ODW_ROI_POOL_PLANE") != nullptr;       // comparison runs: the plane-resident kernels
    if (!no_nhwc && C % 8 == 0 && PH <= 64 && PW <= 64 && PH * PW <= 64 &&
        workspace_bytes >= odw_roi_pool_forward_workspace(B, C, H, W, R, PH, PW) && (((uintptr_t)out) & 15) == 0 &&
        (((uintptr_t)argmax) & 15) == 0 && (long long)B * HW * C < (1ll << 31)) {
        unsigned* ord = (unsigned*)((char*)workspace + odw_roi_pool_workspace(R, PH, PW));
        nchw_to_nhwc_ord_kernel<<<dim3((HW + 31) / 32, (C + 31) / 32, B), 256, 0, stream>>>(feat, C, HW, ord);
        ODW_CHECK_LAUNCH("nchw_to_nhwc_ord_kernel");
        roi_pool_fwd_nhwc_op<2><<<dim3(R, (C + 63) / 64), 512, 0, stream>>>(ord, tab, C, H, W, PH, PW, out, argmax);
        ODW_CHECK_LAUNCH("roi_pool_fwd_nhwc_op");
        return ODW_OK;
    }
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

ODW_EXPORT int odw_roi_pool_backward_det(const float* grad_out, const int32_t* argmax, const float* rois, int B, int C,
                                         int H, int W, int R, int PH, int PW, float* grad_in, void* workspace,
                                         int64_t workspace_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    ODW_REQUIRE(B >= 0 && C > 0 && H > 0 && W > 0 && PH > 0 && PW > 0 && R >= 0, "roi_pool_backward_det: bad dims");
    if (B == 0) return ODW_OK;
    ODW_REQUIRE(grad_in, "roi_pool_backward_det: null grad_in");
    const size_t in_bytes = (size_t)B * C * H * W * 4;
    if (R == 0) {
        ODW_CHECK_HIP(hipMemsetAsync(grad_in, 0, in_bytes, stream), "roi_pool_backward_det memset");
        return ODW_OK;
    }
    ODW_REQUIRE(grad_out && argmax && rois, "roi_pool_backward_det: null pointer");
    ODW_REQUIRE(workspace && workspace_bytes >= 4 && (((uintptr_t)workspace) & 3) == 0, "roi_pool_backward_det: workspace of 4 bytes");
    ODW_REQUIRE(PH * PW <= kPlaneThreads, "roi_pool_backward_det: pooled size too large");
    ODW_REQUIRE((((uintptr_t)grad_out) & 15) == 0, "roi_pool_backward_det: grad_out must be 16-byte aligned");
    unsigned* mx = (unsigned*)workspace;
    ODW_CHECK_HIP(hipMemsetAsync(mx, 0, 4, stream), "roi_pool_backward_det memset");
    const size_t n = (size_t)R * C * PH * PW;
    absmax_kernel<<<2048, 256, 0, stream>>>((const float4*)grad_out, n / 4, grad_out + (n / 4) * 4, (int)(n % 4), mx);
    ODW_CHECK_LAUNCH("absmax_kernel");
    const int HW = H * W;
    const int cap = (ODW_LDS_BYTES - 1024) / 8;                    // 64-bit cells one workgroup holds
    const int cpp = HW < cap ? HW : cap;
    const size_t lds = (size_t)cpp * 8;
    ODW_CHECK_HIP(allow_lds(roi_pool_bwd_plane_det, lds), "roi_pool_bwd_plane_det attr");
    roi_pool_bwd_plane_det<<<B * C, kPlaneThreads, lds, stream>>>(grad_out, argmax, rois, mx, C, H, W, R, PH * PW, cpp, grad_in);
    ODW_CHECK_LAUNCH("roi_pool_bwd_plane_det");
    return ODW_OK;
}

ODW_EXPORT int odw_roi_pool_stack_forward(const float* feat, const float* rois, float spatial_scale, int B, int C,
                                          int H, int W, int R, int PH, int PW, const float* keep, const float* keep_sum,
                                          void* X_bf16, int ld, void* argmax_u16, void* workspace,
                                          int64_t workspace_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    ODW_REQUIRE(B >= 1 && C > 0 && H > 0 && W > 0 && PH > 0 && PW > 0 && R >= 0, "roi_pool_stack_forward: bad dims");
    if (R == 0) return ODW_OK;
    ODW_REQUIRE(feat && rois && X_bf16 && argmax_u16 && (!keep || keep_sum), "roi_pool_stack_forward: null pointer");
    ODW_REQUIRE((long)H * W < 65535, "roi_pool_stack_forward: %dx%d feature map does not fit a 16-bit argmax", H, W);
    ODW_REQUIRE(ld >= C * PH * PW && PH * PW <= kStackThreads / 4, "roi_pool_stack_forward: ld=%d / pooled size", ld);
    if (workspace_bytes < odw_roi_pool_workspace(R, PH, PW) || !workspace) {
        odw_set_error("roi_pool_stack_forward: workspace %lld < %lld bytes", (long long)workspace_bytes,
                      (long long)odw_roi_pool_workspace(R, PH, PW));
        return ODW_EWORKSPACE;
    }
    int* tab = (int*)workspace;
    roi_bins_kernel<<<(R + 255) / 256, 256, 0, stream>>>(rois, spatial_scale, R, PH, PW, H, W, tab);
    ODW_CHECK_LAUNCH("roi_bins_kernel");
    const int HW = H * W;
    ODW_REQUIRE((int64_t)4 * HW * 4 <= ODW_LDS_BYTES, "roi_pool_stack_forward: the 4-level table of a %dx%d plane does not fit in LDS", H, W);
    const int groups = B * C;
    int chunks = (ODW_NUM_CU + groups - 1) / groups;
    chunks = chunks < 1 ? 1 : (chunks > 16 ? 16 : chunks);
    const size_t lds = (size_t)4 * HW * 4;
    const dim3 grid((unsigned)groups, (unsigned)chunks);
    if (PW == 7) {
        ODW_CHECK_HIP(allow_lds(roi_pool_stack_fwd_plane<7>, lds), "roi_pool_stack_fwd_plane<7> attr");
        static const int rowstrip = getenv("ODW_RPS_ROWSTRIP") ? atoi(getenv("ODW_RPS_ROWSTRIP")) : 1;
        roi_pool_stack_fwd_plane<7><<<grid, kPlaneThreads, lds, stream>>>(feat, tab, C, H, W, R, PH, PW, keep, keep_sum,
                                                                         (unsigned short*)X_bf16, ld, (unsigned short*)argmax_u16,
                                                                         PH == 7 ? rowstrip : 0);
    } else {
        ODW_CHECK_HIP(allow_lds(roi_pool_stack_fwd_plane<0>, lds), "roi_pool_stack_fwd_plane<0> attr");
        roi_pool_stack_fwd_plane<0><<<grid, kPlaneThreads, lds, stream>>>(feat, tab, C, H, W, R, PH, PW, keep, keep_sum,
                                                                         (unsigned short*)X_bf16, ld, (unsigned short*)argmax_u16, 0);
    }
    ODW_CHECK_LAUNCH("roi_pool_stack_fwd_plane");
    return ODW_OK;
}

// ---- the same pooling from the backbone's NHWC bf16 map, one workgroup per (ROI, 64-channel chunk) -------------
// The plane-per-workgroup form above owns 98 contiguous output bytes per ROI and array: half of its time goes into
// those partial-line stores.  Here a workgroup owns 64 channels of ONE ROI = 6272 contiguous bytes of each output
// row (column = c*49 + bin): a thread scans one bin for 8 channels with 16-byte loads straight from the NHWC map (L2
// resident, no LDS copy of the plane, no table build), the 49 x 64 results are laid out in LDS in output order and
// leave as full 16-byte vectors.  The key trick is the same: (order-preserving image of the bf16 bits) << 16 |
// (0xFFFF - cell), max over the bin = maximum and first position at once.
// pre-pass: the map in order-preserving form (u16 whose unsigned order is the bf16 order), 16 bytes per thread
__global__ __launch_bounds__(256) void nhwc_ord_kernel(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = in[i];
        unsigned d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            // -0.0 compares EQUAL to +0.0 in the reference's `>` scan (first cell wins): fold it onto +0.0 before the
            // order-preserving map, or a bin whose maximum is zero would prefer the +0.0 cell
            if ((d[q] & 0xFFFFu) == 0x8000u) d[q] &= 0xFFFF0000u;
            if ((d[q] >> 16) == 0x8000u) d[q] &= 0x0000FFFFu;
            d[q] ^= 0x80008000u | (((d[q] >> 15) & 0x00010001u) * 0x7FFFu);
        }
        out[i] = make_uint4(d[0], d[1], d[2], d[3]);
    }
}

__global__ __launch_bounds__(512) void roi_pool_stack_fwd_nhwc(const unsigned short* __restrict__ feat,
                                                               const int* __restrict__ tab, int C, int H, int W, int R,
                                                               const float* __restrict__ keep,
                                                               const float* __restrict__ keep_sum,
                                                               unsigned short* __restrict__ X, int ld,
                                                               unsigned short* __restrict__ argmax) {
    __shared__ __attribute__((aligned(16))) unsigned short s_val[64 * 49], s_arg[64 * 49];
    __shared__ float s_keep[49];
    __shared__ int s_tab[29];
    const int n = blockIdx.x, c0 = blockIdx.y * 64;
    if (threadIdx.x < 29) s_tab[threadIdx.x] = tab[(size_t)n * 29 + threadIdx.x];
    if (threadIdx.x >= 64 && threadIdx.x < 64 + 49) s_keep[threadIdx.x - 64] = keep ? keep[(size_t)n * 49 + threadIdx.x - 64] : 0.0f;
    __syncthreads();
    const int bin = threadIdx.x >> 3, cg = threadIdx.x & 7;
    if (bin < 49) {
        const int ph = bin / 7, pw = bin - ph * 7;
        const int b = s_tab[0], hs = s_tab[1 + ph], he = s_tab[8 + ph], ws = s_tab[15 + pw], we = s_tab[22 + pw];
        unsigned best[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const unsigned short* base = feat + ((size_t)b * H * W) * C + c0 + cg * 8;
        // The window walked as ONE sequence of cells, four loads in flight per round (194 us; eight: 208 us, the repeated tail cells cost more than they hide): a cell per iteration is a serial
        // chain of L2 round trips (13 cells per bin on average, ~1 us each: the kernel was latency-bound at 225 us).
        // Past the end the last cell is repeated -- max is idempotent -- so the loads carry no branch.
        // (`#pragma unroll 4` on the former w loop made it slower: 238 -> 265 us, the tails were predicated.)
        const int nw = we - ws, ncell = (he > hs && nw > 0) ? (he - hs) * nw : 0;
        constexpr int kFly = 4;
        int h = hs, w = ws;
        for (int i = 0; i < ncell; i += kFly) {
            int cell[kFly];
#pragma unroll
            for (int u = 0; u < kFly; ++u) {
                cell[u] = h * W + w;
                if (i + u + 1 < ncell) { if (++w == we) { w = ws; ++h; } }
            }
            uint4 v[kFly];
#pragma unroll
            for (int u = 0; u < kFly; ++u) v[u] = *reinterpret_cast<const uint4*>(base + (size_t)cell[u] * C);
#pragma unroll
            for (int u = 0; u < kFly; ++u) {
                const unsigned pos = 0xFFFFu - (unsigned)cell[u];
                const unsigned d[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {        // `feat` is the pre-pass's order-preserving image of the map
                    best[2 * q] = max(best[2 * q], (d[q] << 16) | pos);
                    best[2 * q + 1] = max(best[2 * q + 1], (d[q] & 0xFFFF0000u) | pos);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            unsigned short vbits = 0, a = 0xFFFF;
            if (best[q]) {
                const unsigned ord = best[q] >> 16;
                vbits = (unsigned short)((ord & 0x8000u) ? (ord & 0x7FFFu) : (~ord & 0xFFFFu));
                a = (unsigned short)(0xFFFFu - (best[q] & 0xFFFFu));
            }
            s_val[(cg * 8 + q) * 49 + bin] = vbits;
            s_arg[(cg * 8 + q) * 49 + bin] = a;
        }
    }
    __syncthreads();
    // 64 channels x 49 bins = 3136 values = 392 vectors of 8, contiguous in the output row
    if (threadIdx.x < 392) {
        const int e0 = threadIdx.x * 8;
        const size_t col = (size_t)c0 * 49 + e0;
        const uint4 vv = *reinterpret_cast<const uint4*>(s_val + e0);
        *reinterpret_cast<uint4*>(X + (size_t)n * ld + col) = vv;
        *reinterpret_cast<uint4*>(argmax + (size_t)n * C * 49 + col) = *reinterpret_cast<const uint4*>(s_arg + e0);
        if (keep) {
            const float numel = (float)((double)R * 49), sum = *keep_sum;
            const unsigned d[4] = {vv.x, vv.y, vv.z, vv.w};
            unsigned o[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int ea = e0 + 2 * q, eb = ea + 1;
                const float ka = s_keep[ea % 49], kb = s_keep[eb % 49];
                const float ra = ((rp_bf2f(d[q] & 0xFFFFu) * ka) * numel) / sum;
                const float rb = ((rp_bf2f(d[q] >> 16) * kb) * numel) / sum;
                o[q] = (unsigned)rp_f2bf(ra) | ((unsigned)rp_f2bf(rb) << 16);
            }
            *reinterpret_cast<uint4*>(X + (size_t)(R + n) * ld + col) = make_uint4(o[0], o[1], o[2], o[3]);
        }
    }
}

ODW_EXPORT int64_t odw_roi_pool_stack_nhwc_f32_workspace(int R, int B, int C, int H, int W) {
    return odw_align_up(odw_roi_pool_workspace(R, 7, 7), 256) + (int64_t)B * H * W * C * 4;
}

ODW_EXPORT int odw_roi_pool_stack_forward_nhwc_f32(const float* feat_nhwc, const float* rois, float spatial_scale, int B, int C,
                                                   int H, int W, int R, const float* keep, const float* keep_sum,
                                                   const int* pattern, int T, void* X_planes, int64_t ld, int block,
                                                   float* pooled_f32, void* argmax_u16, void* workspace,
                                                   int64_t workspace_bytes, void* stream_) {
    return odw_roi_pool_stack_forward_nhwc_f32_cm(feat_nhwc, rois, spatial_scale, B, C, H, W, R, keep, keep_sum, pattern, T,
                                                  X_planes, ld, block, pooled_f32, argmax_u16, nullptr, 0, 0, workspace,
                                                  workspace_bytes, stream_);
}

ODW_EXPORT int odw_roi_pool_stack_forward_nhwc_f32_cm(const float* feat_nhwc, const float* rois, float spatial_scale, int B,
                                                      int C, int H, int W, int R, const float* keep, const float* keep_sum,
                                                      const int* pattern, int T, void* X_planes, int64_t ld, int block,
                                                      float* pooled_f32, void* argmax_u16, void* X_cm, int64_t ld_cm,
                                                      int64_t cm_mid, void* workspace, int64_t workspace_bytes,
                                                      void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    ODW_REQUIRE(!X_cm || ((((uintptr_t)X_cm) & 15) == 0 && cm_mid % 8 == 0 && cm_mid >= (int64_t)C * 49 && ld_cm % 8 == 0 &&
                          ld_cm >= cm_mid + (int64_t)C * 49),
                "roi_pool_stack_forward_nhwc_f32: cell-major planes: alignment / ld / mid offset");
    odwpl::Pattern pat;
    ODW_REQUIRE(odwpl::pattern_ok(pattern, T, pat), "roi_pool_stack_forward_nhwc_f32: pattern = up to %d plane codes in 0..3", odwpl::kMaxTerms);
    ODW_REQUIRE(B >= 1 && C > 0 && C % 64 == 0 && H > 0 && W > 0 && R >= 0, "roi_pool_stack_forward_nhwc_f32: bad dims (C %% 64)");
    if (R == 0) return ODW_OK;
    ODW_REQUIRE(feat_nhwc && rois && X_planes && argmax_u16 && (!keep || keep_sum), "roi_pool_stack_forward_nhwc_f32: null pointer");
    ODW_REQUIRE((long)H * W < 65535, "roi_pool_stack_forward_nhwc_f32: %dx%d feature map does not fit a 16-bit argmax", H, W);
    ODW_REQUIRE(block >= C * 49 && block % 8 == 0 && ld >= (int64_t)T * block && ld % 8 == 0 && (((uintptr_t)X_planes) & 15) == 0 &&
                    (((uintptr_t)argmax_u16) & 15) == 0 && (((uintptr_t)feat_nhwc) & 15) == 0 && (((uintptr_t)pooled_f32) & 15) == 0,
                "roi_pool_stack_forward_nhwc_f32: alignment / ld / block");
    const int64_t need = odw_roi_pool_stack_nhwc_f32_workspace(R, B, C, H, W);
    if (workspace_bytes < need || !workspace) {
        odw_set_error("roi_pool_stack_forward_nhwc_f32: workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)need);
        return ODW_EWORKSPACE;
    }
    ODW_REQUIRE((((uintptr_t)workspace) & 15) == 0, "roi_pool_stack_forward_nhwc_f32: workspace alignment");
    int* tab = (int*)workspace;
    unsigned* ordmap = (unsigned*)((char*)workspace + odw_align_up(odw_roi_pool_workspace(R, 7, 7), 256));
    roi_bins_kernel<<<(R + 255) / 256, 256, 0, stream>>>(rois, spatial_scale, R, 7, 7, H, W, tab);
    ODW_CHECK_LAUNCH("roi_bins_kernel");
    const size_t n16 = (size_t)B * H * W * C / 4;
    nhwc_ord_f32_kernel<<<(unsigned)((n16 + 255) / 256 < 4096 ? (n16 + 255) / 256 : 4096), 256, 0, stream>>>(
        (const uint4*)feat_nhwc, (uint4*)ordmap, n16);
    ODW_CHECK_LAUNCH("nhwc_ord_f32_kernel");
    static const int fly = getenv("ODW_POOL_F32_FLY") ? atoi(getenv("ODW_POOL_F32_FLY")) : 2;
    static const int slice_fast = getenv("ODW_POOL_SLICE_FAST") ? atoi(getenv("ODW_POOL_SLICE_FAST")) : 1;
    const unsigned grid = (unsigned)R * (unsigned)(C / 64);
    if (fly == 4)
        roi_pool_stack_fwd_nhwc_f32<4><<<grid, 512, 0, stream>>>(ordmap, tab, C, H, W, R, keep, keep_sum, pat, (unsigned short*)X_planes,
                                                                 (long long)ld, block, pooled_f32, (unsigned short*)argmax_u16,
                                                                 (unsigned short*)X_cm, (long long)ld_cm, (long long)cm_mid, slice_fast);
    else if (fly == 1)
        roi_pool_stack_fwd_nhwc_f32<1><<<grid, 512, 0, stream>>>(ordmap, tab, C, H, W, R, keep, keep_sum, pat, (unsigned short*)X_planes,
                                                                 (long long)ld, block, pooled_f32, (unsigned short*)argmax_u16,
                                                                 (unsigned short*)X_cm, (long long)ld_cm, (long long)cm_mid, slice_fast);
    else
        roi_pool_stack_fwd_nhwc_f32<2><<<grid, 512, 0, stream>>>(ordmap, tab, C, H, W, R, keep, keep_sum, pat, (unsigned short*)X_planes,
                                                                 (long long)ld, block, pooled_f32, (unsigned short*)argmax_u16,
                                                                 (unsigned short*)X_cm, (long long)ld_cm, (long long)cm_mid, slice_fast);
    ODW_CHECK_LAUNCH("roi_pool_stack_fwd_nhwc_f32");
    return ODW_OK;
}

ODW_EXPORT int64_t odw_roi_pool_stack_nhwc_workspace(int R, int B, int C, int H, int W) {
    return odw_align_up(odw_roi_pool_workspace(R, 7, 7), 256) + (int64_t)B * H * W * C * 2;
}

ODW_EXPORT int odw_roi_pool_stack_forward_nhwc(const void* feat_nhwc_bf16, const float* rois, float spatial_scale, int B,
                                               int C, int H, int W, int R, const float* keep, const float* keep_sum,
                                               void* X_bf16, int ld, void* argmax_u16, void* workspace,
                                               int64_t workspace_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    ODW_REQUIRE(B >= 1 && C > 0 && C % 64 == 0 && H > 0 && W > 0 && R >= 0, "roi_pool_stack_forward_nhwc: bad dims (C %% 64)");
    if (R == 0) return ODW_OK;
    ODW_REQUIRE(feat_nhwc_bf16 && rois && X_bf16 && argmax_u16 && (!keep || keep_sum), "roi_pool_stack_forward_nhwc: null pointer");
    ODW_REQUIRE((long)H * W < 65535, "roi_pool_stack_forward_nhwc: %dx%d feature map does not fit a 16-bit argmax", H, W);
    ODW_REQUIRE(ld >= C * 49 && ld % 8 == 0 && (((uintptr_t)X_bf16) & 15) == 0 && (((uintptr_t)argmax_u16) & 15) == 0 &&
                    (((uintptr_t)feat_nhwc_bf16) & 15) == 0, "roi_pool_stack_forward_nhwc: alignment / ld");
    const int64_t need = odw_roi_pool_stack_nhwc_workspace(R, B, C, H, W);
    if (workspace_bytes < need || !workspace) {
        odw_set_error("roi_pool_stack_forward_nhwc: workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)need);
        return ODW_EWORKSPACE;
    }
    ODW_REQUIRE((((uintptr_t)workspace) & 15) == 0, "roi_pool_stack_forward_nhwc: workspace alignment");
    int* tab = (int*)workspace;
    unsigned short* ordmap = (unsigned short*)((char*)workspace + odw_align_up(odw_roi_pool_workspace(R, 7, 7), 256));
    roi_bins_kernel<<<(R + 255) / 256, 256, 0, stream>>>(rois, spatial_scale, R, 7, 7, H, W, tab);
    ODW_CHECK_LAUNCH("roi_bins_kernel");
    const size_t n16 = (size_t)B * H * W * C / 8;
    nhwc_ord_kernel<<<(unsigned)((n16 + 255) / 256 < 4096 ? (n16 + 255) / 256 : 4096), 256, 0, stream>>>(
        (const uint4*)feat_nhwc_bf16, (uint4*)ordmap, n16);
    ODW_CHECK_LAUNCH("nhwc_ord_kernel");
    roi_pool_stack_fwd_nhwc<<<dim3((unsigned)R, (unsigned)(C / 64)), 512, 0, stream>>>(
        ordmap, tab, C, H, W, R, keep, keep_sum, (unsigned short*)X_bf16, ld, (unsigned short*)argmax_u16);
    ODW_CHECK_LAUNCH("roi_pool_stack_fwd_nhwc");
    return ODW_OK;
}

ODW_EXPORT int odw_roi_pool_stack_backward(const void* dX, int dx_is_f32, int ld, const void* argmax_u16,
                                           const float* rois, const float* keep, const float* keep_sum,
                                           const float* extra, const int* extra_roi, int E, int skip_clean, int B, int C,
                                           int H, int W, int R, int PH, int PW, float* grad_in, void* stream_) {
    return odw_roi_pool_stack_backward_ws(dX, dx_is_f32, ld, argmax_u16, rois, keep, keep_sum, extra, extra_roi, E, skip_clean,
                                          B, C, H, W, R, PH, PW, grad_in, nullptr, 0, stream_);
}

static int roi_pool_stack_backward_launch(const void* dX, int dx_is_f32, int ld, const void* argmax_u16,
                                          const float* rois, const float* keep, const float* keep_sum,
                                          const float* extra, const int* extra_roi, int E, int skip_clean, int B, int C,
                                          int H, int W, int R, int PH, int PW, float* grad_in, void* workspace,
                                          int64_t workspace_bytes, void* stream_, const int* e_dev, int absmax_ready = 0);

ODW_EXPORT int odw_roi_pool_stack_backward_ws(const void* dX, int dx_is_f32, int ld, const void* argmax_u16,
                                              const float* rois, const float* keep, const float* keep_sum,
                                              const float* extra, const int* extra_roi, int E, int skip_clean, int B, int C,
                                              int H, int W, int R, int PH, int PW, float* grad_in, void* workspace,
                                              int64_t workspace_bytes, void* stream_) {
    return roi_pool_stack_backward_launch(dX, dx_is_f32, ld, argmax_u16, rois, keep, keep_sum, extra, extra_roi, E, skip_clean, B, C,
                                          H, W, R, PH, PW, grad_in, workspace, workspace_bytes, stream_, nullptr);
}

// The same backward with the number of side-buffer entries on the device (round 6: the sampled rows and the re-attached
// clean rows of the contrastive loss are counted by loss_lists.hip): E_cap entries of `extra` / `extra_roi` exist as memory,
// the first *e_dev are scattered.  Needs the fixed-point form (a 4-byte workspace).
ODW_EXPORT int odw_roi_pool_stack_backward_dyn(const void* dX, int dx_is_f32, int ld, const void* argmax_u16,
                                               const float* rois, const float* keep, const float* keep_sum,
                                               const float* extra, const int* extra_roi, int E_cap, const int* e_dev,
                                               int skip_clean, int B, int C, int H, int W, int R, int PH, int PW,
                                               float* grad_in, void* workspace, int64_t workspace_bytes, void* stream_) {
    ODW_REQUIRE(e_dev && E_cap >= 1 && extra && extra_roi, "roi_pool_stack_backward_dyn: the side buffer and its device-resident count");
    return roi_pool_stack_backward_launch(dX, dx_is_f32, ld, argmax_u16, rois, keep, keep_sum, extra, extra_roi, E_cap, skip_clean, B,
                                          C, H, W, R, PH, PW, grad_in, workspace, workspace_bytes, stream_, e_dev);
}

// The same backward when the producer of dX already left max |dX| (over the rows this launch reads: [R, 2R) with skip_clean,
// all 2R otherwise) in workspace[0..4) -- odw_gemm_nt_bf16_absmax's epilogue: the 200 MB pre-pass over dX is skipped, the side
// buffer's (small) pre-pass still runs onto the same word.  e_dev null: E is exact.  Fixed-point form only.
ODW_EXPORT int odw_roi_pool_stack_backward_scaled(const void* dX, int dx_is_f32, int ld, const void* argmax_u16,
                                                  const float* rois, const float* keep, const float* keep_sum,
                                                  const float* extra, const int* extra_roi, int E_cap, const int* e_dev,
                                                  int skip_clean, int B, int C, int H, int W, int R, int PH, int PW,
                                                  float* grad_in, void* workspace, int64_t workspace_bytes, void* stream_) {
    ODW_REQUIRE(workspace && workspace_bytes >= 4, "roi_pool_stack_backward_scaled: the word holding max |dX|");
    ODW_REQUIRE(!e_dev || (E_cap >= 1 && extra && extra_roi), "roi_pool_stack_backward_scaled: the side buffer of a device-resident count");
    return roi_pool_stack_backward_launch(dX, dx_is_f32, ld, argmax_u16, rois, keep, keep_sum, extra, extra_roi, E_cap, skip_clean, B,
                                          C, H, W, R, PH, PW, grad_in, workspace, workspace_bytes, stream_, e_dev, 1);
}

static int roi_pool_stack_backward_launch(const void* dX, int dx_is_f32, int ld, const void* argmax_u16,
                                          const float* rois, const float* keep, const float* keep_sum,
                                          const float* extra, const int* extra_roi, int E, int skip_clean, int B, int C,
                                          int H, int W, int R, int PH, int PW, float* grad_in, void* workspace,
                                          int64_t workspace_bytes, void* stream_, const int* e_dev, int absmax_ready) {
    hipStream_t stream = (hipStream_t)stream_;
    ODW_REQUIRE(B >= 1 && C > 0 && H > 0 && W > 0 && PH > 0 && PW > 0 && R >= 1 && E >= 0,
                "roi_pool_stack_backward: bad dims");
    ODW_REQUIRE(dX && argmax_u16 && rois && grad_in && (!keep || keep_sum) && (E == 0 || (extra && extra_roi)),
                "roi_pool_stack_backward: null pointer");
    const int HW = H * W, nb = PH * PW;
    ODW_REQUIRE((int64_t)HW * 4 <= ODW_LDS_BYTES && nb <= kPlaneThreads / 4, "roi_pool_stack_backward: plane / pooled size");
    // fixed-point form (deterministic, 10x the LDS atomic rate): needs 4 bytes of workspace for the launch's scale
    // and a dX without row padding (the max pre-pass reads it as one array)
    static const bool float_atomics = getenv("ODW_POOL_BWD_ATOMIC") != nullptr;
    const size_t el = dx_is_f32 ? 4 : 2;
    if (!float_atomics && workspace && workspace_bytes >= 4 && (((uintptr_t)workspace) & 3) == 0 && ld == C * nb &&
        (int64_t)HW * 8 <= ODW_LDS_BYTES - 1024 && (((uintptr_t)dX) & 15) == 0 && ((size_t)R * ld * el) % 16 == 0 &&
        ((size_t)R * ld) % 8 == 0 && (!extra || (((uintptr_t)extra) & 15) == 0)) {
        unsigned* mx = (unsigned*)workspace;
        const char* first = reinterpret_cast<const char*>(dX) + (skip_clean ? (size_t)R * ld * el : 0);     // rows [0, R) are unset
        const size_t n = (size_t)(skip_clean ? R : 2 * R) * ld;
        if (!absmax_ready) {        // (ready: the producing GEMM's epilogue took the maximum, odw_gemm_nt_bf16_absmax)
            ODW_CHECK_HIP(hipMemsetAsync(mx, 0, 4, stream), "roi_pool_stack_backward memset");
            if (dx_is_f32) odwfx::absmax_kernel<false><<<1024, 256, 0, stream>>>(first, n, mx);
            else odwfx::absmax_kernel<true><<<1024, 256, 0, stream>>>(first, n, mx);
        }
        if (E > 0) odwfx::absmax_kernel<false><<<256, 256, 0, stream>>>(extra, (size_t)E * C * nb, mx, e_dev, (size_t)C * nb);
        ODW_CHECK_LAUNCH("absmax_kernel");
        const size_t lds8 = (size_t)HW * 8;
        if (dx_is_f32) {
            ODW_CHECK_HIP(allow_lds(roi_pool_stack_bwd_plane_fx<true>, lds8), "roi_pool_stack_bwd_plane_fx attr");
            roi_pool_stack_bwd_plane_fx<true><<<B * C, kPlaneThreads, lds8, stream>>>(
                dX, ld, (const unsigned short*)argmax_u16, rois, keep, keep_sum, extra, extra_roi, E, skip_clean, mx, C, H, W, R,
                nb, grad_in, e_dev);
        } else {
            ODW_CHECK_HIP(allow_lds(roi_pool_stack_bwd_plane_fx<false>, lds8), "roi_pool_stack_bwd_plane_fx attr");
            roi_pool_stack_bwd_plane_fx<false><<<B * C, kPlaneThreads, lds8, stream>>>(
                dX, ld, (const unsigned short*)argmax_u16, rois, keep, keep_sum, extra, extra_roi, E, skip_clean, mx, C, H, W, R,
                nb, grad_in, e_dev);
        }
        ODW_CHECK_LAUNCH("roi_pool_stack_bwd_plane_fx");
        return ODW_OK;
    }
    ODW_REQUIRE(!e_dev && !absmax_ready, "roi_pool_stack_backward_dyn / _scaled: the fixed-point form does not apply (workspace, alignment or plane size)");
    const int cg = ((int64_t)2 * HW * 4 <= ODW_LDS_BYTES && (int64_t)B * ((C + 1) / 2) >= 2 * ODW_NUM_CU) ? 2 : 1;
    const int grid = B * ((C + cg - 1) / cg);
    const size_t lds = (size_t)cg * HW * 4;
#define ODW_RPS_BWD(CGV, F32)                                                                                       \
    do {                                                                                                            \
        ODW_CHECK_HIP(allow_lds(roi_pool_stack_bwd_plane<CGV, F32>, lds), "roi_pool_stack_bwd_plane attr");         \
        roi_pool_stack_bwd_plane<CGV, F32><<<grid, kPlaneThreads, lds, stream>>>(                                   \
            dX, ld, (const unsigned short*)argmax_u16, rois, keep, keep_sum, extra, extra_roi, E, skip_clean, C, H, W, R, nb, \
            grad_in);                                                                                               \
    } while (0)
    if (cg == 2) { if (dx_is_f32) ODW_RPS_BWD(2, true); else ODW_RPS_BWD(2, false); }
    else { if (dx_is_f32) ODW_RPS_BWD(1, true); else ODW_RPS_BWD(1, false); }
#undef ODW_RPS_BWD
    ODW_CHECK_LAUNCH("roi_pool_stack_bwd_plane");
    return ODW_OK;
}
