// roi_align.hip -- ROIAlign forward/backward for gfx950 (legacy, un-aligned
// sampling: wetectron/csrc/cuda/ROIAlign_cuda.cu:16-254 ==
// csrc/cpu/ROIAlign_cpu.cpp:18-219).
//
// Same plane-resident structure as roi_pool.hip: CG channel planes of one
// image sit in LDS, every ROI of the image is served from there (bilinear
// gather in LDS, coalesced HBM reads of the planes only); backward scatters
// the 4 taps with LDS float atomics and writes each plane back once.
// Built with -ffp-contract=off: the sample coordinate
//   start + ph*bin + (iy+.5)*bin/grid          (ROIAlign_cuda.cu:109-112)
// must be evaluated as separate fp32 mul/add/div, not FMA, to land on the
// same side of integer boundaries as the reference.
#include "odw_common.h"
#include "odw_fixed.h"
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int kPlaneThreads = 1024;

struct RoiGeom {
    int b;
    float sw, sh, bin_w, bin_h;
    int gw, gh;
    float count;
};

__device__ __forceinline__ RoiGeom roi_geom(const float* __restrict__ roi, float scale, int PH, int PW,
                                            int sampling_ratio) {
    RoiGeom g;
    g.b = (int)roi[0];
    g.sw = roi[1] * scale;
    g.sh = roi[2] * scale;
    float ew = roi[3] * scale, eh = roi[4] * scale;
    float rw = fmaxf(ew - g.sw, 1.0f), rh = fmaxf(eh - g.sh, 1.0f);
    g.bin_h = rh / (float)PH;
    g.bin_w = rw / (float)PW;
    g.gh = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / (float)PH);
    g.gw = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / (float)PW);
    g.count = (float)(g.gh * g.gw);
    return g;
}

// 4 taps of one bilinear sample; false = outside [-1,H]x[-1,W]
__device__ __forceinline__ bool taps(int H, int W, float y, float x, int pos[4], float wg[4]) {
    if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) return false;
    if (y <= 0) y = 0;
    if (x <= 0) x = 0;
    int yl = (int)y, xl = (int)x, yh, xh;
    if (yl >= H - 1) { yh = yl = H - 1; y = (float)yl; } else { yh = yl + 1; }
    if (xl >= W - 1) { xh = xl = W - 1; x = (float)xl; } else { xh = xl + 1; }
    float ly = y - (float)yl, lx = x - (float)xl;
    float hy = 1.0f - ly, hx = 1.0f - lx;
    pos[0] = yl * W + xl; pos[1] = yl * W + xh; pos[2] = yh * W + xl; pos[3] = yh * W + xh;
    wg[0] = hy * hx; wg[1] = hy * lx; wg[2] = ly * hx; wg[3] = ly * lx;
    return true;
}

__device__ __forceinline__ float align_one(const float* p, const RoiGeom& g, int H, int W, int ph, int pw) {
    float acc = 0.0f;
    for (int iy = 0; iy < g.gh; ++iy) {
        float y = g.sh + (float)ph * g.bin_h + ((float)iy + 0.5f) * g.bin_h / (float)g.gh;
        for (int ix = 0; ix < g.gw; ++ix) {
            float x = g.sw + (float)pw * g.bin_w + ((float)ix + 0.5f) * g.bin_w / (float)g.gw;
            int pos[4]; float wg[4];
            if (!taps(H, W, y, x, pos, wg)) continue;
            acc += wg[0] * p[pos[0]] + wg[1] * p[pos[1]] + wg[2] * p[pos[2]] + wg[3] * p[pos[3]];
        }
    }
    return acc / g.count;
}

template <typename AddFn>
__device__ __forceinline__ void align_scatter(const RoiGeom& g, int H, int W, int ph, int pw, float go,
                                              AddFn add) {
    for (int iy = 0; iy < g.gh; ++iy) {
        float y = g.sh + (float)ph * g.bin_h + ((float)iy + 0.5f) * g.bin_h / (float)g.gh;
        for (int ix = 0; ix < g.gw; ++ix) {
            float x = g.sw + (float)pw * g.bin_w + ((float)ix + 0.5f) * g.bin_w / (float)g.gw;
            int pos[4]; float wg[4];
            if (!taps(H, W, y, x, pos, wg)) continue;
#pragma unroll
            for (int k = 0; k < 4; ++k) add(pos[k], go * wg[k] / g.count);  // ROIAlign_cuda.cu:237-240
        }
    }
}

template <int CG>
__global__ __launch_bounds__(kPlaneThreads) void roi_align_fwd_plane(
    const float* __restrict__ feat, const float* __restrict__ rois, float scale, int C, int H, int W,
    int R, int PH, int PW, int sr, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float plane[];
    const int groups = (C + CG - 1) / CG;
    const int b = blockIdx.x / groups;
    const int c0 = (blockIdx.x % groups) * CG;
    const int nc = min(CG, C - c0);
    const int HW = H * W;
    {
        const float* src = feat + ((size_t)b * C + c0) * HW;
        const int count = nc * HW;
        if ((((uintptr_t)src) & 15) == 0 && (count & 3) == 0) {
            for (int i = threadIdx.x; i < count / 4; i += blockDim.x)
                reinterpret_cast<float4*>(plane)[i] = reinterpret_cast<const float4*>(src)[i];
        } else {
            for (int i = threadIdx.x; i < count; i += blockDim.x) plane[i] = src[i];
        }
    }
    __syncthreads();
    const int nb = PH * PW, per_roi = nc * nb;
    int n = threadIdx.x / per_roi, r = threadIdx.x % per_roi;
    const int dn = kPlaneThreads / per_roi, dr = kPlaneThreads % per_roi;
    for (; n < R; n += dn, r += dr) {
        if (r >= per_roi) { r -= per_roi; ++n; if (n >= R) break; }
        const RoiGeom g = roi_geom(rois + (size_t)n * 5, scale, PH, PW, sr);
        if (g.b != b) continue;
        const int cl = r / nb, bin = r - cl * nb;
        const int ph = bin / PW, pw = bin - ph * PW;
        out[((size_t)n * C + c0 + cl) * nb + bin] = align_one(plane + cl * HW, g, H, W, ph, pw);
    }
}

__global__ void roi_align_fwd_direct(const float* __restrict__ feat, const float* __restrict__ rois,
                                     float scale, int C, int H, int W, int R, int PH, int PW, int sr,
                                     float* __restrict__ out) {
    const int nb = PH * PW;
    const size_t total = (size_t)R * C * nb;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        int bin = (int)(i % nb), c = (int)((i / nb) % C), n = (int)(i / nb / C);
        const RoiGeom g = roi_geom(rois + (size_t)n * 5, scale, PH, PW, sr);
        int ph = bin / PW, pw = bin - ph * PW;
        out[i] = align_one(feat + ((size_t)g.b * C + c) * H * W, g, H, W, ph, pw);
    }
}

template <int CG>
__global__ __launch_bounds__(kPlaneThreads) void roi_align_bwd_plane(
    const float* __restrict__ grad_out, const float* __restrict__ rois, float scale, int C, int H, int W,
    int R, int PH, int PW, int sr, float* __restrict__ grad_in) {
    extern __shared__ __attribute__((aligned(16))) float acc[];
    const int groups = (C + CG - 1) / CG;
    const int b = blockIdx.x / groups;
    const int c0 = (blockIdx.x % groups) * CG;
    const int nc = min(CG, C - c0);
    const int HW = H * W;
    for (int i = threadIdx.x; i < nc * HW; i += blockDim.x) acc[i] = 0.0f;
    __syncthreads();
    const int nb = PH * PW, per_roi = nc * nb;
    int n = threadIdx.x / per_roi, r = threadIdx.x % per_roi;
    const int dn = kPlaneThreads / per_roi, dr = kPlaneThreads % per_roi;
    for (; n < R; n += dn, r += dr) {
        if (r >= per_roi) { r -= per_roi; ++n; if (n >= R) break; }
        const RoiGeom g = roi_geom(rois + (size_t)n * 5, scale, PH, PW, sr);
        if (g.b != b) continue;
        const int cl = r / nb, bin = r - cl * nb;
        const int ph = bin / PW, pw = bin - ph * PW;
        float* p = acc + cl * HW;
        const float go = grad_out[((size_t)n * C + c0 + cl) * nb + bin];
        align_scatter(g, H, W, ph, pw, go, [p](int pos, float v) { atomicAdd(p + pos, v); });
    }
    __syncthreads();
    float* dst = grad_in + ((size_t)b * C + c0) * HW;
    for (int i = threadIdx.x; i < nc * HW; i += blockDim.x) dst[i] = acc[i];
}

__global__ void roi_align_bwd_direct(const float* __restrict__ grad_out, const float* __restrict__ rois,
                                     float scale, int C, int H, int W, int R, int PH, int PW, int sr,
                                     float* __restrict__ grad_in) {
    const int nb = PH * PW;
    const size_t total = (size_t)R * C * nb;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        int bin = (int)(i % nb), c = (int)((i / nb) % C), n = (int)(i / nb / C);
        const RoiGeom g = roi_geom(rois + (size_t)n * 5, scale, PH, PW, sr);
        int ph = bin / PW, pw = bin - ph * PW;
        float* p = grad_in + ((size_t)g.b * C + c) * H * W;
        align_scatter(g, H, W, ph, pw, grad_out[i], [p](int pos, float v) { atomicAdd(p + pos, v); });
    }
}

// ---- separable form of the backward -----------------------------------------------------------------------------
// The bilinear weight of sample (iy, ix) on a cell is a PRODUCT of a row weight and a column weight, the sample grid of
// a bin is a product grid, and the validity test (y in [-1,H], x in [-1,W]) is a conjunction: the gradient of bin
// (ph, pw) on cell (cy, cx) is go / count * wy[ph][cy] * wx[pw][cx] with wy / wx summed over the samples of one axis.
// A pre-pass builds the 14 axis vectors of every ROI once (they do not depend on the channel); the plane kernel then
// spends one multiply + one LDS atomic per touched CELL (~(bin+2)^2) instead of re-deriving four taps per SAMPLE
// (4 * ceil(bin)^2, ~60 instructions each) in every one of the C planes: 16.7 ms -> ~1 ms at P = 2000 on 76x76x512.
// Only the backward: its summation order is free (atomics); the forward keeps the reference's order.
constexpr int kAxisLen = 34;                 // cells one bin can touch along an axis (bin extent + 2); longer: fallback
constexpr int kAxisStride = 2 + kAxisLen;    // [first cell, cell count, weights...]; 36 floats: the 7 vectors of an axis
                                             // start in 7 different LDS banks (a 32-float pitch put them in 2)

__global__ void roi_align_axis_kernel(const float* __restrict__ rois, float scale, int R, int PH, int PW, int H, int W,
                                      int sr, float* __restrict__ tab) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int per = PH + PW;
    if (t >= R * per) return;
    const int n = t / per, k = t - n * per;
    const RoiGeom g = roi_geom(rois + (size_t)n * 5, scale, PH, PW, sr);
    const bool is_y = k < PH;
    const int p = is_y ? k : k - PH;
    const int L = is_y ? H : W, grid = is_y ? g.gh : g.gw;
    const float start = is_y ? g.sh : g.sw, bin = is_y ? g.bin_h : g.bin_w;
    float w[kAxisLen + 1];
#pragma unroll
    for (int i = 0; i <= kAxisLen; ++i) w[i] = 0.0f;
    int first = -1, last = -1;
    bool overflow = false;              // a bin wider than the vector (a box far larger than the map): sample form
    for (int i = 0; i < grid; ++i) {
        float v = start + (float)p * bin + ((float)i + 0.5f) * bin / (float)grid;      // ROIAlign_cuda.cu:109-112
        if (v < -1.0f || v > (float)L) continue;
        if (v <= 0) v = 0;
        int lo = (int)v, hi;
        if (lo >= L - 1) { hi = lo = L - 1; v = (float)lo; } else { hi = lo + 1; }
        const float l = v - (float)lo, hwt = 1.0f - l;
        if (first < 0) first = lo;
        const int a = lo - first, b2 = hi - first;
        if (b2 >= kAxisLen) { overflow = true; break; }
        w[a] += hwt;
        w[b2] += l;
        last = hi;
    }
    float* o = tab + (size_t)t * kAxisStride;
    reinterpret_cast<int*>(o)[0] = first < 0 ? 0 : first;
    reinterpret_cast<int*>(o)[1] = overflow ? -1 : (first < 0 ? 0 : last - first + 1);
#pragma unroll
    for (int i = 0; i < kAxisLen; ++i) o[2 + i] = w[i];
}

template <int CG>
__global__ __launch_bounds__(kPlaneThreads) void roi_align_bwd_sep_plane(
    const float* __restrict__ grad_out, const float* __restrict__ rois, float scale, const float* __restrict__ tab,
    int C, int H, int W, int R, int PH, int PW, int sr, float* __restrict__ grad_in) {
    extern __shared__ __attribute__((aligned(16))) float acc[];
    const int groups = (C + CG - 1) / CG;
    const int b = blockIdx.x / groups;
    const int c0 = (blockIdx.x % groups) * CG;
    const int nc = min(CG, C - c0);
    const int HW = H * W;
    for (int i = threadIdx.x; i < nc * HW; i += blockDim.x) acc[i] = 0.0f;
    __syncthreads();
    const int nb = PH * PW, per_roi = nc * nb, per = PH + PW;
    int n = threadIdx.x / per_roi, r = threadIdx.x % per_roi;
    const int dn = kPlaneThreads / per_roi, dr = kPlaneThreads % per_roi;
    for (; n < R; n += dn, r += dr) {
        if (r >= per_roi) { r -= per_roi; ++n; if (n >= R) break; }
        const RoiGeom g = roi_geom(rois + (size_t)n * 5, scale, PH, PW, sr);
        if (g.b != b) continue;
        const int cl = r / nb, bin = r - cl * nb;
        const int ph = bin / PW, pw = bin - ph * PW;
        const float* ty = tab + ((size_t)n * per + ph) * kAxisStride;
        const float* tx = tab + ((size_t)n * per + PH + pw) * kAxisStride;
        const int y0 = reinterpret_cast<const int*>(ty)[0], ny = reinterpret_cast<const int*>(ty)[1];
        const int x0 = reinterpret_cast<const int*>(tx)[0], nx = reinterpret_cast<const int*>(tx)[1];
        if (ny == 0 || nx == 0) continue;
        if (ny < 0 || nx < 0) {         // oversized bin: the sample-by-sample form
            float* q = acc + cl * HW;
            align_scatter(g, H, W, ph, pw, grad_out[((size_t)n * C + c0 + cl) * nb + bin],
                          [q](int pos, float v) { atomicAdd(q + pos, v); });
            continue;
        }
        const float go = grad_out[((size_t)n * C + c0 + cl) * nb + bin] / g.count;
        float* p = acc + cl * HW + y0 * W + x0;
        for (int cy = 0; cy < ny; ++cy) {
            const float gy = go * ty[2 + cy];
            for (int cx = 0; cx < nx; ++cx) atomicAdd(p + cy * W + cx, gy * tx[2 + cx]);
        }
    }
    __syncthreads();
    float* dst = grad_in + ((size_t)b * C + c0) * HW;
    for (int i = threadIdx.x; i < nc * HW; i += blockDim.x) dst[i] = acc[i];
}

// ---- separable forward + backward with the axis vectors staged in LDS ------------------------------------------
// The forward is the same sum regrouped: out = (1/count) sum_cy sum_cx wy[cy] wx[cx] F[y0+cy][x0+cx] -- (bin+2)^2
// cell reads instead of 4 taps and ~60 coordinate instructions per SAMPLE, with the sample coordinates themselves
// still evaluated by roi_align_axis_kernel in the reference's fp32 operation order (which cell a sample lands in is
// what must not move; the order in which fp32 terms are added may: 1e-6 against the reference's own CPU kernel).
// A workgroup owns CG planes in LDS; each of its 16 WAVES takes one ROI at a time from a shared counter (ROI sizes
// span 2.5 .. 75 cells: in lock-step chunks the whole workgroup waited for the largest box of every chunk), keeps the
// ROI's 14 axis vectors in a wave-private LDS slot (the next ROI's vectors travel global -> registers meanwhile; the
// first version read them from global memory inside the cell loop: 5.8 ms backward at P = 2000), and a lane = one bin
// forms each cell weight once for all CG planes.  No workgroup barrier between the plane load and the final store.
constexpr int kSepWaves = kPlaneThreads / 64;

template <int CG, bool BWD>
__global__ __launch_bounds__(kPlaneThreads) void roi_align_sep_plane(
    const float* __restrict__ src, const float* __restrict__ rois, float scale, const float* __restrict__ tab,
    const unsigned* __restrict__ absmax_bits, int C, int H, int W, int R, int PH, int PW, int sr, float* __restrict__ dst) {
    // FWD: src = feat (B,C,H,W), dst = out (R,C,PH,PW), planes fp32;
    // BWD: src = grad_out, dst = grad_in, planes 64-bit fixed point (odw_fixed.h: deterministic, and the integer LDS
    //      atomic runs at 10x the rate of ds_add_f32)
    typedef typename std::conditional<BWD, long long, float>::type cell_t;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int groups = (C + CG - 1) / CG;
    const int b = blockIdx.x / groups;
    const int c0 = (blockIdx.x % groups) * CG;
    const int nc = min(CG, C - c0);
    const int HW = H * W;
    cell_t* plane = reinterpret_cast<cell_t*>(smem_raw);                                  // CG * HW
    const int per = PH + PW, nb = PH * PW;
    const int tab_roi = per * kAxisStride;                                                 // floats per ROI; a multiple of 4
    float* stab = reinterpret_cast<float*>(smem_raw + (((size_t)CG * HW * sizeof(cell_t) + 15) & ~(size_t)15));
    int* next_roi = reinterpret_cast<int*>(stab + (size_t)kSepWaves * 2 * tab_roi);
    odwfx::Scale sc = {0.0f, 0.0f, 1};
    if (BWD) {
        sc = odwfx::scale_of(*absmax_bits);
        if (sc.state != 1) {
            const float fill = sc.state == 0 ? 0.0f : __uint_as_float(0x7fc00000u);
            float* d = dst + ((size_t)b * C + c0) * HW;
            for (int i = threadIdx.x; i < nc * HW; i += blockDim.x) d[i] = fill;
            return;
        }
        for (int i = threadIdx.x; i < nc * HW; i += blockDim.x) plane[i] = 0;
    } else {
        const float* f = src + ((size_t)b * C + c0) * HW;
        for (int i = threadIdx.x; i < nc * HW; i += blockDim.x) plane[i] = (cell_t)f[i];
    }
    if (threadIdx.x == 0) *next_roi = 0;
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* wt = stab + (size_t)wave * 2 * tab_roi;
    const int n4 = tab_roi / 4;                       // <= 128 float4: two per lane
    auto grab = [&]() {
        int n = 0;
        if (lane == 0) n = atomicAdd(next_roi, 1);
        return __builtin_amdgcn_readfirstlane(n);
    };
    float4 pre[2];
    auto fetch = [&](int n) {
        const float4* t4 = reinterpret_cast<const float4*>(tab + (size_t)n * tab_roi);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int i = lane + q * 64;
            pre[q] = i < n4 ? t4[i] : make_float4(0, 0, 0, 0);
        }
    };
    auto stash = [&](int buf) {
        float4* s4 = reinterpret_cast<float4*>(wt + buf * tab_roi);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int i = lane + q * 64;
            if (i < n4) s4[i] = pre[q];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // the wave's own LDS writes, visible to its lanes
        __builtin_amdgcn_wave_barrier();
    };
    int n = grab(), buf = 0;
    if (n < R) { fetch(n); stash(0); }
    while (n < R) {
        const int nn = grab();
        if (nn < R) fetch(nn);                         // in flight while this ROI is processed
        const float* cur = wt + buf * tab_roi;
        const RoiGeom g = roi_geom(rois + (size_t)n * 5, scale, PH, PW, sr);
        if (g.b == b) {
            for (int bin = lane; bin < nb; bin += 64) {
                const int ph = bin / PW, pw = bin - ph * PW;
                const float* ty = cur + ph * kAxisStride;
                const float* tx = cur + (PH + pw) * kAxisStride;
                const int y0 = reinterpret_cast<const int*>(ty)[0], ny = reinterpret_cast<const int*>(ty)[1];
                const int x0 = reinterpret_cast<const int*>(tx)[0], nx = reinterpret_cast<const int*>(tx)[1];
                const size_t o = ((size_t)n * C + c0) * nb + bin;
                if (ny < 0 || nx < 0) {                // a bin wider than an axis vector: the sample-by-sample form
                    for (int cl = 0; cl < nc; ++cl) {
                        cell_t* q = plane + cl * HW;
                        if constexpr (BWD) {
                            const float tf = sc.to_fixed;
                            align_scatter(g, H, W, ph, pw, src[o + (size_t)cl * nb], [q, tf](int pos, float v) { odwfx::add(q + pos, v, tf); });
                        } else {
                            dst[o + (size_t)cl * nb] = align_one(q, g, H, W, ph, pw);
                        }
                    }
                    continue;
                }
                float v[CG];
#pragma unroll
                for (int cl = 0; cl < CG; ++cl) v[cl] = BWD ? (cl < nc ? src[o + (size_t)cl * nb] / g.count : 0.0f) : 0.0f;
                cell_t* p = plane + y0 * W + x0;
                for (int cy = 0; cy < ny; ++cy) {
                    const float wy = ty[2 + cy];
                    for (int cx = 0; cx < nx; ++cx) {
                        const float w = wy * tx[2 + cx];
#pragma unroll
                        for (int cl = 0; cl < CG; ++cl) {
                            if (cl < nc) {
                                if constexpr (BWD) odwfx::add(p + cl * HW + cy * W + cx, v[cl] * w, sc.to_fixed);
                                else v[cl] += w * p[cl * HW + cy * W + cx];
                            }
                        }
                    }
                }
                if (!BWD) {
#pragma unroll
                    for (int cl = 0; cl < CG; ++cl)
                        if (cl < nc) dst[o + (size_t)cl * nb] = (ny == 0 || nx == 0) ? 0.0f : v[cl] / g.count;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();               // every lane is done with slot buf ^ 1's previous content
        if (nn < R) stash(buf ^ 1);
        buf ^= 1;
        n = nn;
    }
    if (BWD) {
        __syncthreads();
        float* d = dst + ((size_t)b * C + c0) * HW;
        for (int i = threadIdx.x; i < nc * HW; i += blockDim.x) d[i] = (float)plane[i] * sc.to_float;
    }
}

// ---- the forward on the (ROI, 64-channel) decomposition of the pooling kernels (round 4) ----------------------------
// The plane-resident forward above keeps 1-4 channel planes in LDS and walks every ROI: a lane = one bin reads its
// (bin + 2)^2 cells one 4-byte LDS word at a time per channel, and writes 196-byte output segments: 0.60 ms at P = 2000 on
// 76 x 76 x 512.  For a fixed ROI the outputs of 64 consecutive channels are ONE contiguous block of `out`, and in an
// NHWC copy of the map a cell's 8 consecutive channels are two 16-byte loads: a workgroup takes (ROI n, channels c0 ..
// c0 + 63), a thread = (bin, 8 channels) forms the same separable sum -- the same axis vectors (roi_align_axis_kernel: the
// sample coordinates in the reference's fp32 operation order), the same cell order, separate multiply and add -- so the
// values are the plane kernel's bit for bit, and the block leaves as full lines.
__global__ __launch_bounds__(256) void nchw_to_nhwc_f32_tile_kernel(const float* __restrict__ in, int C, int HW,
                                                                    float* __restrict__ out) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, c0 = blockIdx.y * 32, p0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float* src = in + (size_t)b * C * HW;
    float* dst = out + (size_t)b * HW * C;
#pragma unroll
    for (int j = 0; j < 32; j += 8) {
        const int c = c0 + ty + j, p = p0 + tx;
        tile[ty + j][tx] = (c < C && p < HW) ? src[(size_t)c * HW + p] : 0.0f;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 32; j += 8) {
        const int p = p0 + ty + j, c = c0 + tx;
        if (p < HW && c < C) dst[(size_t)p * C + c] = tile[tx][ty + j];
    }
}

constexpr int kOpMaxAxes = 32;          // PH + PW of the (ROI, 64-channel) forward

__global__ __launch_bounds__(512) void roi_align_fwd_nhwc_op(const float* __restrict__ nhwc, const float* __restrict__ nchw,
                                                             const float* __restrict__ rois, float scale,
                                                             const float* __restrict__ tab, int C, int H, int W, int PH, int PW,
                                                             int sr, float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) float s_val[64 * 64];
    __shared__ __attribute__((aligned(16))) float s_tab[kOpMaxAxes * kAxisStride];
    const int n = blockIdx.x, c0 = blockIdx.y * 64;
    const int per = PH + PW, nb = PH * PW;
    {
        const float4* t4 = reinterpret_cast<const float4*>(tab + (size_t)n * per * kAxisStride);
        for (int i = threadIdx.x; i < per * kAxisStride / 4; i += blockDim.x) reinterpret_cast<float4*>(s_tab)[i] = t4[i];
    }
    __syncthreads();
    const RoiGeom g = roi_geom(rois + (size_t)n * 5, scale, PH, PW, sr);
    const int bin = threadIdx.x >> 3, cg = threadIdx.x & 7;
    if (bin < nb && c0 + cg * 8 < C) {
        const int ph = bin / PW, pw = bin - ph * PW;
        const float* ty = s_tab + ph * kAxisStride;
        const float* tx = s_tab + (PH + pw) * kAxisStride;
        const int y0 = reinterpret_cast<const int*>(ty)[0], ny = reinterpret_cast<const int*>(ty)[1];
        const int x0 = reinterpret_cast<const int*>(tx)[0], nx = reinterpret_cast<const int*>(tx)[1];
        float v[8];
        if (ny < 0 || nx < 0) {                    // a bin wider than an axis vector (a box far larger than the map): sample form
#pragma unroll
            for (int q = 0; q < 8; ++q)
                v[q] = align_one(nchw + ((size_t)g.b * C + c0 + cg * 8 + q) * H * W, g, H, W, ph, pw);
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = 0.0f;
            const float* base = nhwc + ((size_t)g.b * H * W + (size_t)y0 * W + x0) * C + c0 + cg * 8;
            for (int cy = 0; cy < ny; ++cy) {
                const float wy = ty[2 + cy];
                const float* row = base + (size_t)cy * W * C;
#pragma unroll 2
                for (int cx = 0; cx < nx; ++cx) {
                    const float w = wy * tx[2 + cx];
                    const float4 a = *reinterpret_cast<const float4*>(row + (size_t)cx * C);
                    const float4 b4 = *reinterpret_cast<const float4*>(row + (size_t)cx * C + 4);
                    v[0] += w * a.x; v[1] += w * a.y; v[2] += w * a.z; v[3] += w * a.w;
                    v[4] += w * b4.x; v[5] += w * b4.y; v[6] += w * b4.z; v[7] += w * b4.w;
                }
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = (ny == 0 || nx == 0) ? 0.0f : v[q] / g.count;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) s_val[(cg * 8 + q) * nb + bin] = v[q];
    }
    __syncthreads();
    const int nch = C - c0 < 64 ? C - c0 : 64;
    const int count4 = nch * nb / 4;                  // C % 8 == 0: a multiple of 8 values, 32-byte aligned
    const size_t off = ((size_t)n * C + c0) * nb;
    for (int i = threadIdx.x; i < count4; i += blockDim.x)
        reinterpret_cast<float4*>(out + off)[i] = reinterpret_cast<const float4*>(s_val)[i];
}

// channel planes per workgroup for the chunk-staged kernels: the planes + one chunk of axis vectors must fit in LDS
int pick_cg_sep(int B, int C, int HW, int per, int cell_bytes) {
    const int cands[3] = {4, 2, 1};
    const int64_t tab_bytes = (int64_t)2 * kSepWaves * per * kAxisStride * 4 + 32;
    int fit = 0;
    for (int k = 0; k < 3; ++k) {
        const int cg = cands[k];
        if ((int64_t)cg * HW * cell_bytes + tab_bytes > ODW_LDS_BYTES) continue;
        if (!fit) fit = cg;
        if ((int64_t)B * ((C + cg - 1) / cg) >= ODW_NUM_CU) return cg;
    }
    return fit ? 1 : 0;
}

int pick_cg(int B, int C, int HW) {
    const int cands[3] = {4, 2, 1};
    int fit = 0;
    for (int k = 0; k < 3; ++k) {
        int cg = cands[k];
        if ((int64_t)cg * HW * 4 > ODW_LDS_BYTES) continue;
        if (!fit) fit = cg;
        if ((int64_t)B * ((C + cg - 1) / cg) >= ODW_NUM_CU) return cg;
    }
    return fit ? 1 : 0;
}

template <typename K>
hipError_t allow_lds(K kernel, size_t bytes) {
    return odw_set_max_lds(reinterpret_cast<const void*>(kernel),
                               (int)bytes);
}

}  // namespace

ODW_EXPORT int odw_roi_align_forward(const float* feat, const float* rois, float scale, int B, int C,
                                     int H, int W, int R, int PH, int PW, int sr, float* out,
                                     void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    ODW_REQUIRE(B >= 0 && C > 0 && H > 0 && W > 0 && PH > 0 && PW > 0 && R >= 0,
                "roi_align_forward: bad dims");
    if (R == 0 || B == 0) return ODW_OK;
    ODW_REQUIRE(feat && rois && out, "roi_align_forward: null pointer");
    ODW_REQUIRE(PH * PW <= kPlaneThreads / 4, "roi_align_forward: pooled size too large");
    const int HW = H * W;
    const int cg = pick_cg(B, C, HW);
    if (cg == 0) {
        size_t total = (size_t)R * C * PH * PW;
        int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
        roi_align_fwd_direct<<<grid, 256, 0, stream>>>(feat, rois, scale, C, H, W, R, PH, PW, sr, out);
        ODW_CHECK_LAUNCH("roi_align_fwd_direct");
        return ODW_OK;
    }
    const int grid = B * ((C + cg - 1) / cg);
    const size_t lds = (size_t)cg * HW * 4;
    switch (cg) {
        case 4:
            ODW_CHECK_HIP(allow_lds(roi_align_fwd_plane<4>, lds), "roi_align_fwd_plane attr");
            roi_align_fwd_plane<4><<<grid, kPlaneThreads, lds, stream>>>(feat, rois, scale, C, H, W, R, PH, PW, sr, out);
            break;
        case 2:
            ODW_CHECK_HIP(allow_lds(roi_align_fwd_plane<2>, lds), "roi_align_fwd_plane attr");
            roi_align_fwd_plane<2><<<grid, kPlaneThreads, lds, stream>>>(feat, rois, scale, C, H, W, R, PH, PW, sr, out);
            break;
        default:
            ODW_CHECK_HIP(allow_lds(roi_align_fwd_plane<1>, lds), "roi_align_fwd_plane attr");
            roi_align_fwd_plane<1><<<grid, kPlaneThreads, lds, stream>>>(feat, rois, scale, C, H, W, R, PH, PW, sr, out);
            break;
    }
    ODW_CHECK_LAUNCH("roi_align_fwd_plane");
    return ODW_OK;
}

namespace {
template <bool BWD>
int launch_sep(const float* src, const float* rois, float scale, float* tab, int B, int C, int H, int W, int R, int PH,
               int PW, int sr, float* dst, int cg, hipStream_t stream) {
    static_assert(kAxisStride % 4 == 0, "axis vectors are copied as float4");
    ODW_REQUIRE((PH + PW) * kAxisStride <= 512, "roi_align: pooled size too large for the staged form");
    const int items = R * (PH + PW);
    roi_align_axis_kernel<<<(items + 255) / 256, 256, 0, stream>>>(rois, scale, R, PH, PW, H, W, sr, tab);
    ODW_CHECK_LAUNCH("roi_align_axis_kernel");
    // the launch's fixed-point scale lives right behind the axis vectors (backward only)
    unsigned* mx = reinterpret_cast<unsigned*>(tab + (size_t)R * (PH + PW) * kAxisStride);
    if (BWD) {
        ODW_CHECK_HIP(hipMemsetAsync(mx, 0, 4, stream), "roi_align memset");
        odwfx::absmax_kernel<false><<<1024, 256, 0, stream>>>(src, (size_t)R * C * PH * PW, mx);
        ODW_CHECK_LAUNCH("absmax_kernel");
    }
    const int grid = B * ((C + cg - 1) / cg);
    const size_t lds = (((size_t)cg * H * W * (BWD ? 8 : 4) + 15) & ~(size_t)15) + (size_t)2 * kSepWaves * (PH + PW) * kAxisStride * 4 + 16;
    switch (cg) {
        case 4:
            ODW_CHECK_HIP(allow_lds(roi_align_sep_plane<4, BWD>, lds), "roi_align_sep_plane attr");
            roi_align_sep_plane<4, BWD><<<grid, kPlaneThreads, lds, stream>>>(src, rois, scale, tab, mx, C, H, W, R, PH, PW, sr, dst);
            break;
        case 2:
            ODW_CHECK_HIP(allow_lds(roi_align_sep_plane<2, BWD>, lds), "roi_align_sep_plane attr");
            roi_align_sep_plane<2, BWD><<<grid, kPlaneThreads, lds, stream>>>(src, rois, scale, tab, mx, C, H, W, R, PH, PW, sr, dst);
            break;
        default:
            ODW_CHECK_HIP(allow_lds(roi_align_sep_plane<1, BWD>, lds), "roi_align_sep_plane attr");
            roi_align_sep_plane<1, BWD><<<grid, kPlaneThreads, lds, stream>>>(src, rois, scale, tab, mx, C, H, W, R, PH, PW, sr, dst);
            break;
    }
    ODW_CHECK_LAUNCH("roi_align_sep_plane");
    return ODW_OK;
}
}  // namespace

ODW_EXPORT int64_t odw_roi_align_backward_workspace(int R, int PH, int PW);

// Workspace of the (ROI, 64-channel) forward: the axis vectors + an NHWC copy of the map (C % 8 == 0, PH * PW <= 64,
// PH + PW <= 32).  With odw_roi_align_backward_workspace bytes only, the plane-resident form runs.
ODW_EXPORT int64_t odw_roi_align_forward_workspace(int B, int C, int H, int W, int R, int PH, int PW) {
    return odw_roi_align_backward_workspace(R, PH, PW) + odw_align_up((int64_t)(B > 0 ? B : 1) * C * H * W * 4, 256);
}

ODW_EXPORT int odw_roi_align_forward_ws(const float* feat, const float* rois, float scale, int B, int C, int H, int W,
                                        int R, int PH, int PW, int sr, float* out, void* workspace,
                                        int64_t workspace_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    ODW_REQUIRE(B >= 0 && C > 0 && H > 0 && W > 0 && PH > 0 && PW > 0 && R >= 0, "roi_align_forward: bad dims");
    if (R == 0 || B == 0) return ODW_OK;
    ODW_REQUIRE(feat && rois && out, "roi_align_forward: null pointer");
    static const bool no_nhwc = getenv("ODW_ROI_ALIGN_PLANE") != nullptr;          // comparison runs: the plane-resident form
    if (!no_nhwc && !getenv("ODW_ROI_ALIGN_SAMPLES") && workspace && (((uintptr_t)workspace) & 15) == 0 && C % 8 == 0 &&
        PH * PW <= 64 && PH + PW <= kOpMaxAxes && (((uintptr_t)out) & 15) == 0 && (long long)B * H * W * C < (1ll << 31) &&
        workspace_bytes >= odw_roi_align_forward_workspace(B, C, H, W, R, PH, PW)) {
        float* tab = (float*)workspace;
        float* nhwc = (float*)((char*)workspace + odw_roi_align_backward_workspace(R, PH, PW));
        const int items = R * (PH + PW);
        roi_align_axis_kernel<<<(items + 255) / 256, 256, 0, stream>>>(rois, scale, R, PH, PW, H, W, sr, tab);
        ODW_CHECK_LAUNCH("roi_align_axis_kernel");
        nchw_to_nhwc_f32_tile_kernel<<<dim3((H * W + 31) / 32, (C + 31) / 32, B), 256, 0, stream>>>(feat, C, H * W, nhwc);
        ODW_CHECK_LAUNCH("nchw_to_nhwc_f32_tile_kernel");
        roi_align_fwd_nhwc_op<<<dim3(R, (C + 63) / 64), 512, 0, stream>>>(nhwc, feat, rois, scale, tab, C, H, W, PH, PW, sr, out);
        ODW_CHECK_LAUNCH("roi_align_fwd_nhwc_op");
        return ODW_OK;
    }
    const int cg = pick_cg_sep(B, C, H * W, PH + PW, 4);
    if (cg == 0 || !workspace || workspace_bytes < odw_roi_align_backward_workspace(R, PH, PW) ||
        (((uintptr_t)workspace) & 15) != 0 || PH * PW > kPlaneThreads || getenv("ODW_ROI_ALIGN_SAMPLES"))
        return odw_roi_align_forward(feat, rois, scale, B, C, H, W, R, PH, PW, sr, out, stream_);
    return launch_sep<false>(feat, rois, scale, (float*)workspace, B, C, H, W, R, PH, PW, sr, out, cg, stream);
}

ODW_EXPORT int64_t odw_roi_align_backward_workspace(int R, int PH, int PW) {
    return R > 0 ? odw_align_up((int64_t)R * (PH + PW) * kAxisStride * 4 + 16, 256) : 0;      // axis vectors + the scale word
}

ODW_EXPORT int odw_roi_align_backward(const float* grad_out, const float* rois, float scale, int B, int C,
                                      int H, int W, int R, int PH, int PW, int sr, float* grad_in,
                                      void* stream_) {
    return odw_roi_align_backward_ws(grad_out, rois, scale, B, C, H, W, R, PH, PW, sr, grad_in, nullptr, 0, stream_);
}

ODW_EXPORT int odw_roi_align_backward_ws(const float* grad_out, const float* rois, float scale, int B, int C,
                                         int H, int W, int R, int PH, int PW, int sr, float* grad_in,
                                         void* workspace, int64_t workspace_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    ODW_REQUIRE(B >= 0 && C > 0 && H > 0 && W > 0 && PH > 0 && PW > 0 && R >= 0,
                "roi_align_backward: bad dims");
    if (B == 0) return ODW_OK;
    ODW_REQUIRE(grad_in, "roi_align_backward: null grad_in");
    const size_t in_bytes = (size_t)B * C * H * W * 4;
    if (R == 0) {
        ODW_CHECK_HIP(hipMemsetAsync(grad_in, 0, in_bytes, stream), "roi_align_backward memset");
        return ODW_OK;
    }
    ODW_REQUIRE(grad_out && rois, "roi_align_backward: null pointer");
    ODW_REQUIRE(PH * PW <= kPlaneThreads / 4, "roi_align_backward: pooled size too large");
    const int HW = H * W;
    const int cg = pick_cg(B, C, HW);
    if (cg == 0) {
        ODW_CHECK_HIP(hipMemsetAsync(grad_in, 0, in_bytes, stream), "roi_align_backward memset");
        size_t total = (size_t)R * C * PH * PW;
        int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
        roi_align_bwd_direct<<<grid, 256, 0, stream>>>(grad_out, rois, scale, C, H, W, R, PH, PW, sr, grad_in);
        ODW_CHECK_LAUNCH("roi_align_bwd_direct");
        return ODW_OK;
    }
    const int grid = B * ((C + cg - 1) / cg);
    const size_t lds = (size_t)cg * HW * 4;
    // separable form when the caller brought the table space and no bin can outgrow an axis vector
    if (workspace && workspace_bytes >= odw_roi_align_backward_workspace(R, PH, PW) &&
        (((uintptr_t)workspace) & 15) == 0 && !getenv("ODW_ROI_ALIGN_SAMPLES")) {
        float* tab = (float*)workspace;
        const int cgs = pick_cg_sep(B, C, HW, PH + PW, 8);
        ODW_REQUIRE((((uintptr_t)grad_out) & 15) == 0, "roi_align_backward: grad_out must be 16-byte aligned");
        if (cgs > 0 && !getenv("ODW_ROI_ALIGN_GLOBAL_TAB"))
            return launch_sep<true>(grad_out, rois, scale, tab, B, C, H, W, R, PH, PW, sr, grad_in, cgs, stream);
        const int items = R * (PH + PW);
        roi_align_axis_kernel<<<(items + 255) / 256, 256, 0, stream>>>(rois, scale, R, PH, PW, H, W, sr, tab);
        ODW_CHECK_LAUNCH("roi_align_axis_kernel");
        switch (cg) {
            case 4:
                ODW_CHECK_HIP(allow_lds(roi_align_bwd_sep_plane<4>, lds), "roi_align_bwd_sep_plane attr");
                roi_align_bwd_sep_plane<4><<<grid, kPlaneThreads, lds, stream>>>(grad_out, rois, scale, tab, C, H, W, R, PH, PW, sr, grad_in);
                break;
            case 2:
                ODW_CHECK_HIP(allow_lds(roi_align_bwd_sep_plane<2>, lds), "roi_align_bwd_sep_plane attr");
                roi_align_bwd_sep_plane<2><<<grid, kPlaneThreads, lds, stream>>>(grad_out, rois, scale, tab, C, H, W, R, PH, PW, sr, grad_in);
                break;
            default:
                ODW_CHECK_HIP(allow_lds(roi_align_bwd_sep_plane<1>, lds), "roi_align_bwd_sep_plane attr");
                roi_align_bwd_sep_plane<1><<<grid, kPlaneThreads, lds, stream>>>(grad_out, rois, scale, tab, C, H, W, R, PH, PW, sr, grad_in);
                break;
        }
        ODW_CHECK_LAUNCH("roi_align_bwd_sep_plane");
        return ODW_OK;
    }
    switch (cg) {
        case 4:
            ODW_CHECK_HIP(allow_lds(roi_align_bwd_plane<4>, lds), "roi_align_bwd_plane attr");
            roi_align_bwd_plane<4><<<grid, kPlaneThreads, lds, stream>>>(grad_out, rois, scale, C, H, W, R, PH, PW, sr, grad_in);
            break;
        case 2:
            ODW_CHECK_HIP(allow_lds(roi_align_bwd_plane<2>, lds), "roi_align_bwd_plane attr");
            roi_align_bwd_plane<2><<<grid, kPlaneThreads, lds, stream>>>(grad_out, rois, scale, C, H, W, R, PH, PW, sr, grad_in);
            break;
        default:
            ODW_CHECK_HIP(allow_lds(roi_align_bwd_plane<1>, lds), "roi_align_bwd_plane attr");
            roi_align_bwd_plane<1><<<grid, kPlaneThreads, lds, stream>>>(grad_out, rois, scale, C, H, W, R, PH, PW, sr, grad_in);
            break;
    }
    ODW_CHECK_LAUNCH("roi_align_bwd_plane");
    return ODW_OK;
}
