// preprocess.hip -- the pixel half of the data boundary, fused on the GPU.
// Reference: data/transforms/transforms.py:33-150 (Resize -> RandomHorizontalFlip / RandomVerticalFlip -> ToTensor ->
// Lighting -> Normalize, applied to a PIL image in a CPU worker) + structures/image_list.py:33-76 (zero padding of every
// image into the batch tensor).  Here the host hands over the decoded uint8 HWC image (4x fewer bytes over PCIe than the
// normalised fp32 tensor) and the device does the rest, landing the pixels straight in their slot of the padded batch.
//
// The resize is PIL's (torchvision 0.8.2 F.resize on a PIL image == Image.resize(BILINEAR); Pillow's
// libImaging/Resample.c, a third-party dependency of the reference): separable, support scaled by the down-sampling
// factor (anti-aliasing), double-precision coefficients normalised and rounded to 22-bit fixed point, horizontal pass
// rounded to uint8 before the vertical pass.  All of it is integer arithmetic on the pixel side, so the result is
// bit-identical to Pillow's; the fp32 tail (x/255, + lighting, *255, - mean, / std) is the same sequence of correctly
// rounded fp32 operations torch runs.
#include "odw_common.h"
#include <math.h>

namespace {

constexpr int kPrecisionBits = 32 - 8 - 2;   // Resample.c PRECISION_BITS

struct Axis {
    int in_size, out_size, ksize;
    int* bounds;   // (out_size, 2): first tap, tap count
    int* kk;       // (out_size, ksize) fixed-point weights
};

__host__ __device__ inline int axis_ksize(int in_size, int out_size) {
    double filterscale = (double)((float)in_size - 0.0f) / out_size;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = 1.0 * filterscale;            // BILINEAR support = 1
    return (int)ceil(support) * 2 + 1;
}

// precompute_coeffs + normalize_coeffs_8bpc, one thread per output coordinate; blockIdx.y = axis
__global__ __launch_bounds__(256) void resample_coeffs_kernel(Axis ax0, Axis ax1) {
    const Axis ax = blockIdx.y == 0 ? ax0 : ax1;
    const int xx = blockIdx.x * blockDim.x + threadIdx.x;
    if (xx >= ax.out_size) return;
    double filterscale, scale;
    filterscale = scale = (double)((float)ax.in_size - 0.0f) / ax.out_size;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = 1.0 * filterscale;
    const double center = 0.0 + (xx + 0.5) * scale;
    const double ss = 1.0 / filterscale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > ax.in_size) xmax = ax.in_size;
    xmax -= xmin;
    double ww = 0.0;
    for (int x = 0; x < xmax; ++x) {
        double t = (x + xmin - center + 0.5) * ss;
        if (t < 0.0) t = -t;
        const double w = t < 1.0 ? 1.0 - t : 0.0;
        ww += w;
    }
    int* k = ax.kk + (size_t)xx * ax.ksize;
    for (int x = 0; x < ax.ksize; ++x) {
        double w = 0.0;
        if (x < xmax) {
            double t = (x + xmin - center + 0.5) * ss;
            if (t < 0.0) t = -t;
            w = t < 1.0 ? 1.0 - t : 0.0;
            if (ww != 0.0) w /= ww;
        }
        k[x] = w < 0 ? (int)(-0.5 + w * (1 << kPrecisionBits)) : (int)(0.5 + w * (1 << kPrecisionBits));
    }
    ax.bounds[2 * xx] = xmin;
    ax.bounds[2 * xx + 1] = xmax;
}

__device__ inline int clip8(int v) {
    v >>= kPrecisionBits;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// horizontal pass: src (in_h, in_w, 3) -> tmp (in_h, out_w, 3), uint8
__global__ __launch_bounds__(256) void resample_h_kernel(const uint8_t* __restrict__ src, int in_h, int in_w, Axis ax,
                                                         uint8_t* __restrict__ tmp) {
    const int xx = blockIdx.x * blockDim.x + threadIdx.x;
    const int yy = blockIdx.y;
    if (xx >= ax.out_size) return;
    const int xmin = ax.bounds[2 * xx], n = ax.bounds[2 * xx + 1];
    const int* k = ax.kk + (size_t)xx * ax.ksize;
    const uint8_t* row = src + ((size_t)yy * in_w + xmin) * 3;
    int s0 = 1 << (kPrecisionBits - 1), s1 = s0, s2 = s0;
    for (int x = 0; x < n; ++x) {
        const int w = k[x];
        s0 += row[3 * x + 0] * w;
        s1 += row[3 * x + 1] * w;
        s2 += row[3 * x + 2] * w;
    }
    uint8_t* o = tmp + ((size_t)yy * ax.out_size + xx) * 3;
    o[0] = (uint8_t)clip8(s0); o[1] = (uint8_t)clip8(s1); o[2] = (uint8_t)clip8(s2);
}

struct Finish {
    float light[3];   // Lighting offset per RGB channel (added after /255), zeros when disabled
    float mean[3], std[3];
    int use_light, to_bgr255, hflip, vflip;
};

// vertical pass + flips + ToTensor + Lighting + Normalize + zero padding: one thread per pixel of the padded plane
__global__ __launch_bounds__(256) void resample_v_finish_kernel(const uint8_t* __restrict__ img, int img_w, Axis ay,
                                                                int need_v, int out_h, int out_w, Finish f,
                                                                float* __restrict__ out, int Hp, int Wp) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= Wp) return;
    const size_t plane = (size_t)Hp * Wp;
    float* o = out + (size_t)y * Wp + x;
    if (y >= out_h || x >= out_w) {
        o[0] = 0.0f; o[plane] = 0.0f; o[2 * plane] = 0.0f;
        return;
    }
    const int sx = f.hflip ? out_w - 1 - x : x;     // F.hflip / F.vflip act on the resized image
    const int sy = f.vflip ? out_h - 1 - y : y;
    int v[3];
    if (need_v) {
        const int ymin = ay.bounds[2 * sy], n = ay.bounds[2 * sy + 1];
        const int* k = ay.kk + (size_t)sy * ay.ksize;
        int s0 = 1 << (kPrecisionBits - 1), s1 = s0, s2 = s0;
        for (int t = 0; t < n; ++t) {
            const uint8_t* p = img + ((size_t)(ymin + t) * img_w + sx) * 3;
            const int w = k[t];
            s0 += p[0] * w; s1 += p[1] * w; s2 += p[2] * w;
        }
        v[0] = clip8(s0); v[1] = clip8(s1); v[2] = clip8(s2);
    } else {
        const uint8_t* p = img + ((size_t)sy * img_w + sx) * 3;
        v[0] = p[0]; v[1] = p[1]; v[2] = p[2];
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int c = f.to_bgr255 ? 2 - j : j;       // image[[2, 1, 0]]
        float t = (float)v[c] / 255.0f;              // F.to_tensor
        if (f.use_light) t = t + f.light[c];         // Lighting: img.add(rgb)
        if (f.to_bgr255) t = t * 255.0f;
        t = (t - f.mean[j]) / f.std[j];              // F.normalize: sub_(mean).div_(std)
        o[j * plane] = t;
    }
}

struct Layout { int64_t bh, kh, bv, kv, tmp, total; int ksh, ksv; };

Layout layout(int in_h, int in_w, int out_h, int out_w) {
    Layout L;
    L.ksh = axis_ksize(in_w, out_w);
    L.ksv = axis_ksize(in_h, out_h);
    int64_t off = 0;
    L.bh = off; off = odw_align_up(off + (int64_t)out_w * 2 * 4, 256);
    L.kh = off; off = odw_align_up(off + (int64_t)out_w * L.ksh * 4, 256);
    L.bv = off; off = odw_align_up(off + (int64_t)out_h * 2 * 4, 256);
    L.kv = off; off = odw_align_up(off + (int64_t)out_h * L.ksv * 4, 256);
    L.tmp = off; off = odw_align_up(off + (int64_t)in_h * out_w * 3, 256);
    L.total = off;
    return L;
}

}  // namespace

ODW_EXPORT int64_t odw_image_preprocess_workspace(int in_h, int in_w, int out_h, int out_w) {
    if (in_h <= 0 || in_w <= 0 || out_h <= 0 || out_w <= 0) return 0;
    return layout(in_h, in_w, out_h, out_w).total;
}

ODW_EXPORT int odw_image_preprocess(const uint8_t* rgb, int in_h, int in_w, int out_h, int out_w, int hflip, int vflip,
                                    const float* lighting_rgb, const float* mean, const float* std_, int to_bgr255,
                                    float* out, int Hp, int Wp, void* workspace, int64_t workspace_bytes,
                                    void* stream_) {
    ODW_REQUIRE(in_h > 0 && in_w > 0 && out_h > 0 && out_w > 0, "image_preprocess: empty image %dx%d -> %dx%d", in_h,
                in_w, out_h, out_w);
    ODW_REQUIRE(in_h < 65536 && in_w < (1 << 24) && Hp < 65536, "image_preprocess: image too large");
    ODW_REQUIRE(Hp >= out_h && Wp >= out_w, "image_preprocess: padded plane %dx%d smaller than the image %dx%d", Hp, Wp,
                out_h, out_w);
    ODW_REQUIRE(rgb && mean && std_ && out, "image_preprocess: null pointer");
    const Layout L = layout(in_h, in_w, out_h, out_w);
    ODW_REQUIRE(workspace && workspace_bytes >= L.total, "image_preprocess: workspace %lld < %lld bytes",
                (long long)workspace_bytes, (long long)L.total);
    hipStream_t stream = (hipStream_t)stream_;
    char* ws = (char*)workspace;
    Axis ah{in_w, out_w, L.ksh, (int*)(ws + L.bh), (int*)(ws + L.kh)};
    Axis av{in_h, out_h, L.ksv, (int*)(ws + L.bv), (int*)(ws + L.kv)};
    const int need_h = out_w != in_w, need_v = out_h != in_h;     // Resample.c ImagingResample: box == whole image
    const int longest = out_w > out_h ? out_w : out_h;
    if (need_h || need_v) {
        resample_coeffs_kernel<<<dim3((longest + 255) / 256, 2), 256, 0, stream>>>(ah, av);
        ODW_CHECK_LAUNCH("resample_coeffs_kernel");
    }
    const uint8_t* img = rgb;
    int img_w = in_w;
    if (need_h) {
        // Pillow only resamples the rows the vertical pass reads; the rows it skips are never read here either
        resample_h_kernel<<<dim3((out_w + 255) / 256, in_h), 256, 0, stream>>>(rgb, in_h, in_w, ah,
                                                                                (uint8_t*)(ws + L.tmp));
        ODW_CHECK_LAUNCH("resample_h_kernel");
        img = (const uint8_t*)(ws + L.tmp);
        img_w = out_w;
    }
    Finish f;
    for (int i = 0; i < 3; ++i) {
        f.light[i] = lighting_rgb ? lighting_rgb[i] : 0.0f;
        f.mean[i] = mean[i];
        f.std[i] = std_[i];
    }
    f.use_light = lighting_rgb != nullptr;
    f.to_bgr255 = to_bgr255;
    f.hflip = hflip;
    f.vflip = vflip;
    resample_v_finish_kernel<<<dim3((Wp + 255) / 256, Hp), 256, 0, stream>>>(img, img_w, av, need_v, out_h, out_w, f, out,
                                                                             Hp, Wp);
    ODW_CHECK_LAUNCH("resample_v_finish_kernel");
    return ODW_OK;
}
