/*
 * odw_oracle.c -- CPU restatement of the native operators on OD-WSCL's
 * proposal-feature hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / timed CPU baseline.
 *
 * Every function cites the reference file:line whose behaviour it restates
 * (paths relative to /root/reference/wetectron).  Plain C99, scalar,
 * single-threaded, written from the behavioural description in SURVEY.md --
 * not copied.  Built with -ffp-contract=off so that no mul+add is fused: the
 * reference CUDA/CPU kernels evaluate these expressions as separate fp32 ops.
 *
 * Pinning: roi_align_fwd and nms_wt(ge=1) are checked against the reference's
 * own compiled csrc/cpu sources (oracle/_ref, built by oracle/build_ref.py);
 * roi_pool_* and roi_align_bwd restate CUDA-only reference kernels that cannot
 * be compiled here (THC headers) and are pinned through the Python reference's
 * end-to-end golden vectors (tests/golden/) in which they are the injected
 * `_C.roi_pool_*`; nms_tv restates torchvision 0.8.2 (absent from the
 * reference tree): parity unpinned by the reference for that function alone.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ODW_API __attribute__((visibility("default")))

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

/* ------------------------------------------------------------------------ */
/* ROIPool forward: csrc/cuda/ROIPool_cuda.cu:17-77                          */
/* rois (R,5) = [batch, x1, y1, x2, y2]; out/argmax (R,C,PH,PW).             */
ODW_API void oracle_roi_pool_fwd(const float* feat, const float* rois, float scale,
                                 int B, int C, int H, int W, int R, int PH, int PW,
                                 float* out, int32_t* argmax) {
    (void)B;
    for (int n = 0; n < R; ++n) {
        const float* roi = rois + (size_t)n * 5;
        int b = (int)roi[0];
        /* C round(): half away from zero (ROIPool_cuda.cu:30-33) */
        int sw = (int)roundf(roi[1] * scale);
        int sh = (int)roundf(roi[2] * scale);
        int ew = (int)roundf(roi[3] * scale);
        int eh = (int)roundf(roi[4] * scale);
        int rw = imax(ew - sw + 1, 1);
        int rh = imax(eh - sh + 1, 1);
        float bin_h = (float)rh / (float)PH;
        float bin_w = (float)rw / (float)PW;
        for (int c = 0; c < C; ++c) {
            const float* plane = feat + ((size_t)b * C + c) * H * W;
            for (int ph = 0; ph < PH; ++ph) {
                int hs = (int)floorf((float)ph * bin_h);
                int he = (int)ceilf((float)(ph + 1) * bin_h);
                hs = imin(imax(hs + sh, 0), H);
                he = imin(imax(he + sh, 0), H);
                for (int pw = 0; pw < PW; ++pw) {
                    int ws = (int)floorf((float)pw * bin_w);
                    int we = (int)ceilf((float)(pw + 1) * bin_w);
                    ws = imin(imax(ws + sw, 0), W);
                    we = imin(imax(we + sw, 0), W);
                    int empty = (he <= hs) || (we <= ws);
                    float best = empty ? 0.0f : -FLT_MAX;
                    int besti = -1;
                    for (int h = hs; h < he; ++h)
                        for (int w = ws; w < we; ++w) {
                            float v = plane[h * W + w];
                            if (v > best) { best = v; besti = h * W + w; } /* strict >, first max */
                        }
                    size_t o = (((size_t)n * C + c) * PH + ph) * PW + pw;
                    out[o] = best;
                    argmax[o] = besti;
                }
            }
        }
    }
}

/* ROIPool backward: csrc/cuda/ROIPool_cuda.cu:80-108 (scatter-add by argmax).
 * The reference uses atomicAdd (order undefined); here the adds happen in
 * ascending output-index order.  grad_in (B,C,H,W) is zeroed first
 * (ROIPool_cuda.cu:172). */
ODW_API void oracle_roi_pool_bwd(const float* grad_out, const int32_t* argmax, const float* rois,
                                 int B, int C, int H, int W, int R, int PH, int PW,
                                 float* grad_in) {
    memset(grad_in, 0, sizeof(float) * (size_t)B * C * H * W);
    for (int n = 0; n < R; ++n) {
        int b = (int)rois[(size_t)n * 5];
        for (int c = 0; c < C; ++c) {
            float* plane = grad_in + ((size_t)b * C + c) * H * W;
            size_t base = ((size_t)n * C + c) * PH * PW;
            for (int k = 0; k < PH * PW; ++k) {
                int a = argmax[base + k];
                if (a != -1) plane[a] += grad_out[base + k];
            }
        }
    }
}

/* ------------------------------------------------------------------------ */
/* ROIAlign: one bilinear sample's 4 taps.  csrc/cpu/ROIAlign_cpu.cpp:46-105
 * == csrc/cuda/ROIAlign_cuda.cu:16-62,126-175.  Returns 0 when the sample is
 * outside [-1,H] x [-1,W] (contributes nothing). */
static int bilinear_taps(int H, int W, float y, float x, int pos[4], float wgt[4]) {
    if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) return 0;
    if (y <= 0) y = 0;
    if (x <= 0) x = 0;
    int yl = (int)y, xl = (int)x, yh, xh;
    if (yl >= H - 1) { yh = yl = H - 1; y = (float)yl; } else { yh = yl + 1; }
    if (xl >= W - 1) { xh = xl = W - 1; x = (float)xl; } else { xh = xl + 1; }
    float ly = y - (float)yl, lx = x - (float)xl;
    float hy = 1.0f - ly, hx = 1.0f - lx;
    pos[0] = yl * W + xl; pos[1] = yl * W + xh; pos[2] = yh * W + xl; pos[3] = yh * W + xh;
    wgt[0] = hy * hx; wgt[1] = hy * lx; wgt[2] = ly * hx; wgt[3] = ly * lx;
    return 1;
}

/* ROIAlign forward: csrc/cpu/ROIAlign_cpu.cpp:114-219 (legacy, un-aligned:
 * no -0.5 offset, no rounding, roi extent clamped to >= 1). */
ODW_API void oracle_roi_align_fwd(const float* feat, const float* rois, float scale,
                                  int B, int C, int H, int W, int R, int PH, int PW,
                                  int sampling_ratio, float* out) {
    (void)B;
    for (int n = 0; n < R; ++n) {
        const float* roi = rois + (size_t)n * 5;
        int b = (int)roi[0];
        float sw = roi[1] * scale, sh = roi[2] * scale;
        float ew = roi[3] * scale, eh = roi[4] * scale;
        float rw = fmaxf(ew - sw, 1.0f), rh = fmaxf(eh - sh, 1.0f);
        float bin_h = rh / (float)PH, bin_w = rw / (float)PW;
        int gh = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / (float)PH);
        int gw = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / (float)PW);
        float count = (float)(gh * gw);
        for (int c = 0; c < C; ++c) {
            const float* plane = feat + ((size_t)b * C + c) * H * W;
            for (int ph = 0; ph < PH; ++ph)
                for (int pw = 0; pw < PW; ++pw) {
                    float acc = 0.0f;
                    for (int iy = 0; iy < gh; ++iy) {
                        float y = sh + (float)ph * bin_h + ((float)iy + 0.5f) * bin_h / (float)gh;
                        for (int ix = 0; ix < gw; ++ix) {
                            float x = sw + (float)pw * bin_w + ((float)ix + 0.5f) * bin_w / (float)gw;
                            int pos[4]; float wg[4];
                            if (!bilinear_taps(H, W, y, x, pos, wg)) continue;
                            /* same association as ROIAlign_cpu.cpp:197-200 */
                            acc += wg[0] * plane[pos[0]] + wg[1] * plane[pos[1]] +
                                   wg[2] * plane[pos[2]] + wg[3] * plane[pos[3]];
                        }
                    }
                    out[(((size_t)n * C + c) * PH + ph) * PW + pw] = acc / count;
                }
        }
    }
}

/* ROIAlign backward: csrc/cuda/ROIAlign_cuda.cu:178-254 (CUDA only in the
 * reference).  g_k = grad * w_k / count, scatter-added to the 4 taps. */
ODW_API void oracle_roi_align_bwd(const float* grad_out, const float* rois, float scale,
                                  int B, int C, int H, int W, int R, int PH, int PW,
                                  int sampling_ratio, float* grad_in) {
    memset(grad_in, 0, sizeof(float) * (size_t)B * C * H * W);
    for (int n = 0; n < R; ++n) {
        const float* roi = rois + (size_t)n * 5;
        int b = (int)roi[0];
        float sw = roi[1] * scale, sh = roi[2] * scale;
        float ew = roi[3] * scale, eh = roi[4] * scale;
        float rw = fmaxf(ew - sw, 1.0f), rh = fmaxf(eh - sh, 1.0f);
        float bin_h = rh / (float)PH, bin_w = rw / (float)PW;
        int gh = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / (float)PH);
        int gw = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / (float)PW);
        float count = (float)(gh * gw);
        for (int c = 0; c < C; ++c) {
            float* plane = grad_in + ((size_t)b * C + c) * H * W;
            for (int ph = 0; ph < PH; ++ph)
                for (int pw = 0; pw < PW; ++pw) {
                    float g = grad_out[(((size_t)n * C + c) * PH + ph) * PW + pw];
                    for (int iy = 0; iy < gh; ++iy) {
                        float y = sh + (float)ph * bin_h + ((float)iy + 0.5f) * bin_h / (float)gh;
                        for (int ix = 0; ix < gw; ++ix) {
                            float x = sw + (float)pw * bin_w + ((float)ix + 0.5f) * bin_w / (float)gw;
                            int pos[4]; float wg[4];
                            if (!bilinear_taps(H, W, y, x, pos, wg)) continue;
                            for (int k = 0; k < 4; ++k) plane[pos[k]] += g * wg[k] / count;
                        }
                    }
                }
        }
    }
}

/* ------------------------------------------------------------------------ */
/* Stable descending argsort of scores (ties: lower index first).            */
typedef struct { float s; int i; } sidx_t;
static int cmp_desc(const void* a, const void* b) {
    const sidx_t* x = (const sidx_t*)a; const sidx_t* y = (const sidx_t*)b;
    if (x->s > y->s) return -1;
    if (x->s < y->s) return 1;
    return x->i - y->i;
}
static int* order_desc(const float* scores, int n) {
    sidx_t* t = (sidx_t*)malloc(sizeof(sidx_t) * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; ++i) { t[i].s = scores[i]; t[i].i = i; }
    qsort(t, (size_t)n, sizeof(sidx_t), cmp_desc);
    int* o = (int*)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; ++i) o[i] = t[i].i;
    free(t);
    return o;
}

/* wetectron `_C.nms`: csrc/cpu/nms_cpu.cpp:6-65 (use_ge=1: suppress when
 * ovr >= thr, nms_cpu.cpp:60) and csrc/cuda/nms.cu:13-67 (use_ge=0: ovr > thr,
 * nms.cu:60).  +1 pixel areas.  Returns kept ORIGINAL indices ascending
 * (nms_cpu.cpp:64, nms.cu:127-130).  keep must hold n entries. */
ODW_API int oracle_nms_wt(const float* boxes, const float* scores, int n, float thr, int use_ge,
                          int64_t* keep) {
    if (n == 0) return 0;
    int* ord = order_desc(scores, n);
    uint8_t* dead = (uint8_t*)calloc((size_t)n, 1);
    for (int a = 0; a < n; ++a) {
        int i = ord[a];
        if (dead[i]) continue;
        const float* bi = boxes + (size_t)i * 4;
        float ai = (bi[2] - bi[0] + 1) * (bi[3] - bi[1] + 1);
        for (int c = a + 1; c < n; ++c) {
            int j = ord[c];
            if (dead[j]) continue;
            const float* bj = boxes + (size_t)j * 4;
            float aj = (bj[2] - bj[0] + 1) * (bj[3] - bj[1] + 1);
            float w = fmaxf(0.0f, fminf(bi[2], bj[2]) - fmaxf(bi[0], bj[0]) + 1);
            float h = fmaxf(0.0f, fminf(bi[3], bj[3]) - fmaxf(bi[1], bj[1]) + 1);
            float inter = w * h;
            float ovr = inter / (ai + aj - inter);
            if (use_ge ? (ovr >= thr) : (ovr > thr)) dead[j] = 1;
        }
    }
    int k = 0;
    for (int i = 0; i < n; ++i) if (!dead[i]) keep[k++] = i;
    free(ord); free(dead);
    return k;
}

/* torchvision.ops.nms (third-party, torchvision==0.8.2 per README.md:32; call
 * sites structures/boxlist_ops.py:32,57).  Published semantics restated: IoU
 * without +1, suppress when IoU > thr, kept indices in descending-score order. */
ODW_API int oracle_nms_tv(const float* boxes, const float* scores, int n, float thr, int64_t* keep) {
    if (n == 0) return 0;
    int* ord = order_desc(scores, n);
    uint8_t* dead = (uint8_t*)calloc((size_t)n, 1);
    int k = 0;
    for (int a = 0; a < n; ++a) {
        int i = ord[a];
        if (dead[i]) continue;
        keep[k++] = i;
        const float* bi = boxes + (size_t)i * 4;
        float ai = (bi[2] - bi[0]) * (bi[3] - bi[1]);
        for (int c = a + 1; c < n; ++c) {
            int j = ord[c];
            if (dead[j]) continue;
            const float* bj = boxes + (size_t)j * 4;
            float aj = (bj[2] - bj[0]) * (bj[3] - bj[1]);
            float w = fmaxf(0.0f, fminf(bi[2], bj[2]) - fmaxf(bi[0], bj[0]));
            float h = fmaxf(0.0f, fminf(bi[3], bj[3]) - fmaxf(bi[1], bj[1]));
            float inter = w * h;
            float iou = inter / (ai + aj - inter);
            if (iou > thr) dead[j] = 1;
        }
    }
    free(ord); free(dead);
    return k;
}

/* boxlist_iou: structures/boxlist_ops.py:127-160 (TO_REMOVE = 1), with
 * BoxList.area() = (x2-x1+1)*(y2-y1+1) (structures/bounding_box.py:223-233).
 * a (N,4), b (M,4) -> iou (N,M). */
ODW_API void oracle_box_iou(const float* a, int N, const float* b, int M, float* iou) {
    for (int i = 0; i < N; ++i) {
        const float* p = a + (size_t)i * 4;
        float ap = (p[2] - p[0] + 1) * (p[3] - p[1] + 1);
        for (int j = 0; j < M; ++j) {
            const float* q = b + (size_t)j * 4;
            float aq = (q[2] - q[0] + 1) * (q[3] - q[1] + 1);
            float w = fminf(p[2], q[2]) - fmaxf(p[0], q[0]) + 1;
            float h = fminf(p[3], q[3]) - fmaxf(p[1], q[1]) + 1;
            if (w < 0) w = 0;
            if (h < 0) h = 0;
            float inter = w * h;
            iou[(size_t)i * M + j] = inter / (ap + aq - inter);
        }
    }
}

/* sim_mat = E E^T: roi_heads/weak_head/loss.py:319.  E (P,D) -> S (P,P).
 * k-ordered fp32 fma-free accumulation. */
ODW_API void oracle_pairwise_sim(const float* E, int P, int D, float* S) {
    for (int i = 0; i < P; ++i)
        for (int j = 0; j < P; ++j) {
            float acc = 0.0f;
            for (int k = 0; k < D; ++k) acc += E[(size_t)i * D + k] * E[(size_t)j * D + k];
            S[(size_t)i * P + j] = acc;
        }
}

/* SupConLossV2: roi_heads/sim_head/sim_loss.py:49-80.  F (N,D) unit rows,
 * labels (N) int, w (N) detached weights, temperature tau.  Returns the mean
 * loss and, when dF != NULL, dL/dF (SURVEY.md section 8a math notes).  Double
 * accumulation: this is the high-precision checker for the fp32 kernels. */
ODW_API double oracle_supcon_v2(const float* F, const int32_t* labels, const float* w,
                                int N, int D, float tau, float* dF) {
    double* S = (double*)malloc(sizeof(double) * (size_t)N * N);
    double* A = (double*)calloc((size_t)N, sizeof(double));
    double* Bs = (double*)calloc((size_t)N, sizeof(double));
    double loss = 0.0;
    for (int i = 0; i < N; ++i) {
        double m = -1e300;
        for (int j = 0; j < N; ++j) {
            double acc = 0.0;
            for (int k = 0; k < D; ++k) acc += (double)F[(size_t)i * D + k] * (double)F[(size_t)j * D + k];
            acc /= (double)tau;
            S[(size_t)i * N + j] = acc;
            if (acc > m) m = acc;
        }
        for (int j = 0; j < N; ++j) {
            double e = exp(S[(size_t)i * N + j] - m);
            S[(size_t)i * N + j] = e;
            if (j == i) continue;
            Bs[i] += e;
            if (labels[j] == labels[i]) A[i] += e;
        }
        loss += -log(A[i] / Bs[i]) * (double)w[i];
    }
    loss /= (double)N;
    if (dF) {
        /* G_ij = (w_i/N) e_ij (1/B_i - [y_i==y_j]/A_i), j != i; dF = (G+G^T) F / tau */
        double* acc = (double*)calloc((size_t)N * D, sizeof(double));
        for (int i = 0; i < N; ++i)
            for (int j = 0; j < N; ++j) {
                if (i == j) continue;
                double g = ((double)w[i] / N) * S[(size_t)i * N + j] *
                           (1.0 / Bs[i] - (labels[i] == labels[j] ? 1.0 / A[i] : 0.0));
                for (int k = 0; k < D; ++k) {
                    acc[(size_t)i * D + k] += g * F[(size_t)j * D + k];
                    acc[(size_t)j * D + k] += g * F[(size_t)i * D + k];
                }
            }
        for (size_t t = 0; t < (size_t)N * D; ++t) dF[t] = (float)(acc[t] / (double)tau);
        free(acc);
    }
    free(S); free(A); free(Bs);
    return loss;
}
