// wgrad_timeline.hip -- experiment: phase timeline of conv_wgrad_halo_kernel (compiles the library's gemm_bf16.hip with
// ODW_WH_TIMELINE; tools/exp, not shipped).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iod_wscl_amd/csrc -Iinclude tools/exp/wgrad_timeline.hip \
//         od_wscl_amd/csrc/odw_common.hip -o tools/exp/wgrad_timeline.bin
#define ODW_WH_TIMELINE 1
#include "../../od_wscl_amd/csrc/gemm_bf16.hip"
#include <cstdio>
#include <vector>
#include <algorithm>

int main(int argc, char** argv) {
    const int C = argc > 1 ? atoi(argv[1]) : 512, HW = argc > 2 ? atoi(argv[2]) : 76, dil = argc > 3 ? atoi(argv[3]) : 1;
    const int m = HW * HW;
    std::vector<unsigned short> h((size_t)m * C);
    unsigned s = 12345;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (unsigned short)(0x3c00 + ((s >> 20) & 0xff)); }
    unsigned short *dz, *x, *zero; float* dw; void* ws;
    hipMalloc(&dz, h.size() * 2); hipMalloc(&x, h.size() * 2); hipMalloc(&zero, 256); hipMemset(zero, 0, 256);
    hipMemcpy(dz, h.data(), h.size() * 2, hipMemcpyHostToDevice); hipMemcpy(x, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipMalloc(&dw, (size_t)C * C * 9 * 4);
    const int64_t wsb = odw_conv_wgrad_tn_workspace(C, C, m);
    hipMalloc(&ws, wsb);
    for (int r = 0; r < 5; ++r)
        if (odw_conv_wgrad_tn(dz, C, x, m, HW, HW, C, dil, C, C, dw, 0, zero, ws, wsb, nullptr) != 0) { printf("failed: %s\n", odw_last_error()); return 1; }
    hipDeviceSynchronize();
    std::vector<long long> tl(1024 * 8 * 32, 0);
    hipMemcpyToSymbol(HIP_SYMBOL(g_wh_tl), tl.data(), tl.size() * 8);
    odw_conv_wgrad_tn(dz, C, x, m, HW, HW, C, dil, C, C, dw, 0, zero, ws, wsb, nullptr);
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(tl.data(), HIP_SYMBOL(g_wh_tl), tl.size() * 8);
    long long t0 = 0;
    for (size_t i = 0; i < tl.size(); ++i) if (tl[i] && (!t0 || tl[i] < t0)) t0 = tl[i];
    printf("C=%d %dx%d dil %d: event: waves, mean / min / max us after the first stamp\n", C, HW, HW, dil);
    for (int e = 0; e < 32; ++e) {
        double sum = 0; long long mn = 1ll << 60, mx = 0; int n = 0;
        for (int w = 0; w < 1024 * 8; ++w) { const long long t = tl[(size_t)w * 32 + e]; if (!t) continue; sum += t - t0; mn = std::min(mn, t - t0); mx = std::max(mx, t - t0); ++n; }
        if (!n) continue;
        const char* nm = e == 30 ? "K loops done" : e == 31 ? "exit" : (e % 4 == 0 ? "landed" : e % 4 == 1 ? "reads + DMA issued" : "K loop done");
        printf("  %2d tile %d %-20s %5d  %7.2f %7.2f %7.2f\n", e, e / 4, nm, n, sum / n / 100.0, mn / 100.0, mx / 100.0);
    }
    return 0;
}
