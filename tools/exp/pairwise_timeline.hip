// pairwise_timeline.hip -- experiment: where does pairwise_sim_panel_kernel spend its time?  Compiles the library's
// contrastive.hip with ODW_PW_TIMELINE (lane 0 of every wave stamps wall_clock64() at its phase boundaries) and prints,
// per event, the mean / min / max offset from the earliest entry stamp over all workgroups.  (tools/exp, not shipped.)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iod_wscl_amd/csrc -Iinclude tools/exp/pairwise_timeline.hip \
//         od_wscl_amd/csrc/odw_common.hip -o tools/exp/pairwise_timeline.bin
#define ODW_PW_TIMELINE 1
#include "../../od_wscl_amd/csrc/contrastive.hip"
#include <cstdio>
#include <vector>
#include <algorithm>

int main(int argc, char** argv) {
    const int P = argc > 1 ? atoi(argv[1]) : 4000;
    std::vector<float> h((size_t)P * 128);
    unsigned s = 12345;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.0f - 0.5f; }
    const bool dma = argc > 2 && argv[2][0] == 'd';        // "dma": the planes + LDS-DMA kernel (planes split beforehand)
    float *E, *S;
    void* ws;
    const long long wsb = odw_pairwise_sim_workspace(P, 128);
    hipMalloc(&E, h.size() * 4); hipMalloc(&S, (size_t)P * P * 4); hipMalloc(&ws, wsb);
    hipMemcpy(E, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    odw_pairwise_split_planes(E, P, ws, nullptr);
    auto launch = [&]() { return dma ? odw_pairwise_sim_planes(ws, P, S, nullptr) : odw_pairwise_sim(E, P, 128, S, nullptr); };
    for (int r = 0; r < 10; ++r) if (launch() != 0) { printf("launch failed: %s\n", odw_last_error()); return 1; }
    hipDeviceSynchronize();
    std::vector<long long> tl(1024 * 8 * 32, 0);
    hipMemcpyToSymbol(HIP_SYMBOL(g_pw_tl), tl.data(), tl.size() * 8);
    launch();
    hipDeviceSynchronize();
    printf("%s kernel\n", dma ? "planes + DMA" : "panel");
    hipMemcpyFromSymbol(tl.data(), HIP_SYMBOL(g_pw_tl), tl.size() * 8);
    long long t0 = 0;
    for (int w = 0; w < 256 * 8; ++w) if (tl[w * 32] && (!t0 || tl[w * 32] < t0)) t0 = tl[w * 32];
    const char* names[32] = {"entry", "panel staged"};
    printf("P=%d   event: waves, mean / min / max us after the first entry (100 MHz clock)\n", P);
    for (int role = 0; role < 2; ++role) {
        printf(role ? "loader waves\n" : "compute waves\n");
        for (int e = 0; e < 32; ++e) {
            double sum = 0; long long mn = 1ll << 60, mx = 0; int n = 0;
            for (int w = 0; w < 256 * 8; ++w) {
                if ((w % 8 == 7) != (role == 1)) continue;
                const long long t = tl[w * 32 + e];
                if (!t) continue;
                sum += t - t0; mn = std::min(mn, t - t0); mx = std::max(mx, t - t0); ++n;
            }
            if (!n) continue;
            char nm[32];
            if (e < 2) snprintf(nm, 32, "%s", names[e]);
            else if (e == 31) snprintf(nm, 32, "exit");
            else snprintf(nm, 32, "it %d %s", (e - 2) / 3, (e - 2) % 3 == 0 ? "block ready" : (e - 2) % 3 == 1 ? "MFMAs done" : "stores issued");
            printf("  %-22s %5d  %7.2f %7.2f %7.2f\n", nm, n, sum / n / 100.0, mn / 100.0, mx / 100.0);
        }
    }
    return 0;
}
