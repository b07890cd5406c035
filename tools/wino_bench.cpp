// Stand-alone timing harness of the Winograd k3 kernel (3d-sis_amd/csrc/conv3d_wino.hip) for compile-time experiments:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -I3d-sis_amd/csrc -DWN_EXP=<bits> tools/wino_bench.cpp \
//         3d-sis_amd/csrc/conv3d_wino.hip 3d-sis_amd/csrc/api.hip -o tools/_bin/wino_bench_<bits>
//   tools/_bin/wino_bench_<bits> [cin cout X Y Z nprob]      -> us per launch (HIP events around 20 back-to-back launches, best of 5)
// No torch, no Python: the variants with WN_EXP != 0 compute garbage on purpose (they remove one cost at a time).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include "sis3d.h"

int main(int argc, char **argv)
{
    int cin = argc > 1 ? atoi(argv[1]) : 128, cout = argc > 2 ? atoi(argv[2]) : 256;
    int X = argc > 3 ? atoi(argv[3]) : 24, Y = argc > 4 ? atoi(argv[4]) : 12, Z = argc > 5 ? atoi(argv[5]) : 24;
    int nprob = argc > 6 ? atoi(argv[6]) : 1;
    size_t nin = (size_t)X * Y * Z * cin, nout = (size_t)X * Y * Z * cout, nw = (size_t)cout * cin * 27;
    std::vector<float> h(nin > nw ? nin : nw);
    // own generator: the HIP runtime's threads call rand() too, so srand / rand are not reproducible here
    unsigned long long rs = 88172645463325252ull;
    auto rnd = [&]() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (int)((rs >> 20) % 2001); };
    float *in[4], *out[4], *wp[4], *w, *bias;
    const float *cin_p[4], *cwp[4], *cb[4];
    hipMalloc(&w, nw * 4);
    hipMalloc(&bias, cout * 4);
    hipMemset(bias, 0, cout * 4);
    size_t np = sis3d_conv_k3wino_packed_floats(cout, cin);
    for (int p = 0; p < nprob; ++p) {
        hipMalloc(&in[p], nin * 4); hipMalloc(&out[p], (nout + 65536) * 4); hipMalloc(&wp[p], np * 4);
        for (size_t i = 0; i < nin; ++i) { float v = (rnd() - 1000) * 1e-3f; h[i] = v > 0 ? v : 0; }
        hipMemcpy(in[p], h.data(), nin * 4, hipMemcpyHostToDevice);
        for (size_t i = 0; i < nw; ++i) h[i] = (rnd() - 1000) * 5e-5f;
        hipMemcpy(w, h.data(), nw * 4, hipMemcpyHostToDevice);
        if (sis3d_conv_k3wino_pack_weight(w, cout, cin, wp[p], nullptr)) return 1;
        cin_p[p] = in[p]; cwp[p] = wp[p]; cb[p] = bias;
    }
    hipDeviceSynchronize();
    auto launch = [&]() { return sis3d_conv3d_k3wino(nprob, cin_p, X, Y, Z, cin, cin, cwp, cb, cout, SIS3D_EPI_RELU, out, cout, 0, nullptr); };
    for (int i = 0; i < 30; ++i) if (launch()) { printf("launch failed: %s\n", sis3d_last_hip_error()); return 1; }
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0, nullptr);
        for (int i = 0; i < 20; ++i) launch();
        hipEventRecord(e1, nullptr);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms / 20 < best) best = ms / 20;
    }
    double fl = 2.0 * X * Y * Z * cout * (double)cin * 27 * nprob;
    printf("WN_EXP=%d %d->%d %dx%dx%d x%d: %.1f us  (%.1f TF algorithmic, %.1f TF executed)\n", WN_EXP, cin, cout, X, Y, Z, nprob, best * 1e3,
           fl / best / 1e9, fl / 3.375 / best / 1e9);
    {   // order-sensitive checksum of problem 0's output: equal between variants that promise bit-identical results (SIS3D_WINO_WC=1 / 2)
        std::vector<float> o(nout);
        hipMemcpy(o.data(), out[0], nout * 4, hipMemcpyDeviceToHost);
        unsigned long long hsh = 1469598103934665603ull;
        double sum = 0;
        for (size_t i = 0; i < nout; ++i) { unsigned u; memcpy(&u, &o[i], 4); hsh = (hsh ^ u) * 1099511628211ull; sum += o[i]; }
        printf("  output checksum %016llx  sum %.6f  first %g %g %g %g  in0 %g %g\n", hsh, sum, o[0], o[1], o[1000], o[nout - 1], h[0], h[1]);
    }
    if (WN_EXP & 64) {          // phase timestamps (ns) written by thread 0 of every workgroup BEHIND the output: 20 floats each
        launch();
        hipDeviceSynchronize();
        int nwg = ((X + 7) / 8) * ((Y + 3) / 4) * ((Z + 7) / 8) * ((cout / 16 + 1) / 2);
        const int R = 20;
        std::vector<float> t((size_t)nwg * R * nprob);
        for (int p = 0; p < nprob; ++p) hipMemcpy(t.data() + (size_t)p * nwg * R, out[p] + nout, (size_t)nwg * R * 4, hipMemcpyDeviceToHost);
        const int n = nwg * nprob;
        static const char *nm[15] = {"prologue", "loop", "epilogue", "", "pro:setup+loads issued", "pro:zero fill+acc clear", "pro:wait for loads", "pro:stores+barrier",
                                     "pro:first V", "epi:wait+barrier", "epi:transform+partner stores", "epi:barrier", "epi:finish rows", "epi:tail+stores issued",
                                     "epi:stores drained"};
        static const int order[17] = {0, 1, 2, 18, 17, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, -1};
        for (int q = 0; order[q] >= 0; ++q) {
            const int j = order[q];
            double s = 0, mx = 0;
            for (int i = 0; i < n; ++i) { s += t[R * i + j]; if (t[R * i + j] > mx) mx = t[R * i + j]; }
            printf("  %-30s %7.0f ns (max %.0f)\n", j == 18 ? "kernel entry -> wave body" : j == 17 ? "pro:(of setup) U DMA issued" : nm[j], s / n, mx);
        }
        // hand-over on a CU: workgroups that ran on the same CU, by start time: gap = next start - previous end (stores drained)
        std::vector<int> idx(n);
        for (int i = 0; i < n; ++i) idx[i] = i;
        auto cu = [&](int i) { return (unsigned)t[R * i + 16]; };
        std::sort(idx.begin(), idx.end(), [&](int a, int b) { return cu(a) != cu(b) ? cu(a) < cu(b) : t[R * a + 3] < t[R * b + 3]; });
        double gs = 0, gmx = 0, gmn = 1e30; int ng = 0, ncu = 0;
        for (int k = 0; k < n; ++k) {
            if (k == 0 || cu(idx[k]) != cu(idx[k - 1])) { ++ncu; continue; }
            double gap = t[R * idx[k] + 3] - t[R * idx[k - 1] + 15];
            if (gap < -5e7) gap += 1e8;
            gs += gap; ++ng; if (gap > gmx) gmx = gap; if (gap < gmn) gmn = gap;
        }
        double t0min = 1e30, t1max = 0;
        for (int i = 0; i < n; ++i) { if (t[R * i + 3] < t0min) t0min = t[R * i + 3]; if (t[R * i + 15] > t1max) t1max = t[R * i + 15]; }
        printf("  %d workgroups on %d distinct CUs; %d hand-overs on a CU: gap mean %.0f ns (min %.0f, max %.0f); first start -> last end %.0f ns\n", n, ncu, ng,
               ng ? gs / ng : 0.0, ng ? gmn : 0.0, gmx, t1max - t0min);
    }
    return 0;
}
