// Phase timing of the one-launch Bottleneck kernel (csrc/bottleneck.hip built with -DBN_TIMING): wall_clock64() (100 MHz)
// at the phase boundaries of every wave, 20 launches; prints the median span of each phase over workgroups.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DBN_TIMING -I include -I 3d-sis_amd/csrc tools/bn_timing.cpp -o gpurun_out/bn_timing
#include "../3d-sis_amd/csrc/bottleneck.hip"
#include <algorithm>
#include <stdio.h>
#include <vector>
long long *g_bn_dbg = nullptr;
void sis3d_record_hip_error(hipError_t) {}
extern "C" size_t sis3d_conv_k3t16_packed_floats(int cout, int cin) { return (size_t)((cout + 15) / 16) * (cin / 32) * 4 * 27 * 128; }

static void run(int planes, int cio, int c2, int X, int Y, int Z, const char *tag)
{
    const size_t nv = (size_t)X * Y * Z;
    float *y1, *res, *out, *y1n, *w2, *w3, *wn, *b;
    hipMalloc(&y1, nv * planes * 4); hipMalloc(&res, nv * cio * 4); hipMalloc(&out, nv * cio * 4); hipMalloc(&y1n, nv * 64 * 4);
    hipMalloc(&w2, sis3d_conv_k3t16_packed_floats(planes, planes) * 4); hipMalloc(&w3, (size_t)cio * planes * 4);
    hipMalloc(&wn, (size_t)cio * 64 * 4); hipMalloc(&b, 1024);
    hipMemset(y1, 0x3c, nv * planes * 4); hipMemset(res, 0x3c, nv * cio * 4); hipMemset(w2, 0x3c, sis3d_conv_k3t16_packed_floats(planes, planes) * 4);
    hipMemset(w3, 0x3c, (size_t)cio * planes * 4); hipMemset(wn, 0x3c, (size_t)cio * 64 * 4); hipMemset(b, 0, 1024);
    const int brick = sis3d_bottleneck16_brick(X, Y, Z, planes);
    const int bs = brick == 0 ? 6 : 3;
    const int nwg = ((X + bs - 1) / bs) * ((Y + bs - 1) / bs) * ((Z + bs - 1) / bs);
    hipMalloc(&g_bn_dbg, (size_t)nwg * 4 * 8 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<long long> h((size_t)nwg * 32);
    const char *names[7] = {"addr+issue loads -> LDS stores issued", "barrier (loads landed)", "conv2 main loop", "tail operand requests + barrier",
                            "reduction LDS writes", "barrier", "1x1x1 tail"};
    std::vector<double> acc[8];
    float ms_sum = 0;
    for (int it = 0; it < 25; ++it) {
        hipMemset(g_bn_dbg, 0, (size_t)nwg * 256);
        hipEventRecord(e0, 0);
        int rc = sis3d_bottleneck16(y1, X, Y, Z, planes, w2, b, w3, b, cio, res, cio, out, cio, 0, wn, b, c2, y1n, -1, nullptr);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        if (rc) { printf("rc %d\n", rc); return; }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (it < 5) continue;
        ms_sum += ms;
        hipMemcpy(h.data(), g_bn_dbg, (size_t)nwg * 256, hipMemcpyDeviceToHost);
        long long t0 = h[0], tend = 0;
        for (int w = 0; w < nwg * 4; ++w) { t0 = std::min(t0, h[w * 8]); tend = std::max(tend, h[w * 8 + 7]); }
        std::vector<double> ph[7], start, end;
        for (int w = 0; w < nwg * 4; ++w) {
            for (int k = 0; k < 7; ++k) ph[k].push_back((h[w * 8 + k + 1] - h[w * 8 + k]) * 0.01);
            start.push_back((h[w * 8] - t0) * 0.01); end.push_back((h[w * 8 + 7] - t0) * 0.01);
        }
        auto med = [](std::vector<double> &v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
        for (int k = 0; k < 7; ++k) acc[k].push_back(med(ph[k]));
        std::sort(start.begin(), start.end());
        acc[7].push_back((tend - t0) * 0.01);
        if (it == 24) printf("%s: wave start spread (us) median %.2f max %.2f; last wave ends %.2f after the first starts\n", tag, start[start.size() / 2],
                             start.back(), (tend - t0) * 0.01);
    }
    printf("%s: event time %.1f us per launch (incl. launch gap)\n", tag, ms_sum / 20 * 1e3);
    for (int k = 0; k < 7; ++k) { std::sort(acc[k].begin(), acc[k].end()); printf("  %-42s %6.2f us\n", names[k], acc[k][acc[k].size() / 2]); }
    std::sort(acc[7].begin(), acc[7].end());
    printf("  %-42s %6.2f us\n", "first start -> last end (in-kernel)", acc[7][acc[7].size() / 2]);
    hipFree(g_bn_dbg); g_bn_dbg = nullptr;
}

int main()
{
    run(32, 32, 32, 48, 24, 48, "bneck 32/32 @48x24x48 6x6x6");
    run(32, 128, 32, 24, 12, 24, "bneck 128/32 @24x12x24 3x3x3");
    return 0;
}
