// Standalone timing harness for the implicit-GEMM kernels (no Python / torch): includes gemm.hip directly so that
// experiment macros (-DEXP_...) can be tried without touching the product library.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I tacotron_multispeaker_amd/csrc -I include scripts/gemm_bench.hip -o scripts/_mb/gemm_bench
#include "../tacotron_multispeaker_amd/csrc/gemm.hip"
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <cmath>
#include <vector>

static float* dalloc(size_t n, float scale) {
    std::vector<float> h(n);
    unsigned s = 12345u + (unsigned)n;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = scale * ((int)(s >> 9) % 2001 - 1000) / 1000.0f; }
    float* d; hipMalloc(&d, n * sizeof(float)); hipMemcpy(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice);
    return d;
}

struct Case { const char* name; int M, T, Cin, Cout, kw, bank; };

int main() {
    const Case cases[] = {
        {"post proj_1", 20480, 640, 1024, 256, 3, 0},
        {"pp1 M=8192", 8192, 512, 1024, 256, 3, 0},
        {"pp1 M=12288", 12288, 512, 1024, 256, 3, 0},
        {"pp1 M=16384", 16384, 512, 1024, 256, 3, 0},
        {"pp1 M=18432", 18432, 512, 1024, 256, 3, 0},
        {"pp1 M=24576", 24576, 512, 1024, 256, 3, 0},
        {"pp1 M=32768", 32768, 512, 1024, 256, 3, 0},
        {"post bank", 20480, 640, 80, 1024, 1, 8},
        {"enc bank", 4096, 128, 128, 2048, 1, 16},
        {"dense 8k", 8192, 8192, 2048, 8192, 1, 0},
        {"linear", 20480, 20480, 256, 1028, 1, 0},
    };
    hipStream_t st; hipStreamCreate(&st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* only = getenv("ONLY");
    for (const Case& c : cases) {
        if (only && strcmp(only, c.name)) continue;
        const int Cout = c.Cout & ~3;
        size_t wn = c.bank ? (size_t)c.bank * (c.bank + 1) / 2 * c.Cin * 128 : (size_t)c.kw * c.Cin * Cout;
        float* X = dalloc((size_t)c.M * c.Cin, 1.0f);
        float* W = dalloc(wn, 0.05f);
        float* Y = dalloc((size_t)c.M * Cout, 0.0f);
        double flop = 0;
        if (c.bank) for (int k = 1; k <= c.bank; ++k) flop += 2.0 * c.M * c.Cin * 128 * k;
        else flop = 2.0 * c.M * c.Cin * Cout * c.kw;
        for (int pass = 0; pass < 2; ++pass) {
            const int it = pass ? (getenv("ITER") ? atoi(getenv("ITER")) : 20) : 3;
            hipEventRecord(e0, st);
            for (int i = 0; i < it; ++i) {
                int e = taco_conv_gemm_fwd(X, W, nullptr, Y, c.M, c.T, c.Cin, Cout, c.kw, c.bank, c.Cin, Cout, Cout, 0, 0, st);
                if (e) { printf("%s: error %d\n", c.name, e); return 1; }
            }
            hipEventRecord(e1, st); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (pass) printf("%-12s fwd %8.1f us %6.1f TF\n", c.name, ms * 1e3 / it, flop / (ms / it * 1e-3) / 1e12);
        }
        std::vector<float> y((size_t)c.M * Cout); hipMemcpy(y.data(), Y, y.size() * 4, hipMemcpyDeviceToHost);
        double cs = 0, ca = 0;
        for (size_t i = 0; i < y.size(); ++i) { cs += y[i] * (double)((i % 7) + 1); ca += fabs(y[i]); }
        printf("   chk %.6e %.6e\n", cs, ca);
        hipFree(X); hipFree(W); hipFree(Y);
    }
    return 0;
}
