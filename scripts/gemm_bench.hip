// Standalone timing harness for the implicit-GEMM kernels (no Python / torch): includes gemm.hip directly so that
// experiment macros can be tried without touching the product library.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -I tacotron_multispeaker_amd/csrc -I include scripts/gemm_bench.hip -o scripts/_mb/gemm_bench
// env: ONLY=<case name>, ITER=<timed iterations>, MODE=fwd|bwd_data|bwd_weight|all (default all)
// The checksums let two builds / env settings (e.g. TACO_NT_V1=1) be compared on identical inputs.
#include "../tacotron_multispeaker_amd/csrc/gemm.hip"
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <cmath>
#include <vector>

static float* dalloc(size_t n, float scale, unsigned seed) {
    std::vector<float> h(n);
    unsigned s = 12345u + seed;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = scale * ((int)(s >> 9) % 2001 - 1000) / 1000.0f; }
    float* d; hipMalloc(&d, n * sizeof(float)); hipMemcpy(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice);
    return d;
}

static void checksum(const float* d, size_t n) {
    std::vector<float> y(n); hipMemcpy(y.data(), d, n * 4, hipMemcpyDeviceToHost);
    double cs = 0, ca = 0;
    for (size_t i = 0; i < n; ++i) { cs += y[i] * (double)((i % 7) + 1); ca += fabs(y[i]); }
    printf("   chk %.6e %.6e\n", cs, ca);
}

struct Case { const char* name; int M, T, Cin, Cout, kw, bank; };

int main() {
    const Case cases[] = {
        {"post proj_1", 20480, 640, 1024, 256, 3, 0},
        {"post proj_2", 20480, 640, 256, 80, 3, 0},
        {"post bank", 20480, 640, 80, 1024, 1, 8},
        {"post hw", 20480, 20480, 128, 256, 1, 0},
        {"post xp", 20480, 20480, 128, 768, 1, 0},
        {"linear", 20480, 20480, 256, 1028, 1, 0},
        {"enc bank", 4096, 128, 128, 2048, 1, 16},
        {"enc proj_1", 4096, 128, 2048, 128, 3, 0},
        {"enc proj_2", 4096, 128, 128, 128, 3, 0},
        {"dec 256", 4096, 4096, 256, 256, 1, 0},
        {"dense 8k", 8192, 8192, 2048, 8192, 1, 0},
    };
    const char* only = getenv("ONLY");
    const char* mode = getenv("MODE");
    const int iters = getenv("ITER") ? atoi(getenv("ITER")) : 100;
    hipStream_t st; hipStreamCreate(&st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (const Case& c : cases) {
        if (only && strcmp(only, c.name)) continue;
        const int Cout = c.Cout;
        size_t wn = c.bank ? (size_t)c.bank * (c.bank + 1) / 2 * c.Cin * 128 : (size_t)c.kw * c.Cin * Cout;
        float* X = dalloc((size_t)c.M * c.Cin, 1.0f, 1);
        float* W = dalloc(wn, 0.05f, 2);
        float* Y = dalloc((size_t)c.M * Cout, 1.0f, 3);      // doubles as dY
        float* dX = dalloc((size_t)c.M * c.Cin, 0.0f, 4);
        float* dW = dalloc(wn, 0.0f, 5);
        double flop = 0;
        if (c.bank) for (int k = 1; k <= c.bank; ++k) flop += 2.0 * c.M * c.Cin * 128 * k;
        else flop = 2.0 * c.M * c.Cin * Cout * c.kw;
        for (int m = 0; m < 3; ++m) {
            const char* mname = m == 0 ? "fwd" : m == 1 ? "bwd_data" : "bwd_weight";
            if (mode && strcmp(mode, "all") && strcmp(mode, mname)) continue;
            if (m == 2) hipMemsetAsync(dW, 0, wn * 4, st);
            for (int pass = 0; pass < 2; ++pass) {
                const int it = pass ? iters : 3;
                hipEventRecord(e0, st);
                for (int i = 0; i < it; ++i) {
                    int e;
                    if (m == 0) e = taco_conv_gemm_fwd(X, W, nullptr, Y, c.M, c.T, c.Cin, Cout, c.kw, c.bank, c.Cin, Cout, Cout, 0, 0, st);
                    else if (m == 1) e = taco_conv_gemm_bwd_data(Y, W, dX, c.M, c.T, c.Cin, Cout, c.kw, c.bank, Cout, Cout, c.Cin, 0, st);
                    else e = taco_conv_gemm_bwd_weight(X, Y, dW, c.M, c.T, c.Cin, Cout, c.kw, c.bank, c.Cin, Cout, Cout, st);
                    if (e) { printf("%s %s: error %d\n", c.name, mname, e); break; }
                }
                hipEventRecord(e1, st); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (pass) printf("%-12s %-10s %8.1f us %6.1f TF\n", c.name, mname, ms * 1e3 / it, flop / (ms / it * 1e-3) / 1e12);
            }
            if (m == 0) { checksum(Y, (size_t)c.M * Cout); hipFree(Y); Y = dalloc((size_t)c.M * Cout, 1.0f, 3); }
            else if (m == 1) checksum(dX, (size_t)c.M * c.Cin);
            else checksum(dW, wn);     // accumulated over 3 + iters calls
        }
        hipFree(X); hipFree(W); hipFree(Y); hipFree(dX); hipFree(dW);
    }
    return 0;
}
