// Phase timing of the LDS panel kernel (cycles of s_memtime): hipcc -DDRE_PANEL_PROBE ... see tools/README in DESIGN.md.
#include "../differentialriccatiequations.jl_amd/csrc/dense.hip"
#include <cstdio>
#include <vector>
using namespace dre;
int main() {
    for (int m : {120, 300, 600, 1000}) {
        const int b = 16;
        std::vector<double> h((size_t)m * b);
        for (size_t i = 0; i < h.size(); ++i) h[i] = sin(0.37 * i) + 0.01 * (i % 7);
        double *A, *V, *T, *VT, *part; AdiState* st;
        hipMalloc(&A, h.size() * 8); hipMalloc(&V, h.size() * 8); hipMalloc(&VT, h.size() * 8); hipMalloc(&T, b * b * 8); hipMalloc(&part, 64 * 8); hipMalloc(&st, sizeof(AdiState));
        hipMemset(V, 0, h.size() * 8); hipMemset(st, 0, sizeof(AdiState));
        std::vector<double> hp(64, 1.0); hipMemcpy(part, hp.data(), 64 * 8, hipMemcpyHostToDevice);
        hipFuncSetAttribute((const void*)k_qr_panel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 132 * 1024);
        for (int withpart = 0; withpart < 2; ++withpart)
            for (int rep = 0; rep < 3; ++rep) {
                hipMemcpy(A, h.data(), h.size() * 8, hipMemcpyHostToDevice);
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                hipEventRecord(e0, 0);
                hipLaunchKernelGGL((k_qr_panel<true>), dim3(1), dim3(1024), (size_t)(m | 1) * b * 8, 0, A, m, m, 0, b, V, m, T, b, VT, m, st,
                                   withpart ? part : (const double*)nullptr, 64, 0, 4.0, (double*)nullptr);
                hipEventRecord(e1, 0);
                hipDeviceSynchronize();
                float ms; hipEventElapsedTime(&ms, e0, e1);
                long long g[20] = {0}; hipError_t er = hipMemcpyFromSymbol(g, HIP_SYMBOL(dre::g_probe), sizeof(g)); if (er != hipSuccess) printf("memcpyFromSymbol: %s\n", hipGetErrorString(er));
                if (rep == 2) {
                    printf("   column 3, wave 4: scalar %lld, dot %lld, wsum+scal %lld, axpy %lld, nn-wsum %lld, barrier %lld\n", g[11]-g[10], g[12]-g[11], g[13]-g[12], g[14]-g[13], g[15]-g[14], g[16]-g[15]);
                    printf("m=%d part=%d event %.1f us | cycles: prologue %lld, decl %lld, load %lld, init-norm %lld, columns %lld, finish %lld, writeAV+T %lld, VT %lld, total %lld\n", m, withpart,
                           ms * 1e3, g[1] - g[0], g[2] - g[1], 0LL, g[3] - g[2], g[4] - g[3], g[5] - g[4], g[6] - g[5], g[7] - g[6], g[7] - g[0]);
                }
            }
    }
    return 0;
}
