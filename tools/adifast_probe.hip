// Chain timing of the fast ADI iteration kernel (k_adi_fast) on synthetic data: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include tools/adifast_probe.hip -o tools/adifast_probe
#include "../differentialriccatiequations.jl_amd/csrc/dense.hip"
#include <cstdio>
#include <vector>
using namespace dre;
__global__ void k_empty(AdiFastArgs a) { if (a.st->done) return; }
int main(int argc, char** argv) {
    Ctx ctx; hipStreamCreate(&ctx.stream);
    for (int n : {371, 1357}) for (int k : {64, 160, 294}) {
        const int nstrip = adi_fast_nstrip(n), kst = adi_fast_kst(n), NIT = 200;
        std::vector<double> hp(adi_fast_pack_doubles(n)), hr((size_t)n * k), ht((size_t)k * k, 0.0);
        for (size_t i = 0; i < hp.size(); ++i) hp[i] = 1e-3 * sin(0.37 * i);
        for (size_t i = 0; i < hr.size(); ++i) hr[i] = cos(0.11 * i);
        for (int i = 0; i < k; ++i) ht[i + (size_t)i * k] = 1.0;
        double *P[10], *R, *V, *G, *T, *W; AdiState* st; hipMalloc(&W, ADI_FAST_NWS * 8); hipMemset(W, 0, ADI_FAST_NWS * 8);
        for (int j = 0; j < 10; ++j) { hipMalloc(&P[j], hp.size() * 8); hipMemcpy(P[j], hp.data(), hp.size() * 8, hipMemcpyHostToDevice); }
        hipMalloc(&R, hr.size() * 8 * (NIT + 1)); hipMalloc(&V, hr.size() * 8 * NIT); hipMalloc(&G, (size_t)k * k * 16); hipMalloc(&T, ht.size() * 8); hipMalloc(&st, sizeof(AdiState));
        hipMemcpy(R, hr.data(), hr.size() * 8, hipMemcpyHostToDevice); hipMemcpy(T, ht.data(), ht.size() * 8, hipMemcpyHostToDevice);
        AdiState h; memset(&h, 0, sizeof(h)); h.maxiters = 100000; h.abstol = 0.0; hipMemcpy(st, &h, sizeof(h), hipMemcpyHostToDevice);
        for (int mode = 0; mode < 4; ++mode) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0, ctx.stream);
                for (int j = 1; j <= NIT; ++j) {
                    AdiFastArgs a; memset(&a, 0, sizeof(a));
                    a.n = n; a.k = k; a.nstrip = nstrip; a.kst = kst; a.Apack = P[j % 10]; adi_fast_pick(n, k, &a.mode, &a.nt); if (getenv("NT")) a.nt = atoi(getenv("NT")); if (getenv("MODE")) a.mode = atoi(getenv("MODE"));
                    a.Rcur = R + (size_t)(j - 1) * n * k; a.ldr = n; a.Rnext = R + (size_t)j * n * k; a.ldr_next = n; a.V = V + (size_t)(j - 1) * n * k; a.ldv = n;
                    a.two_mu = 1e-3; a.T = T; a.ldt = k; a.tdiag = 0; a.alpha = 1.0; a.st = st; a.nws = W; a.it_prev2 = j - 2;
                    a.G_prev = (mode == 0 && j >= 2) ? G + (size_t)((j - 1) & 1) * k * k : nullptr;
                    a.G_prev2 = (mode == 0 && j >= 3) ? G + (size_t)(j & 1) * k * k : nullptr;
                    a.do_strips = mode <= 1 ? 1 : 0;
                    if (mode == 2) { a.G_prev = G; a.G_prev2 = G + (size_t)k * k; }
                    if (mode == 3) hipLaunchKernelGGL(k_empty, dim3(200), dim3(256), 0, ctx.stream, a);
                    else adi_fast_iter(&ctx, a);
                }
                hipEventRecord(e1, ctx.stream);
                hipStreamSynchronize(ctx.stream);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep == 1) printf("n=%d k=%d mode=%s: %.2f us per launch\n", n, k, mode == 0 ? "iteration+riders" : mode == 1 ? "iteration only" : mode == 2 ? "riders only (flush)" : "empty kernel", ms * 1e3 / NIT);
            }
        }
        for (int j = 0; j < 10; ++j) hipFree(P[j]);
        hipFree(R); hipFree(V); hipFree(G); hipFree(T); hipFree(st);
    }
    return 0;
}
