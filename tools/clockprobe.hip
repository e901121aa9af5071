// Micro-probe: what clock does a single-workgroup kernel run at, and what does a dependent f64 op cost?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
__global__ void chain(double* out, long long* cyc, int n, int mode) {
    double x = out[0], y = 1.0000001;
    long long t0 = clock64();
    long long w0 = wall_clock64();
    if (mode == 0) for (int i = 0; i < n; ++i) x = fma(x, y, 1e-9);
    if (mode == 1) for (int i = 0; i < n; ++i) x = sqrt(x * x + 1.0);
    if (mode == 2) for (int i = 0; i < n; ++i) x = 1.0 / (x + 2.0);
    if (mode == 3) for (int i = 0; i < n; ++i) x = rsqrt(x * x + 1.0);
    long long t1 = clock64();
    long long w1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = w1 - w0; }
    out[threadIdx.x + blockIdx.x * blockDim.x] = x;
}
__global__ void ldschain(double* out, long long* cyc, int n) {
    __shared__ double buf[256];
    buf[threadIdx.x] = threadIdx.x;
    __syncthreads();
    long long t0 = clock64();
    double x = 0; int idx = threadIdx.x;
    for (int i = 0; i < n; ++i) { x += buf[idx & 255]; idx = (int)x + i; }
    long long t1 = clock64();
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
    out[threadIdx.x] = x;
}
__global__ void barriers(long long* cyc, int n) {
    long long t0 = clock64();
    for (int i = 0; i < n; ++i) __syncthreads();
    long long t1 = clock64();
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    double* out; long long* cyc;
    hipMalloc(&out, 1 << 24); hipMalloc(&cyc, 64);
    hipMemset(out, 0, 1 << 24);
    const char* names[] = {"fma", "sqrt", "div", "rsqrt"};
    for (int blocks : {1, 1024}) {
        for (int mode = 0; mode < 4; ++mode) {
            int n = 200000;
            for (int rep = 0; rep < 2; ++rep) {
                auto a = std::chrono::high_resolution_clock::now();
                hipLaunchKernelGGL(chain, dim3(blocks), dim3(blocks == 1 ? 64 : 256), 0, 0, out, cyc, n, mode);
                hipDeviceSynchronize();
                auto b = std::chrono::high_resolution_clock::now();
                long long h[2]; hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
                double us = std::chrono::duration<double, std::micro>(b - a).count();
                printf("blocks=%d %s: %.1f cycles/op (s_memtime), wall_clock64 ticks/op %.2f, host %.1f us total => %.1f ns/op\n", blocks, names[mode],
                       (double)h[0] / n, (double)h[1] / n, us, us * 1e3 / n);
            }
        }
    }
    hipLaunchKernelGGL(ldschain, dim3(1), dim3(64), 0, 0, out, cyc, 100000);
    hipDeviceSynchronize();
    long long h[2]; hipMemcpy(h, cyc, 8, hipMemcpyDeviceToHost);
    printf("dependent LDS read+add: %.1f cycles/op\n", (double)h[0] / 100000);
    for (int th : {64, 256, 1024}) {
        hipLaunchKernelGGL(barriers, dim3(1), dim3(th), 0, 0, cyc, 100000);
        hipDeviceSynchronize();
        hipMemcpy(h, cyc, 8, hipMemcpyDeviceToHost);
        printf("__syncthreads with %d threads: %.1f cycles\n", th, (double)h[0] / 100000);
    }
    return 0;
}
