// Microbenchmark: issue rate of v_add_f64 / v_mul_f64 / v_fma_f64 on gfx950 WITH the
// in-kernel clock (s_memtime / s_memrealtime stamps, MI355X_MICROARCH.md DVFS item 6),
// so that "cycles per wave-instruction" is measured and not derived from an assumed
// clock; and of conflict-free ds_read_b64 (answers: what bounds the column kernel).
//   hipcc --offload-arch=gfx950 -O3 -o fp64_rate fp64_rate.hip && ./fp64_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

template <int OP>
__global__ void __launch_bounds__(256) k_rate(double *out, unsigned long long *stamps, double a, double b, int iters)
{
    double x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    unsigned long long t0 = 0, r0 = 0;
    if (threadIdx.x == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    for (int i = 0; i < iters; ++i) {
#define STEP(x)                                                                        \
        if (OP == 0) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x) : "v"(a));          \
        else if (OP == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(b));     \
        else asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(a));
        STEP(x0) STEP(x1) STEP(x2) STEP(x3) STEP(x4) STEP(x5) STEP(x6) STEP(x7)
    }
    if (threadIdx.x == 0) {
        stamps[blockIdx.x * 2 + 0] = __builtin_amdgcn_s_memtime() - t0;
        stamps[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

__global__ void __launch_bounds__(256) k_lds(double *out, int iters, int stride)
{
    __shared__ double T[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) T[i] = i;
    __syncthreads();
    typedef __attribute__((address_space(3))) double lds_d;
    const volatile lds_d *p = (const volatile lds_d *)(T + (threadIdx.x & 63) * stride);
    double acc = 0;
    for (int i = 0; i < iters; ++i) {
        double v0 = p[0], v1 = p[256], v2 = p[512], v3 = p[768], v4 = p[1024], v5 = p[1280], v6 = p[1536], v7 = p[1792];
        acc += v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

static void launch(int op, int blocks, double *out, unsigned long long *st, int iters)
{
    if (op == 0) hipLaunchKernelGGL(k_rate<0>, dim3(blocks), dim3(256), 0, 0, out, st, 1e-9, 1.0000001, iters);
    if (op == 1) hipLaunchKernelGGL(k_rate<1>, dim3(blocks), dim3(256), 0, 0, out, st, 1e-9, 1.0000001, iters);
    if (op == 2) hipLaunchKernelGGL(k_rate<2>, dim3(blocks), dim3(256), 0, 0, out, st, 1e-9, 1.0000001, iters);
}

int main()
{
    double *out;
    unsigned long long *st;
    const int iters = 20000, blocks = 256 * 8;     // 32 waves per CU, all resident at once
    hipMalloc(&out, sizeof(double) * 256 * 4096);
    hipMalloc(&st, sizeof(unsigned long long) * 2 * blocks);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const char *names[] = {"v_add_f64", "v_mul_f64", "v_fma_f64"};
    std::vector<unsigned long long> h(2 * blocks);
    for (int op = 0; op < 3; ++op) {
        // >= 2 s of back-to-back launches first: the chip settles on its clock under this load
        hipEventRecord(e0);
        for (int rep = 0; rep < 800; ++rep) launch(op, blocks, out, st, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float warm; hipEventElapsedTime(&warm, e0, e1);
        hipEventRecord(e0);
        launch(op, blocks, out, st, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h.data(), st, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost);
        std::vector<double> ghz(blocks), cyc(blocks);
        for (int b = 0; b < blocks; ++b) {
            ghz[b] = (double)h[2 * b] / (double)h[2 * b + 1] * 0.1;
            cyc[b] = (double)h[2 * b] / ((double)iters * 8 * 2);       // 2 waves of the workgroup share each SIMD
        }
        std::sort(ghz.begin(), ghz.end()); std::sort(cyc.begin(), cyc.end());
        double insts = (double)blocks * 4 * iters * 8;      // wave-instructions
        double rate = insts / (ms * 1e-3);
        printf("%s: %.3f ms (after %.1f s warm), %.4g wave-instr/s chip-wide, in-kernel clock median %.3f GHz "
               "(p10 %.3f, p90 %.3f) => %.2f clk per wave-instr per SIMD (events x clock), %.2f (stamps, per SIMD with 8 waves)\n",
               names[op], ms, warm * 1e-3, rate, ghz[blocks / 2], ghz[blocks / 10], ghz[blocks * 9 / 10],
               1024.0 * ghz[blocks / 2] * 1e9 / rate, cyc[blocks / 2] / 4.0);
    }
    for (int stride = 1; stride <= 2; ++stride)
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_lds, dim3(256 * 4), dim3(512), 0, 0, out, 4000, stride);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double insts = (double)256 * 4 * 8 * 4000 * 8;
        if (rep) printf("ds_read_b64 stride %d: %.3f ms, LDS cycles per wave-instr per CU at 2.4GHz: %.2f\n", stride, ms, 256 * 2.4e9 / (insts / (ms * 1e-3)));
    }
    return 0;
}
