// Microbenchmark: issue rate of v_add_f64 / v_mul_f64 / v_fma_f64 and of
// conflict-free ds_read_b64 on gfx950 (answers: what bounds the column kernel).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int OP>
__global__ void __launch_bounds__(256) k_rate(double *out, double a, double b, int iters)
{
    double x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < iters; ++i) {
#define STEP(x)                                                                        \
        if (OP == 0) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x) : "v"(a));          \
        else if (OP == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(b));     \
        else asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(a));
        STEP(x0) STEP(x1) STEP(x2) STEP(x3) STEP(x4) STEP(x5) STEP(x6) STEP(x7)
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

int main()
{
    double *out;
    hipMalloc(&out, sizeof(double) * 256 * 4096);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000, blocks = 256 * 8;     // 32 waves per CU
    const char *names[] = {"v_add_f64", "v_mul_f64", "v_fma_f64"};
    for (int op = 0; op < 3; ++op) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (op == 0) hipLaunchKernelGGL(k_rate<0>, dim3(blocks), dim3(256), 0, 0, out, 1e-9, 1.0000001, iters);
            if (op == 1) hipLaunchKernelGGL(k_rate<1>, dim3(blocks), dim3(256), 0, 0, out, 1e-9, 1.0000001, iters);
            if (op == 2) hipLaunchKernelGGL(k_rate<2>, dim3(blocks), dim3(256), 0, 0, out, 1e-9, 1.0000001, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double insts = (double)blocks * 4 * iters * 8;      // wave-instructions
            if (rep) printf("%s: %.3f ms, %.3g wave-instr/s, cycles per wave-instr per SIMD at 2.4GHz: %.2f\n", names[op], ms,
                            insts / (ms * 1e-3), 1024 * 2.4e9 / (insts / (ms * 1e-3)));
        }
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
