// Microbenchmark: sustained rate of conflict-free ds_read_b64 (64 consecutive
// rows, like phase B of the column kernel) at the column kernel's occupancy
// (2 workgroups x 512 threads per CU, 64 KiB LDS each).
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int READS_PER_ITER, bool WITH_FMA>
__global__ void __launch_bounds__(512, 4) k_lds(double *out, int iters, int spread)
{
    __shared__ double T[8192];                     // 64 KiB
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) T[i] = i * 0.001;
    __syncthreads();
    typedef __attribute__((address_space(3))) double lds_d;
    const int lane = threadIdx.x & 63;
    const volatile lds_d *p = (const volatile lds_d *)(T + lane * spread + (threadIdx.x >> 6));
    double acc = 0, a = 0.5, b = 0.25;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < READS_PER_ITER; r += 2) {
            const double lo = p[r * 128];
            const double hi = p[r * 128 + 1];
            if (WITH_FMA) acc = fma(a, hi, fma(b, lo, acc));
            else acc += lo + hi;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int R, bool F>
static void run(const char *name, double *out, int spread)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 2000, blocks = 512;
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k_lds<R, F>), dim3(blocks), dim3(512), 0, 0, out, iters, spread);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        double insts = (double)blocks * 8 * iters * R;     // wave-level ds_read_b64
        if (rep) printf("%s spread %d: %.3f ms  %.3g ds_read_b64/s chip-wide, %.1f B/clk/CU at 2.4 GHz, err=%s\n", name, spread,
                        ms, insts / (ms * 1e-3), insts * 512 / (ms * 1e-3) / 256 / 2.4e9, hipGetErrorString(hipGetLastError()));
    }
}

int main()
{
    double *out;
    (void)hipMalloc(&out, sizeof(double) * 512 * 512);
    run<32, false>("32 reads/iter, add  ", out, 1);
    run<32, true>("32 reads/iter, 2 fma", out, 1);
    run<32, true>("32 reads/iter, 2 fma", out, 2);
    run<64, true>("64 reads/iter, 2 fma", out, 1);
    return 0;
}
