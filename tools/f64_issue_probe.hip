// Diagnostic (not part of the product): how fast does ONE compute unit of an MI355X issue binary64 VALU instructions, as a
// function of the number of waves of the block (1, 2, 4, 8) and of the instruction-level parallelism inside a wave (a single
// dependent chain, or four independent ones)? The planner's stage kernels and the single-call kernel are chains of dependent
// f64 operations on few waves per CU, so this is the figure that bounds them.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o f64_issue_probe f64_issue_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int kIters = 4096;

template <int ILP, int OP>   // OP 0 = fma, 1 = mul then add (two instructions per step), 2 = 32-bit fma
__global__ void chain(double* out, unsigned long long* ticks, double a, double b)
{
    double x[ILP];
    float xf[ILP];
#pragma unroll
    for (int k = 0; k < ILP; ++k) { x[k] = threadIdx.x * 1e-3 + k; xf[k] = (float)x[k]; }
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    const unsigned long long c0 = clock64();
    for (int i = 0; i < kIters; ++i) {
#pragma unroll
        for (int k = 0; k < ILP; ++k) {
            if (OP == 0) x[k] = __builtin_fma(x[k], a, b);
            else if (OP == 1) x[k] = x[k] * a + b;
            else xf[k] = __builtin_fmaf(xf[k], (float)a, (float)b);
        }
    }
    const unsigned long long c1 = clock64();
    const unsigned long long t1 = wall_clock64();
    double s = 0;
#pragma unroll
    for (int k = 0; k < ILP; ++k) s += x[k] + xf[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) {
        ticks[(blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64) * 2] = t1 - t0;
        ticks[(blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64) * 2 + 1] = c1 - c0;
    }
}

template <int ILP, int OP>
static void run(const char* what, int instr_per_step)
{
    double* out;
    unsigned long long* ticks;
    CK(hipMalloc((void**)&out, 8 * 1024 * sizeof(double)));
    CK(hipMalloc((void**)&ticks, 64 * sizeof(unsigned long long)));
    for (int waves : {1, 2, 4, 8, 16}) {
        unsigned long long h[64];
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL((chain<ILP, OP>), dim3(1), dim3(64 * waves), 0, nullptr, out, ticks, 1.0000001, 1e-9);
            CK(hipDeviceSynchronize());
        }
        CK(hipMemcpy(h, ticks, sizeof(unsigned long long) * 2 * waves, hipMemcpyDeviceToHost));
        unsigned long long worst = 0, worst_c = 0;
        for (int w = 0; w < waves; ++w) { worst = h[2 * w] > worst ? h[2 * w] : worst; worst_c = h[2 * w + 1] > worst_c ? h[2 * w + 1] : worst_c; }
        const double instr = (double)kIters * ILP * instr_per_step;
        printf("%-34s ILP %d, %2d waves in one block: %7.2f ns per instruction and wave (%5.2f shader clocks), CU total %6.2f instr/us\n", what, ILP, waves,
               worst * 10.0 / instr, (double)worst_c / instr, instr * waves / (worst * 0.01));
    }
    CK(hipFree(out));
    CK(hipFree(ticks));
}

int main()
{
    run<1, 0>("v_fma_f64, one dependent chain", 1);
    run<4, 0>("v_fma_f64, four chains", 1);
    run<1, 1>("v_mul_f64 + v_add_f64, one chain", 2);
    run<4, 1>("v_mul_f64 + v_add_f64, four chains", 2);
    run<1, 2>("v_fma_f32, one dependent chain", 1);
    run<4, 2>("v_fma_f32, four chains", 1);
    return 0;
}
