// Scratch micro-benchmark: issue rate of fp64 vector instructions (gfx950).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rate tools/valu_rate.hip && /tmp/valu_rate
// One workgroup per CU, W waves per SIMD; every wave runs N rounds of ILP independent
// chains of one instruction kind; cycles from s_memtime / wall clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int KIND, int ILP>
__global__ void __launch_bounds__(1024) k_rate(double* out, int rounds, double a, double b) {
    double v[ILP];
#pragma unroll
    for (int i = 0; i < ILP; ++i) v[i] = a + i + threadIdx.x;
    for (int r = 0; r < rounds; ++r) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < ILP; ++i) {
                if (KIND == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b));
                if (KIND == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(v[i]) : "v"(a));
                if (KIND == 2) asm volatile("v_add_f64 %0, %0, %1" : "+v"(v[i]) : "v"(b));
                if (KIND == 3) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(*(float*)&v[i]) : "v"((float)a), "v"((float)b));
            }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < ILP; ++i) s += v[i];
    if (s == 12345.678) out[0] = s;
}

template <int KIND, int ILP>
static void run(const char* name, int waves_per_simd) {
    double* out; hipMalloc(&out, 8);
    const int rounds = 2000, threads = 256 * waves_per_simd, blocks = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k_rate<KIND, ILP><<<blocks, threads>>>(out, 10, 1.0000001, 1e-9);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k_rate<KIND, ILP><<<blocks, threads>>>(out, rounds, 1.0000001, 1e-9);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)rounds * 8 * ILP * waves_per_simd;
    printf("%-10s ILP %d, %d wave(s) per SIMD: %.2f ns per wave instruction per SIMD (%.2f cycles at 2.4 GHz)\n",
           name, ILP, waves_per_simd, ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
    hipFree(out);
}

int main() {
    run<0, 1>("fma_f64", 1); run<0, 2>("fma_f64", 1); run<0, 4>("fma_f64", 1); run<0, 8>("fma_f64", 1);
    run<0, 1>("fma_f64", 2); run<0, 4>("fma_f64", 2); run<0, 8>("fma_f64", 2);
    run<1, 8>("mul_f64", 1); run<1, 8>("mul_f64", 2);
    run<2, 8>("add_f64", 1); run<2, 8>("add_f64", 2);
    run<3, 8>("fma_f32", 1); run<3, 8>("fma_f32", 2);
    return 0;
}
