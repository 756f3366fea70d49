// fp64_rate.hip -- measurement aid: cycles per wave64 FP64 instruction on gfx950, one SIMD, 1 or 2 waves.
//   hipcc --offload-arch=gfx950 -O3 -o fp64_rate bench/fp64_rate.hip && ./fp64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int OP>
__global__ __launch_bounds__(512) void k(double* out, long long* cyc, double a, double b, int iters) {
    double r[8];
    for (int j = 0; j < 8; ++j) r[j] = a + j * 1e-3 + threadIdx.x * 1e-6;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (OP == 0) r[j] = __builtin_fma(r[j], b, a);          // v_fma_f64
            if (OP == 1) r[j] = r[j] * b;                            // v_mul_f64
            if (OP == 2) r[j] = r[j] + b;                            // v_add_f64
            if (OP == 3) r[j] = a / r[j];                            // IEEE division sequence
            if (OP == 4) r[j] = __builtin_amdgcn_rcp(r[j]);          // v_rcp_f64
            if (OP == 5) r[j] = sin(r[j]);                           // ocml sin
            if (OP == 6) r[j] = __builtin_fmaf((float)r[j], 1.0001f, 0.5f);  // f32 fma for reference
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int j = 0; j < 8; ++j) s += r[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x % 64 == 0) cyc[blockIdx.x * 8 + threadIdx.x / 64] = t1 - t0;
}

template <int OP>
int run(const char* name, int iters) {
    double* out; long long* cyc;
    CK(hipMalloc(&out, 512 * 8)); CK(hipMalloc(&cyc, 64));
    for (int waves : {4, 8}) {  // 4 waves = 1 per SIMD of one CU, 8 = 2 per SIMD
        k<OP><<<1, waves * 64>>>(out, cyc, 1.000001, 0.999999, iters);
        CK(hipDeviceSynchronize());
        long long h[8]; CK(hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost));
        double c = 0; for (int i = 0; i < waves; ++i) c += h[i]; c /= waves;
        printf("%-10s %d wave(s)/SIMD: %.1f cycles per wave-instruction (per wave), %.1f per SIMD\n", name, waves / 4,
               c / (iters * 8.0), c / (iters * 8.0) / (waves / 4));
    }
    return 0;
}

int main() {
    run<0>("fma_f64", 2000); run<1>("mul_f64", 2000); run<2>("add_f64", 2000); run<3>("div_f64", 300);
    run<4>("rcp_f64", 1000); run<5>("sin_f64", 100); run<6>("fma_f32", 2000);
    return 0;
}
