// store_ceiling.hip -- measurement aid (not part of the product): what the MI355X sustains for the
// store shapes the evaluator uses, and a known-byte-count calibration for FETCH_SIZE/WRITE_SIZE.
//   hipcc --offload-arch=gfx950 -O3 -o store_ceiling bench/store_ceiling.hip && ./store_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

// A: flat grid-stride fill, 16 B/lane
__global__ void k_fill_flat(double2* __restrict__ p, size_t n2) {
    const double2 v = make_double2(1.0, 2.0);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
// B: one wave per contiguous region of `region2` double2 (the evaluator's shape: 1 wave = 1 problem)
__global__ __launch_bounds__(64) void k_fill_region(double2* __restrict__ p, int region2, size_t stride2) {
    const double2 v = make_double2(1.0, 2.0);
    double2* q = p + blockIdx.x * stride2;
    for (int i = threadIdx.x; i < region2; i += 64) q[i] = v;
}
// C: copy with 8 B/lane loads and stores (calibration of FETCH_SIZE for the Z staging loads)
__global__ void k_copy8(const double* __restrict__ a, double* __restrict__ b, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
// D: copy with 16 B/lane
__global__ void k_copy16(const double2* __restrict__ a, double2* __restrict__ b, size_t n2) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}

template <class F> float time_ms(F f, int iters = 10) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    std::vector<float> t;
    for (int i = 0; i < iters; ++i) { CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); t.push_back(ms); }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

int main() {
    const int B = 65536;
    const size_t region = 11740, stride = 12880;            // doubles: dynamic Jacobian values / padded nnz per problem
    const size_t nbytes = (size_t)B * stride * 8;
    double *buf, *src;
    CK(hipMalloc(&buf, nbytes));
    CK(hipMalloc(&src, (size_t)B * 800 * 8));
    CK(hipMemset(buf, 0, nbytes)); CK(hipMemset(src, 0, (size_t)B * 800 * 8));
    const size_t wbytes = (size_t)B * region * 8;
    for (int blocks : {2048, 4096, 8192}) {
        float ms = time_ms([&] { k_fill_flat<<<blocks, 256>>>((double2*)buf, wbytes / 16); });
        printf("fill_flat   blocks=%5d x256: %.3f ms  %.1f GB/s\n", blocks, ms, wbytes / ms / 1e6);
    }
    {
        float ms = time_ms([&] { k_fill_region<<<B, 64>>>((double2*)buf, (int)(region / 2), stride / 2); });
        printf("fill_region 1 wave/problem   : %.3f ms  %.1f GB/s\n", ms, wbytes / ms / 1e6);
    }
    const size_t n = (size_t)B * 795;
    {
        float ms = time_ms([&] { k_copy8<<<4096, 256>>>(src, buf, n); });
        printf("copy8  %zu B read + write : %.3f ms  %.1f GB/s (r+w)\n", n * 8, ms, 2.0 * n * 8 / ms / 1e6);
        ms = time_ms([&] { k_copy16<<<4096, 256>>>((double2*)src, (double2*)buf, n / 2); });
        printf("copy16 %zu B read + write : %.3f ms  %.1f GB/s (r+w)\n", n / 2 * 16, ms, 2.0 * (n / 2) * 16 / ms / 1e6);
    }
    // big copy for an HBM-resident calibration (3.2 GB read)
    {
        const size_t nb = (size_t)400 * 1000 * 1000;  // doubles
        float ms = time_ms([&] { k_copy8<<<8192, 256>>>(buf, buf + nb, nb); }, 5);
        printf("copy8_big  %zu B read + write : %.3f ms  %.1f GB/s (r+w)\n", nb * 8, ms, 2.0 * nb * 8 / ms / 1e6);
        ms = time_ms([&] { k_copy16<<<8192, 256>>>((double2*)buf, (double2*)(buf + nb), nb / 2); }, 5);
        printf("copy16_big %zu B read + write : %.3f ms  %.1f GB/s (r+w)\n", nb * 8, ms, 2.0 * nb * 8 / ms / 1e6);
    }
    return 0;
}
