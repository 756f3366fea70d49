// structural_floor.hip -- measurement aid (not part of the product): what the MI355X sustains for the MEMORY SHAPE of the
// structural-format launch at config 3 -- per workgroup (one wave): read 795 doubles of Z, write 722 doubles of c and
// 2 430 doubles of vals (three separate buffers, XCD-contiguous block -> problem map, the evaluator's 16-B-per-lane drain)
// -- with no arithmetic at all, and with a stand-in for the arithmetic (a dependent FMA chain of a given length between
// the load and the stores).  If the kernel without arithmetic is no faster than the real one, the real one sits on the
// memory system's ceiling for this read/write mix and launch length; if it is, the difference is what overlap could win.
//   hipcc --offload-arch=gfx950 -O3 -o structural_floor bench/structural_floor.hip && ./structural_floor
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

typedef double v2f64 __attribute__((ext_vector_type(2)));
constexpr int kZ = 795, kC = 722, kCs = 736 /* c_off stride: 722 rounded up to 16 */, kV = 2430, kVs = 3520 /* j stride: nnz 3519 -> 16 */;

template <bool NT>
__device__ __forceinline__ void st16(v2f64* p, v2f64 v) {
    if (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

// MODE 0: read + both writes; 1: no read (writes only); 2: read + vals only; 3: read + c only
template <int MODE, bool NT, int WAVES_PER_CU_LDS>
__global__ __launch_bounds__(64) void k_shape(const double* __restrict__ Z, double* __restrict__ C, double* __restrict__ V, int B,
                                              int chain) {
    extern __shared__ double2 tile2[];
    double* tile = reinterpret_cast<double*>(tile2);
    const int lane = threadIdx.x;
    const int per = (B + 7) >> 3;
    const int b = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (b >= B) return;
    double acc = 0.0;
    if (MODE != 1) {
        const double* zb = Z + (size_t)b * kZ;
        double r[13];
#pragma unroll
        for (int it = 0; it < 13; ++it) r[it] = zb[min(it * 64 + lane, kZ - 1)];
#pragma unroll
        for (int it = 0; it < 13; ++it) tile[it * 64 + lane] = r[it];
        acc = r[0];
    }
    // stand-in for the arithmetic: a dependent chain (one FP64 FMA per link)
    for (int i = 0; i < chain; ++i) acc = fma(acc, 1.0000001, 1e-9);
    for (int i = lane; i < 1300; i += 64) tile2[i] = make_double2(acc, 2.0);
    __syncthreads();
    if (MODE == 0 || MODE == 1 || MODE == 3) {
        v2f64* d = reinterpret_cast<v2f64*>(C + (size_t)b * kCs);
        for (int i = lane; i < kC / 2; i += 64) st16<NT>(d + i, v2f64{tile2[i].x, tile2[i].y});
    }
    if (MODE == 0 || MODE == 1 || MODE == 2) {
        v2f64* d = reinterpret_cast<v2f64*>(V + (size_t)b * kVs);
        for (int i = lane; i < kV / 2; i += 64) st16<NT>(d + i, v2f64{tile2[i].x, tile2[i].y});
    }
}

// P consecutive problems per workgroup, one after the other (the next problem's slice requested before this one's stores):
// the same bytes, 1/P as many workgroups, every wave's write fronts P times as long
template <int P, bool NT>
__global__ __launch_bounds__(64) void k_shape_multi(const double* __restrict__ Z, double* __restrict__ C, double* __restrict__ V, int B,
                                                    int chain) {
    extern __shared__ double2 tile2[];
    double* tile = reinterpret_cast<double*>(tile2);
    const int lane = threadIdx.x;
    const int G = B / P;
    const int per = (G + 7) >> 3;
    const int g = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (g >= G) return;
    double r[13];
    {
        const double* zb = Z + (size_t)(g * P) * kZ;
#pragma unroll
        for (int it = 0; it < 13; ++it) r[it] = zb[min(it * 64 + lane, kZ - 1)];
    }
    for (int q = 0; q < P; ++q) {
        const int b = g * P + q;
#pragma unroll
        for (int it = 0; it < 13; ++it) tile[it * 64 + lane] = r[it];
        double acc = r[0];
        if (q + 1 < P) {
            const double* zb = Z + (size_t)(b + 1) * kZ;
#pragma unroll
            for (int it = 0; it < 13; ++it) r[it] = zb[min(it * 64 + lane, kZ - 1)];
        }
        for (int i = 0; i < chain; ++i) acc = fma(acc, 1.0000001, 1e-9);
        for (int i = lane; i < 1300; i += 64) tile2[i] = make_double2(acc, 2.0);
        __syncthreads();
        v2f64* d = reinterpret_cast<v2f64*>(C + (size_t)b * kCs);
        for (int i = lane; i < kC / 2; i += 64) st16<NT>(d + i, v2f64{tile2[i].x, tile2[i].y});
        d = reinterpret_cast<v2f64*>(V + (size_t)b * kVs);
        for (int i = lane; i < kV / 2; i += 64) st16<NT>(d + i, v2f64{tile2[i].x, tile2[i].y});
        __syncthreads();
    }
}

template <class K>
float time_ms(K launch, int iters = 20) {
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    std::vector<float> t;
    for (int i = 0; i < iters; ++i) {
        CK(hipEventRecord(a));
        launch();
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

int main(int argc, char** argv) {
    const int B = 65536;
    double *Z, *C, *V;
    CK(hipMalloc(&Z, (size_t)B * kZ * 8));
    CK(hipMalloc(&C, (size_t)B * kCs * 8));
    CK(hipMalloc(&V, (size_t)B * kVs * 8));
    CK(hipMemset(Z, 0, (size_t)B * kZ * 8));
    const double gb_all = (double)B * (kZ + kC + kV) * 8 / 1e9, gb_w = (double)B * (kC + kV) * 8 / 1e9;
    const unsigned grid = 8u * ((B + 7) / 8);
    printf("structural-format memory shape, config 3 (B = %d): read %.3f GB, write %.3f GB per launch; plain hipMalloc buffers\n", B,
           (double)B * kZ * 8 / 1e9, gb_w);
    for (int lds_kb : {19, 22, 38}) {
        const size_t lds = (size_t)lds_kb * 1024;
        printf("-- %d KB of LDS per workgroup (%d waves per CU)\n", lds_kb, (int)(160 / lds_kb));
        auto run = [&](const char* name, auto kern, double gb, int chain) {
            CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            float ms = time_ms([&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(64), lds, 0, Z, C, V, B, chain); });
            printf("  %-58s chain %5d : %.4f ms  %.2f TB/s\n", name, chain, ms, gb / ms);
        };
        for (int chain : {0, 500, 1500}) {
            run("read Z + write c + write vals, non-temporal stores", k_shape<0, true, 0>, gb_all, chain);
            run("read Z + write c + write vals, plain stores", k_shape<0, false, 0>, gb_all, chain);
        }
        run("writes only (c + vals), non-temporal", k_shape<1, true, 0>, gb_w, 0);
        run("read Z + write vals only, non-temporal", k_shape<2, true, 0>, (double)B * (kZ + kV) * 8 / 1e9, 0);
        run("read Z + write c only, non-temporal", k_shape<3, true, 0>, (double)B * (kZ + kC) * 8 / 1e9, 0);
    }
    printf("-- P consecutive problems per workgroup (non-temporal stores, no arithmetic)\n");
    for (int lds_kb : {19, 38}) {
        const size_t lds = (size_t)lds_kb * 1024;
        auto runm = [&](const char* name, auto kern, int P) {
            CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            const unsigned gridm = 8u * ((B / P + 7) / 8);
            float ms = time_ms([&] { hipLaunchKernelGGL(kern, dim3(gridm), dim3(64), lds, 0, Z, C, V, B, 0); });
            printf("  %d KB LDS, %-40s : %.4f ms  %.2f TB/s\n", lds_kb, name, ms, gb_all / ms);
        };
        runm("P = 1", k_shape_multi<1, true>, 1);
        runm("P = 2", k_shape_multi<2, true>, 2);
        runm("P = 4", k_shape_multi<4, true>, 4);
        runm("P = 8", k_shape_multi<8, true>, 8);
        runm("P = 16", k_shape_multi<16, true>, 16);
    }
    return 0;
}
