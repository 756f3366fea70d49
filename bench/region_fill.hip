// region_fill.hip -- measurement aid (not part of the product): is the speed class of the evaluator's store stream a
// property of the REGION of device memory, and does it help to give every XCD's write front its own region?
// One large allocation; the evaluator's store shape (one wave per 103 040-byte problem segment, XCD-contiguous
// map: XCD x fills problems [x*8192, (x+1)*8192)) with XCD x's range placed at slab offset off[x].
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/region_fill bench/region_fill.hip && /tmp/region_fill [slab_GiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

struct Offs { long long o[8]; };  // byte offset of every XCD's range inside the slab

__global__ __launch_bounds__(64) void k_fill(char* slab, Offs offs, int per_xcd, int region2, size_t stride2) {
    const int x = blockIdx.x & 7, i = blockIdx.x >> 3;
    double2* q = reinterpret_cast<double2*>(slab + offs.o[x]) + (size_t)i * stride2;
    const double2 v = make_double2(1.0, 2.0);
    for (int j = threadIdx.x; j < region2; j += 64) q[j] = v;
}

// flat grid-stride fill of one contiguous range (what hipMemset-like code does): the ceiling of any layout
__global__ void k_flat(double2* p, size_t n2) {
    const double2 v = make_double2(1.0, 2.0);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

// the same shape, reading instead of writing (does a consumer of the Jacobian see the regions too?)
__global__ __launch_bounds__(64) void k_read(const char* slab, Offs offs, int per_xcd, int region2, size_t stride2, double* sink) {
    const int x = blockIdx.x & 7, i = blockIdx.x >> 3;
    const double2* q = reinterpret_cast<const double2*>(slab + offs.o[x]) + (size_t)i * stride2;
    double acc = 0.0;
    for (int j = threadIdx.x; j < region2; j += 64) {
        const double2 v = q[j];
        acc += v.x + v.y;
    }
    if (acc == 1.2345e300) sink[0] = acc;  // never true; keeps the loads
}

int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IOLBF, 0);
    const double gib = argc > 1 ? atof(argv[1]) : 192;
    const size_t slab_bytes = (size_t)(gib * (1ull << 30));
    const int B = 65536, per_xcd = B / 8;
    const size_t region = 11740, stride = 12880;  // doubles written / doubles between problems
    const size_t range_bytes = (size_t)per_xcd * stride * 8;  // 0.79 GiB per XCD
    char* slab;
    CK(hipMalloc(&slab, slab_bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const bool reading = getenv("READ") != nullptr;
    double* sink;
    CK(hipMalloc(&sink, 8));
    if (reading) { CK(hipMemset(slab, 0, slab_bytes)); printf("READ mode\n"); }
    auto run = [&](const Offs& o) {
        std::vector<float> t;
        for (int r = 0; r < 6; ++r) {
            CK(hipEventRecord(e0));
            if (reading) k_read<<<B, 64>>>(slab, o, per_xcd, (int)(region / 2), stride / 2, sink);
            else k_fill<<<B, 64>>>(slab, o, per_xcd, (int)(region / 2), stride / 2);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 2) t.push_back(ms);
        }
        std::sort(t.begin(), t.end());
        return t[t.size() / 2];
    };
    const double wbytes = (double)B * region * 8;
    printf("slab %.0f GiB; one XCD range = %.3f GiB; written per launch %.2f GB\n", gib, range_bytes / 1073741824.0, wbytes / 1e9);
    // (1) the eight ranges back to back (what one contiguous vals buffer is), at increasing slab offsets
    const double step = getenv("SWEEP") ? 48.0 : 8.0;
    std::vector<double> cls;
    for (double s = 0; (s + 6.5) <= gib; s += step) {
        Offs o;
        for (int x = 0; x < 8; ++x) o.o[x] = (long long)(s * (1ull << 30)) + (long long)x * range_bytes;
        float ms = run(o);
        cls.push_back(wbytes / ms / 1e6);
        printf("contiguous at %6.1f GiB: %.3f ms  %.0f GB/s\n", s, ms, wbytes / ms / 1e6);
    }
    // (2) one range per region: XCD x at x * spread
    for (double spread : {1.0, 4.0, 8.0, 16.0, 23.0}) {
        if (7 * spread + 1 > gib) continue;
        for (double base : {0.0, 5.0}) {
            Offs o;
            for (int x = 0; x < 8; ++x) o.o[x] = (long long)((base + x * spread) * (1ull << 30));
            float ms = run(o);
            printf("spread %5.1f GiB apart, base %4.1f GiB: %.3f ms  %.0f GB/s\n", spread, base, ms, wbytes / ms / 1e6);
        }
    }
    // (2b) finer: which spacings help?
    if (getenv("SWEEP")) {
        for (double spread : {2.0, 6.0, 9.0, 10.0, 11.0, 12.0, 13.0, 14.0, 15.0, 16.0, 17.0, 18.0, 20.0, 24.0}) {
            if (7 * spread + 1 > gib) continue;
            Offs o;
            for (int x = 0; x < 8; ++x) o.o[x] = (long long)((3.0 + x * spread) * (1ull << 30));
            float ms = run(o);
            printf("spread %5.1f GiB apart: %.3f ms  %.0f GB/s\n", spread, ms, wbytes / ms / 1e6);
        }
        // two groups of four back-to-back ranges, D GiB apart
        for (double D : {4.0, 8.0, 12.0, 16.0, 24.0, 32.0, 48.0, 64.0, 96.0, 128.0}) {
            if (D + 4 > gib) continue;
            Offs o;
            for (int x = 0; x < 8; ++x) o.o[x] = (long long)(((x & 1) ? D : 0.0) * (1ull << 30)) + (long long)(x >> 1) * range_bytes;
            float ms = run(o);
            printf("two groups of four, %5.1f GiB apart: %.3f ms  %.0f GB/s\n", D, ms, wbytes / ms / 1e6);
        }
        // four groups of two
        for (double D : {8.0, 16.0, 32.0, 40.0}) {
            if (3 * D + 2 > gib) continue;
            Offs o;
            for (int x = 0; x < 8; ++x) o.o[x] = (long long)(((x & 3) * D) * (1ull << 30)) + (long long)(x >> 2) * range_bytes;
            float ms = run(o);
            printf("four groups of two, %5.1f GiB apart: %.3f ms  %.0f GB/s\n", D, ms, wbytes / ms / 1e6);
        }
    }
    if (getenv("FLAT")) {
        // flat fill of 6.29 GiB starting at offset s: inside one region, and across a boundary
        for (double sgib = 0; sgib + 6.5 <= gib && sgib <= 72; sgib += 2.0) {
            std::vector<float> t;
            for (int r = 0; r < 5; ++r) {
                CK(hipEventRecord(e0));
                k_flat<<<4096, 256>>>(reinterpret_cast<double2*>(slab + (size_t)(sgib * (1ull << 30))), (size_t)8 * range_bytes / 16);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (r >= 1) t.push_back(ms);
            }
            std::sort(t.begin(), t.end());
            printf("flat fill at %5.1f GiB: %.3f ms  %.0f GB/s\n", sgib, t[t.size() / 2], 8.0 * range_bytes / t[t.size() / 2] / 1e6);
        }
    }
    // (3) all eight ranges inside the best and inside the worst 8-GiB window found in (1)
    const int best = (int)(std::max_element(cls.begin(), cls.end()) - cls.begin());
    const int worst = (int)(std::min_element(cls.begin(), cls.end()) - cls.begin());
    printf("best window at %.0f GiB (%.0f GB/s), worst at %.0f GiB (%.0f GB/s)\n", best * step, cls[best], worst * step, cls[worst]);
    // (4) half of the XCDs in the best window, half in the worst
    {
        Offs o;
        for (int x = 0; x < 8; ++x)
            o.o[x] = (long long)(((x & 1) ? best : worst) * step * (1ull << 30)) + (long long)(x >> 1) * range_bytes;
        float ms = run(o);
        printf("XCDs alternating between best and worst window: %.3f ms  %.0f GB/s\n", ms, wbytes / ms / 1e6);
    }
    return 0;
}
