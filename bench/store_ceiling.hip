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
// B2: the evaluator's store structure without its arithmetic: one wave per region, a T*2400-B LDS tile
// (limits residency exactly like the real kernel), sub-tiles drained with ds_read_b128 + 1-KiB stores
template <int T>
__global__ __launch_bounds__(64) void k_fill_region_lds(double2* __restrict__ p, int nknots, size_t stride2) {
    __shared__ double2 tile[T * 150];
    for (int i = threadIdx.x; i < T * 150; i += 64) tile[i] = make_double2(1.0, 2.0);
    __syncthreads();
    double2* q = p + blockIdx.x * stride2;
    for (int k0 = 0; k0 < nknots; k0 += T) {
        const int np = min(T, nknots - k0) * 150;
        double2* d = q + (size_t)k0 * 150;
        for (int i = threadIdx.x; i < np; i += 64) d[i] = tile[i];
    }
}

// B3: cache-policy variants of B (one wave per region, 16 B per lane)
typedef double v2f64 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(64) void k_fill_region_policy(double2* __restrict__ p, int region2, size_t stride2) {
    v2f64 v = {1.0, 2.0};
    v2f64* q = reinterpret_cast<v2f64*>(p + blockIdx.x * stride2);
    for (int i = threadIdx.x; i < region2; i += 64) {
        if (MODE == 0) q[i] = v;
        if (MODE == 1) __builtin_nontemporal_store(v, &q[i]);
        if (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(&q[i]), "v"(v) : "memory");
        if (MODE == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(&q[i]), "v"(v) : "memory");
        if (MODE == 4) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(&q[i]), "v"(v) : "memory");
        if (MODE == 5) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(&q[i]), "v"(v) : "memory");
    }
}

// B4: block -> region mappings and cooperative variants (how much does the width of the write front matter?)
//  MAP 0: region = block; MAP 1: XCD-contiguous (blocks b, b+8, .. share an XCD: give each XCD a contiguous range)
template <int MAP, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_fill_region_map(double2* __restrict__ p, int region2, size_t stride2, int nregions) {
    const double2 v = make_double2(1.0, 2.0);
    int r = blockIdx.x;
    if (MAP == 1) r = (blockIdx.x % 8) * (nregions / 8) + blockIdx.x / 8;
    double2* q = p + (size_t)r * stride2;
    // WAVES waves share the region; wave w writes 1200-piece (19.2 KB) sub-tiles w, w+WAVES, ...
    const int w = threadIdx.x / 64, lane = threadIdx.x % 64;
    for (int t0 = w * 1200; t0 < region2; t0 += WAVES * 1200) {
        const int n = min(1200, region2 - t0);
        for (int i = lane; i < n; i += 64) q[t0 + i] = v;
    }
}

// B5: interleaved-batch layout: regions of a group of G problems are stored piece-major (piece = 1 KiB = one
// wave instruction): address = group_base + (piece * G + member) KiB.  Concurrent waves then write adjacent KiBs.
template <int MAP>
__global__ __launch_bounds__(64) void k_fill_interleaved(double2* __restrict__ p, int pieces, int G, int nregions) {
    const double2 v = make_double2(1.0, 2.0);
    int r = blockIdx.x;
    if (MAP == 1) r = (blockIdx.x % 8) * (nregions / 8) + blockIdx.x / 8;
    const int grp = r / G, mem = r % G;
    double2* q = p + ((size_t)grp * pieces * G + mem) * 64 + threadIdx.x;
    for (int i = 0; i < pieces; ++i) q[(size_t)i * G * 64] = v;
}

// B6: the proposed interleaved layout with the evaluator's real store structure: LDS tile of 16 knots (38400 B),
// 512-B pieces (64 doubles), G members per group; one 1-KiB wave store covers two pieces (lanes 0-31 / 32-63).
template <int G>
__global__ __launch_bounds__(64) void k_fill_tile_interleaved(double2* __restrict__ p, int nknots, int nregions) {
    __shared__ double2 tile[16 * 150];
    for (int i = threadIdx.x; i < 16 * 150; i += 64) tile[i] = make_double2(1.0, 2.0);
    __syncthreads();
    const int r = (blockIdx.x % 8) * (nregions / 8) + blockIdx.x / 8;
    const int grp = r / G, mem = r % G;
    const int Q = (nknots * 2400 + 511) / 512;  // pieces per problem
    const int lane = threadIdx.x;
    // double2 index of this lane inside piece (2i + lane/32)
    double2* base = p + ((size_t)grp * Q * G + mem) * 32 + (size_t)(lane >> 5) * G * 32 + (lane & 31);
    for (int k0 = 0; k0 < nknots; k0 += 16) {
        const int nb = min(16, nknots - k0) * 2400;        // bytes in this sub-tile
        const size_t q0 = (size_t)k0 * 2400 / 512;          // first piece (16*2400 = 75 pieces exactly)
        for (int i = 0; i * 1024 < nb; ++i) {
            if (i * 1024 + lane * 16 < nb) base[(q0 + 2 * i) * G * 32] = tile[i * 64 + lane];
        }
    }
}

// B7: chunked XCD map: XCD x owns chunks of CH consecutive regions, chunks dealt round-robin over XCDs
__global__ __launch_bounds__(64) void k_fill_region_chunked(double2* __restrict__ p, int region2, size_t stride2, int CH) {
    const double2 v = make_double2(1.0, 2.0);
    const int x = blockIdx.x & 7, s = blockIdx.x >> 3;       // XCD label, sequence number on that XCD
    const int r = ((s / CH) * 8 + x) * CH + (s % CH);
    double2* q = p + (size_t)r * stride2;
    for (int i = threadIdx.x; i < region2; i += 64) q[i] = v;
}

// B8: XCD-contiguous ranges, each XCD walking its range from a different phase (skew, in regions): the 8 write
// fronts are then not a multiple of a large power of two apart
__global__ __launch_bounds__(64) void k_fill_region_skew(double2* __restrict__ p, int region2, size_t stride2, int nregions, int skew) {
    const double2 v = make_double2(1.0, 2.0);
    const int per = nregions / 8;
    const int x = blockIdx.x & 7, sq = blockIdx.x >> 3;
    const int r = x * per + (sq + x * skew) % per;
    double2* q = p + (size_t)r * stride2;
    for (int i = threadIdx.x; i < region2; i += 64) q[i] = v;
}

// B9: the evaluator's store structure (T=16 tile, XCD-contiguous) with a synthetic non-store phase in front of each
// problem: a dependent 6.4-KB read plus DELAY x 64 clocks of s_sleep.  How much do the evaluator's load/compute
// phases cost at 4 waves per CU if the store stream itself is unchanged?
template <int DELAY>
__global__ __launch_bounds__(64) void k_fill_tile_delay(double2* __restrict__ p, const double* __restrict__ src, int nknots,
                                                        size_t stride2, int nregions) {
    __shared__ double2 tile[16 * 150];
    const int r = (blockIdx.x % 8) * (nregions / 8) + blockIdx.x / 8;
    double acc = 0.0;
    if (DELAY >= 0) {
        const double* z = src + (size_t)r * 795;
        for (int i = threadIdx.x; i < 795; i += 64) acc += z[i];
        for (int d = 0; d < DELAY; ++d) __builtin_amdgcn_s_sleep(64);
    }
    for (int i = threadIdx.x; i < 16 * 150; i += 64) tile[i] = make_double2(1.0 + acc, 2.0);
    __syncthreads();
    double2* q = p + (size_t)r * stride2;
    for (int k0 = 0; k0 < nknots; k0 += 16) {
        const int np = min(16, nknots - k0) * 150;
        double2* d = q + (size_t)k0 * 150;
        for (int i = threadIdx.x; i < np; i += 64) d[i] = tile[i];
    }
}

// B10: persistent waves with the NEXT problem's 6.4-KB read prefetched into registers while the current problem's
// tile stream is stored (1024 single-wave blocks, XCD-contiguous ranges, 128 waves per XCD walking their range)
template <bool PREFETCH, int LDMODE = 0>
__global__ __launch_bounds__(64) void k_fill_tile_persistent(double2* __restrict__ p, const double* __restrict__ src, int nknots,
                                                             size_t stride2, int nregions, int period = 0) {
    __shared__ double2 tile[16 * 150];
    const int per = nregions / 8, x = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
    const int lane = threadIdx.x;
    double zr[13];
    int r = x * per + slot;
    auto issue = [&](int rr) {
        const double* z = src + (size_t)rr * 795;
#pragma unroll
        for (int it = 0; it < 13; ++it) {
            const double* a = z + min(it * 64 + lane, 794);
            if (LDMODE == 0) zr[it] = *a;
            if (LDMODE == 1) zr[it] = __builtin_nontemporal_load(a);
            if (LDMODE == 2) asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(zr[it]) : "v"(a) : "memory");
            if (LDMODE == 3) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1" : "=v"(zr[it]) : "v"(a) : "memory");
        }
        if (LDMODE >= 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // asm loads are not counted by hipcc
    };
    if (slot < per) issue(r);
    for (int s = slot; s < per; s += nslots) {
        r = x * per + s;
        double acc = 0.0;
#pragma unroll
        for (int it = 0; it < 13; ++it) acc += zr[it];   // consume the staged read (waits for it)
        for (int i = lane; i < 16 * 150; i += 64) tile[i] = make_double2(1.0 + acc, 2.0);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (PREFETCH && s + nslots < per) {
            if (period > 0) {
                // time-division experiment: every wave of the chip issues its reads only in the first 1/8 of each
                // `period` ticks of the 100-MHz real-time counter, so the DRAM channels see reads in bursts
                int guard = 0;
                while ((__builtin_amdgcn_s_memrealtime() % (unsigned long long)period) >= (unsigned long long)(period / 8) &&
                       ++guard < 100000)
                    __builtin_amdgcn_s_sleep(2);
            }
            issue(r + nslots);  // next problem's read flies under this problem's stores
        }
        double2* q = p + (size_t)r * stride2;
        for (int k0 = 0; k0 < nknots; k0 += 16) {
            const int np = min(16, nknots - k0) * 150;
            double2* d = q + (size_t)k0 * 150;
            for (int i = lane; i < np; i += 64) d[i] = tile[i];
        }
        if (!PREFETCH && s + nslots < per) issue(r + nslots);
        __builtin_amdgcn_wave_barrier();
    }
}

// B11: the north star's literal mapping, store side only: ONE WAVE PER KNOT POINT, each wave writes its 2400-B block
// (150 x 16 B: two full wave instructions + 22 lanes), blocks of a problem contiguous.  No arithmetic, no reads.
__global__ __launch_bounds__(64) void k_fill_wave_per_knot(double2* __restrict__ p, int nknots, size_t stride2, int nregions) {
    const double2 v = make_double2(1.0, 2.0);
    const int g = blockIdx.x;                       // flat knot index, problem-major
    const int r = g / nknots, k = g - r * nknots;
    double2* q = p + (size_t)r * stride2 + (size_t)k * 150;
    for (int i = threadIdx.x; i < 150; i += 64) q[i] = v;
}
// same, 4 knots (waves) per 256-thread workgroup
__global__ __launch_bounds__(256) void k_fill_wave_per_knot4(double2* __restrict__ p, int nknots, size_t stride2, int nregions) {
    const double2 v = make_double2(1.0, 2.0);
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= nknots * nregions) return;
    const int r = g / nknots, k = g - r * nknots;
    double2* q = p + (size_t)r * stride2 + (size_t)k * 150;
    for (int i = threadIdx.x & 63; i < 150; i += 64) q[i] = v;
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
    setvbuf(stdout, nullptr, _IOLBF, 0);
    const int B = 65536;
    const size_t region = 11740, stride = 12880;            // doubles: dynamic Jacobian values / padded nnz per problem
    const size_t nbytes = (size_t)B * stride * 8;
    double *buf, *src;
    if (getenv("CONTIG")) { CK(hipExtMallocWithFlags((void**)&buf, (size_t)B * 16384 * 8, hipDeviceMallocContiguous)); printf("contiguous allocation\n"); }
    else CK(hipMalloc(&buf, (size_t)B * 16384 * 8));  // room for the pitch sweep
    CK(hipMalloc(&src, (size_t)B * 800 * 8));
    CK(hipMemset(buf, 0, nbytes)); CK(hipMemset(src, 0, (size_t)B * 800 * 8));
    const size_t wbytes = (size_t)B * region * 8;
    for (int blocks : {256, 512, 1024, 2048, 4096, 8192}) {
        float ms = time_ms([&] { k_fill_flat<<<blocks, 256>>>((double2*)buf, wbytes / 16); });
        printf("fill_flat   blocks=%5d x256: %.3f ms  %.1f GB/s\n", blocks, ms, wbytes / ms / 1e6);
    }
    {
        float ms = time_ms([&] { k_fill_region<<<B, 64>>>((double2*)buf, (int)(region / 2), stride / 2); });
        printf("fill_region 1 wave/problem   : %.3f ms  %.1f GB/s\n", ms, wbytes / ms / 1e6);
    }
    {
        float ms = time_ms([&] { k_fill_region_lds<8><<<B, 64>>>((double2*)buf, 39, stride / 2); });
        printf("fill_region_lds T=8  (8 waves/CU) : %.3f ms  %.1f GB/s\n", ms, (double)B * 39 * 2400 / ms / 1e6);
        ms = time_ms([&] { k_fill_region_lds<16><<<B, 64>>>((double2*)buf, 39, stride / 2); });
        printf("fill_region_lds T=16 (4 waves/CU) : %.3f ms  %.1f GB/s\n", ms, (double)B * 39 * 2400 / ms / 1e6);
        ms = time_ms([&] { k_fill_region_lds<4><<<B, 64>>>((double2*)buf, 39, stride / 2); });
        printf("fill_region_lds T=4  (16 waves/CU): %.3f ms  %.1f GB/s\n", ms, (double)B * 39 * 2400 / ms / 1e6);
        ms = time_ms([&] { k_fill_region_lds<2><<<B, 64>>>((double2*)buf, 39, stride / 2); });
        printf("fill_region_lds T=2  (32 waves/CU): %.3f ms  %.1f GB/s\n", ms, (double)B * 39 * 2400 / ms / 1e6);
    }
    for (int rep = 0; rep < 2; ++rep) {
        float ms;
        ms = time_ms([&] { k_fill_region_policy<0><<<B, 64>>>((double2*)buf, (int)(region / 2), stride / 2); });
        printf("fill_region plain      : %.3f ms  %.1f GB/s\n", ms, wbytes / ms / 1e6);
        ms = time_ms([&] { k_fill_region_policy<1><<<B, 64>>>((double2*)buf, (int)(region / 2), stride / 2); });
        printf("fill_region nt         : %.3f ms  %.1f GB/s\n", ms, wbytes / ms / 1e6);
        ms = time_ms([&] { k_fill_region_policy<2><<<B, 64>>>((double2*)buf, (int)(region / 2), stride / 2); });
        printf("fill_region sc1        : %.3f ms  %.1f GB/s\n", ms, wbytes / ms / 1e6);
        ms = time_ms([&] { k_fill_region_policy<3><<<B, 64>>>((double2*)buf, (int)(region / 2), stride / 2); });
        printf("fill_region sc0 sc1    : %.3f ms  %.1f GB/s\n", ms, wbytes / ms / 1e6);
        ms = time_ms([&] { k_fill_region_policy<4><<<B, 64>>>((double2*)buf, (int)(region / 2), stride / 2); });
        printf("fill_region sc1 nt     : %.3f ms  %.1f GB/s\n", ms, wbytes / ms / 1e6);
        ms = time_ms([&] { k_fill_region_policy<5><<<B, 64>>>((double2*)buf, (int)(region / 2), stride / 2); });
        printf("fill_region sc0 sc1 nt : %.3f ms  %.1f GB/s\n", ms, wbytes / ms / 1e6);
        ms = time_ms([&] { k_fill_region_map<1, 1><<<B, 64>>>((double2*)buf, (int)(region / 2), stride / 2, B); });
        printf("fill_region XCD-contig : %.3f ms  %.1f GB/s\n", ms, wbytes / ms / 1e6);
        ms = time_ms([&] { k_fill_region_map<0, 4><<<B, 256>>>((double2*)buf, (int)(region / 2), stride / 2, B); });
        printf("fill_region 4 waves/rgn: %.3f ms  %.1f GB/s\n", ms, wbytes / ms / 1e6);
        ms = time_ms([&] { k_fill_region_map<1, 4><<<B, 256>>>((double2*)buf, (int)(region / 2), stride / 2, B); });
        printf("fill_region 4w + XCD   : %.3f ms  %.1f GB/s\n", ms, wbytes / ms / 1e6);
        ms = time_ms([&] { k_fill_region_map<0, 2><<<B, 128>>>((double2*)buf, (int)(region / 2), stride / 2, B); });
        printf("fill_region 2 waves/rgn: %.3f ms  %.1f GB/s\n", ms, wbytes / ms / 1e6);
        // XCD-contiguous + each XCD's range walked by a tight front: region r of XCD x = x*(B/8) + j
        ms = time_ms([&] { k_fill_region_map<1, 1><<<B, 64>>>((double2*)buf, 1200 * 2, 1200 * 2, B * 2); });
        printf("fill 38KB regions XCDc : %.3f ms  %.1f GB/s\n", ms, (double)B * 2 * 2400 * 16 / ms / 1e6);
        ms = time_ms([&] { k_fill_tile_interleaved<64><<<B, 64>>>((double2*)buf, 39, B); });
        printf("tile-interleaved 512B G=64 : %.3f ms  %.1f GB/s\n", ms, (double)B * 39 * 2400 / ms / 1e6);
        ms = time_ms([&] { k_fill_tile_interleaved<32><<<B, 64>>>((double2*)buf, 39, B); });
        printf("tile-interleaved 512B G=32 : %.3f ms  %.1f GB/s\n", ms, (double)B * 39 * 2400 / ms / 1e6);
        ms = time_ms([&] { k_fill_tile_interleaved<128><<<B, 64>>>((double2*)buf, 39, B); });
        printf("tile-interleaved 512B G=128: %.3f ms  %.1f GB/s\n", ms, (double)B * 39 * 2400 / ms / 1e6);
        ms = time_ms([&] { k_fill_region_lds<16><<<B, 64>>>((double2*)buf, 39, stride / 2); });
        printf("fill_region_lds T=16 rr    : %.3f ms  %.1f GB/s\n", ms, (double)B * 39 * 2400 / ms / 1e6);
        ms = time_ms([&] { k_fill_tile_delay<-1><<<B, 64>>>((double2*)buf, src, 39, stride / 2, B); });
        printf("tile16 XCDc, no front phase       : %.3f ms  %.1f GB/s\n", ms, (double)B * 39 * 2400 / ms / 1e6);
        ms = time_ms([&] { k_fill_tile_delay<0><<<B, 64>>>((double2*)buf, src, 39, stride / 2, B); });
        printf("tile16 XCDc, 6.4 KB read in front : %.3f ms  %.1f GB/s\n", ms, (double)B * 39 * 2400 / ms / 1e6);
        ms = time_ms([&] { k_fill_tile_delay<2><<<B, 64>>>((double2*)buf, src, 39, stride / 2, B); });
        printf("tile16 XCDc, read + 8k clk idle   : %.3f ms  %.1f GB/s\n", ms, (double)B * 39 * 2400 / ms / 1e6);
        ms = time_ms([&] { k_fill_tile_delay<4><<<B, 64>>>((double2*)buf, src, 39, stride / 2, B); });
        printf("tile16 XCDc, read + 16k clk idle  : %.3f ms  %.1f GB/s\n", ms, (double)B * 39 * 2400 / ms / 1e6);
        ms = time_ms([&] { k_fill_tile_delay<8><<<B, 64>>>((double2*)buf, src, 39, stride / 2, B); });
        printf("tile16 XCDc, read + 32k clk idle  : %.3f ms  %.1f GB/s\n", ms, (double)B * 39 * 2400 / ms / 1e6);
        ms = time_ms([&] { k_fill_tile_persistent<false><<<1024, 64>>>((double2*)buf, src, 39, stride / 2, B); });
        printf("tile16 persistent, read after stores : %.3f ms  %.1f GB/s\n", ms, (double)B * 39 * 2400 / ms / 1e6);
        ms = time_ms([&] { k_fill_tile_persistent<true><<<1024, 64>>>((double2*)buf, src, 39, stride / 2, B); });
        printf("tile16 persistent, read prefetched   : %.3f ms  %.1f GB/s\n", ms, (double)B * 39 * 2400 / ms / 1e6);
        for (int period : {200, 400, 800, 1600, 3200}) {
            ms = time_ms([&] { k_fill_tile_persistent<true, 0><<<1024, 64>>>((double2*)buf, src, 39, stride / 2, B, period); });
            printf("tile16 persistent, prefetched, read window every %4d x10ns : %.3f ms  %.1f GB/s\n", period, ms, (double)B * 39 * 2400 / ms / 1e6);
        }
        ms = time_ms([&] { k_fill_tile_persistent<true, 1><<<1024, 64>>>((double2*)buf, src, 39, stride / 2, B); });
        printf("tile16 persistent, prefetched, nt loads : %.3f ms  %.1f GB/s\n", ms, (double)B * 39 * 2400 / ms / 1e6);
        ms = time_ms([&] { k_fill_tile_persistent<false, 2><<<1024, 64>>>((double2*)buf, src, 39, stride / 2, B); });
        printf("tile16 persistent, sc1 loads (waited)   : %.3f ms  %.1f GB/s\n", ms, (double)B * 39 * 2400 / ms / 1e6);
        ms = time_ms([&] { k_fill_tile_persistent<false, 3><<<1024, 64>>>((double2*)buf, src, 39, stride / 2, B); });
        printf("tile16 persistent, sc0 sc1 loads (wait) : %.3f ms  %.1f GB/s\n", ms, (double)B * 39 * 2400 / ms / 1e6);
        ms = time_ms([&] { k_fill_wave_per_knot<<<B * 39, 64>>>((double2*)buf, 39, stride / 2, B); });
        printf("wave per knot (2.56M single-wave workgroups)  : %.3f ms  %.1f GB/s\n", ms, (double)B * 39 * 2400 / ms / 1e6);
        ms = time_ms([&] { k_fill_wave_per_knot4<<<(B * 39 + 3) / 4, 256>>>((double2*)buf, 39, stride / 2, B); });
        printf("wave per knot, 4 waves per workgroup          : %.3f ms  %.1f GB/s\n", ms, (double)B * 39 * 2400 / ms / 1e6);
        for (int skew : {0}) {
            ms = time_ms([&] { k_fill_region_skew<<<B, 64>>>((double2*)buf, (int)(region / 2), stride / 2, B, skew); });
            printf("fill_region XCD-contig skew=%-5d: %.3f ms  %.1f GB/s\n", skew, ms, wbytes / ms / 1e6);
        }
        for (size_t pitch : {(size_t)11744, (size_t)12880, (size_t)12896, (size_t)13056, (size_t)13312, (size_t)16384}) {
            if ((size_t)B * pitch * 8 > nbytes + (size_t)B * 800 * 8) continue;
            ms = time_ms([&] { k_fill_region_map<1, 1><<<B, 64>>>((double2*)buf, (int)(region / 2), pitch / 2, B); });
            printf("fill_region XCD-contig pitch=%zu doubles: %.3f ms  %.1f GB/s\n", pitch, ms, wbytes / ms / 1e6);
        }
        for (int CH : {2048}) {
            ms = time_ms([&] { k_fill_region_chunked<<<B, 64>>>((double2*)buf, (int)(region / 2), stride / 2, CH); });
            printf("fill_region chunked XCD map CH=%-4d: %.3f ms  %.1f GB/s\n", CH, ms, wbytes / ms / 1e6);
        }
        for (int G : {64}) {
            const int pieces = 92;  // 94208 B per region
            ms = time_ms([&] { k_fill_interleaved<1><<<B, 64>>>((double2*)buf, pieces, G, B); });
            printf("interleaved G=%-5d XCDc: %.3f ms  %.1f GB/s\n", G, ms, (double)B * pieces * 1024 / ms / 1e6);
            ms = time_ms([&] { k_fill_interleaved<0><<<B, 64>>>((double2*)buf, pieces, G, B); });
            printf("interleaved G=%-5d rr  : %.3f ms  %.1f GB/s\n", G, ms, (double)B * pieces * 1024 / ms / 1e6);
        }
        ms = time_ms([&] { CK(hipMemsetAsync(buf, 0, wbytes, 0)); });
        printf("hipMemsetAsync         : %.3f ms  %.1f GB/s\n", ms, wbytes / ms / 1e6);
        ms = time_ms([&] { k_fill_flat<<<8192, 256>>>((double2*)buf, wbytes / 16); });
        printf("fill_flat 8192x256     : %.3f ms  %.1f GB/s\n", ms, wbytes / ms / 1e6);
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
