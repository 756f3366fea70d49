// vmm_va_reuse.cpp -- one-shot diagnostic for the placed-buffer code (qln_vals_alloc_placed): does a virtual range
// that is unmapped -- and optionally handed back with hipMemAddressFree and reserved again -- ever serve accesses
// through the translations of its PREVIOUS mapping?  (Round 1 saw wrong data after exactly that sequence and stopped
// freeing the range; the cause was never isolated.)
//
// Safe by construction: the physical chunks of the first mapping (A) are never released while the test runs, so a
// stale translation cannot reach freed memory -- it would show up as B's pattern inside A.  Every HIP return code is
// printed.  Scale matches the failing case: `nchunks` x 256 MiB behind one reservation (default 130 = 32.5 GiB).
//
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/vmm_va_reuse bench/vmm_va_reuse.cpp && /tmp/vmm_va_reuse [nchunks]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e_ = (x);                                                   \
        printf("  %-58s -> %s\n", #x, hipGetErrorName(e_));                    \
        if (e_ != hipSuccess) {                                                \
            printf("ABORT: unexpected HIP error, nothing concluded\n");        \
            exit(2);                                                           \
        }                                                                      \
    } while (0)
#define CKQ(x)                                                                 \
    do {                                                                       \
        hipError_t e_ = (x);                                                   \
        if (e_ != hipSuccess) {                                                \
            printf("  %s -> %s\nABORT\n", #x, hipGetErrorName(e_));            \
            exit(2);                                                           \
        }                                                                      \
    } while (0)

__global__ void fill(unsigned long long* p, size_t n, unsigned long long tag) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = tag ^ i;
}
__global__ void count_tag(const unsigned long long* p, size_t n, unsigned long long tag, unsigned long long* hits) {
    unsigned long long local = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        local += (p[i] == (tag ^ i));
    if (local) atomicAdd(hits, local);
}

static hipMemAllocationProp g_prop;
static hipMemAccessDesc g_ad;
static size_t g_chunk;

static std::vector<hipMemGenericAllocationHandle_t> create_chunks(size_t n) {
    std::vector<hipMemGenericAllocationHandle_t> v(n);
    for (size_t i = 0; i < n; ++i) CKQ(hipMemCreate(&v[i], g_chunk, &g_prop, 0));
    return v;
}
static void map_all(char* va, const std::vector<hipMemGenericAllocationHandle_t>& hs) {
    for (size_t i = 0; i < hs.size(); ++i) CKQ(hipMemMap(va + i * g_chunk, g_chunk, 0, hs[i], 0));
    CK(hipMemSetAccess(va, hs.size() * g_chunk, &g_ad, 1));
}
static unsigned long long count(const void* p, size_t words, unsigned long long tag, unsigned long long* d_hits) {
    CKQ(hipMemset(d_hits, 0, 8));
    count_tag<<<4096, 256>>>((const unsigned long long*)p, words, tag, d_hits);
    CKQ(hipDeviceSynchronize());
    unsigned long long h = 0;
    CKQ(hipMemcpy(&h, d_hits, 8, hipMemcpyDeviceToHost));
    return h;
}

// one scenario; returns the number of words of A that were overwritten with B's pattern (0 = translations are fresh)
static unsigned long long scenario(const char* name, size_t nchunks, bool address_free, bool unmap_piecewise,
                                   bool release_first = false) {
    printf("== %s: %zu chunks, hipMemAddressFree between the mappings: %s, unmap: %s, A released before B exists: %s\n", name,
           nchunks, address_free ? "yes" : "no", unmap_piecewise ? "chunk by chunk" : "whole range", release_first ? "yes" : "no");
    const size_t bytes = nchunks * g_chunk, words = bytes / 8;
    const unsigned long long TA = 0xA1A1A1A100000000ull, TB = 0xB2B2B2B200000000ull;
    unsigned long long* d_hits = nullptr;
    CKQ(hipMalloc(&d_hits, 8));
    void* va = nullptr;
    CK(hipMemAddressReserve(&va, bytes, 0, nullptr, 0));
    auto A = create_chunks(nchunks);
    map_all((char*)va, A);
    fill<<<4096, 256>>>((unsigned long long*)va, words, TA);
    CK(hipDeviceSynchronize());
    printf("  A through its first mapping: %llu / %zu words carry A's pattern\n", count(va, words, TA, d_hits), words);
    if (unmap_piecewise) {
        for (size_t i = 0; i < nchunks; ++i) CKQ(hipMemUnmap((char*)va + i * g_chunk, g_chunk));
        printf("  hipMemUnmap x %zu -> hipSuccess\n", nchunks);
    } else {
        CK(hipMemUnmap(va, bytes));
    }
    if (release_first) {  // the literal round-1 sequence: the physical memory goes back to the driver as well
        for (auto h : A) CKQ(hipMemRelease(h));
        A.clear();
    }
    void* va_b = va;
    if (address_free) {
        CK(hipMemAddressFree(va, bytes));
        CK(hipMemAddressReserve(&va_b, bytes, 0, nullptr, 0));
        printf("  second reservation %s the first one's addresses (%p vs %p)\n", va_b == va ? "REUSES" : "does not reuse", va_b, va);
    }
    auto Bc = create_chunks(nchunks);
    map_all((char*)va_b, Bc);
    fill<<<4096, 256>>>((unsigned long long*)va_b, words, TB);
    CK(hipDeviceSynchronize());
    const unsigned long long b_ok = count(va_b, words, TB, d_hits);
    if (release_first) {
        // nothing of A is left to inspect: B must read back whole, from a second pass over every XCD
        const unsigned long long b_again = count(va_b, words, TB, d_hits);
        printf("  B through the reused range: %llu and %llu / %zu words carry B's pattern\n", b_ok, b_again, words);
        CK(hipMemUnmap(va_b, bytes));
        for (auto h : Bc) CKQ(hipMemRelease(h));
        CK(hipMemAddressFree(va_b, bytes));
        CKQ(hipFree(d_hits));
        const bool ok = (b_ok == words && b_again == words);
        printf("  RESULT %s: %s\n", name, ok ? "clean" : "WRONG DATA THROUGH THE REUSED RANGE");
        return ok ? 0 : 1;
    }
    // A, still alive, mapped somewhere else: must still hold A's pattern everywhere
    void* va_a2 = nullptr;
    CK(hipMemAddressReserve(&va_a2, bytes, 0, nullptr, 0));
    map_all((char*)va_a2, A);
    const unsigned long long a_ok = count(va_a2, words, TA, d_hits);
    const unsigned long long a_hit_by_b = count(va_a2, words, TB, d_hits);
    printf("  B through the reused range: %llu / %zu words carry B's pattern\n", b_ok, words);
    printf("  A through a fresh mapping:  %llu / %zu words carry A's pattern, %llu carry B's\n", a_ok, words, a_hit_by_b);
    CK(hipMemUnmap(va_a2, bytes));
    CK(hipMemUnmap(va_b, bytes));
    for (auto h : A) CKQ(hipMemRelease(h));
    for (auto h : Bc) CKQ(hipMemRelease(h));
    CK(hipMemAddressFree(va_a2, bytes));
    CK(hipMemAddressFree(va_b, bytes));
    CKQ(hipFree(d_hits));
    const bool clean = (b_ok == words && a_ok == words && a_hit_by_b == 0);
    printf("  RESULT %s: %s\n", name, clean ? "clean (no access went through a stale translation)" : "STALE TRANSLATIONS OBSERVED");
    return clean ? 0 : 1;
}

int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IOLBF, 0);
    const size_t nchunks = argc > 1 ? (size_t)atoi(argv[1]) : 130;
    g_prop = {};
    g_prop.type = hipMemAllocationTypePinned;
    g_prop.location.type = hipMemLocationTypeDevice;
    g_prop.location.id = 0;
    g_ad = {};
    g_ad.location = g_prop.location;
    g_ad.flags = hipMemAccessFlagsProtReadWrite;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &g_prop, hipMemAllocationGranularityRecommended));
    g_chunk = ((size_t)256 << 20) / gran * gran;
    printf("granularity %zu, chunk %zu MiB\n", gran, g_chunk >> 20);
    unsigned long long bad = 0;
    bad += scenario("remap-same-range", nchunks, false, false);
    bad += scenario("free-and-rereserve", nchunks, true, false);
    bad += scenario("free-and-rereserve-piecewise-unmap", nchunks, true, true);
    bad += scenario("release-free-and-rereserve (round-1 sequence)", nchunks, true, true, true);
    printf("SUMMARY: %llu of 4 scenarios saw stale translations\n", bad);
    return bad ? 1 : 0;
}
