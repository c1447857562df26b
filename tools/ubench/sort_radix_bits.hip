// development microbenchmark: rocPRIM onesweep radix sort, bits per pass (8 = library default); profiles/r04_sort_radix_bits.txt
// build: hipcc --offload-arch=gfx950 -O3 -DTRY10 -o tools/_old/sort_radix_bits tools/ubench/sort_radix_bits.hip ; run on the GPU box
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_fill(uint64_t* p, uint64_t n, uint64_t seed) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t z = (i + seed) * 0x9E3779B97F4A7C15ull; z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27; z *= 0x94D049BB133111EBull; z ^= z >> 31;
    p[i] = z;
}
__global__ void k_fill32(uint32_t* p, uint64_t n, uint64_t seed, uint32_t mask) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t z = (i + seed) * 0x9E3779B97F4A7C15ull; z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27;
    p[i] = (uint32_t)z & mask;
}
__global__ void k_check(const uint64_t* p, uint64_t n, int lo, uint32_t* bad) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i + 1 >= n) return;
    if ((p[i] >> lo) > (p[i + 1] >> lo)) atomicAdd(bad, 1u);
}
template <class Cfg, class K>
static int run_keys(const char* what, K* a, K* b, uint64_t n, int lo, int hi, void* tmp, size_t tmpcap) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        k_fill<<<(unsigned)((n + 255) / 256), 256>>>((uint64_t*)a, n * sizeof(K) / 8, 1234 + rep);
        rocprim::double_buffer<K> db(a, b);
        size_t tb = 0;
        CK((rocprim::radix_sort_keys<Cfg>(nullptr, tb, db, (size_t)n, lo, hi, 0)));
        if (tb > tmpcap) { printf("tmp too small %zu\n", tb); return 1; }
        hipEventRecord(e0);
        CK((rocprim::radix_sort_keys<Cfg>(tmp, tb, db, (size_t)n, lo, hi, 0)));
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
        if (rep == 0 && sizeof(K) == 8) {
            uint32_t* bad; hipMalloc(&bad, 4); hipMemset(bad, 0, 4);
            k_check<<<(unsigned)((n + 255) / 256), 256>>>((const uint64_t*)db.current(), n, lo, bad);
            uint32_t h = 0; hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost); hipFree(bad);
            if (h) printf("  NOT SORTED (%u)\n", h);
        }
    }
    printf("%-44s n=%llu bits[%d,%d): %.2f ms\n", what, (unsigned long long)n, lo, hi, best);
    return 0;
}
template <class Cfg, class K, class V>
static int run_pairs(const char* what, K* a, K* b, V* va, V* vb, uint64_t n, int lo, int hi, void* tmp, size_t tmpcap) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        k_fill32<<<(unsigned)((n + 255) / 256), 256>>>((uint32_t*)a, n, 99 + rep, 0xFFFFFFFFu);
        rocprim::double_buffer<K> dk(a, b);
        rocprim::double_buffer<V> dv(va, vb);
        size_t tb = 0;
        CK((rocprim::radix_sort_pairs<Cfg>(nullptr, tb, dk, dv, (size_t)n, lo, hi, 0)));
        if (tb > tmpcap) { printf("tmp too small %zu\n", tb); return 1; }
        hipEventRecord(e0);
        CK((rocprim::radix_sort_pairs<Cfg>(tmp, tb, dk, dv, (size_t)n, lo, hi, 0)));
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("%-44s n=%llu bits[%d,%d): %.2f ms\n", what, (unsigned long long)n, lo, hi, best);
    return 0;
}
using namespace rocprim;
template <unsigned B, unsigned I, unsigned R> using OS = radix_sort_config<default_config, default_config, radix_sort_onesweep_config<kernel_config<B, I>, kernel_config<B, I>, R, block_radix_rank_algorithm::match>>;
int main() {
    const uint64_t n = 1043374415ull, m = 205450045ull;
    uint64_t *a, *b; void* tmp; const size_t tmpcap = 1ull << 30;
    CK(hipMalloc(&a, 8 * n)); CK(hipMalloc(&b, 8 * n)); CK(hipMalloc(&tmp, tmpcap));
    run_keys<default_config>("u64 keys default (8 bits)", a, b, n, 32, 64, tmp, tmpcap);
    run_keys<OS<512, 12, 8>>("u64 keys 512x12 r8", a, b, n, 32, 64, tmp, tmpcap);
    run_keys<OS<1024, 6, 8>>("u64 keys 1024x6 r8", a, b, n, 32, 64, tmp, tmpcap);
    run_keys<OS<512, 12, 9>>("u64 keys 512x12 r9 (27 bits)", a, b, n, 37, 64, tmp, tmpcap);
#ifdef TRY10
    run_keys<OS<1024, 6, 10>>("u64 keys 1024x6 r10 (30 bits)", a, b, n, 34, 64, tmp, tmpcap);
    run_keys<OS<1024, 4, 10>>("u64 keys 1024x4 r10 (30 bits)", a, b, n, 34, 64, tmp, tmpcap);
    run_keys<OS<1024, 8, 10>>("u64 keys 1024x8 r10 (30 bits)", a, b, n, 34, 64, tmp, tmpcap);
#endif
    // layout: u32 key (27 bits) + u64 value; position sort: u32 key (30 bits) + u32 value
    uint32_t* k1 = (uint32_t*)a; uint32_t* k2 = k1 + m; uint64_t* v1 = b; uint64_t* v2 = b + m;
    run_pairs<default_config>("u32 key + u64 val default", k1, k2, v1, v2, m, 0, 27, tmp, tmpcap);
    run_pairs<OS<512, 12, 9>>("u32 key + u64 val 512x12 r9", k1, k2, v1, v2, m, 0, 27, tmp, tmpcap);
    run_pairs<OS<1024, 6, 9>>("u32 key + u64 val 1024x6 r9", k1, k2, v1, v2, m, 0, 27, tmp, tmpcap);
    uint32_t* w1 = (uint32_t*)b; uint32_t* w2 = w1 + m;
    run_pairs<default_config>("u32 key + u32 val default", k1, k2, w1, w2, m, 0, 30, tmp, tmpcap);
#ifdef TRY10
    run_pairs<OS<1024, 8, 10>>("u32 key + u32 val 1024x8 r10", k1, k2, w1, w2, m, 0, 30, tmp, tmpcap);
    run_pairs<OS<1024, 6, 10>>("u32 key + u32 val 1024x6 r10", k1, k2, w1, w2, m, 0, 30, tmp, tmpcap);
#endif
    return 0;
}
