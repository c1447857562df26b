// development aid: cost of random-address LDS operations on gfx950 as the row kernels issue them (per CU, by waves per CU).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/_old/lds_ops tools/ubench/lds_ops.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
constexpr int SIZE = 4096;   // words of LDS per workgroup
template <int OP, int U, bool DEP>
__global__ void k(uint32_t* out, int iters) {
    __shared__ uint32_t t[SIZE];
    for (int i = threadIdx.x; i < SIZE; i += blockDim.x) t[i] = i * 2654435761u;
    __syncthreads();
    uint32_t x[U];
    for (int u = 0; u < U; ++u) x[u] = (threadIdx.x * 97u + u * 1013u + blockIdx.x) * 2654435761u;
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        uint32_t r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t a = (x[u] >> 7) % SIZE;
            if (OP == 0) r[u] = t[a];
            else if (OP == 1) r[u] = atomicAdd(&t[a], 1u);
            else if (OP == 2) r[u] = atomicCAS(&t[a], x[u], x[u] + 1);
            else if (OP == 3) r[u] = atomicMin(&t[a], x[u]);
            else if (OP == 4) { atomicAdd(&t[a], 1u); r[u] = 0; }          // no return
            else if (OP == 5) { t[a] = x[u]; r[u] = 0; }                   // plain store
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            acc += r[u];
            x[u] = DEP ? (x[u] * 1664525u + 1013904223u + r[u]) : (x[u] * 1664525u + 1013904223u);
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}
template <int OP, int U, bool DEP>
void run(const char* name, uint32_t* d) {
    for (int wg_per_cu : {1, 2, 4}) {
        for (int threads : {512, 1024}) {
            if (wg_per_cu * threads > 2048) continue;
            const int iters = 2000;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            k<OP, U, DEP><<<256 * wg_per_cu, threads>>>(d, 10);
            hipEventRecord(e0);
            k<OP, U, DEP><<<256 * wg_per_cu, threads>>>(d, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double waves = wg_per_cu * threads / 64.0;
            const double wave_ops = waves * iters * U;                       // wave-instructions per CU
            const double cyc = ms * 1e-3 * 2.4e9;
            printf("%-22s U=%d dep=%d waves/CU=%2.0f : %.1f cycles per wave-instruction per CU, %.0f cycles latency per round\n", name, U, (int)DEP, waves,
                   cyc / wave_ops, cyc / iters);
        }
    }
}
int main() {
    uint32_t* d; hipMalloc(&d, 64);
    run<0, 1, true>("read b32 random", d);
    run<0, 4, false>("read b32 random", d);
    run<1, 1, true>("atomicAdd rtn", d);
    run<1, 4, false>("atomicAdd rtn", d);
    run<2, 1, true>("atomicCAS rtn", d);
    run<2, 4, false>("atomicCAS rtn", d);
    run<3, 4, false>("atomicMin rtn", d);
    run<4, 4, false>("atomicAdd noret", d);
    run<5, 4, false>("store b32 random", d);
    return 0;
}
