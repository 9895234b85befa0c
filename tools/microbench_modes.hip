// microbench_modes.hip - does any cache-policy modifier on global_load_ubyte make a one-byte
// gather fetch less than a full 128-byte line (i.e. raise the line rate)?
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/mbm tools/microbench_modes.hip && /tmp/mbm
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// Eight independent loads and their wait live in ONE asm statement with early-clobber outputs:
// hipcc treats an asm's outputs as ready when the statement ends, so a load left in flight
// across statements would let it reuse (and corrupt) the destination registers.
#define LOADS8(MOD)                                                                        \
    asm volatile("global_load_ubyte %0, %8, off " MOD "\n\t"                                \
                 "global_load_ubyte %1, %9, off " MOD "\n\t"                                \
                 "global_load_ubyte %2, %10, off " MOD "\n\t"                               \
                 "global_load_ubyte %3, %11, off " MOD "\n\t"                               \
                 "global_load_ubyte %4, %12, off " MOD "\n\t"                               \
                 "global_load_ubyte %5, %13, off " MOD "\n\t"                               \
                 "global_load_ubyte %6, %14, off " MOD "\n\t"                               \
                 "global_load_ubyte %7, %15, off " MOD "\n\t"                               \
                 "s_waitcnt vmcnt(0)"                                                      \
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]),   \
                   "=&v"(v[6]), "=&v"(v[7])                                                \
                 : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), \
                   "v"(p[7])                                                               \
                 : "memory")

template <int MODE>
__global__ void k_gather(const uint8_t *buf, uint32_t line_mask, uint32_t n, uint32_t *sink)
{
    uint32_t acc = 0;
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i + 7 * stride < n; i += 8 * stride) {
        uint32_t v[8];
        const uint8_t *p[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const uint32_t line = ((i + u * stride) * 2654435761u) & line_mask;
            p[u] = buf + (size_t)line * 128u + (line & 63u);
        }
        if (MODE == 0) LOADS8("");
        if (MODE == 1) LOADS8("nt");
        if (MODE == 2) LOADS8("sc0");
        if (MODE == 3) LOADS8("sc1");
        if (MODE == 4) LOADS8("sc0 sc1");
        if (MODE == 5) LOADS8("sc0 sc1 nt");
#pragma unroll
        for (int u = 0; u < 8; u++) acc += v[u];
    }
    if (acc == 0xFFFFFFFFu) *sink = acc;
}

int main()
{
    const size_t span = 4ull << 30;
    const uint32_t n_lines = (uint32_t)(span / 128), n = n_lines / 2;
    uint8_t *buf; uint32_t *sink;
    CK(hipMalloc(&buf, span)); CK(hipMalloc(&sink, 4)); CK(hipMemset(buf, 1, span)); CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    dim3 g(256 * 16), b(256);
    const char *names[6] = {"plain", "nt", "sc0", "sc1", "sc0 sc1", "sc0 sc1 nt"};
    auto run = [&](int m) {
        switch (m) {
        case 0: hipLaunchKernelGGL(k_gather<0>, g, b, 0, 0, buf, n_lines - 1, n, sink); break;
        case 1: hipLaunchKernelGGL(k_gather<1>, g, b, 0, 0, buf, n_lines - 1, n, sink); break;
        case 2: hipLaunchKernelGGL(k_gather<2>, g, b, 0, 0, buf, n_lines - 1, n, sink); break;
        case 3: hipLaunchKernelGGL(k_gather<3>, g, b, 0, 0, buf, n_lines - 1, n, sink); break;
        case 4: hipLaunchKernelGGL(k_gather<4>, g, b, 0, 0, buf, n_lines - 1, n, sink); break;
        default: hipLaunchKernelGGL(k_gather<5>, g, b, 0, 0, buf, n_lines - 1, n, sink); break;
        }
    };
    for (int m = 0; m < 6; m++) {
        run(m); CK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int r = 0; r < 3; r++) {
            CK(hipEventRecord(e0)); run(m); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        printf("%-12s %.3f ms  %.1f Glines/s\n", names[m], best, n / best / 1e6);
    }
    return 0;
}
