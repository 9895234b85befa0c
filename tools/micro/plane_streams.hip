// Microbenchmark: what the HBM gives a kernel that reads P planes (stride apart) in step - every workgroup takes
// the same chunk of wells from each plane, as k_dense_pack does - against the chunk a workgroup reads per plane
// and the loads a thread keeps in flight.  Build + run: bash tools/micro/plane_streams.sh
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// W = dwords per lane and load (1: 1 KB per workgroup and plane, 4: 4 KB), F = loads in flight per thread
template <int W, int F>
__global__ __launch_bounds__(256) void k_planes(const uint32_t *base, size_t stride_dw, int planes, size_t n_dw, uint32_t *sink)
{
    typedef uint32_t V __attribute__((ext_vector_type(W)));
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * W;
    if (i + W > n_dw)
        return;
    uint32_t acc = 0;
    for (int p0 = 0; p0 + F <= planes; p0 += F) {
        V v[F];
#pragma unroll
        for (int f = 0; f < F; f++)
            v[f] = __builtin_nontemporal_load((const __attribute__((address_space(1))) V *)(base + (size_t)(p0 + f) * stride_dw + i));
#pragma unroll
        for (int f = 0; f < F; f++) {
#pragma unroll
            for (int w = 0; w < W; w++)
                acc ^= v[f][w];
        }
    }
    if (acc == 0x9E3779B9u)
        sink[blockIdx.x & 63] = acc;
}

template <int W, int F>
static void run(const uint32_t *buf, size_t plane_bytes, int planes, uint32_t *sink, const char *name)
{
    const size_t n_dw = plane_bytes / 4;
    const unsigned grid = (unsigned)((n_dw / W + 255) / 256);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_planes<W, F>), dim3(grid), dim3(256), 0, 0, buf, n_dw, planes, n_dw, sink);
    CK(hipEventRecord(e0, 0));
    const int reps = 5;
    for (int r = 0; r < reps; r++)
        hipLaunchKernelGGL((k_planes<W, F>), dim3(grid), dim3(256), 0, 0, buf, n_dw, planes, n_dw, sink);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const int used = planes / F * F;
    printf("%-28s planes %3d: %7.1f GB/s  (%.1f us per pass)\n", name, planes, (double)used * plane_bytes * reps / (ms * 1e-3) / 1e9, ms / reps * 1e3);
}

int main()
{
    const size_t plane_bytes = 4309760;            // a HiSeq 4000 tile's plane, padded to 256
    const int planes = 160, tiles = 8;
    uint32_t *buf, *sink;
    CK(hipMalloc((void **)&buf, plane_bytes * planes * tiles));
    CK(hipMalloc((void **)&sink, 256));
    CK(hipMemset(buf, 5, plane_bytes * planes * tiles));
    for (int rep = 0; rep < 2; rep++) {
        run<1, 8>(buf, plane_bytes, 160, sink, "dword, 8 in flight");
        run<1, 32>(buf, plane_bytes, 160, sink, "dword, 32 in flight");
        run<4, 8>(buf, plane_bytes, 160, sink, "dwordx4, 8 in flight");
        run<4, 16>(buf, plane_bytes, 160, sink, "dwordx4, 16 in flight");
        run<2, 16>(buf, plane_bytes, 160, sink, "dwordx2, 16 in flight");
        run<1, 8>(buf, plane_bytes, 16, sink, "dword, 8 in flight");
        run<4, 8>(buf, plane_bytes, 16, sink, "dwordx4, 8 in flight");
    }
    return 0;
}
