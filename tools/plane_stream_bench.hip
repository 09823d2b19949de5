// tools/plane_stream_bench.hip -- what a pass shaped like the stress update can reach on HBM when it does
// nothing but move its bytes: NR planes read and NW planes written per element, one element per lane,
//   (a) 8 B per lane and plane   (the engine's SoA planes [k][ne] of doubles),
//   (b) 16 B per lane and plane  (the same bytes as planes of double2),
// against the plain device copy (one 16-B stream in, one out).  No arithmetic but the sum that ties the
// loads to the stores.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/plane_stream_bench tools/plane_stream_bench.hip && /tmp/plane_stream_bench [nelem]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int NR, int NW>
__global__ void __launch_bounds__(256) k_planes8(const double *__restrict__ src, double *__restrict__ dst, int ne)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= ne) return;
    double v[NR];
#pragma unroll
    for (int k = 0; k < NR; ++k) v[k] = src[(size_t)k * ne + e];
    double s = 0;
#pragma unroll
    for (int k = 0; k < NR; ++k) s += v[k];
#pragma unroll
    for (int k = 0; k < NW; ++k) dst[(size_t)k * ne + e] = s + v[k % NR];
}

template <int NR, int NW>          // NR, NW: planes of double2
__global__ void __launch_bounds__(256) k_planes16(const double2 *__restrict__ src, double2 *__restrict__ dst, int ne)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= ne) return;
    double2 v[NR];
#pragma unroll
    for (int k = 0; k < NR; ++k) v[k] = src[(size_t)k * ne + e];
    double s = 0;
#pragma unroll
    for (int k = 0; k < NR; ++k) s += v[k].x + v[k].y;
#pragma unroll
    for (int k = 0; k < NW; ++k) { double2 o; o.x = s + v[k % NR].x; o.y = s + v[k % NR].y; dst[(size_t)k * ne + e] = o; }
}

// two elements per lane, 8-B planes read as 16 B (lane l takes elements 2l, 2l+1)
template <int NR, int NW>
__global__ void __launch_bounds__(256) k_planes8x2(const double *__restrict__ src, double *__restrict__ dst, int ne)
{
    const int e2 = blockIdx.x * 256 + threadIdx.x;
    if (2 * e2 + 1 >= ne) return;
    double2 v[NR];
#pragma unroll
    for (int k = 0; k < NR; ++k) v[k] = ((const double2 *)(src + (size_t)k * ne))[e2];
    double s = 0;
#pragma unroll
    for (int k = 0; k < NR; ++k) s += v[k].x + v[k].y;
#pragma unroll
    for (int k = 0; k < NW; ++k) { double2 o; o.x = s + v[k % NR].x; o.y = s + v[k % NR].y; ((double2 *)(dst + (size_t)k * ne))[e2] = o; }
}

// TILE-MAJOR layout: the NR (NW) values of the 64 elements of a wave-tile lie together, [tile][plane][64] -- one contiguous
// NR * 512-byte record per wavefront instead of NR streams `ne * 8` bytes apart (what a re-laid-out element store would be)
template <int NR, int NW>
__global__ void __launch_bounds__(256) k_tiles8(const double *__restrict__ src, double *__restrict__ dst, int ne)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= ne) return;
    const size_t tile = (size_t)(e >> 6), lane = (size_t)(e & 63);
    const double *sp = src + tile * (NR * 64) + lane;
    double *dp = dst + tile * (NW * 64) + lane;
    double v[NR];
#pragma unroll
    for (int k = 0; k < NR; ++k) v[k] = sp[k * 64];
    double s = 0;
#pragma unroll
    for (int k = 0; k < NR; ++k) s += v[k];
#pragma unroll
    for (int k = 0; k < NW; ++k) dp[k * 64] = s + v[k % NR];
}

__global__ void __launch_bounds__(256) k_copy16(const double2 *__restrict__ src, double2 *__restrict__ dst, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

template <class F>
static float time_ms(F launch, int reps)
{
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    launch();
    (void)hipEventRecord(a, 0);
    for (int r = 0; r < reps; ++r) launch();
    (void)hipEventRecord(b, 0);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    return ms / reps;
}

int main(int argc, char **argv)
{
    const int ne = argc > 1 ? std::atoi(argv[1]) : 1001310 / 2 * 2;
    constexpr int NR = 18, NW = 15;                 // E2<GEO>, interior step: 141 B read, 120 B written per element
    double *src = nullptr, *dst = nullptr;
    CK(hipMalloc((void **)&src, (size_t)NR * ne * 8 + 64));
    CK(hipMalloc((void **)&dst, (size_t)(NW + 1) * ne * 8 + 64));
    CK(hipMemset(src, 0, (size_t)NR * ne * 8));
    const int reps = 50;
    const dim3 g((ne + 255) / 256), g2((ne / 2 + 255) / 256);
    const double bytes = (double)(NR + NW) * 8 * ne;
    float t;
    t = time_ms([&] { hipLaunchKernelGGL((k_planes8<NR, NW>), g, dim3(256), 0, 0, src, dst, ne); }, reps);
    std::printf("%d + %d planes of 8 B, one element per lane      : %7.1f us  %6.0f GB/s\n", NR, NW, t * 1e3, bytes / t / 1e6);
    t = time_ms([&] { hipLaunchKernelGGL((k_planes16<NR / 2, (NW + 1) / 2>), g, dim3(256), 0, 0, (const double2 *)src, (double2 *)dst, ne); }, reps);
    std::printf("%d + %d planes of 16 B, one element per lane      : %7.1f us  %6.0f GB/s\n", NR / 2, (NW + 1) / 2, t * 1e3,
                (double)(NR / 2 + (NW + 1) / 2) * 16 * ne / t / 1e6);
    t = time_ms([&] { hipLaunchKernelGGL((k_planes8x2<NR, NW>), g2, dim3(256), 0, 0, src, dst, ne); }, reps);
    std::printf("%d + %d planes of 8 B, two elements per lane     : %7.1f us  %6.0f GB/s\n", NR, NW, t * 1e3, bytes / t / 1e6);
    t = time_ms([&] { hipLaunchKernelGGL((k_tiles8<NR, NW>), dim3(ne / 256), dim3(256), 0, 0, src, dst, ne / 256 * 256); }, reps);
    std::printf("%d + %d values of 8 B, TILE-MAJOR [tile][plane][64]  : %7.1f us  %6.0f GB/s\n", NR, NW, t * 1e3, (double)(NR + NW) * 8 * (ne / 256 * 256) / t / 1e6);
    const size_t n16 = (size_t)NW * ne / 2;
    t = time_ms([&] { hipLaunchKernelGGL(k_copy16, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, 0, (const double2 *)src, (double2 *)dst, n16); }, reps);
    std::printf("plain copy, 16 B per lane, %zu MB each way         : %7.1f us  %6.0f GB/s\n", n16 * 16 / 1000000, t * 1e3, 2.0 * n16 * 16 / t / 1e6);
    CK(hipGetLastError());
    (void)hipFree(src); (void)hipFree(dst);
    return 0;
}
