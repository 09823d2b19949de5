// tools/launch_floor.hip -- what ONE dependent launch costs a stream on this device, whatever it does: the floor under a step that
// is made of a few short kernels (a strong-scaling shard of the 1M-tet mesh: 4 launches of ~600 workgroups each).
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/launch_floor tools/launch_floor.hip && /tmp/launch_floor [workgroups] [lanes]
//
// Back-to-back launches on one stream (each waits for the one before: the kernels of a step depend on each other), timed with
// HIP events around 2000 of them:
//   empty      the kernel returns at once                                   -> dispatch + drain of a grid
//   touch      every workgroup reads 16 KB it has not seen and writes 4 KB  -> + first loads from a cold L2 / HBM, + end-of-kernel write-back
//   chain3     the same, three dependent loads deep (index -> index -> record), as a patch pass's staging
//   life9us    every wavefront additionally spins ~9 us (s_sleep), the measured lifetime of a patch-pass workgroup on the shard
// `life9us` minus 9 us = what a launch adds to the one workgroup lifetime a single-round pass cannot go below.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

__global__ void k_empty(const double *, double *, const int *, int) {}

__global__ void k_touch(const double *in, double *out, const int *, int round)
{
    const size_t base = ((size_t)blockIdx.x + (size_t)round * gridDim.x) * 2048;     // 16 KB per workgroup, new every launch
    double acc = 0;
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) acc += in[base + i];
    if (threadIdx.x < 512) out[(size_t)blockIdx.x * 512 + threadIdx.x] = acc;
}

__global__ void k_chain3(const double *in, double *out, const int *idx, int round)
{
    const size_t base = ((size_t)blockIdx.x + (size_t)round * gridDim.x) * 2048;
    const int a = idx[(base + threadIdx.x) & ((1u << 22) - 1)];
    const int b = idx[a];
    double acc = in[(size_t)b];
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) acc += in[base + i];
    if (threadIdx.x < 512) out[(size_t)blockIdx.x * 512 + threadIdx.x] = acc;
}

__global__ void k_life(const double *in, double *out, const int *idx, int round)
{
    const size_t base = ((size_t)blockIdx.x + (size_t)round * gridDim.x) * 2048;
    const int a = idx[(base + threadIdx.x) & ((1u << 22) - 1)];
    const int b = idx[a];
    double acc = in[(size_t)b];
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) acc += in[base + i];
    const unsigned long long t0 = wall_clock64();                     // 100 MHz
    while (wall_clock64() - t0 < 900) __builtin_amdgcn_s_sleep(8);     // ~9 us
    if (threadIdx.x < 512) out[(size_t)blockIdx.x * 512 + threadIdx.x] = acc;
}

int main(int argc, char **argv)
{
    const int wgs = argc > 1 ? std::atoi(argv[1]) : 596, lanes = argc > 2 ? std::atoi(argv[2]) : 256;
    const int N = 2000;
    const size_t n_in = (size_t)1 << 28;                               // 2 GiB of doubles: every launch reads lines nothing has touched
    double *in, *out; int *idx;
    CK(hipMalloc(&in, n_in * sizeof(double))); CK(hipMalloc(&out, (size_t)wgs * 512 * sizeof(double))); CK(hipMalloc(&idx, sizeof(int) << 22));
    CK(hipMemset(in, 0, n_in * sizeof(double)));
    std::vector<int> h(1u << 22);
    unsigned s = 12345;
    for (auto &v : h) { s = s * 1664525u + 1013904223u; v = (int)(s >> 10); }
    CK(hipMemcpy(idx, h.data(), sizeof(int) << 22, hipMemcpyHostToDevice));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    struct { const char *name; void (*k)(const double *, double *, const int *, int); } ks[] = {
        {"empty", k_empty}, {"touch", k_touch}, {"chain3", k_chain3}, {"life9us", k_life}};
    std::printf("# %d workgroups of %d lanes, %d dependent launches on one stream, HIP events; us per launch\n", wgs, lanes, N);
    for (auto &k : ks) {
        const size_t per = (size_t)wgs * 2048;
        const int rounds = (int)(n_in / per) - 1;
        for (int w = 0; w < 50; ++w) hipLaunchKernelGGL(k.k, dim3(wgs), dim3(lanes), 0, st, in, out, idx, w % rounds);
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(a, st));
        for (int w = 0; w < N; ++w) hipLaunchKernelGGL(k.k, dim3(wgs), dim3(lanes), 0, st, in, out, idx, (50 + w) % rounds);
        CK(hipEventRecord(b, st));
        CK(hipEventSynchronize(b));
        float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
        std::printf("%-8s %7.2f\n", k.name, 1e3 * ms / N);
    }
    return 0;
}
