// passes/n2.hpp -- Pass N2 (nodes): NMD gather.
// Part of the single translation unit des_dev.hip (included inside namespace des_hip, after
// DevClock / struct des_dev); not a stand-alone header.

// ---- N2 --------------------------------------------------------------------------
// NMD_stress gather (geometry.cxx:302-309)
__global__ void __launch_bounds__(DES_BLOCK)
N2_nmd_gather(int o0, int nn, int nblocks, int npb, const int *__restrict__ sup_idx, const int *__restrict__ sup_pack,
     const double *__restrict__ etmp2, const double *__restrict__ volume_n, double *__restrict__ ntmp)
{
    __shared__ double lds[DES_TILE_LDS(DES_TILE_N2)];
    const int TILE = DES_TILE_N2;
    const int n0 = o0 + desk::logical_block(nblocks) * npb;
    const int n = (threadIdx.x < npb) ? n0 + threadIdx.x : nn;
    if (n0 >= nn) return;
    const int nlast = min(n0 + npb, nn);
    const int kb = sup_idx[n0], ke = sup_idx[nlast];
    int r0 = ke, r1 = ke;
    if (n < nn) { r0 = sup_idx[n]; r1 = sup_idx[n+1]; }
    double acc = 0;
#if DES_PIPE
    constexpr int PER = DES_TILE_N2 / DES_BLOCK;
    double rv[PER];
    auto fetch = [&](int t0, int tn) {
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int j = threadIdx.x + u * DES_BLOCK;
            if (j < tn) rv[u] = etmp2[sup_pack[t0 + j] >> 2];
        }
    };
    if (kb < ke) fetch(kb, min(TILE, ke - kb));
#endif
    for (int t0 = kb; t0 < ke; t0 += TILE) {
        const int tn = min(TILE, ke - t0);
#if DES_PIPE
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int j = threadIdx.x + u * DES_BLOCK;
            if (j < tn) lds[lds_slot(j)] = rv[u];
        }
        __syncthreads();
        if (t0 + TILE < ke) fetch(t0 + TILE, min(TILE, ke - t0 - TILE));
#else
        for (int j = threadIdx.x; j < tn; j += DES_BLOCK)
            lds[lds_slot(j)] = etmp2[sup_pack[t0 + j] >> 2];
        __syncthreads();
#endif
        const int a = max(r0, t0) - t0, b = min(r1, t0 + tn) - t0;
        for (int j = a; j < b; ++j) acc += lds[lds_slot(j)];
        __syncthreads();
    }
    if (n < nn) ntmp[n] = acc / volume_n[n];
}
