// passes/en2.hpp -- Pass EN2 (node patches): N2 over node-block patches.
// Part of the single translation unit des_dev.hip (included inside namespace des_hip, after
// DevClock / struct des_dev); not a stand-alone header.

// ---- EN2 -------------------------------------------------------------------------
// NMD_stress gather (geometry.cxx:302-309): ntmp[n] = sum over supp(n) of dp x volume / volume_n[n].
// N2 fetches etmp2 once per incidence (4.4M scattered 8-byte reads at 1M tets); here a workgroup
// fetches it once per element of its block's patch (1.96M, in ascending element order) into LDS
// and the nodes sum their incidences from there, in CSR order (same association, same bits).
template <int INC, int PE>
__global__ void __launch_bounds__(256)
EN2_nmd_gather(int nn, int nblocks, int npb, const int *__restrict__ pe_ptr, const ulonglong2 *__restrict__ pe_pack,
               const int *__restrict__ sup_idx,
               const double *__restrict__ etmp2, const double *__restrict__ volume_n, double *__restrict__ ntmp)
{
    __shared__ double lv[PE];
    __shared__ unsigned short lidx[INC];
    const int lb = desk::logical_block(nblocks);
    const int n0 = lb * npb;
    if (n0 >= nn) return;                                  // grid padding
    const int nown = min(npb, nn - n0);
    const int e_begin = pe_ptr[lb], e_end = pe_ptr[lb + 1];
    const int n = n0 + threadIdx.x;
    const bool has_node = (int)threadIdx.x < nown;
    int r0 = 0, r1 = 0;
    double vn = 1.0;
    if (has_node) {
        const int kb = sup_idx[n0];
        r0 = sup_idx[n] - kb; r1 = sup_idx[n + 1] - kb;
        vn = volume_n[n];
    }
    for (int i = e_begin + threadIdx.x; i < e_end; i += 256) {
        const int q = i - e_begin;
        const PatchElem E = patch_elem_unpack(pe_pack[i]);    // (the record EN3 streams next: it finds it in the caches)
        lv[q] = etmp2[E.ew & 0x3fffffff];
        if (E.sl[0] >= 0) lidx[E.sl[0]] = (unsigned short)q;
        if (E.sl[1] >= 0) lidx[E.sl[1]] = (unsigned short)q;
        if (E.sl[2] >= 0) lidx[E.sl[2]] = (unsigned short)q;
        if (E.sl[3] >= 0) lidx[E.sl[3]] = (unsigned short)q;
    }
    __syncthreads();
    if (!has_node) return;
    double acc = 0;
    int k = r0;
    for (; k + 8 <= r1; k += 8) {                           // eight indices, then eight terms, then the sums in list order
        int q[8];
        double t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) q[u] = lidx[k + u];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = lv[q[u]];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += t[u];
    }
    for (; k < r1; ++k) acc += lv[lidx[k]];
    ntmp[n] = acc / vn;
}
