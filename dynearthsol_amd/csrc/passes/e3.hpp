// passes/e3.hpp -- Pass E3 (elements) and the stress-bc facet work that rides in its launch.
// Part of the single translation unit des_dev.hip (included inside namespace des_hip, after
// DevClock / struct des_dev); not a stand-alone header.

// ---- E3 --------------------------------------------------------------------------
// NMD_stress apply (geometry.cxx:316-331), update_force element part (fields.cxx:623-653).
// The 12 force terms of an element are one 96-byte record; a wavefront's records are staged
// in LDS and written back as contiguous 16-byte pieces instead of 64 strided 8-byte stores.
// Workgroups past the element range compute the stress-bc facet terms (bc_facet_work):
// both only read the nodal records, and N3 consumes both.
__device__ void bc_facet_work(const des_params *__restrict__ p, int g, const int4 *__restrict__ conn,
                              const d4 *__restrict__ xt, const MatData &md,
                              const int *__restrict__ f_elem, const int *__restrict__ f_facet,
                              const int *__restrict__ f_kind, const double *__restrict__ f_val,
                              double *__restrict__ f_tmp);

__global__ void __launch_bounds__(DES_BLOCK, DES_E3_WAVES)
E3_nmd_force(const des_params *__restrict__ p, int nmd, int ne, int e_begin, int e_count, int nblocks, int nblocks8,
     const int4 *__restrict__ conn,
     const d4 *__restrict__ xt, const double *__restrict__ ntmp, const MatData md,
     const double *__restrict__ volume,
     const double *__restrict__ dpressure, double *__restrict__ stress, double *__restrict__ ftmp,
     int nbcf, const int *__restrict__ f_elem, const int *__restrict__ f_facet,
     const int *__restrict__ f_kind, const double *__restrict__ f_val, double *__restrict__ f_tmp)
{
    if ((int)blockIdx.x >= nblocks8) {                    // facet blocks (uniform per workgroup)
        const int g = ((int)blockIdx.x - nblocks8) * DES_BLOCK + threadIdx.x;
        if (g < nbcf) bc_facet_work(p, g, conn, xt, md, f_elem, f_facet, f_kind, f_val, f_tmp);
        return;
    }
    __shared__ double stage[DES_BLOCK * 13];              // 12 doubles per element, row stride 13
    const int e_end = e_begin + e_count;                  // this launch's element range
    const int e0 = e_begin + desk::logical_block(nblocks) * DES_BLOCK;
    const int e = e0 + threadIdx.x;
    if (e0 >= e_end) return;
    if (e < e_end) {
        const int4 cn = conn[e];
        d4 c[4];
        c[0] = xt[cn.x]; c[1] = xt[cn.y]; c[2] = xt[cn.z]; c[3] = xt[cn.w];
        double s[6];
        for (int i = 0; i < 6; ++i) s[i] = stress[(size_t)i*ne + e];
        if (nmd) {                                          // is_using_mixed_stress, outside the isostasy loop
            double dp = 0;
            dp += ntmp[cn.x]; dp += ntmp[cn.y]; dp += ntmp[cn.z]; dp += ntmp[cn.w];
            double dp_el = dp / 4;
            double dp_orig = dpressure[e];
            double ddp = (-dp_orig + dp_el) / 3;
            for (int i = 0; i < 3; ++i) { s[i] += ddp; stress[(size_t)i*ne + e] = s[i]; }
        }
        const double vol = volume[e];
        double sx[4], sy[4], sz[4];
        desk::shape_fn(c, vol, sx, sy, sz);
        double buoy = 0;
        if (p->gravity != 0) {
            double T = 0;
            T += c[0].w; T += c[1].w; T += c[2].w; T += c[3].w;
            T /= 4;
            const desk::Mix mx = mix_of(md, p->nmat, e);
            const double rho = desk::mat_rho(p, mx, T);
            const double phi = load_props(p, md, mx, ne, e).phi;
            buoy = (rho * (1 - phi) + 1000.0 * phi) * p->gravity / 4;
        }
        double *out = stage + threadIdx.x * 13;
        for (int i = 0; i < 4; ++i) {
            out[i*3 + 0] = (s[0]*sx[i] + s[3]*sy[i] + s[4]*sz[i]) * vol;
            out[i*3 + 1] = (s[3]*sx[i] + s[1]*sy[i] + s[5]*sz[i]) * vol;
            out[i*3 + 2] = (s[4]*sx[i] + s[5]*sy[i] + s[2]*sz[i] + buoy) * vol;
        }
    }
    __syncthreads();
    const int nvalid = min(DES_BLOCK, e_end - e0) * 12;
    double *dst = ftmp + (size_t)e0 * 12;
    for (int idx = threadIdx.x; idx < nvalid; idx += DES_BLOCK) {
        const int t = idx / 12, k = idx - t * 12;
        dst[idx] = stage[t * 13 + k];
    }
}

// ---- stress-bc facets ------------------------------------------------------------
// apply_stress_bcs facet loop (bc.cxx:707-777) and apply_stress_bcs_neumann (bc.cxx:846-905)
__device__ void bc_facet_work(const des_params *__restrict__ p, int g, const int4 *__restrict__ conn,
                              const d4 *__restrict__ xt, const MatData &md,
                              const int *__restrict__ f_elem, const int *__restrict__ f_facet,
                              const int *__restrict__ f_kind, const double *__restrict__ f_val,
                              double *__restrict__ f_tmp)
{
    const int e = f_elem[g], f = f_facet[g], kind = f_kind[g];
    const int4 cn = conn[e];
    const int cna[4] = {cn.x, cn.y, cn.z, cn.w};
    d4 fc[3];
    for (int j = 0; j < 3; ++j) fc[j] = xt[cna[NODE_OF_FACET_D[f][j]]];
    // normal_vector_of_facet, bc.cxx:24-54
    double v01[3] = {fc[1].x - fc[0].x, fc[1].y - fc[0].y, fc[1].z - fc[0].z};
    double v02[3] = {fc[2].x - fc[0].x, fc[2].y - fc[0].y, fc[2].z - fc[0].z};
    double normal[3];
    normal[0] = (v01[1] * v02[2] - v01[2] * v02[1]) / 2;
    normal[1] = (v01[2] * v02[0] - v01[0] * v02[2]) / 2;
    normal[2] = (v01[0] * v02[1] - v01[1] * v02[0]) / 2;
    double zcenter = (fc[0].z + fc[1].z + fc[2].z) / 3;
    double *out = f_tmp + (size_t)g * 9;
    if (kind >= 3) {
        double traction[3] = {0, 0, 0};
        traction[kind - 3] = f_val[g];
        for (int j = 0; j < 3; ++j)
            for (int d = 0; d < 3; ++d) out[j*3 + d] = traction[d] * normal[d] / 3;
        return;
    }
    double pr;
    if (kind == 0) {
        double T = 0;
        T += xt[cn.x].w; T += xt[cn.y].w; T += xt[cn.z].w; T += xt[cn.w].w;
        T /= 4;
        double rho_effective = desk::mat_rho(p, mix_of(md, p->nmat, e), T);
        pr = p->compensation_pressure -
             (rho_effective + p->winkler_delta_rho) * p->gravity * (zcenter + p->zlength);
    } else if (kind == 1) {
        pr = 0;
        if (zcenter < p->surf_base_level)
            pr = p->sea_water_density * p->gravity * (p->surf_base_level - zcenter);
    } else {
        pr = desk::ref_pressure(p, zcenter);
        if (pr < 0.0) pr = 0.0;
    }
    for (int j = 0; j < 3; ++j)
        for (int d = 0; d < 3; ++d) out[j*3 + d] = pr * normal[d] / 3;
}
