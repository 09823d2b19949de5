// passes/n1.hpp -- Pass N1 (nodes).
// Part of the single translation unit des_dev.hip (included inside namespace des_hip, after
// DevClock / struct des_dev); not a stand-alone header.

// ---- N1 --------------------------------------------------------------------------
// compute_mass gather (geometry.cxx:1846-1864), update_temperature node loop
// (fields.cxx:245-262), compute_dvoldt gather (geometry.cxx:231-238).
// Also advances the clock: steps++, time += dt (dynearthsol.cxx:773-774).
// FULL = 0: compute_mass only (init_geometry and the end of a des_dev_step call).
// CONSTM = 1: quasi-static run with one material -- the inertial mass of an element is
// (K / pseudo_speed^2) * V / 4 with a run constant factor (geometry.cxx:1814-1816, 1829), so it
// is formed from the gathered volume instead of being gathered itself (one LDS plane less).
template <int FULL, int CONSTM>
__global__ void __launch_bounds__(DES_BLOCK)
N1_mass_temperature_dvoldt(const des_params *__restrict__ p, DevClock *__restrict__ clk, int o0, int nn, int nblocks, int npb,
     const int *__restrict__ sup_idx, const int *__restrict__ sup_pack, const unsigned *__restrict__ bcflag,
     const d4 *__restrict__ mrec, const d4 *__restrict__ ttmp, const MatData md, int ne,
     d4 *__restrict__ xt, d4 *__restrict__ vm, double *__restrict__ volume_n, double *__restrict__ tmass,
     double *__restrict__ ymass, double *__restrict__ ntmp)
{
    constexpr int TILE = CONSTM ? DES_TILE_N1C : DES_TILE_N1;
    constexpr int NPL = CONSTM ? 4 : 5;
    __shared__ double lds[NPL][DES_TILE_LDS(TILE)];
    // nodes [o0, nn) are this rank's owned nodes (the whole mesh on one GPU)
    // a workgroup owns `npb` consecutive nodes (256, or 64 on small meshes so that there are
    // enough workgroups: all 256 lanes still share the gather phase, the first npb do the sums)
    const int lb = desk::logical_block(nblocks);
    const int n0 = o0 + lb * npb;
    const int n = (threadIdx.x < npb) ? n0 + threadIdx.x : nn;
    const double dt = clk->dt;
    if (FULL && blockIdx.x == 0 && threadIdx.x == 0) {
        if (!clk->iso && !clk->pt) {                       // the isostasy and PT loops do not advance the clock
            clk->steps += 1;
            clk->time += dt;
        }
        clk->maxdh = 0.0;
        clk->n_defer = 0;
    }
    if (n0 >= nn) return;                                   // whole block idle (grid padding)
    const bool thermal = p->has_thermal_diffusion;
    const bool need_ym = p->damping_option == 4;
    const double pseudo_speed = p->max_vbc_val * p->inertial_scaling;
    const double rho_m = p->bulk_modulus[0] / (pseudo_speed * pseudo_speed);
    const int nlast = min(n0 + npb, nn);
    const int kb = sup_idx[n0], ke = sup_idx[nlast];
    int r0 = ke, r1 = ke;
    if (n < nn) { r0 = sup_idx[n]; r1 = sup_idx[n+1]; }
    double vn = 0, ms = 0, tms = 0, acc = 0, tdot = 0, yms = 0;
#if DES_PIPE
    constexpr int PER = TILE / DES_BLOCK;
    static_assert(TILE % DES_BLOCK == 0, "tile must be a multiple of the block");
    d4 rr[PER]; double r3[PER];
    auto fetch = [&](int t0, int tn) {
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int j = threadIdx.x + u * DES_BLOCK;
            if (j < tn) {
                const int pk = sup_pack[t0 + j];
                const int e = pk >> 2;
                rr[u] = mrec[e];
                if (FULL && thermal) r3[u] = (&ttmp[e].x)[pk & 3];
                else if (need_ym)    r3[u] = elem_ym(p, md, ne, e);
            }
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
            if (j < tn) {
                const int sl = lds_slot(j);
                lds[0][sl] = rr[u].x; lds[1][sl] = rr[u].z; lds[2][sl] = rr[u].w;
                if ((FULL && thermal) || need_ym) lds[3][sl] = r3[u];
                if (!CONSTM) lds[NPL - 1][sl] = rr[u].y;
            }
        }
        __syncthreads();
        if (t0 + TILE < ke) fetch(t0 + TILE, min(TILE, ke - t0 - TILE));
#else
        for (int j = threadIdx.x; j < tn; j += DES_BLOCK) {
            const int pk = sup_pack[t0 + j];
            const int e = pk >> 2;
            const d4 r = mrec[e];
            const int sl = lds_slot(j);
            lds[0][sl] = r.x; lds[1][sl] = r.z; lds[2][sl] = r.w;
            if (FULL && thermal) lds[3][sl] = (&ttmp[e].x)[pk & 3];
            else if (need_ym)    lds[3][sl] = elem_ym(p, md, ne, e);
            if (!CONSTM) lds[NPL - 1][sl] = r.y;
        }
        __syncthreads();
#endif
        const int a = max(r0, t0) - t0, b = min(r1, t0 + tn) - t0;
        for (int j = a; j < b; ++j) {
            const int sl = lds_slot(j);
            const double vol = lds[0][sl];
            vn += vol;
            if (CONSTM) ms += rho_m * vol / 4;
            else        ms += lds[NPL - 1][sl];
            if (thermal) tms += lds[1][sl];
            if (FULL) {
                if (thermal) tdot += lds[3][sl];
                acc += lds[2][sl];
            }
            if (need_ym && !(FULL && thermal)) yms += lds[3][sl];
        }
        __syncthreads();
    }
    if (n >= nn) return;
    if (need_ym && FULL && thermal) {
        // damping option 4 together with thermal diffusion: the spare LDS plane is taken by
        // the conduction term, so the Young's-modulus mass is summed straight from memory
        for (int k = r0; k < r1; ++k) yms += elem_ym(p, md, ne, sup_pack[k] >> 2);
    }
    volume_n[n] = vn;
    tmass[n] = tms;
    if (need_ym) ymass[n] = yms;
    d4 m4 = vm[n];
    m4.w = ms;
    vm[n] = m4;
    if (FULL) {
        if (thermal && !clk->iso && !clk->pt) {            // the isostasy / PT loops have no update_temperature
            d4 x4 = xt[n];
            if (bcflag[n] & (1u << 5))
                x4.w = p->surface_temperature;
            else
                x4.w -= dt * tdot / tms;
            xt[n] = x4;
        }
        ntmp[n] = acc / vn;
    }
}
