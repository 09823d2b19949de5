// passes/e2.hpp -- Pass E2 (elements): update_stress, first pass and return-mapping pass.
// Part of the single translation unit des_dev.hip (included inside namespace des_hip, after
// DevClock / struct des_dev); not a stand-alone header.

__device__ void bc_facet_work(const des_params *__restrict__ p, int g, const int4 *__restrict__ conn,
                              const d4 *__restrict__ xt, const MatData &md,
                              const int *__restrict__ f_elem, const int *__restrict__ f_facet,
                              const int *__restrict__ f_kind, const double *__restrict__ f_val,
                              double *__restrict__ f_tmp);

// ---- E2 --------------------------------------------------------------------------
// compute_edvoldt (geometry.cxx:264-272), update_stress (rheology.cxx:728-1026),
// NMD_stress element part (geometry.cxx:294-296)
// The stress update of element e.  DEFER = 1 (first pass): returns true WITHOUT having stored
// anything but viscosity[e] when the element needs the Mohr-Coulomb return mapping; the second
// pass then runs the same code with DEFER = 0 for exactly those elements.
// what E1<MODE_DEFER> left for this pass to finish (all null / 0: nothing)
// The strain planes are read once and written once per step and nothing touches them in between (~0.6 GB of other
// traffic): loaded and stored non-temporally they do not evict what the next passes re-read (the stress, the patch
// lists): -3..5 us per step, spread over E2<GEO>, EN3 and EN1.  (The same on the stress / plstrain / volume / ddp
// loads and on the output-only stores: no further gain, EN3 a little slower.)
#ifndef DES_E2_PIPE_LANDED
#define DES_E2_PIPE_LANDED 1
#endif
#define DES_STRAIN_LD pl_ld_nt
#define DES_STRAIN_ST pl_st_nt
// avg_*: Output::average_fields (output.cxx:327-370) folded into this pass when the end-of-step pass is (engine/launch.hpp):
// stress_avg / strain0 take the end-of-step stress / strain of the step BEFORE, which this pass holds in registers right
// after the pending rotation; dplstrain_avg takes this step's delta_plstrain where it is formed (avg_dpl: unless the
// classic end-of-step pass + k_average_fields follow this step, i.e. the last step of a call).  Null: not averaging.
// edv_*: the edvacc_surf update of a surface step whose S2 / S3 launches were left out and whose diffusion EN1 has
// just done (passes/en1.hpp: SurfPending): extra workgroups behind the stress-bc facet ones.  edv_etop = 0: none.
struct RotPending { const double *spin, *ddp; const unsigned char *topflag; int prev_dt; const d4 *vm; int outputs;
                    double *stress_avg, *strain0, *dplstrain_avg; int avg_dpl, qcsi;
                    int edv_etop; const int *edv_conn_surf; const double *edv_dh_n; double *edv_edvacc;
                    int fresh; };

// GEO = 1: this pass also does what is left of the end-of-step pass of the step before AND the strain
// rate of this step, from the nodal records it gathers anyway: compute_volume after the volume swap
// (geometry.cxx:170-201, dynearthsol.cxx:466-470), correct_surface_element for the elements of the top
// surface (bc.cxx:1670-1687), NMD_stress' pending increment, rotate_stress (fields.cxx:827-902),
// update_strain_rate (fields.cxx:415-476) -- in the reference's order, on the stress it holds in
// registers.  No E1 launch between two such steps (engine/launch.hpp: e2geo_ok); the coordinates,
// velocities and dt it uses are the ones that pass would have seen (nothing but the temperature moves
// in between, and steps with a compute_dt keep the fused E1).
// RotPending::fresh = 1 (the first step of a call that follows a finished call with nothing uploaded in between,
// engine/launch.hpp: fresh_ok): the end-of-step pass of the step before HAS run -- nothing is pending, volume[] / volume_old[]
// hold this step's values already -- but the strain rate is still formed here from the nodal records instead of being
// read back from an E1 launch that only exists to store it.
// RotPending::outputs = 0 (a step of a call that is not its last one, engine/launch.hpp): what no pass reads
// before the next E2<GEO> overwrites it -- strain_rate (recomputed from the nodal records every step),
// viscosity, delta_plstrain, volume_old -- is not stored: 72 of the pass's 192 B of stores per element.
// The last step of every call stores them all, so what a caller can download is what the reference holds.
// With GEO the first pass of two (DEFER = 1) stores the strain and the corrected strain-rate diagonal as
// soon as they are final -- before the constitutive law, whose registers they would otherwise sit next to --
// also for the elements it sets aside; the return-mapping pass (RM = 1) then leaves the strain alone.
// ---- the pipelined form of E2<GEO> (E2_update_stress_pipe below) ---------------------------------------------------
// What an element's stress update reads at its OWN index (planes, connectivity, marker word, top flag) -- there it arrives
// in LDS one tile AHEAD of the arithmetic -- and what it gathers through the connectivity (nodal records), requested at the
// top of its own tile.  Same values, same operations on them: e2_element<..., PIPE = 1> only takes them from here instead
// of loading them itself.
// (the stress and strain planes stay in LDS until the arithmetic wants them -- lp[k * 64], k = 0-5 stress, 6-11 strain --:
//  held in registers from the top of the tile they cost 24 VGPRs across the geometry, i.e. spills)
struct E2Pre { int4 cn; int mono, top; double vol_prev, pls, dd; const double *lp; };
// what the pipelined kernel does once an element has taken its last value out of LDS (request the next tile's pieces)
struct E2NoAfter { __device__ __forceinline__ void operator()() const {} };
struct E2Gath { d4 c[4], v[4]; double nt[4]; };

__device__ __forceinline__ void e2_gather(E2Gath &G, const int4 cn, const d4 *__restrict__ xt, const d4 *__restrict__ vm,
                                          const double *__restrict__ ntmp)
{
    G.c[0] = rec_ld(xt, cn.x); G.c[1] = rec_ld(xt, cn.y); G.c[2] = rec_ld(xt, cn.z); G.c[3] = rec_ld(xt, cn.w);
    G.v[0] = rec_ld(vm, cn.x); G.v[1] = rec_ld(vm, cn.y); G.v[2] = rec_ld(vm, cn.z); G.v[3] = rec_ld(vm, cn.w);
    G.nt[0] = rec_ld(ntmp, cn.x); G.nt[1] = rec_ld(ntmp, cn.y); G.nt[2] = rec_ld(ntmp, cn.z); G.nt[3] = rec_ld(ntmp, cn.w);
}

#ifdef DES_STAMPS
// instrumented builds: where inside the element code a wavefront of the pipelined kernel spends its tile (sums per wavefront)
struct E2Stamps { unsigned long long t, geo, dma, strain, visc, law, store; };
#define DES_E2_STAMP(field) do { if (PIPE && est) { const unsigned long long now_ = wall_clock64(); est->field += now_ - est->t; est->t = now_; } } while (0)
#else
struct E2Stamps;
#define DES_E2_STAMP(field) do {} while (0)
#endif
template <class M, int DEFER, int GEO, int RM = 0, int RH = 0, int PIPE = 0, class After = E2NoAfter>
__device__ __forceinline__ bool e2_element(const int e, const des_params *__restrict__ p, const desk::ViscTerms *__restrict__ vt,
     const DevClock *__restrict__ clk, int ne, const int4 *__restrict__ conn, const d4 *__restrict__ xt,
     const double *__restrict__ ntmp, const MatData &md,
     double *__restrict__ volume, double *__restrict__ volume_old,
     double *__restrict__ stress, double *__restrict__ strain, double *__restrict__ strain_rate,
     double *__restrict__ plstrain, double *__restrict__ delta_plstrain, double *__restrict__ viscosity,
     double *__restrict__ dpressure, double *__restrict__ etmp2, const RotPending rp,
     const E2Pre *__restrict__ pre = nullptr, const E2Gath *__restrict__ gath = nullptr, const After after = After(),
     E2Stamps *est = nullptr)
{
    static_assert(!PIPE || (GEO && !DEFER && !RM), "the pipelined form is the one-pass E2<GEO>");
    const double dt = clk->dt;
    const unsigned eo = (unsigned)e * 8u;
    const int4 cn = PIPE ? pre->cn : rec_ld(conn, e);
    // (RH != 0: the launch is the common one of that rheology -- NMD_stress on, no averaged output fields -- and the kernel
    //  holds that path only: the law and the two switches known at compile time, -2.5 us of 78 at 1M tets for evp)
    const int rheol = RH ? RH : p->rheol_type;
    const bool nmd_on = RH ? true : (bool)p->is_using_mixed_stress, averaging = RH ? false : true;
    const desk::Mix mx = PIPE ? mix_from_mono(md, p->nmat, e, pre->mono) : mix_of(md, p->nmat, e);
    const ElemProps pr = load_props(p, md, mx, ne, e);

    double dj = 0;
    if (PIPE) { dj += gath->nt[0]; dj += gath->nt[1]; dj += gath->nt[2]; dj += gath->nt[3]; }
    else { dj += rec_ld(ntmp, cn.x); dj += rec_ld(ntmp, cn.y); dj += rec_ld(ntmp, cn.z); dj += rec_ld(ntmp, cn.w); }
    const double edvoldt = dj / 4;

    const bool outs = !GEO || rp.outputs;
    // strain / strain-rate diagonal stored before the law: by the only pass (nothing re-reads them), and by the first
    // of two passes when the second can tell (GEO: it recomputes the strain rate and skips the strain, ES_DONE)
    constexpr bool ES_EARLY = !RM && (GEO || !DEFER);
    constexpr bool ES_DONE = GEO && RM;              // ... by the first pass: not touched here
    double s[6], es[6] = {0, 0, 0, 0, 0, 0}, edot[6];
    double g_vol = 0, g_vol_old = 0, g_pls = 0, g_T = 0;
    bool g_rescaled = false, g_top = false;
    if (GEO) {
        d4 c[4], v[4];
        if (PIPE) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { c[i] = gath->c[i]; v[i] = gath->v[i]; }
        } else {
            c[0] = rec_ld(xt, cn.x); c[1] = rec_ld(xt, cn.y); c[2] = rec_ld(xt, cn.z); c[3] = rec_ld(xt, cn.w);
            v[0] = rec_ld(rp.vm, cn.x); v[1] = rec_ld(rp.vm, cn.y); v[2] = rec_ld(rp.vm, cn.z); v[3] = rec_ld(rp.vm, cn.w);
        }
        g_T += c[0].w; g_T += c[1].w; g_T += c[2].w; g_T += c[3].w;
        g_T /= 4;
        const double vol_prev = PIPE ? pre->vol_prev : pl_ld(volume, 0, ne, eo);
        g_vol = desk::tet_volume(c);
        double rdv = 0.0;
        g_top = (PIPE ? pre->top : (int)rp.topflag[e]) != 0;
        if (rp.fresh) g_vol_old = pl_ld(volume_old, 0, ne, eo);      // (the swap is done: vol_prev is this step's volume)
        else if (g_top) { rdv = g_vol / vol_prev; g_vol_old = g_vol; }     // correct_surface_element stored the new volume before the swap
        else g_vol_old = vol_prev;
        double sx[4], sy[4], sz[4];
        desk::shape_fn(c, g_vol, sx, sy, sz);
        e1_strain_rate_diag(v, sx, sy, sz, edot[0], edot[1], edot[2]);
        edot[3] = 0; for (int i = 0; i < 4; ++i) edot[3] += 0.5 * (v[i].x * sy[i] + v[i].y * sx[i]);
        edot[4] = 0; for (int i = 0; i < 4; ++i) edot[4] += 0.5 * (v[i].x * sz[i] + v[i].z * sx[i]);
        edot[5] = 0; for (int i = 0; i < 4; ++i) edot[5] += 0.5 * (v[i].y * sz[i] + v[i].z * sy[i]);
        // the shear components are final (only the diagonal is corrected below): out now, not held to the end
        // (the return-mapping pass recomputes and rewrites the same values for the elements set aside)
        if (outs) for (int i = 3; i < 6; ++i) pl_st(strain_rate, i, ne, eo, edot[i]);
        double w3 = 0, w4 = 0, w5 = 0;
        for (int i = 0; i < 4; ++i) w3 += 0.5 * (v[i].x * sy[i] - v[i].y * sx[i]);
        for (int i = 0; i < 4; ++i) w4 += 0.5 * (v[i].x * sz[i] - v[i].z * sx[i]);
        for (int i = 0; i < 4; ++i) w5 += 0.5 * (v[i].y * sz[i] - v[i].z * sy[i]);
        for (int i = 0; i < 6; ++i) {
            s[i] = PIPE ? pre->lp[i * 64] : pl_ld(stress, i, ne, eo);
            if (!ES_DONE) es[i] = PIPE ? pre->lp[(6 + i) * 64] : DES_STRAIN_LD(strain, i, ne, eo);
        }
        DES_E2_STAMP(geo);
        if (PIPE) after();                 // everything of this tile is out of LDS: the next tile's pieces may land
        DES_E2_STAMP(dma);
        g_pls = PIPE ? pre->pls : pl_ld(plstrain, 0, ne, eo);
        const double dd = PIPE ? pre->dd : (rp.ddp ? pl_ld(rp.ddp, 0, ne, eo) : 0.0);
        if (dd != 0.0) for (int i = 0; i < 3; ++i) s[i] += dd;
        if (rdv >= 1.0) {                                                      // bc.cxx:1677
            g_pls /= rdv;
            for (int i = 0; i < 6; ++i) { s[i] /= rdv; if (!ES_DONE) es[i] /= rdv; }
            g_rescaled = true;
        }
        if ((rheol & DES_RH_ELASTIC) && !rp.fresh) {
            const double dtr = rp.prev_dt ? clk->dt_prev : dt;             // the dt of the step being finished
            desk::jaumann_rate_3d(s, dtr, w3, w4, w5);
            if (!ES_DONE) desk::jaumann_rate_3d(es, dtr, w3, w4, w5);
        }
        if (averaging && rp.stress_avg && !RM) {
            // average_fields of the step this block has just finished (number clk->steps - 1: EN1 has counted on)
            const bool first = (clk->steps - 1) % rp.qcsi == 1;
            for (int i = 0; i < 6; ++i) pl_st(rp.stress_avg, i, ne, eo, first ? s[i] : pl_ld(rp.stress_avg, i, ne, eo) + s[i]);
            if (first) for (int i = 0; i < 6; ++i) pl_st(rp.strain0, i, ne, eo, es[i]);
        }
    } else {
    for (int i = 0; i < 6; ++i) {
        s[i] = pl_ld(stress, i, ne, eo);
        es[i] = DES_STRAIN_LD(strain, i, ne, eo);
        edot[i] = pl_ld(strain_rate, i, ne, eo);
    }
    }
    if (!GEO && rp.spin && !rp.topflag[e]) {
        // the end of the step before, left here by E1<MODE_DEFER>: NMD_stress' increment of the diagonal
        // (geometry.cxx:316-331), then rotate_stress (fields.cxx:827-902) with that step's dt -- the same
        // operations in the same order as E1 does them in place
        const double dd = rp.ddp ? rp.ddp[e] : 0.0;
        if (dd != 0.0) for (int i = 0; i < 3; ++i) s[i] += dd;
        const double dtr = rp.prev_dt ? clk->dt_prev : dt;
        const double w3 = rp.spin[e], w4 = rp.spin[(size_t)ne + e], w5 = rp.spin[(size_t)2*ne + e];
        desk::jaumann_rate_3d(s, dtr, w3, w4, w5);
        desk::jaumann_rate_3d(es, dtr, w3, w4, w5);
    }
    const double old_s = desk::trace3(s);
    {
        double div = desk::trace3(edot);
        for (int i = 0; i < 3; ++i) edot[i] += (edvoldt - div) / 3;
    }
    if (!ES_DONE) for (int i = 0; i < 6; ++i) es[i] += edot[i] * dt;
    if (ES_EARLY) {
        for (int i = 0; i < 6; ++i) DES_STRAIN_ST(strain, i, ne, eo, es[i]);
        if (outs) for (int i = 0; i < 3; ++i) pl_st(strain_rate, i, ne, eo, edot[i]);
    }
    double de[6];
    for (int i = 0; i < 6; ++i) de[i] = edot[i] * dt;
    double dpl = 0.;
    bool defer = false;
    const double vol = GEO ? g_vol : pl_ld(volume, 0, ne, eo);
    const double vol_old = GEO ? g_vol_old : ((rheol == DES_RH_MAXWELL || rheol == DES_RH_EVP) ? pl_ld(volume_old, 0, ne, eo) : 0.0);   // (dies at dv)

    if (!PIPE) M::stage_end();         // (the pipelined kernel stages the libm tables once per workgroup, ahead of its tile loop)
    DES_E2_STAMP(strain);
    double visc = 0;
    if (rheol & DES_RH_VISCOUS) {
        double T = 0;
        if (GEO) T = g_T;
        else {
            T += xt[cn.x].w; T += xt[cn.y].w; T += xt[cn.z].w; T += xt[cn.w].w;
            T /= 4;
        }
        visc = desk::mat_visc<M>(p, vt, mx, T, s, edot);
        if (outs) pl_st(viscosity, 0, ne, eo, visc);
    }

    DES_E2_STAMP(visc);
    switch (rheol) {
    case DES_RH_ELASTIC:
        desk::elastic(pr.bulkm, pr.shearm, de, s);
        break;
    case DES_RH_VISCOUS:
        desk::viscous(pr.bulkm, visc, desk::trace3(es), edot, s);
        break;
    case DES_RH_MAXWELL: {
        double dv = vol / vol_old - 1;
        desk::maxwell(pr.bulkm, pr.shearm, visc, dt, dv, de, s);
        break;
    }
    case DES_RH_EP: {
        double amc, anphi, anpsi, hardn, ten_max;
        double pls = GEO ? g_pls : pl_ld(plstrain, 0, ne, eo);
        desk::plastic_props<M>(p, mx, pls, amc, anphi, anpsi, hardn, ten_max, md.pptab);
        double depls = desk::elasto_plastic<M, DEFER>(pr.bulkm, pr.shearm, amc, anphi, anpsi, hardn, ten_max, de, s, &defer);
        if (DEFER && defer) return true;
        if (depls != 0 || g_rescaled) pl_st(plstrain, 0, ne, eo, pls + depls);       // plstrain += 0 is the identity
        g_rescaled = false;
        dpl = depls;
        break;
    }
    case DES_RH_EVP: {
        double dv = vol / vol_old - 1;
        double sv[6];
        for (int i = 0; i < 6; ++i) sv[i] = s[i];
        desk::maxwell(pr.bulkm, pr.shearm, visc, dt, dv, de, sv);
        double svII = desk::second_invariant2(sv);
        double amc, anphi, anpsi, hardn, ten_max;
        double pls = GEO ? g_pls : pl_ld(plstrain, 0, ne, eo);
        desk::plastic_props<M>(p, mx, pls, amc, anphi, anpsi, hardn, ten_max, md.pptab);
        double sp[6];
        for (int i = 0; i < 6; ++i) sp[i] = s[i];
        double depls = desk::elasto_plastic<M, DEFER>(pr.bulkm, pr.shearm, amc, anphi, anpsi, hardn, ten_max, de, sp, &defer);
        if (DEFER && defer) return true;
        double spII = desk::second_invariant2(sp);
        if (svII < spII) {
            for (int i = 0; i < 6; ++i) s[i] = sv[i];
        } else {
            for (int i = 0; i < 6; ++i) s[i] = sp[i];
            pl_st(plstrain, 0, ne, eo, pls + depls);
            dpl = depls;
            g_rescaled = false;
        }
        break;
    }
    default: break;
    }
    DES_E2_STAMP(law);
#if DES_E2_PIPE_LANDED
    // Pipelined form: the next tile's LDS-DMA pieces were requested half a tile ago (after(), behind the geometry) and the
    // early stores of this tile before the law -- by now all of them have long completed, so this wait costs nothing; what it
    // buys is an INVARIANT instead of a count: every piece has landed before the last group of stores is issued, whatever
    // the compiler makes of the stores (the first form of this kernel waited at the head of the next tile with vmcnt(12),
    // i.e. relied on at least 13 younger stores in every tile).
    if (PIPE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    if (GEO) {
        if (g_rescaled) pl_st(plstrain, 0, ne, eo, g_pls);                 // rescaled by correct_surface_element, not changed by the law
        if (outs && !rp.fresh) pl_st(volume_old, 0, ne, eo, g_top ? vol : pl_ld(volume, 0, ne, eo));             // (re-read rather than held in registers through the update)
        pl_st(volume, 0, ne, eo, vol);
    }
    if (outs) pl_st(delta_plstrain, 0, ne, eo, dpl);
    if (averaging && rp.dplstrain_avg && rp.avg_dpl) {
        const bool first = clk->steps % rp.qcsi == 1;
        pl_st(rp.dplstrain_avg, 0, ne, eo, first ? dpl : pl_ld(rp.dplstrain_avg, 0, ne, eo) + dpl);
    }
    for (int i = 0; i < 6; ++i) {
        pl_st(stress, i, ne, eo, s[i]);
        if (!ES_EARLY && !ES_DONE) DES_STRAIN_ST(strain, i, ne, eo, es[i]);
    }
    if (!ES_EARLY && !ES_DONE && outs)
        for (int i = 0; i < 3; ++i) pl_st(strain_rate, i, ne, eo, edot[i]);   // only the diagonal changed (GEO: all six are new)
    if (nmd_on) {
        double dp = desk::trace3(s) - old_s;
        pl_st(dpressure, 0, ne, eo, dp);
        pl_st(etmp2, 0, ne, eo, dp * vol);
    }
    DES_E2_STAMP(store);
    return defer;                  // went past the yield pre-filter
}

// First pass: every element [e_begin, e_begin + e_count) (the whole local mesh, or a sub-range:
// ne stays the SoA plane stride).  DEFER = 1: elements that need the return mapping are appended
// to `list` (wave-aggregated: one atomic per wavefront) for E2_return_mapping.
// first pass with the geometry part: 206 VGPRs; held to the 168 of three waves per SIMD it spills 150 B per
// lane and takes 137 us instead of 89 (1M tets)
#ifndef DES_E2GEO_WAVES
#define DES_E2GEO_WAVES DES_E2_WAVES
#endif
// W3 = 1: the same kernel held to three waves per SIMD (168 VGPRs, a few dozen bytes of scratch).  At 1M tets that is
// no faster than two (the pass is not short of waves), but a SHARD of a strong-scaling run can be: 152,589 local tets
// are 2,385 wavefronts against 2,048 resident at two per SIMD -- a second, nearly empty round of workgroups -- and
// 3,072 at three: one round.  engine/launch.hpp (e2_three_waves) picks by the launch's wavefront count.
template <class M, int DEFER, int GEO, int RH = 0, int W3 = 0>
__global__ void __launch_bounds__(DES_BLOCK, W3 ? 3 : (DEFER ? (GEO ? DES_E2GEO_WAVES : DES_E2_WAVES_FAST) : DES_E2_WAVES))
E2_update_stress(const des_params *__restrict__ p, const desk::ViscTerms *__restrict__ vt, const DevClock *__restrict__ clk,
     int ne, int e_begin, int e_count, int e_begin2, int e_count2, int nblocks, const int4 *__restrict__ conn, const d4 *__restrict__ xt,
     const double *__restrict__ ntmp, const MatData md,
     double *__restrict__ volume, double *__restrict__ volume_old,
     double *__restrict__ stress, double *__restrict__ strain, double *__restrict__ strain_rate,
     double *__restrict__ plstrain, double *__restrict__ delta_plstrain, double *__restrict__ viscosity,
     double *__restrict__ dpressure, double *__restrict__ etmp2, int *__restrict__ list, int *__restrict__ count,
     int nblocks8, int nbcf, const int *__restrict__ f_elem, const int *__restrict__ f_facet,
     const int *__restrict__ f_kind, const double *__restrict__ f_val, double *__restrict__ f_tmp, const RotPending rp)
{
    if ((int)blockIdx.x >= nblocks8) {
        // workgroups past the element range: the stress-bc facet terms (passes/e3.hpp, bc_facet_work).
        // They need the temperature N1 has just updated and the coordinates of this step, and the
        // force pass consumes them; with EN3 there is no E3 launch for them to ride in.
        const int g = ((int)blockIdx.x - nblocks8) * DES_BLOCK + threadIdx.x;
        const int nbcf_pad = (nbcf + DES_BLOCK - 1) / DES_BLOCK * DES_BLOCK;
        if (g < nbcf) bc_facet_work(p, g, conn, xt, md, f_elem, f_facet, f_kind, f_val, f_tmp);
        else if (g >= nbcf_pad && g - nbcf_pad < rp.edv_etop)
            edvacc_facet(g - nbcf_pad, rp.edv_etop, rp.edv_conn_surf, xt, rp.edv_dh_n, rp.edv_edvacc);
        return;
    }
    M::stage_begin();
    // (two element ranges: the overlapped multi-GPU schedule runs the deep elements first, then the two groups around them)
    const int el = desk::logical_block(nblocks) * DES_BLOCK + threadIdx.x;
    if (el >= e_count + e_count2) return;
    const int e = el < e_count ? e_begin + el : e_begin2 + (el - e_count);
    const bool defer = e2_element<M, DEFER, GEO, 0, RH>(e, p, vt, clk, ne, conn, xt, ntmp, md, volume, volume_old, stress, strain, strain_rate,
                                            plstrain, delta_plstrain, viscosity, dpressure, etmp2, rp);
    // one atomic per wavefront that has such elements; without DEFER only the count is kept
    // (des_scalars::n_return_mapping, and what the host picks the next call's mode from)
    const unsigned long long mask = __ballot(defer);
    if (defer) {
        const int lane = threadIdx.x & 63;
        const int leader = __ffsll((long long)mask) - 1;
        int base = 0;
        if (lane == leader) base = atomicAdd(count, __popcll(mask));
        if (DEFER) {
            base = __shfl(base, leader);
            list[base + __popcll(mask & ((1ull << lane) - 1))] = e;
        }
    }
}

// E2<GEO>, one pass, PIPELINED (round 4).  The plain kernel gives every wavefront one tile of 64 elements: index -> planes
// and nodal records -> arithmetic -> stores, four to five dependent trips to memory, ~1150 fp64 instructions, and at two
// waves per SIMD (175 VGPRs) the two hardly overlap -- the pass takes what its memory shape takes alone PLUS most of its
// issue time (profiles/r03_b_pmc_counters_1M.txt: VALUBusy 38 %, a wave waits 60 % of its life).  Here 2 x CUs
// workgroups stay resident and walk the tiles of their XCD's share of the mesh, and every wavefront has the own-index data
// of its NEXT 64 elements -- fifteen planes, connectivity, marker word, top flag: 9.3 KB -- on its way into a private LDS
// region while it computes the current ones: LDS-DMA (global_load_lds: no destination registers, so nothing is held in
// VGPRs across the 1150-instruction body; holding them there spilled 200 B per lane).  Per tile a wavefront
//   waits for its own DMA (counted: vmcnt is in issue order and only this tile's stores are younger), reads the 36
//   values per lane out of LDS, issues the nodal gathers, THEN the next tile's DMA (so that the gathers' waits never
//   cover it), computes, stores.
// No workgroup barrier in the loop: a wavefront only ever reads what it requested itself.  The libm tables are staged
// once per workgroup instead of once per tile.  Same element code (e2_element<..., PIPE = 1>): same operations, same bits.
// Needs: one element range starting at 0 (not the split parts of the overlapped schedule), an even plane stride (16-byte
// DMA pieces), and the 4 KB of slack dev_alloc() leaves behind every array (the last tile reads whole 64-element pieces).
// npers: resident workgroups (a multiple of 8: blockIdx.x & 7 = the XCD under round-robin placement, for locality only).
// DES_E2_PIPE_LANDED = 1 (default): "this tile's pieces have landed" is established inside the tile before (e2_element: one
// s_waitcnt vmcnt(0) in front of the last group of stores); 0: the first form, a counted wait at the head of the tile.
// (Round 5, measured and dropped: the wave-tiles handed out one by one from a counter per XCD share -- one atomic per
//  wavefront and tile, drawn a tile ahead -- 195 against 67 us: two thousand same-address device-scope atomics per counter
//  and launch serialise at the memory side; profiles/r05_c_ab_patch_variants.txt.  The 7-or-8-tiles tail it was aimed at is
//  granularity, not imbalance: 15,646 wave-tiles over 2,048 resident wavefronts are 7.64 rounds however they are dealt.)
typedef const __attribute__((address_space(1))) void *des_gptr;
typedef __attribute__((address_space(3))) void *des_lptr;
// NW = wavefronts per workgroup: 4 (two workgroups per CU = two waves per SIMD).  Tried: 12 -- ONE workgroup of 768 lanes per
// CU, three waves per SIMD at 168 VGPRs, LDS 136 KB -- spills 350 B per lane (the loop wants ~250 registers); not instantiated.
template <class M, int RH, int NW>
__global__ void __launch_bounds__(NW * 64, NW == 4 ? DES_E2_WAVES : 3)
E2_update_stress_pipe(const des_params *__restrict__ p, const desk::ViscTerms *__restrict__ vt, const DevClock *__restrict__ clk,
     int ne, int ntiles, int npers, const int4 *__restrict__ conn, const d4 *__restrict__ xt,
     const double *__restrict__ ntmp, const MatData md,
     double *__restrict__ volume, double *__restrict__ volume_old,
     double *__restrict__ stress, double *__restrict__ strain, double *__restrict__ strain_rate,
     double *__restrict__ plstrain, double *__restrict__ delta_plstrain, double *__restrict__ viscosity,
     double *__restrict__ dpressure, double *__restrict__ etmp2, int *__restrict__ count,
     int nbcf, const int *__restrict__ f_elem, const int *__restrict__ f_facet,
     const int *__restrict__ f_kind, const double *__restrict__ f_val, double *__restrict__ f_tmp, const RotPending rp)
{
    if ((int)blockIdx.x >= npers) {
        // workgroups past the resident ones: the stress-bc facet terms and the edvacc_surf update (as in E2_update_stress)
        const int g = ((int)blockIdx.x - npers) * (NW * 64) + threadIdx.x;
        const int nbcf_pad = (nbcf + NW * 64 - 1) / (NW * 64) * (NW * 64);
        if (g < nbcf) bc_facet_work(p, g, conn, xt, md, f_elem, f_facet, f_kind, f_val, f_tmp);
        else if (g >= nbcf_pad && g - nbcf_pad < rp.edv_etop)
            edvacc_facet(g - nbcf_pad, rp.edv_etop, rp.edv_conn_surf, xt, rp.edv_dh_n, rp.edv_edvacc);
        return;
    }
    __shared__ double lpl[NW][16][64];                   // per wavefront: stress 0-5, strain 6-11, volume 12, plstrain 13, ddp 14, (pad 15)
    __shared__ int4 lcn[NW][64];
    __shared__ int lmono[NW][64];
    __shared__ unsigned char ltop[NW][256];
    M::stage_begin();
    M::stage_end();
    // (w through readfirstlane: the LDS base of a DMA piece goes through M0 and must be known to be wave-uniform)
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = (int)(threadIdx.x & 63);
    // Every wavefront walks WAVE-TILES of 64 elements.  XCD x sweeps the x-th eighth of the 256-element tiles (desk::logical_block's
    // chunks), i.e. the wave-tiles [x per NW, wt_end); its npers / 8 resident workgroups -- nwx wavefronts -- share them: statically
    // interleaved (the four wavefronts of a workgroup side by side on one 256-element tile).
    const int per = (ntiles + 7) >> 3, wx = npers >> 3;
    const int x = (int)(blockIdx.x & 7), j = (int)(blockIdx.x >> 3);
    const int wt_begin = x * per * NW, wt_end = min((x + 1) * per, ntiles) * NW;
    const int nwx = wx * NW;
    const bool have_ddp = rp.ddp != nullptr;
    // The DMA of one wave-tile.  A piece = one instruction = 64 lanes x 16 B landing at the LDS base + 16 lane.
    // 64 elements of a plane are 512 B, so a piece carries TWO planes: lanes 0-31 two consecutive elements of plane k each,
    // lanes 32-63 of plane k + 1 (the same array's next plane, or the partner array's) -- every lane active, no divergent
    // region around the requests.  Plane 15 is padding (the partner of ddp).  Then one connectivity record and one marker word
    // per lane, and the top flags.
    auto dma = [&](int wtile) {
        const size_t eb = (size_t)wtile * 64;
        const size_t o = eb + 2 * (size_t)(lane & 31);
        const size_t hi = (size_t)(lane >> 5);
#pragma unroll
        for (int k = 0; k < 6; k += 2) {
            __builtin_amdgcn_global_load_lds((des_gptr)(stress + ((size_t)k + hi) * ne + o), (des_lptr)&lpl[w][k][0], 16, 0, 0);
            // (aux 2 = non-temporal, as DES_STRAIN_LD: the strain is read once and written once per step)
            __builtin_amdgcn_global_load_lds((des_gptr)(strain + ((size_t)k + hi) * ne + o), (des_lptr)&lpl[w][6 + k][0], 16, 0, 2);
        }
        __builtin_amdgcn_global_load_lds((des_gptr)((hi ? plstrain : volume) + o), (des_lptr)&lpl[w][12][0], 16, 0, 0);
        if (have_ddp) __builtin_amdgcn_global_load_lds((des_gptr)(rp.ddp + o), (des_lptr)&lpl[w][14][0], 16, 0, 0);
        __builtin_amdgcn_global_load_lds((des_gptr)(conn + eb + lane), (des_lptr)&lcn[w][0], 16, 0, 0);
        __builtin_amdgcn_global_load_lds((des_gptr)(md.mono + eb + lane), (des_lptr)&lmono[w][0], 4, 0, 0);
        // (the top flags are bytes: a dword piece of 64 x 4 B covers them four times over -- bytes 0-63 are this tile's)
        __builtin_amdgcn_global_load_lds((des_gptr)(rp.topflag + eb + 4 * (size_t)lane), (des_lptr)&ltop[w][0], 4, 0, 0);
    };
    const int wt_first = wt_begin + j * NW + w, wt_step = nwx;
    int t = wt_first;
    if (t < wt_end) dma(t);
#ifdef DES_STAMPS
    unsigned long long st_wait = 0, st_issue = 0, st_gather = 0, st_body = 0, st_n = 0, st_a, st_b;
    E2Stamps est = {0, 0, 0, 0, 0, 0, 0};
    const unsigned long long st_begin = wall_clock64();
#endif
    while (t < wt_end) {
        const int e = t * 64 + lane;
        const int tn = t + wt_step;
#ifdef DES_STAMPS
        st_a = wall_clock64();
#endif
#if DES_E2_PIPE_LANDED
        // this tile's pieces: landed before the last stores of the tile before were issued (e2_element); the first tile's here
        if (t == wt_first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
        // This tile's DMA has landed once at most the vector-memory operations issued BEHIND it are outstanding: the stores
        // of the tile before (at least 13: stress 6, strain 6, volume) -- or nothing, for the first tile.
        if (t == wt_first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else               asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
#endif
#ifdef DES_STAMPS
        st_b = wall_clock64(); st_wait += st_b - st_a; st_a = st_b;
#endif
        E2Pre cur;
        cur.cn = lcn[w][lane]; cur.mono = lmono[w][lane]; cur.top = ltop[w][lane];
        cur.vol_prev = lpl[w][12][lane]; cur.pls = lpl[w][13][lane];
        cur.dd = have_ddp ? lpl[w][14][lane] : 0.0;
        cur.lp = &lpl[w][0][lane];
        const bool valid = e < ne;
        if (!valid) cur.cn = make_int4(0, 0, 0, 0);          // (lanes past the mesh: harmless gathers, nothing stored)
        E2Gath G;
        e2_gather(G, cur.cn, xt, rp.vm, ntmp);
        asm volatile("" ::: "memory");                     // the gathers are issued before the next tile's pieces (vmcnt is in order)
        // The region is free again once the element has read its stress and strain out of it -- behind the geometry, inside
        // e2_element --: there the next tile's pieces are requested (all lanes of the wavefront get there together: a lane
        // past the mesh runs the same code on its harmless gathers and stores nothing).
        bool dma_done = false;
        auto after = [&]() {
            // Every vector load of this tile (the gathers) must have ARRIVED before the next tile's pieces are requested, and the
            // compiler must know it: vmcnt counts in issue order, so a gather result first used behind the DMA would be waited
            // for with vmcnt(0) -- the DMA's whole trip to HBM, in the middle of the arithmetic (the first build of this kernel
            // had such a wait; taking it out changed nothing measurable, 71.9 us either way: the pass is issue-bound,
            // DESIGN.md section 5 -- it stays out because it is the right order).  The builtin (not inline asm) puts an s_waitcnt the
            // compiler's own scoreboard sees: vmcnt(0), expcnt / lgkmcnt untouched (gfx9 encoding 0x0F70).
            __builtin_amdgcn_s_waitcnt(0x0F70);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (tn < wt_end) dma(tn);
            asm volatile("" ::: "memory");
            dma_done = true;
        };
#ifdef DES_STAMPS
        st_b = wall_clock64(); st_issue += st_b - st_a; st_a = st_b;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the gathers have arrived
        st_b = wall_clock64(); st_gather += st_b - st_a; st_a = st_b;
#endif
        bool defer = false;
#ifdef DES_STAMPS
        est.t = wall_clock64();
        if (valid)
            defer = e2_element<M, 0, 1, 0, RH, 1>(e, p, vt, clk, ne, conn, xt, ntmp, md, volume, volume_old, stress, strain, strain_rate,
                                                  plstrain, delta_plstrain, viscosity, dpressure, etmp2, rp, &cur, &G, after, &est);
#else
        if (valid)
            defer = e2_element<M, 0, 1, 0, RH, 1>(e, p, vt, clk, ne, conn, xt, ntmp, md, volume, volume_old, stress, strain, strain_rate,
                                                  plstrain, delta_plstrain, viscosity, dpressure, etmp2, rp, &cur, &G, after);
#endif
        if (!dma_done) after();                            // (a wavefront wholly past the mesh)
        // the count of elements past the yield pre-filter (des_scalars::n_return_mapping): one atomic per wavefront that has any
        const unsigned long long mask = __ballot(defer);
        if (defer && lane == __ffsll((long long)mask) - 1) atomicAdd(count, __popcll(mask));
#ifdef DES_STAMPS
        st_b = wall_clock64(); st_body += st_b - st_a; ++st_n;
#endif
        t = tn;
    }
#ifdef DES_STAMPS
    if (lane == 0 && blockIdx.x * NW + w < DES_STAMP_WG) {
        const int q = blockIdx.x * NW + w;
        g_stamps[2][0][q] = st_begin; g_stamps[2][1][q] = wall_clock64(); g_stamps[2][2][q] = st_wait; g_stamps[2][3][q] = st_issue;
        g_stamps[2][4][q] = st_gather; g_stamps[2][5][q] = st_body; g_stamps[2][6][q] = st_n;
        // inside the element code (pass 3 of the stamp array): geometry, DMA issue, strain update, viscosity, law, stores
        g_stamps[3][0][q] = est.geo; g_stamps[3][1][q] = est.dma; g_stamps[3][2][q] = est.strain; g_stamps[3][3][q] = est.visc;
        g_stamps[3][4][q] = est.law; g_stamps[3][5][q] = est.store; g_stamps[3][6][q] = st_n;
    }
#endif
}

// Second pass: the elements the first pass set aside, full stress update with the return mapping
// (same code, same arithmetic; the order of the list does not matter, every element is its own).
template <class M, int GEO>
__global__ void __launch_bounds__(DES_BLOCK, DES_E2_WAVES)
E2_return_mapping(const des_params *__restrict__ p, const desk::ViscTerms *__restrict__ vt, const DevClock *__restrict__ clk,
     int ne, const int4 *__restrict__ conn, const d4 *__restrict__ xt,
     const double *__restrict__ ntmp, const MatData md,
     double *__restrict__ volume, double *__restrict__ volume_old,
     double *__restrict__ stress, double *__restrict__ strain, double *__restrict__ strain_rate,
     double *__restrict__ plstrain, double *__restrict__ delta_plstrain, double *__restrict__ viscosity,
     double *__restrict__ dpressure, double *__restrict__ etmp2, const int *__restrict__ list, const int *__restrict__ count,
     const RotPending rp)
{
    const int n = *count;
    if (n == 0) return;                 // (uniform) nothing was set aside: not even the libm tables are staged
    M::stage_begin();
    M::stage_end();
    for (int i = blockIdx.x * DES_BLOCK + threadIdx.x; i < n; i += gridDim.x * DES_BLOCK)
        e2_element<M, 0, GEO, 1>(list[i], p, vt, clk, ne, conn, xt, ntmp, md, volume, volume_old, stress, strain, strain_rate,
                         plstrain, delta_plstrain, viscosity, dpressure, etmp2, rp);
}
