// passes/e1.hpp -- Pass E1 (elements): end of step t + start of step t+1.
// Part of the single translation unit des_dev.hip (included inside namespace des_hip, after
// DevClock / struct des_dev); not a stand-alone header.

// Element terms that both E1 and the node-patch pass EN1 form -- one source, so that the two passes
// produce the same bits from the same nodal records.
// compute_mass, element part (geometry.cxx:1795-1840): inertial and thermal mass of a quarter element
__device__ __forceinline__ void e1_mass_terms(const des_params *__restrict__ p, const ElemProps &pr, double rho, double vol,
                                              double &m, double &tm)
{
    const double pseudo_speed = p->max_vbc_val * p->inertial_scaling;
    double rho_m = p->is_quasi_static ? pr.bulkm / (pseudo_speed * pseudo_speed) : rho;
    m = rho_m * vol / 4;
    tm = rho * pr.cp * vol / 4;
}
// update_temperature, element part (fields.cxx:211-239)
__device__ __forceinline__ void e1_thermal_terms(const d4 c[4], const double sx[4], const double sy[4], const double sz[4],
                                                 double k, double vol, double radiogenic, double rho, double tr[4])
{
    double kv = k * vol;
    double rh = radiogenic * vol * rho / 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        double diffusion = 0.;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            diffusion += (sx[i] * sx[j] + sy[i] * sy[j] + sz[i] * sz[j]) * c[j].w;
        tr[i] = diffusion * kv - rh;
    }
}
// update_strain_rate, the three diagonal components (fields.cxx:415-440): their sum times the
// volume is compute_dvoldt's element term (geometry.cxx:218-224)
__device__ __forceinline__ void e1_strain_rate_diag(const d4 v[4], const double sx[4], const double sy[4], const double sz[4],
                                                    double &s0, double &s1, double &s2)
{
    s0 = 0; for (int i = 0; i < 4; ++i) s0 += v[i].x * sx[i];
    s1 = 0; for (int i = 0; i < 4; ++i) s1 += v[i].y * sy[i];
    s2 = 0; for (int i = 0; i < 4; ++i) s2 += v[i].z * sz[i];
}

// ---- E1 --------------------------------------------------------------------------
// MODE_C: compute_volume (geometry.cxx:170-201) after the volume swap (dynearthsol.cxx:466-470),
//         compute_mass element part (geometry.cxx:1795-1840), rotate_stress (fields.cxx:827-902);
//         MODE_DT adds the compute_dt reduction (geometry.cxx:1513-1593).
// MODE_A: update_temperature element part (fields.cxx:211-239), update_strain_rate
//         (fields.cxx:415-476), compute_dvoldt element part (geometry.cxx:218-224).
template <int MODE>
__global__ void __launch_bounds__(DES_BLOCK, DES_E1_WAVES)
E1_geom_rotate_strainrate(const des_params *__restrict__ p, DevClock *__restrict__ clk, int ne, int nblocks,
     int b0, int c0, int b1, int c1, const int4 *__restrict__ conn, const d4 *__restrict__ xt, const d4 *__restrict__ vm,
     const MatData md,
     const double *__restrict__ radiogenic, const unsigned char *__restrict__ topflag,
     double *__restrict__ stress, const double *__restrict__ ddp, double *__restrict__ strain, double *__restrict__ plstrain,
     double *__restrict__ volume, double *__restrict__ volume_old,
     double *__restrict__ strain_rate, d4 *__restrict__ mrec, d4 *__restrict__ ttmp, double *__restrict__ spin,
     double *__restrict__ dt_part, int dt_cap, int dt_base)
{
    // this launch covers the elements [b0, b0 + c0) and [b1, b1 + c1) (the whole mesh: 0, ne, 0, 0;
    // the overlapped multi-GPU schedule runs the interior elements while the ghost region is
    // still on its way, then the two groups that touch it); ne stays the SoA plane stride
    const int el = desk::logical_block(nblocks) * DES_BLOCK + threadIdx.x;
    const bool active = el < c0 + c1;
    const int e = el < c0 ? b0 + el : b1 + (el - c0);

    double r_minl = DBL_MAX, r_maxw = DBL_MAX, r_diff = DBL_MAX, r_gdt = DBL_MAX, r_vem = 0.0;

    if (active) {
        const int4 cn = conn[e];
        d4 c[4], v[4];
        c[0] = xt[cn.x]; c[1] = xt[cn.y]; c[2] = xt[cn.z]; c[3] = xt[cn.w];
        v[0] = vm[cn.x]; v[1] = vm[cn.y]; v[2] = vm[cn.z]; v[3] = vm[cn.w];
        const desk::Mix mx = mix_of(md, p->nmat, e);
        const ElemProps pr = load_props(p, md, mx, ne, e);

        // mean nodal temperature, matprops.cxx:338-343
        double T = 0;
        T += c[0].w; T += c[1].w; T += c[2].w; T += c[3].w;
        T /= 4;
        const double rho = desk::mat_rho(p, mx, T);

        double vol;
        d4 rec;
        double rdv = 0.0;                        // >= 1: correct_surface_element rescales this element
        // update_mesh only runs on a moving mesh (dynearthsol.cxx:870-873; always in the isostasy
        // loop): without it volumes and masses keep their values, rotate_stress below still runs
        const bool remesh_geom = (MODE & MODE_INIT) || p->has_moving_mesh || clk->iso;
        if ((MODE & MODE_C) && !remesh_geom) {
            vol = volume[e];
            const d4 old = mrec[e];
            rec.x = old.x; rec.y = old.y; rec.z = old.z;
        } else if (MODE & MODE_C) {
            const double vol_prev = volume[e];
            vol = desk::tet_volume(c);
            if (!(MODE & MODE_INIT) && topflag[e] && !clk->pt) {     // (no surface processes inside the PT loop)
                // correct_surface_element (bc.cxx:1670-1687) runs before the swap: it already
                // stored the new volume, so the swap moves the NEW volume into volume_old
                rdv = vol / vol_prev;
                volume_old[e] = vol;
            } else {
                volume_old[e] = vol_prev;        // pointer swap of dynearthsol.cxx:466-470
            }
            volume[e] = vol;
            // compute_mass, element part
            rec.x = vol;
            e1_mass_terms(p, pr, rho, vol, rec.y, rec.z);
        } else {
            vol = (MODE & MODE_VOLX) ? desk::tet_volume(c) : volume[e];
            if (MODE & MODE_A) {
                const d4 old = mrec[e];
                rec.x = old.x; rec.y = old.y; rec.z = old.z;
            }
        }

        double sx[4], sy[4], sz[4];
        desk::shape_fn(c, vol, sx, sy, sz);

        if ((MODE & MODE_DEFER) && !topflag[e]) {
            // rotate_stress of this element happens in the next step's stress update (passes/e2.hpp), which
            // has its stress and strain in registers anyway: 24 B written here instead of 192 B moved.  The
            // top elements (correct_surface_element may rescale them first) are finished here as usual.
            double w3 = 0, w4 = 0, w5 = 0;
            for (int i = 0; i < 4; ++i) w3 += 0.5 * (v[i].x * sy[i] - v[i].y * sx[i]);
            for (int i = 0; i < 4; ++i) w4 += 0.5 * (v[i].x * sz[i] - v[i].z * sx[i]);
            for (int i = 0; i < 4; ++i) w5 += 0.5 * (v[i].y * sz[i] - v[i].z * sy[i]);
            spin[e] = w3; spin[(size_t)ne + e] = w4; spin[(size_t)2*ne + e] = w5;
        } else if ((MODE & MODE_C) && !(MODE & MODE_INIT)) {
            const bool rescale = rdv >= 1.0;                         // bc.cxx:1677
            const bool rotate = (p->rheol_type & DES_RH_ELASTIC) != 0 && !clk->iso && !clk->pt;   // not in the isostasy / PT loops
            // NMD_stress' increment of the diagonal (geometry.cxx:316-331), left here by EN3 -- the
            // operation E3 would have done in place, before anything else touches the stress
            double dd = 0.0;
            if (ddp && p->is_using_mixed_stress && !clk->iso) dd = ddp[e];
            if (dd != 0.0 && !(rescale || rotate))
                for (int i = 0; i < 3; ++i) stress[(size_t)i*ne + e] += dd;
            if (rescale || rotate) {
                double s[6], es[6];
                for (int i = 0; i < 6; ++i) { s[i] = stress[(size_t)i*ne + e]; es[i] = strain[(size_t)i*ne + e]; }
                if (dd != 0.0) for (int i = 0; i < 3; ++i) s[i] += dd;
                if (rescale) {
                    plstrain[e] /= rdv;
                    for (int i = 0; i < 6; ++i) { s[i] /= rdv; es[i] /= rdv; }
                    if (!(MODE & MODE_A))            // otherwise update_strain_rate overwrites it below
                        for (int i = 0; i < 6; ++i) strain_rate[(size_t)i*ne + e] /= rdv;
                }
                if (rotate) {
                    const double dt = clk->dt;
                    double w3 = 0, w4 = 0, w5 = 0;
                    for (int i = 0; i < 4; ++i) w3 += 0.5 * (v[i].x * sy[i] - v[i].y * sx[i]);
                    for (int i = 0; i < 4; ++i) w4 += 0.5 * (v[i].x * sz[i] - v[i].z * sx[i]);
                    for (int i = 0; i < 4; ++i) w5 += 0.5 * (v[i].y * sz[i] - v[i].z * sy[i]);
                    desk::jaumann_rate_3d(s, dt, w3, w4, w5);
                    desk::jaumann_rate_3d(es, dt, w3, w4, w5);
                }
                if (rescale || rotate)
                    for (int i = 0; i < 6; ++i) { stress[(size_t)i*ne + e] = s[i]; strain[(size_t)i*ne + e] = es[i]; }
            }
        }

        if (MODE & MODE_DT) {
            double vx = 0.0, vy = 0.0, vz = 0.0;
            const double weight = 1.0 / 4;
            for (int j = 0; j < 4; ++j) { vx += v[j].x * weight; vy += v[j].y * weight; vz += v[j].z * weight; }
            r_vem = vx*vx + vy*vy + vz*vz;           // its square root is taken after the maximum over the block (monotone)
            // max of the four facet areas = sqrt(max of the radicands) / 2, to the bit
            double maxa = sqrt(fmax(fmax(desk::tri_area_sq4(c[0], c[1], c[2]), desk::tri_area_sq4(c[0], c[1], c[3])),
                                    fmax(desk::tri_area_sq4(c[2], c[3], c[0]), desk::tri_area_sq4(c[2], c[3], c[1])))) / 2;
            double minh = 3 * vol / maxa;
            r_maxw = 0.5 * p->visc_min / (1e-40 + pr.shearm);
            if (p->has_thermal_diffusion) r_diff = 0.5 * minh * minh / p->therm_diff_max;
            r_minl = minh;
            r_gdt = minh / sqrt(pr.shearm / rho) / 5.0;
        }

        if (MODE & MODE_A) {
            if (p->has_thermal_diffusion && !(MODE & MODE_NOREC)) {
                double trp[4];
                e1_thermal_terms(c, sx, sy, sz, pr.k, vol, radiogenic[e], rho, trp);
                d4 tr = {trp[0], trp[1], trp[2], trp[3]};
                ttmp[e] = tr;
            }
            double s[6];
            e1_strain_rate_diag(v, sx, sy, sz, s[0], s[1], s[2]);
            s[3] = 0; for (int i = 0; i < 4; ++i) s[3] += 0.5 * (v[i].x * sy[i] + v[i].y * sx[i]);
            s[4] = 0; for (int i = 0; i < 4; ++i) s[4] += 0.5 * (v[i].x * sz[i] + v[i].z * sx[i]);
            s[5] = 0; for (int i = 0; i < 4; ++i) s[5] += 0.5 * (v[i].y * sz[i] + v[i].z * sy[i]);
            for (int i = 0; i < 6; ++i) strain_rate[(size_t)i*ne + e] = s[i];
            double dj = s[0] + s[1] + s[2];
            rec.w = dj * vol;
        } else {
            rec.w = 0;
        }
        // MODE_NOREC (with C | A): the next pass is EN1, which forms mrec / ttmp itself from the nodal records
        if ((MODE & (MODE_C | MODE_A)) && !(MODE & MODE_NOREC)) mrec[e] = rec;
    }

    if (MODE & MODE_DT) {
        __shared__ double red[5][DES_BLOCK / 64];
        r_minl = desk::wave_min(r_minl); r_maxw = desk::wave_min(r_maxw); r_diff = desk::wave_min(r_diff);
        r_gdt = desk::wave_min(r_gdt);   r_vem = desk::wave_max(r_vem);
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        if (lane == 0) { red[0][w] = r_minl; red[1][w] = r_maxw; red[2][w] = r_diff; red[3][w] = r_gdt; red[4][w] = r_vem; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int i = 1; i < DES_BLOCK / 64; ++i) {
                red[0][0] = fmin(red[0][0], red[0][i]); red[1][0] = fmin(red[1][0], red[1][i]);
                red[2][0] = fmin(red[2][0], red[2][i]); red[3][0] = fmin(red[3][0], red[3][i]);
                red[4][0] = fmax(red[4][0], red[4][i]);
            }
            // one slot per workgroup, reduced by the next k_dt_finalize / k_dt_pack (dt_reduce_partials): thousands
            // of atomics on the five words of one cache line took longer than the pass itself
            double *q = dt_part + dt_base + blockIdx.x;
            q[0] = red[0][0]; q[dt_cap] = red[1][0]; q[2 * (size_t)dt_cap] = red[2][0]; q[3 * (size_t)dt_cap] = red[3][0];
            q[4 * (size_t)dt_cap] = sqrt(red[4][0]);
        }
    }
}

// the per-workgroup compute_dt partials of the E1<MODE_DT> launches since the last reduction -> clk->r_*
// (min / max are order-independent); one workgroup of DES_BLOCK lanes, ends with a barrier
__device__ void dt_reduce_partials(DevClock *clk, const double *__restrict__ dt_part, int dt_cap, int count)
{
    __shared__ double red[5][DES_BLOCK / 64];
    if (count <= 0) return;                      // (uniform)
    double r[5] = {DBL_MAX, DBL_MAX, DBL_MAX, DBL_MAX, 0.0};
    for (int i = threadIdx.x; i < count; i += DES_BLOCK) {
        for (int k = 0; k < 4; ++k) r[k] = fmin(r[k], dt_part[(size_t)k * dt_cap + i]);
        r[4] = fmax(r[4], dt_part[(size_t)4 * dt_cap + i]);
    }
    for (int k = 0; k < 4; ++k) r[k] = desk::wave_min(r[k]);
    r[4] = desk::wave_max(r[4]);
    if ((threadIdx.x & 63) == 0) for (int k = 0; k < 5; ++k) red[k][threadIdx.x >> 6] = r[k];
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < DES_BLOCK / 64; ++i) {
            for (int k = 0; k < 4; ++k) red[k][0] = fmin(red[k][0], red[k][i]);
            red[4][0] = fmax(red[4][0], red[4][i]);
        }
        clk->r_minl = fmin(clk->r_minl, red[0][0]); clk->r_dt_maxwell = fmin(clk->r_dt_maxwell, red[1][0]);
        clk->r_dt_diffusion = fmin(clk->r_dt_diffusion, red[2][0]); clk->r_global_dt_min = fmin(clk->r_global_dt_min, red[3][0]);
        clk->r_max_vem = fmax(clk->r_max_vem, red[4][0]);
    }
    __syncthreads();
}

// compute_dt tail (geometry.cxx:1597-1646); one workgroup: the reduction of the partials, then one thread
__global__ void __launch_bounds__(DES_BLOCK)
k_dt_finalize(const des_params *p, DevClock *clk, const double *red, const double *__restrict__ dt_part, int dt_cap, int dt_count)
{
    dt_reduce_partials(clk, dt_part, dt_cap, dt_count);
    if (threadIdx.x != 0) return;
    if (red) {          // partials min-reduced over the ranks (k_dt_pack layout)
        clk->r_minl = red[0]; clk->r_dt_maxwell = red[1]; clk->r_dt_diffusion = red[2];
        clk->r_global_dt_min = red[3]; clk->r_max_vem = -red[4]; clk->max_surf_vel = -red[5];
    }
    double dt_maxwell = clk->r_dt_maxwell, dt_diffusion = clk->r_dt_diffusion, minl = clk->r_minl;
    const double dt_hydro_diffusion = DBL_MAX;
    double global_max_vem = clk->r_max_vem;
    double max_vbc_val;
    if (p->characteristic_speed == 0) {
        max_vbc_val = p->max_vbc_val;
        if (p->surface_process_option > 0)
            max_vbc_val = fmax(max_vbc_val, clk->max_surf_vel * 5e-1);
    } else
        max_vbc_val = p->characteristic_speed;
    global_max_vem = fmax(global_max_vem, p->max_vbc_val);
    clk->max_global_vel_mag = global_max_vem;
    clk->global_dt_min = clk->r_global_dt_min;
    double dt_advection = 0.5 * minl / max_vbc_val;
    double dt_elastic = p->is_quasi_static
        ? 0.5 * minl / (max_vbc_val * p->inertial_scaling)
        : 0.5 * minl / sqrt(p->bulk_modulus[p->mattype_ref] / p->rho0[p->mattype_ref]);
    double dt = fmin(fmin(fmin(dt_elastic, dt_maxwell), fmin(dt_advection, dt_diffusion)), dt_hydro_diffusion)
                * p->dt_fraction;
    if (p->fixed_dt != 0) dt = p->fixed_dt;
    if (!(dt > 0)) clk->status = DES_ERR_RUNTIME_NAN;
    clk->dt_prev = clk->dt;
    clk->dt = dt;
    clk->r_minl = DBL_MAX; clk->r_dt_maxwell = DBL_MAX; clk->r_dt_diffusion = DBL_MAX;
    clk->r_global_dt_min = DBL_MAX; clk->r_max_vem = 0.0;
}
