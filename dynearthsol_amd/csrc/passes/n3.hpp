// passes/n3.hpp -- Pass N3 (nodes).
// Part of the single translation unit des_dev.hip (included inside namespace des_hip, after
// DevClock / struct des_dev); not a stand-alone header.

// ---- N3 --------------------------------------------------------------------------
// apply_vbcs for one node (bc.cxx:400-651, THREED)
// hold: Param::control.PT_jump -- inside the pseudo-transient loop the x / y / z boundaries are held
// at rest (bc.cxx:330-343: bc_vx0 .. bc_vz1 = 0; the lateral _l values are not touched)
__device__ __forceinline__ void apply_vbcs_node(const des_params *p, unsigned flag, double time,
                                                const double *bnormals, const double *edge_vec,
                                                const int *edge_slot, double v[3], bool hold = false)
{
    for (int lf = 0; lf < 4; ++lf) {
        if (!(flag & (1u << lf))) continue;
        const int ni = (lf < 2) ? 0 : 1, li = (lf < 2) ? 1 : 0;
        const double val = hold ? 0.0 : p->vbc_values[lf], val_l = p->vbc_val_l[lf];
        switch (p->vbc_types[lf]) {
        case 0: break;
        case 1: v[ni] = val; break;
        case 2: v[li] = 0; v[2] = 0; break;
        case 3: v[ni] = val; v[li] = 0; v[2] = 0; break;
        case 4: v[li] = val; v[2] = 0; break;
        case 5: v[ni] = 0; v[li] = val; v[2] = 0; break;
        case 6: v[ni] = val; v[li] = val_l; break;
        case 7: v[ni] = val; v[li] = 0; break;
        }
    }
    if (flag & 0x3c0u) {
        for (int ib = 6; ib <= 9; ib++) {
            if (!(flag & (1u << ib))) continue;
            const double n[3] = {bnormals[ib], bnormals[DES_NBDRY + ib], bnormals[2*DES_NBDRY + ib]};
            const int type = p->vbc_types[ib];
            double fac = 0;
            if (type == 1 || type == 11) {
                const int nd = (type == 1) ? 3 : 2;
                double target = p->vbc_values[ib];
                if (type == 11) {
                    fac = 1 / sqrt(1 - n[2]*n[2]);
                    target = p->vbc_values[ib] * fac;
                }
                if (flag == (1u << ib)) {
                    double vn = 0;
                    for (int d = 0; d < nd; d++) vn += v[d] * n[d];
                    for (int d = 0; d < nd; d++) v[d] += (target - vn) * n[d];
                } else {
                    for (int ic = 0; ic < ib; ic++) {
                        if (!(flag & (1u << ic))) continue;
                        if (p->vbc_types[ic] == 0) {
                            double vn = 0;
                            for (int d = 0; d < nd; d++) vn += v[d] * n[d];
                            for (int d = 0; d < nd; d++) v[d] += (target - vn) * n[d];
                        } else if (p->vbc_types[ic] == 1) {
                            const int slot = edge_slot[ic*DES_NBDRY + ib];
                            if (slot < 0) continue;
                            const double *edge = &edge_vec[slot*3];
                            double ve = 0;
                            for (int d = 0; d < 3; d++) ve += v[d] * edge[d];
                            for (int d = 0; d < 3; d++) v[d] = ve * edge[d];
                        }
                    }
                }
            } else if (type == 3) {
                for (int d = 0; d < 3; d++) v[d] = p->vbc_values[ib] * n[d];
            } else if (type == 13) {
                fac = 1 / sqrt(1 - n[2]*n[2]);
                for (int d = 0; d < 2; d++) v[d] = p->vbc_values[ib] * fac * n[d];
                v[2] = 0;
            }
        }
    }
    int bc_z0 = p->vbc_types[4], bc_z1 = p->vbc_types[5];
    if (time > p->vbc_val_z1_loading_period) bc_z1 = 0;
    if (bc_z0 == 0 && bc_z1 == 0) return;
    const double bc_vz0 = hold ? 0.0 : p->vbc_values[4], bc_vz1 = hold ? 0.0 : p->vbc_values[5];
    if (flag & (1u << 4)) {
        switch (bc_z0) {
        case 1: v[2] = bc_vz0; break;
        case 2: v[0] = 0; v[1] = 0; break;
        case 3: v[0] = 0; v[1] = 0; v[2] = bc_vz0; break;
        }
    }
    if (flag & (1u << 5)) {
        switch (bc_z1) {
        case 1: v[2] = bc_vz1; break;
        case 2: v[0] = 0; v[1] = 0; break;
        case 3: v[0] = 0.0; v[1] = 0; v[2] = bc_vz1; break;
        case 4: v[0] = bc_vz1; v[1] = 0; v[2] = 0; break;
        }
    }
}

// xt != NULL: followed by update_coordinate (fields.cxx:761-784) -- the two nodal operations that
// N3 leaves out when the pseudo-transient loop sits between update_velocity and them
__global__ void __launch_bounds__(DES_BLOCK)
k_apply_vbcs(const des_params *__restrict__ p, const DevClock *__restrict__ clk, int nn,
             const unsigned *__restrict__ bcflag, const double *__restrict__ bnormals,
             const double *__restrict__ edge_vec, const int *__restrict__ edge_slot, d4 *__restrict__ vm,
             d4 *__restrict__ xt)
{
    const int n = blockIdx.x * DES_BLOCK + threadIdx.x;
    if (n >= nn) return;
    const unsigned flag = bcflag[n];
    if (!xt && !(flag & 0x3ffu)) return;
    d4 m4 = vm[n];
    double v[3] = {m4.x, m4.y, m4.z};
    if (flag & 0x3ffu) {
        apply_vbcs_node(p, flag, clk->time, bnormals, edge_vec, edge_slot, v, clk->pt != 0);
        m4.x = v[0]; m4.y = v[1]; m4.z = v[2];
        vm[n] = m4;
    }
    if (xt && p->has_moving_mesh) {
        const double dt = clk->dt;
        d4 x4 = xt[n];
        x4.x += v[0] * dt; x4.y += v[1] * dt; x4.z += v[2] * dt;
        xt[n] = x4;
    }
}

// What follows the force sums of a node, shared by N3 and EN3: apply_stress_bcs node loop
// (bc.cxx:783-802, 817-823), apply_stress_bcs_neumann, apply_damping (fields.cxx:483-579),
// update_velocity (fields.cxx:725-742), the node's share of the residual (fields.cxx:700-722),
// apply_vbcs, update_coordinate (fields.cxx:761-784).  Returns that share.
__device__ __forceinline__ double n3_finish_node(const des_params *__restrict__ p, const DevClock *__restrict__ clk,
     int n, int nn, int o0, int nn_own_end, int nn_global, double f[3], const double fr[3],
     const unsigned *__restrict__ bcflag, unsigned bc_mask, const int *__restrict__ bcn_idx,
     const int *__restrict__ bcn_ent, const double *__restrict__ bcf_tmp,
     const double *__restrict__ coord0, const double *__restrict__ ymass,
     const double *__restrict__ bnormals, const double *__restrict__ edge_vec, const int *__restrict__ edge_slot,
     const unsigned flag, d4 x4, d4 m4, d4 *xt_out, bool always_store_xt, d4 *__restrict__ vm,
     double *__restrict__ force, double *__restrict__ fres, bool outs = true)
{
    // outs = false (EN3, a step of a multi-step call that is not its last): force and force_residual are not stored -- the next
    // step's pass forms them anew before anything reads them, and what this step needs of them it has in registers.
    // flag, x4, m4: bcflag[n], the node's {x,y,z,T} and {vx,vy,vz,mass} records, loaded by the caller
    // (EN3 has them in flight long before the force sums are ready).  xt_out: the array itself for
    // N3; EN3 reads the coordinates of other blocks' nodes in the same launch, so it writes the
    // records of its own nodes to the other buffer of a pair (and then always, moving mesh or not)
    double l2 = 0.0;
    const double dt = clk->dt;
    if (flag & bc_mask) {
        // the node's facet terms: first the ones apply_stress_bcs subtracts, then the elastic-foundation term,
        // then (entries with bit 0 set) the ones apply_stress_bcs_neumann adds -- in list order.  Four entries
        // at a time, every entry and every term of a batch requested before the first is used: a boundary
        // node's lane walks its list alone, two dependent look-ups per entry
        const int b0 = bcn_idx[n], b1 = bcn_idx[n+1];
        bool plus = false;
        auto foundation = [&]() {
            if (p->has_elastic_foundation && (flag & (1u << 4)))
                f[2] -= p->elastic_foundation_constant * (x4.z - coord0[(size_t)2*nn + n]);
        };
        for (int b = b0; b < b1; b += 4) {
            int ent[4];
            double t[4][3];
#pragma unroll
            for (int u = 0; u < 4; ++u) ent[u] = (b + u < b1) ? bcn_ent[b + u] : 0;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const double *q = bcf_tmp + (size_t)(ent[u] >> 1) * 3;       // (entry 0 for the padding: a valid row)
                t[u][0] = q[0]; t[u][1] = q[1]; t[u][2] = q[2];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (b + u >= b1) break;
                if ((ent[u] & 1) && clk->no_neumann) continue;        // initial_body_force_adjustment: fields.cxx:690
                if (!plus && (ent[u] & 1)) { plus = true; foundation(); }
                if (plus) { f[0] += t[u][0]; f[1] += t[u][1]; f[2] += t[u][2]; }
                else      { f[0] -= t[u][0]; f[1] -= t[u][1]; f[2] -= t[u][2]; }
            }
        }
        if (!plus) foundation();
    }
    double v[3] = {m4.x, m4.y, m4.z};
    const double small_vel = 1e-13;
    const double dfac = p->damping_factor;
    switch (p->damping_option) {
    case 1:
        for (int j = 0; j < 3; j++)
            if (fabs(v[j]) > small_vel) f[j] -= dfac * copysign(f[j], v[j]);
        break;
    case 2:
        for (int j = 0; j < 3; j++) f[j] -= dfac * f[j];
        break;
    case 3:
        for (int j = 0; j < 3; j++) {
            if ((f[j] < 0) == (v[j] < 0)) f[j] -= dfac * f[j];   // fields.cxx:538 (comma operator)
            else                          f[j] += (1 - dfac) * f[j];
        }
        break;
    case 4: {
        double critical_coeff = 2.0 * sqrt(m4.w * ymass[n]);
        for (int j = 0; j < 3; j++)
            if (fabs(v[j]) > small_vel) {
                double f_C = dfac * copysign(f[j], v[j]);
                double f_V = critical_coeff * v[j];
                double f_damping = (fabs(f_C) < fabs(f_V)) ? f_V : f_C;
                f[j] -= f_damping;
            }
        break;
    }
    default: break;
    }
    for (int j = 0; j < 3; j++) {
        if (outs) { force[(size_t)j*nn + n] = f[j]; fres[(size_t)j*nn + n] = fr[j]; }
        v[j] += dt * f[j] / m4.w;
    }
    if (n >= o0 && n < nn_own_end) {
        const double num = (double)nn_global * 3;
        l2 = fr[0]*fr[0] / num;
        l2 += fr[1]*fr[1] / num;
        l2 += fr[2]*fr[2] / num;
    }
    if (p->has_PT) {
        // the pseudo-transient loop comes between update_velocity / the residual and apply_vbcs /
        // update_coordinate (dynearthsol.cxx:798-872): those two follow in k_apply_vbcs
        m4.x = v[0]; m4.y = v[1]; m4.z = v[2];
        vm[n] = m4;
        if (always_store_xt) xt_out[n] = x4;           // (EN3: the record goes to the other buffer of the pair, moved or not)
        return l2;
    }
    if (clk->iso) {
        // isostasy_adjustment (dynearthsol.cxx:521-535): no velocity bcs, vertical motion only,
        // a bottom without Winkler foundation is held
        v[0] = 0; v[1] = 0;
        if (!p->has_winkler_foundation && (flag & (1u << 4))) v[2] = 0;       // BOUNDZ0
    } else if (flag & 0x3ffu)
        apply_vbcs_node(p, flag, clk->time, bnormals, edge_vec, edge_slot, v);
    m4.x = v[0]; m4.y = v[1]; m4.z = v[2];
    vm[n] = m4;
    if (p->has_moving_mesh || clk->iso) {
        x4.x += v[0] * dt; x4.y += v[1] * dt; x4.z += v[2] * dt;
        xt_out[n] = x4;
    } else if (always_store_xt)
        xt_out[n] = x4;
    return l2;
}

// update_force node loop (fields.cxx:659-676), apply_stress_bcs node loop (bc.cxx:783-802,
// 817-823), apply_stress_bcs_neumann, apply_damping (fields.cxx:483-579), update_velocity
// (fields.cxx:725-742), residual partial sums (fields.cxx:700-722), apply_vbcs,
// update_coordinate (fields.cxx:761-784)
__global__ void __launch_bounds__(DES_BLOCK)
N3_force_velocity_coord(const des_params *__restrict__ p, const DevClock *__restrict__ clk, int o0, int nn_own_end,
     int nn, int nn_global, int nblocks, int npb,
     const int *__restrict__ sup_idx, const int *__restrict__ sup_pack, const unsigned *__restrict__ bcflag,
     const double *__restrict__ ftmp, unsigned bc_mask, const int *__restrict__ bcn_idx,
     const int *__restrict__ bcn_ent, const double *__restrict__ bcf_tmp,
     const double *__restrict__ coord0, const double *__restrict__ ymass,
     const double *__restrict__ bnormals, const double *__restrict__ edge_vec, const int *__restrict__ edge_slot,
     d4 *__restrict__ xt, d4 *__restrict__ vm, double *__restrict__ force, double *__restrict__ fres,
     double *__restrict__ res_part)
{
    __shared__ double lds[3][DES_TILE_LDS(DES_TILE_N3)];
    __shared__ double red[DES_BLOCK / 64];
    // every local node is updated (nn = local node count = stride of the SoA planes); the owned
    // nodes [o0, nn_own_end) alone enter the residual
    const int lb = desk::logical_block(nblocks);
    const int n0 = lb * npb;
    const int n = (threadIdx.x < npb) ? n0 + threadIdx.x : nn;
    if (n0 >= nn) return;
    const int nlast = min(n0 + npb, nn);
    const int kb = sup_idx[n0], ke = sup_idx[nlast];
    int r0 = ke, r1 = ke;
    if (n < nn) { r0 = sup_idx[n]; r1 = sup_idx[n+1]; }
    double f[3] = {0, 0, 0}, fr[3] = {0, 0, 0};
#if DES_PIPE
    constexpr int PER = DES_TILE_N3 / DES_BLOCK;
    static_assert(DES_TILE_N3 % DES_BLOCK == 0, "tile must be a multiple of the block");
    double q0[PER], q1[PER], q2[PER];
    auto fetch = [&](int t0, int tn) {
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int j = threadIdx.x + u * DES_BLOCK;
            if (j < tn) {
                const int pk = sup_pack[t0 + j];
                const double *tr = ftmp + (size_t)(pk >> 2) * 12 + (pk & 3) * 3;
                q0[u] = tr[0]; q1[u] = tr[1]; q2[u] = tr[2];
            }
        }
    };
    if (kb < ke) fetch(kb, min(DES_TILE_N3, ke - kb));
#endif
    for (int t0 = kb; t0 < ke; t0 += DES_TILE_N3) {
        const int tn = min(DES_TILE_N3, ke - t0);
#if DES_PIPE
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int j = threadIdx.x + u * DES_BLOCK;
            if (j < tn) {
                const int sl = lds_slot(j);
                lds[0][sl] = q0[u]; lds[1][sl] = q1[u]; lds[2][sl] = q2[u];
            }
        }
        __syncthreads();
        if (t0 + DES_TILE_N3 < ke) fetch(t0 + DES_TILE_N3, min(DES_TILE_N3, ke - t0 - DES_TILE_N3));
#else
        for (int j = threadIdx.x; j < tn; j += DES_BLOCK) {
            const int pk = sup_pack[t0 + j];
            const double *tr = ftmp + (size_t)(pk >> 2) * 12 + (pk & 3) * 3;
            const int sl = lds_slot(j);
            lds[0][sl] = tr[0]; lds[1][sl] = tr[1]; lds[2][sl] = tr[2];
        }
        __syncthreads();
#endif
        const int a = max(r0, t0) - t0, b = min(r1, t0 + tn) - t0;
        for (int j = a; j < b; ++j) {
            const int sl = lds_slot(j);
            const double t0v = lds[0][sl], t1v = lds[1][sl], t2v = lds[2][sl];
            f[0] -= t0v; f[1] -= t1v; f[2] -= t2v;
            fr[0] = t0v; fr[1] = t1v; fr[2] = t2v;          // assignment: fields.cxx:673
        }
        __syncthreads();
    }
    double l2 = 0.0;
    if (n < nn)
        l2 = n3_finish_node(p, clk, n, nn, o0, nn_own_end, nn_global, f, fr, bcflag, bc_mask, bcn_idx, bcn_ent, bcf_tmp,
                            coord0, ymass, bnormals, edge_vec, edge_slot, bcflag[n], xt[n], vm[n], xt, false, vm, force, fres);
    // per-block partial of the residual; the partials are added in block order afterwards
    l2 = desk::wave_sum(l2);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = l2;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = red[0];
        for (int i = 1; i < DES_BLOCK / 64; ++i) t += red[i];
        res_part[lb] = t;
    }
}
