// passes/en1.hpp -- Pass EN1 (node patches): N1 without the element temporaries.
// Part of the single translation unit des_dev.hip (included inside namespace des_hip, after
// DevClock / struct des_dev); not a stand-alone header.

// ---- EN1 -------------------------------------------------------------------------
// compute_mass gather (geometry.cxx:1846-1864), update_temperature (fields.cxx:211-262),
// compute_dvoldt (geometry.cxx:218-238), and the clock (dynearthsol.cxx:773-774) -- what N1 does,
// for one block of `npb` nodes per workgroup over its patch (engine/patch.hpp), in the manner of
// EN3: the element terms N1 gathers (volume, inertial / thermal mass, conduction term of each of the
// four nodes, tr(strain rate) x volume) are functions of the nodal records {x,y,z,T}, {vx,vy,vz}
// and of per-element constants only, so the workgroup stages the records of its patch's nodes in
// LDS and RECOMPUTES those terms (the very expressions of E1: e1_mass_terms, e1_thermal_terms,
// e1_strain_rate_diag) instead of E1 writing 64 B per element and N1 gathering them back per
// incidence.  Sums per node run over the block's slice of the CSR list in order -> same bits.
//
// Valid when nothing but the nodal records has changed since the end-of-step pass (engine/launch.hpp,
// en1_ok): inside a multi-step call.  The new temperatures go to the other buffer of the
// {x,y,z,T} pair (another block may still be reading this block's nodes); the host swaps.

// The surface step of the step BEFORE, when its S2 / S3 launches were left out (engine/launch.hpp: s2_defer_ok):
// simple_diffusion of every surface node of the patch (passes/surface.hpp: s2_node_dh, from the coordinates
// update_coordinate left -- the buffer this pass only reads), the new height applied to the staged record, so that
// every element term below and the record this block stores for its own nodes hold the committed surface, as
// after k_s2 + k_s3_finalize; dh / dhacc / the nodal dh are stored for the block's own nodes.  Surface nodes of
// the patch that belong to other blocks are recomputed here (their owners store them): a few tens of nodes per
// block, against two launches of ~5 us each.  topidx == nullptr: nothing pending.
// tfan[node] = {position in top_nodes | facets in its fan << 27, first entry of its fan in ssup_nodes}, {-1, 0} for
// the nodes below the surface: one look-up instead of topidx -> ssup_idx.
// pb_top[block] != 0: the block's patch holds a surface node at all (the others never touch tfan); the surface nodes
// among a patch's foreign nodes carry bit 31 in pn_id (engine/patch.hpp), so only they cost a look-up.
struct SurfPending { const int2 *tfan; const int *ssup_nodes; double *dh, *dhacc, *dh_n; };
#ifndef DES_EN1_S2_BATCH
#define DES_EN1_S2_BATCH 2
#endif

#ifndef DES_EN1_NB
#define DES_EN1_NB 8              // incidences per batch of LDS requests in the node phase
#endif
// DES_EN1_SPLIT = 1: the node phase on three wavefronts, one group of sums each, instead of one lane per node walking all
// five (round 4, the round-3 review's proposal).  Same bits (every sum keeps its CSR order; the GPU suite passes either way).
// Measured with phase stamps (profiles/r04_c_phase_*.txt, tools/patch_phase_timing.py): the node phase goes 2.18 -> 1.85 us
// (it is bound by LDS latency under the other workgroups' element phases, not by the length of one lane's walk), the
// element phase 5.98 -> 6.34 us (two more wavefronts compete for LDS and issue slots), a workgroup's life 10.79 -> 10.93 us,
// the kernel 51.7 -> 51.3 us: nothing.  The pass is issue-bound (1100 fp64-heavy instructions per wavefront at 65-70 % issue
// utilisation), not bound by a workgroup's critical path.  Off by default; kept as the measured alternative.
#ifndef DES_EN1_SPLIT
#define DES_EN1_SPLIT 0
#endif
#ifndef DES_PATCH_PE
#define DES_PATCH_PE 1280         // elements of a patch (LDS records)
#endif
#ifndef DES_EXP_EN1
#define DES_EXP_EN1 0             // timing experiments (wrong results; profiles/r05_d_patch_phase_removal.txt), bits: 1 no element arithmetic,
                                  // 2 no node sums, 4 no staging gathers, 8 no conduction terms, 16 no LDS reads in the element phase; on top of 1|2|4 (the skeleton):
                                  // 128 no list-entry loads, 256 no LDS stores of the element phase, 512 no node stores, 1024 header loads only
#endif
#ifndef DES_EN1_MINWAVES
#define DES_EN1_MINWAVES 3        // waves per SIMD the register budget of the 256-lane kernel is held to (168 VGPRs; 4: 128)
#endif

// THERM = 1: has_thermal_diffusion known to be on (the common launch: that path only in the kernel), 0: read at run time
// LDS is DYNAMIC, sized by the host to the mesh's largest block (cap_inc incidences, cap_pn patch nodes, cap_pe patch elements:
// engine/launch.hpp, en1_lds_bytes) -- the number of workgroups a CU holds follows from the mesh and the block size, not from a
// handful of compiled shapes (round 5; rounds 2-4: template shapes <1600, 296, 872> / <2048, 512, 1280>).
__host__ __device__ inline size_t en1_lds_bytes(int cap_inc, int cap_pn, int cap_pe, bool constm)
{
    return (size_t)cap_pn * (32 + 24) + (size_t)cap_pe * (constm ? 24 : 32) + (size_t)cap_inc * 10;
}
template <int THREADS, int CONSTM, int THERM = 0>
__global__ void __launch_bounds__(THREADS, THREADS == 512 ? 4 : DES_EN1_MINWAVES)
EN1_mass_temperature_dvoldt(const des_params *__restrict__ p, DevClock *__restrict__ clk, int nn, int ne, int b0, int c0, int b1, int c1, int do_clock, int npb,
     int cap_inc, int cap_pn, int cap_pe,
     const int *__restrict__ pe_ptr, const ulonglong2 *__restrict__ pe_pack, const int *__restrict__ pn_ptr, const int *__restrict__ pn_id,
     const int *__restrict__ sup_idx, const unsigned *__restrict__ bcflag, const MatData md,
     const double *__restrict__ radiogenic, const d4 *__restrict__ xt, d4 *__restrict__ xt_out, d4 *__restrict__ vm,
     double *__restrict__ volume_n, double *__restrict__ tmass, double *__restrict__ ntmp, const SurfPending sp,
     const int *__restrict__ pb_top, const int *__restrict__ bperm)
{
    extern __shared__ __attribute__((aligned(32))) unsigned char des_smem[];
    unsigned char *sm = des_smem;
    d4 *const lxt = (d4 *)sm; sm += (size_t)cap_pn * 32;
    double *const lvx = (double *)sm; sm += (size_t)cap_pn * 8;
    double *const lvy = (double *)sm; sm += (size_t)cap_pn * 8;
    double *const lvz = (double *)sm; sm += (size_t)cap_pn * 8;
    double *const lvol = (double *)sm; sm += (size_t)cap_pe * 8;
    double *const ltm = (double *)sm; sm += (size_t)cap_pe * 8;
    double *const ldv = (double *)sm; sm += (size_t)cap_pe * 8;
    double *const lm = (double *)sm; sm += CONSTM ? 0 : (size_t)cap_pe * 8;
    double *const ltd = (double *)sm; sm += (size_t)cap_inc * 8;
    unsigned short *const lidx = (unsigned short *)sm;
    // this launch covers the node blocks [b0, b0 + c0) and [b1, b1 + c1): all of them (0, nb, 0, 0), or -- overlapped
    // multi-GPU schedule -- first the blocks deep inside the slab, later the ones near its cuts (engine/launch.hpp)
    const int L = desk::logical_block(c0 + c1);
    // (bperm: the launch order of a launch over all blocks -- every XCD its share of the surface blocks, engine/patch.hpp)
    int lb = L < c0 ? b0 + L : b1 + (L - c0);
    if (bperm && L < c0 + c1) lb = bperm[lb];
    const int n0 = lb * npb;
    const double dt = clk->dt;
    if (do_clock && blockIdx.x == 0 && threadIdx.x == 0) { // never in the isostasy loop (en1_ok); once per step
        // Output::average_fields' time0 (output.cxx:332) of the step that has just ended, when its end-of-step pass
        // (and with it k_average_fields) was left to the next stress update
        if (p->is_outputting_averaged_fields && clk->steps % p->quality_check_step_interval == 1) clk->avg_time0 = clk->time;
        clk->steps += 1;
        clk->time += dt;
        clk->maxdh = 0.0;
        clk->n_defer = 0;
    }
    if (L >= c0 + c1 || n0 >= nn) return;                  // grid padding
    DES_STAMP0(0, 0);
    const int nown = min(npb, nn - n0);
    const int h0 = pn_ptr[lb], nh = pn_ptr[lb + 1] - h0;
    const int e_begin = pe_ptr[lb], e_end = pe_ptr[lb + 1];
    const bool thermal = THERM ? true : (bool)p->has_thermal_diffusion;
    const int nmat = p->nmat;
#if DES_EXP_EN1 & 1024
    if (nh + e_end == -12345) volume_n[n0] = nown;         // timing experiment only: header loads, then out
    if (nh + e_end != -12345) return;
#endif
    // the node this lane works for.  DES_EN1_SPLIT (blocks of up to 64 nodes): lane l of EVERY wavefront belongs to node
    // n0 + l, and the node's independent sums go to different wavefronts (node phase below); otherwise lane t < nown alone.
#if DES_EN1_SPLIT
    const bool split = npb <= 64;
#else
    const bool split = false;
#endif
    const int nl = split ? (int)(threadIdx.x & 63) : (int)threadIdx.x;         // node of the block
    const int part = split ? (int)(threadIdx.x >> 6) : 0;                      // which sums (split), else all
    const int n = n0 + nl;
    const bool has_node = nl < nown && part < 3;
    // Requests first, uses later (round 5): the node's list bounds and flags, the block's surface flag and the ids of the patch's
    // other nodes all go out in ONE round behind the block's scalar bounds, and nothing waits before the ids are needed.  (Used
    // where they were written -- the bounds subtracted at once, the flag tested in the loop head -- each was a round of its own:
    // six dependent trips to memory from the kernel arguments to the first nodal record, now four.)
    // (a lane without a node reads the block's first node's -- no branch around the requests: merging its results made the
    //  compiler wait for them on the spot)
    const int top_raw = pb_top[lb];                        // (an argument of its own, __restrict__: a scalar load beside the bounds)
    const int nc = has_node ? n : n0;
    const int s_k = sup_idx[n0], s_a = sup_idx[nc], s_b = sup_idx[nc + 1];
    const unsigned flag_raw = bcflag[nc];
    // the patch's nodes into LDS: own range first (local id = n - n0), then the listed others
    // (every record of the patch is requested before anything waits: the surface nodes, which need more, come after)
    constexpr int ROUNDS = DES_PATCH_PN / THREADS;          // (cap_pn <= DES_PATCH_PN = 512)
    int2 tf[ROUNDS];
    int ids[ROUNDS];
    int raws[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int j = threadIdx.x + r * THREADS;
        // (no branch around the request; pn_id ends in a spare entry for a block without foreign nodes)
        const int raw = pn_id[h0 + min(max(j - nown, 0), max(nh - 1, 0))];
        raws[r] = j < nown ? n0 + j : raw;
    }
    // (pinned here -- an empty asm that "changes" them --: the compiler otherwise sinks the request into the conditional block
    //  that uses the ids)
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) asm volatile("" : "+v"(raws[r]));
    const bool blk_top = sp.tfan && top_raw != 0;
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int j = threadIdx.x + r * THREADS;
        tf[r] = make_int2(-1, 0);
        ids[r] = 0;
        if (j < nown + nh) {
#if DES_EXP_EN1 & 4
            // timing experiment only (wrong results): the staging without its gathers
            const int raw = n0 + (j & 63);
            const int id = raw & 0x7fffffff;
            ids[r] = id;
            d4 xr; xr.x = 1.0 * j; xr.y = 2.0 * (j & 7); xr.z = 3.0 * (j & 3) + 0.5 * j; xr.w = 300.0;
            lxt[j] = xr;
            lvx[j] = 1e-9 * j; lvy[j] = 2e-9; lvz[j] = 3e-9;
#else
            const int raw = raws[r];
            const int id = raw & 0x7fffffff;
            ids[r] = id;
            lxt[j] = xt[id];
            const d4 v = vm[id];
            lvx[j] = v.x; lvy[j] = v.y; lvz[j] = v.z;
#endif
            if (blk_top && (j < nown || raw < 0)) tf[r] = sp.tfan[id];
        }
    }
    if (blk_top) {
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            if (tf[r].x < 0) continue;
            const int j = threadIdx.x + r * THREADS;
            const int id = ids[r];
            const int ti = tf[r].x & 0x7ffffff, nf = tf[r].x >> 27;
            // (dt: the step before ran with it too -- a step with a compute_dt keeps its own S2 / S3 launches)
            const double dhacc_old = j < nown ? sp.dhacc[id] : 0.0;
            const double d = s2_node_dh_range<DES_EN1_S2_BATCH>(id, tf[r].y, tf[r].y + nf, sp.ssup_nodes, xt, p->surface_diffusivity, dt);
            lxt[j].z = lxt[j].z + d;                                // (this lane staged the record itself)
            if (j < nown) { sp.dh[ti] = d; sp.dhacc[id] = dhacc_old + d; sp.dh_n[id] = d; }
        }
    }
    int r0 = has_node ? s_a - s_k : 0, r1 = has_node ? s_b - s_k : 0;
    const unsigned flag = has_node ? flag_raw : 0u;
    __syncthreads();
    DES_STAMP0(0, 1);
    // the patch's elements: E1's element terms, recomputed
    auto do_elem = [&](const int i) {
#if DES_EXP_EN1 & 128
        // timing experiment only (wrong results): the list entry made up from the position (no load)
        const unsigned long long q_ = (unsigned long long)(i - e_begin);
        const PatchElem PE_ = patch_elem_unpack(make_ulonglong2((unsigned long long)((lb * 345 + (int)q_) % ne) | ((q_ % 64) << 31) | (((q_ + 1) % 64) << 40) | (((q_ + 2) % 64) << 49),
                                                                ((q_ + 3) % 64) | ((q_ % 1200) << 9) | (0xfffull << 21) | (0xfffull << 33) | (0xfffull << 45)));
#else
        const PatchElem PE_ = patch_elem_unpack(pe_pack[i]);
#endif
        const int e = PE_.ew & 0x3fffffff;
        const ushort4 ln = make_ushort4(PE_.ln[0], PE_.ln[1], PE_.ln[2], PE_.ln[3]);
        const short4 sl = make_short4(PE_.sl[0], PE_.sl[1], PE_.sl[2], PE_.sl[3]);
        const int q = i - e_begin;                          // position in the patch
#if DES_EXP_EN1 & 1
        // timing experiment only (wrong results): the element phase without its LDS reads and arithmetic (list entry, marker
        // word and the LDS stores stay)
        {
            const double fake = (double)(e & 1023) + md.mono[e];
#if DES_EXP_EN1 & 256
            if (fake == -1.2345) lvol[q] = fake;             // (keeps the marker-word load; no LDS stores)
            return;
#endif
            lvol[q] = fake + 1.0; ltm[q] = fake; ldv[q] = fake;
            if (!CONSTM) lm[q] = fake;
            if (sl.x >= 0) { ltd[sl.x] = fake; lidx[sl.x] = (unsigned short)q; }
            if (sl.y >= 0) { ltd[sl.y] = fake; lidx[sl.y] = (unsigned short)q; }
            if (sl.z >= 0) { ltd[sl.z] = fake; lidx[sl.z] = (unsigned short)q; }
            if (sl.w >= 0) { ltd[sl.w] = fake; lidx[sl.w] = (unsigned short)q; }
            return;
        }
#endif
        d4 c[4], v[4];
#if DES_EXP_EN1 & 16
        // timing experiment only (wrong results): the arithmetic on values made up in registers instead of read from LDS
        for (int u = 0; u < 4; ++u) {
            const int l = u == 0 ? ln.x : (u == 1 ? ln.y : (u == 2 ? ln.z : ln.w));
            c[u].x = 1.0 * l + 0.25 * u * u; c[u].y = 2.0 * (l & 7) + 1.5 * u; c[u].z = 0.5 * l + (u == 3 ? 7.0 : 0.0); c[u].w = 300.0 + l;
            v[u].x = 1e-9 * l; v[u].y = 2e-9 * u; v[u].z = 3e-9 * l; v[u].w = 0;
        }
#else
        c[0] = lxt[ln.x]; c[1] = lxt[ln.y]; c[2] = lxt[ln.z]; c[3] = lxt[ln.w];
        v[0].x = lvx[ln.x]; v[0].y = lvy[ln.x]; v[0].z = lvz[ln.x]; v[0].w = 0;
        v[1].x = lvx[ln.y]; v[1].y = lvy[ln.y]; v[1].z = lvz[ln.y]; v[1].w = 0;
        v[2].x = lvx[ln.z]; v[2].y = lvy[ln.z]; v[2].z = lvz[ln.z]; v[2].w = 0;
        v[3].x = lvx[ln.w]; v[3].y = lvy[ln.w]; v[3].z = lvz[ln.w]; v[3].w = 0;
#endif
        const desk::Mix mx = mix_of(md, nmat, e);
        const ElemProps pr = load_props(p, md, mx, ne, e);
        double T = 0;
        T += c[0].w; T += c[1].w; T += c[2].w; T += c[3].w;
        T /= 4;
        const double rho = desk::mat_rho(p, mx, T);
        const double vol = desk::tet_volume(c);             // = volume[e]: E1 formed it from the same records
        double m, tm;
        e1_mass_terms(p, pr, rho, vol, m, tm);
        double sx[4], sy[4], sz[4];
        desk::shape_fn(c, vol, sx, sy, sz);
        double tr[4] = {0, 0, 0, 0};
        // (radiogenic == nullptr: every heat source is +0.0, engine/launch.hpp -- the same arithmetic without the fetch)
#if DES_EXP_EN1 & 8
        tr[0] = sx[0]; tr[1] = sy[1]; tr[2] = sz[2]; tr[3] = vol;     // timing experiment only (wrong results): no conduction terms
#else
        if (thermal) e1_thermal_terms(c, sx, sy, sz, pr.k, vol, radiogenic ? radiogenic[e] : 0.0, rho, tr);
#endif
        double s0, s1, s2;
        e1_strain_rate_diag(v, sx, sy, sz, s0, s1, s2);
        double dj = s0 + s1 + s2;
        lvol[q] = vol; ltm[q] = tm; ldv[q] = dj * vol;
        if (!CONSTM) lm[q] = m;
        if (sl.x >= 0) { ltd[sl.x] = tr[0]; lidx[sl.x] = (unsigned short)q; }
        if (sl.y >= 0) { ltd[sl.y] = tr[1]; lidx[sl.y] = (unsigned short)q; }
        if (sl.z >= 0) { ltd[sl.z] = tr[2]; lidx[sl.z] = (unsigned short)q; }
        if (sl.w >= 0) { ltd[sl.w] = tr[3]; lidx[sl.w] = (unsigned short)q; }
    };
    for (int i = e_begin + threadIdx.x; i < e_end; i += THREADS) do_elem(i);
    DES_STAMP0(0, 2);                                       // (this wavefront's own elements done)
    __syncthreads();
    DES_STAMP0(0, 3);
    if (!has_node) return;
    const double pseudo_speed = p->max_vbc_val * p->inertial_scaling;
    const double rho_m = p->bulk_modulus[0] / (pseudo_speed * pseudo_speed);
    constexpr int NB = DES_EN1_NB;
#if DES_EN1_SPLIT
    if (split) {
        // The node's five sums are independent of each other and each keeps its own CSR order -- the reference's
        // association, the same bits -- whichever lane forms it: wavefront 0 takes {volume_n, dvoldt}, wavefront 1 the
        // inertial mass, wavefront 2 {thermal mass, conduction}, each with one index + at most two terms per incidence
        // instead of one lane walking index + five terms (64 of 256 lanes busy); every wavefront stores what its sums
        // complete, so nothing has to change hands.
        if (part == 0) {
            double vn = 0, acc = 0;
            int k = r0;
            for (; k + NB <= r1; k += NB) {
                int q[NB]; double a[NB], b[NB];
#pragma unroll
                for (int u = 0; u < NB; ++u) q[u] = lidx[k + u];
#pragma unroll
                for (int u = 0; u < NB; ++u) { a[u] = lvol[q[u]]; b[u] = ldv[q[u]]; }
#pragma unroll
                for (int u = 0; u < NB; ++u) { vn += a[u]; acc += b[u]; }
            }
            for (; k < r1; ++k) { const int q = lidx[k]; vn += lvol[q]; acc += ldv[q]; }
            volume_n[n] = vn;
            ntmp[n] = acc / vn;
            DES_STAMP(0, 4);
        } else if (part == 1) {
            double ms = 0;
            int k = r0;
            for (; k + NB <= r1; k += NB) {
                int q[NB]; double a[NB];
#pragma unroll
                for (int u = 0; u < NB; ++u) q[u] = lidx[k + u];
#pragma unroll
                for (int u = 0; u < NB; ++u) a[u] = CONSTM ? lvol[q[u]] : lm[q[u]];
#pragma unroll
                for (int u = 0; u < NB; ++u) { if (CONSTM) ms += rho_m * a[u] / 4; else ms += a[u]; }
            }
            for (; k < r1; ++k) { const int q = lidx[k]; if (CONSTM) ms += rho_m * lvol[q] / 4; else ms += lm[q]; }
            d4 m4;                                               // the velocity is the staged one (own node: local id nl)
            m4.x = lvx[nl]; m4.y = lvy[nl]; m4.z = lvz[nl]; m4.w = ms;
            vm[n] = m4;
            DES_STAMP(0, 5);
        } else {
            double tms = 0, tdot = 0;
            if (thermal) {
                int k = r0;
                for (; k + NB <= r1; k += NB) {
                    int q[NB]; double a[NB], b[NB];
#pragma unroll
                    for (int u = 0; u < NB; ++u) q[u] = lidx[k + u];
#pragma unroll
                    for (int u = 0; u < NB; ++u) { a[u] = ltm[q[u]]; b[u] = ltd[k + u]; }
#pragma unroll
                    for (int u = 0; u < NB; ++u) { tms += a[u]; tdot += b[u]; }
                }
                for (; k < r1; ++k) { tms += ltm[lidx[k]]; tdot += ltd[k]; }
            }
            tmass[n] = tms;
            d4 x4 = lxt[nl];
            if (thermal) {
                if (flag & (1u << 5))
                    x4.w = p->surface_temperature;
                else
                    x4.w -= dt * tdot / tms;
            }
            xt_out[n] = x4;
            DES_STAMP(0, 6);
        }
        return;
    }
#endif
#if DES_EXP_EN1 & 2
    // timing experiment only (wrong results): the node phase without its sums (one term each; the stores stay)
    if (r1 > r0) r1 = r0 + 1;
#endif
    // the node: sums in CSR order (compute_mass / update_temperature / compute_dvoldt node loops)
    double vn = 0, ms = 0, tms = 0, acc = 0, tdot = 0;
    // DES_EN1_NB incidences at a time: their indices, then every term they point at, are requested from LDS
    // before the first is used (one lane walks its list alone; a dependent look-up per incidence made this
    // phase a third of the workgroup's life); the sums themselves stay in list order
    int k = r0;
    for (; k + NB <= r1; k += NB) {
        int q[NB];
        double vol[NB], mq[NB], tq[NB], td[NB], dv[NB];
#pragma unroll
        for (int u = 0; u < NB; ++u) q[u] = lidx[k + u];
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            vol[u] = lvol[q[u]];
            mq[u] = CONSTM ? 0.0 : lm[q[u]];
            tq[u] = thermal ? ltm[q[u]] : 0.0;
            td[u] = thermal ? ltd[k + u] : 0.0;
            dv[u] = ldv[q[u]];
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            vn += vol[u];
            if (CONSTM) ms += rho_m * vol[u] / 4;
            else        ms += mq[u];
            if (thermal) { tms += tq[u]; tdot += td[u]; }
            acc += dv[u];
        }
    }
    for (; k < r1; ++k) {
        const int q = lidx[k];
        const double vol = lvol[q];
        vn += vol;
        if (CONSTM) ms += rho_m * vol / 4;
        else        ms += lm[q];
        if (thermal) { tms += ltm[q]; tdot += ltd[k]; }
        acc += ldv[q];
    }
#if DES_EXP_EN1 & 512
    if (vn != -1.2345) return;                               // timing experiment only: no stores of the node phase
#endif
    volume_n[n] = vn;
    tmass[n] = tms;
    d4 m4;                                                   // the velocity is the staged one (own node: local id = lane)
    m4.x = lvx[threadIdx.x]; m4.y = lvy[threadIdx.x]; m4.z = lvz[threadIdx.x]; m4.w = ms;
    vm[n] = m4;
    d4 x4 = lxt[threadIdx.x];
    if (thermal) {
        if (flag & (1u << 5))
            x4.w = p->surface_temperature;
        else
            x4.w -= dt * tdot / tms;
    }
    xt_out[n] = x4;
    ntmp[n] = acc / vn;
    DES_STAMP0(0, 4);
}
