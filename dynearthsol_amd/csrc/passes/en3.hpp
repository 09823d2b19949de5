// passes/en3.hpp -- Pass EN3 (node patches): E3 + N3 in one launch, without the force temporaries.
// Part of the single translation unit des_dev.hip (included inside namespace des_hip, after
// DevClock / struct des_dev); not a stand-alone header.

// ---- EN3 -------------------------------------------------------------------------
// NMD_stress apply (geometry.cxx:316-331), update_force (fields.cxx:609-698) and everything N3
// does after the force sums, for one block of `npb` consecutive nodes per workgroup.
//
// The classic pair E3 -> N3 writes the 12 force terms of every element to HBM (96 B) and gathers
// them back per incidence (4.4M scattered 24-byte reads at 1M tets: N3 is bound by the rate of
// those gathers, not by bandwidth).  Here a workgroup RECOMPUTES the force terms of its node
// block's PATCH -- every element that touches one of its nodes, listed once per mesh by
// build_patches() -- straight into LDS: the nodal records {x,y,z,T} and the NMD nodal pressure of
// the patch's nodes are staged in LDS first (one gather per node instead of one per incidence),
// each lane then takes patch elements (stress, volume, dpressure: the only per-element global
// reads), and stores the terms of the element's nodes that belong to this block into the LDS
// slot of that incidence = its position in the block's slice of the CSR support list.  After a
// barrier each node's lane sums its slots in CSR order, i.e. in the reference's ascending element
// order (fields.cxx:667-675): same association, same bits as the gather.  An element is
// recomputed by every block it touches (1.96 blocks on average at 64 nodes per block on the 1M-tet
// TetGen mesh): ~150 flops each, against 192 B of HBM traffic saved per element.
//
// The NMD correction of the stress diagonal is not stored into `stress` here (another block may
// still be reading that element's old stress): the block that OWNS the element (the one holding its
// lowest node) stores the increment ddp[e]; the end-of-step pass E1 -- the next reader of the
// stress -- adds it with the same operation E3 would have used (and the ghost-exchange pack does
// the same for what it sends).
// Launch shapes: THREADS lanes per workgroup, LDS for INC incidences and PN patch nodes; the host
// picks the smallest that holds the mesh's largest block (engine/launch.hpp, launch_en3).
#define DES_PATCH_INC 2048        // caps of the largest shape = what build_patches() accepts
#define DES_PATCH_PN 512
// DES_EN3_SPLIT = 1: the three force sums of a node on three wavefronts (round 4, the round-3 review's proposal; same bits).
// Measured (profiles/r04_c_phase_*.txt): the force sums were 1.1 us of a workgroup's 11.3-us life to begin with, the second
// barrier and the extra LDS round trip give the gain back (life 11.34 -> 11.71 us, kernel 54.1 -> 54.6 us).  Off by default.
#ifndef DES_EN3_SPLIT
#define DES_EN3_SPLIT 0
#endif

// (512 lanes: three workgroups per CU are 6 waves per SIMD, i.e. at most 80 VGPRs)
// KNOWN = 1: the launch is the common one -- NMD_stress on, gravity on -- and the kernel holds that path only (the two flags
// known at compile time: -2 us of 56 at 1M tets); KNOWN = 0: both read at run time.
// LDS is DYNAMIC, sized by the host to the mesh's largest block (cap_inc incidences, cap_pn patch nodes; engine/launch.hpp).
#ifndef DES_EXP_EN3
#define DES_EXP_EN3 0             // timing experiments (wrong results; profiles/r05_d_patch_phase_removal.txt), bits: 1 no element arithmetic, 2 no force sums, 4 no staging gathers, 8 no element loads, 16 the element loads as one 64-byte record
#endif
__host__ __device__ inline size_t en3_lds_bytes(int cap_inc, int cap_pn) { return (size_t)cap_pn * (32 + 8) + (size_t)cap_inc * 24; }
#ifndef DES_EN3_MINWAVES
#define DES_EN3_MINWAVES 3        // 256-lane form: waves per SIMD its register budget is held to
#endif
template <int THREADS, int KNOWN = 0>
__global__ void __launch_bounds__(THREADS, THREADS == 512 ? 6 : DES_EN3_MINWAVES)
EN3_force_nodes(const des_params *__restrict__ p, const DevClock *__restrict__ clk, int nmd_arg, int o0, int nn_own_end,
     int nn, int nn_global, int ne, int nblocks, int npb, int cap_inc, int cap_pn,
     const int *__restrict__ pe_ptr, const ulonglong2 *__restrict__ pe_pack, const int *__restrict__ pn_ptr, const int *__restrict__ pn_id,
     const int *__restrict__ sup_idx, const unsigned *__restrict__ bcflag,
     const double *__restrict__ ntmp, const MatData md, const double *__restrict__ volume,
     const double *__restrict__ dpressure, const double *__restrict__ stress, double *__restrict__ ddp_out,
     unsigned bc_mask, const int *__restrict__ bcn_idx, const int *__restrict__ bcn_ent, const double *__restrict__ bcf_tmp,
     const double *__restrict__ coord0, const double *__restrict__ ymass,
     const double *__restrict__ bnormals, const double *__restrict__ edge_vec, const int *__restrict__ edge_slot,
     const d4 *__restrict__ xt, d4 *__restrict__ xt_out, d4 *__restrict__ vm, double *__restrict__ force,
     double *__restrict__ fres, double *__restrict__ res_part, int outs)
{
    // xt: the nodal records as the last pass left them (read for the whole patch); xt_out: the other
    // buffer of the pair, where this block stores the records of its own nodes (the host swaps the
    // two after the launch) -- a block must not move a node another block may still be reading
    extern __shared__ __attribute__((aligned(32))) unsigned char des_smem[];
    unsigned char *sm = des_smem;
    d4 *const lxt = (d4 *)sm; sm += (size_t)cap_pn * 32;
    double *const lnt = (double *)sm; sm += (size_t)cap_pn * 8;
    double *const lf[3] = {(double *)sm, (double *)sm + cap_inc, (double *)sm + 2 * (size_t)cap_inc};
    __shared__ double red[THREADS / 64];
    // (EN1's launch order -- every XCD its share of the surface blocks, engine/launch.hpp -- tried here too: 52.1 -> 53.1 us)
    const int lb = desk::logical_block(nblocks);
    const int n0 = lb * npb;
    if (n0 >= nn) return;                                 // grid padding
    DES_STAMP0(1, 0);
    const int nown = min(npb, nn - n0);
    const int h0 = pn_ptr[lb], nh = pn_ptr[lb + 1] - h0;
    const int e_begin = pe_ptr[lb], e_end = pe_ptr[lb + 1];
    const double gravity = p->gravity;
    const bool nmd = KNOWN ? true : nmd_arg != 0, grav = KNOWN ? true : gravity != 0;
    const int nmat = p->nmat;

    // Everything below is ordered so that the global loads that do not depend on each other are in
    // flight TOGETHER: a workgroup's critical path is two round trips to memory (index, then record),
    // not one per phase -- with three workgroups per CU there is little else to hide them behind.
    struct Elem { int ew, mono; ushort4 ln; short4 sl; double s[6], vol, dpo; };
    auto load_elem_rec = [&](const ulonglong2 rec_, Elem &E) {
        const PatchElem PE_ = patch_elem_unpack(rec_);
        E.ew = PE_.ew; E.ln = make_ushort4(PE_.ln[0], PE_.ln[1], PE_.ln[2], PE_.ln[3]);
        E.sl = make_short4(PE_.sl[0], PE_.sl[1], PE_.sl[2], PE_.sl[3]);
        int e = E.ew & 0x3fffffff;
#ifdef DES_EXP_EN3_FAKE_ELEM
        e = (e & 1023) + (lb & 7) * 1024; // timing experiment only (wrong results): the element loads hit a few hot lines
#endif
        const unsigned eo = (unsigned)e * 8u;               // scalar plane base + one 32-bit offset (passes/common.hpp)
#if DES_EXP_EN3 & 16
        // timing experiment only (wrong results): the element's stress / volume / dpressure as ONE 64-byte record (two 32-byte
        // loads from consecutive addresses, read out of the stress planes' memory) instead of eight loads from eight planes
        {
            const d4 *rec = (const d4 *)stress;
            const size_t q = (size_t)(e % (ne / 4 * 3)) * 2;
            const d4 ra = rec[q], rb = rec[q + 1];
            E.s[0] = ra.x; E.s[1] = ra.y; E.s[2] = ra.z; E.s[3] = ra.w; E.s[4] = rb.x; E.s[5] = rb.y;
            E.vol = 1e7 + fabs(rb.z) * 1e-30; E.dpo = rb.w * 1e-30; E.mono = grav ? md.mono[e] : 0;
        }
#elif DES_EXP_EN3 & 8
        // timing experiment only (wrong results): no per-element global loads
        for (int k = 0; k < 6; ++k) E.s[k] = 1e6 * (k + 1) + e;
        E.vol = 1e7 + e; E.dpo = 1e3; E.mono = (0 << 16) | 4;
#else
        for (int k = 0; k < 6; ++k) E.s[k] = pl_ld(stress, k, ne, eo);
        E.vol = pl_ld(volume, 0, ne, eo);
        E.dpo = nmd ? pl_ld(dpressure, 0, ne, eo) : 0.0;
        E.mono = grav ? md.mono[e] : 0;
#endif
    };
    auto load_elem = [&](int i, Elem &E) { load_elem_rec(pe_pack[i], E); };
    // Round 5: REQUESTS first, uses later.  The waits count requests in order, so a result used where it was asked for holds the
    // lane until everything asked for before it is back: written phase by phase -- list entry, element data, list bounds (used on the
    // spot), ids, records -- the pass made five dependent trips to memory from the block's scalar bounds to the barrier.  Now three:
    //   1. this lane's list entry, the ids of the patch's other nodes, the node's list bounds and flags (none needs another);
    //   2. the element's stress / volume / dpressure / marker word (behind 1's first) and the nodal records (behind 1's second);
    //   3. nothing -- the barrier.
    Elem E0;
    const int i0 = e_begin + threadIdx.x, i1 = i0 + THREADS;
    // (no branches around the requests -- a lane past the end of a list asks for the list's last entry and drops it: where two
    //  paths meet, the compiler waits for the count that is safe on both, i.e. for nearly everything)
    const ulonglong2 rec0 = pe_pack[min(i0, e_end - 1)];
    constexpr int ROUNDS = DES_PATCH_PN / THREADS;          // (cap_pn <= DES_PATCH_PN = 512)
    int ids[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int j = threadIdx.x + r * THREADS;
        // (pn_id ends in a spare entry: a block without foreign nodes reads that; used -- and bit 31 masked -- in the staging loop)
        const int raw = pn_id[h0 + min(max(j - nown, 0), max(nh - 1, 0))];
        ids[r] = j < nown ? n0 + j : raw;
    }
    // (b) the node this lane will finish: its CSR segment and its records.  DES_EN3_SPLIT (blocks of up to 64 nodes):
    //     lane l of the first three wavefronts sums ONE force component of node n0 + l (below), so those lanes need the
    //     segment too; the first wavefront's lanes finish the nodes as before.
#if DES_EN3_SPLIT
    const bool split = npb <= 64;
#else
    const bool split = false;
#endif
    const int n = n0 + threadIdx.x;
    const bool has_node = (int)threadIdx.x < nown;
    const int nl = split ? (int)(threadIdx.x & 63) : (int)threadIdx.x, part = split ? (int)(threadIdx.x >> 6) : 0;
    const bool sums = nl < nown && part < 3;
    // (a lane without a node reads the block's first node's: no branch around the requests)
    const int nc = sums ? n0 + nl : n0;
    const int s_k = sup_idx[n0], s_a = sup_idx[nc], s_b = sup_idx[nc + 1];
    const unsigned flag_raw = bcflag[has_node ? n : n0];
    // (the ids are pinned HERE -- an empty asm that "changes" them: the compiler otherwise sinks their request into the
    //  conditional block that uses them, behind the element data, and the records wait a trip longer)
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) asm volatile("" : "+v"(ids[r]));
    // (c) the patch's nodes: own range first (local id = n - n0), then the listed others -- requested into registers, stored to LDS
    //     behind (a)'s requests (a lane past the patch asks for the block's first node and drops it)
    d4 xr[ROUNDS];
    double ntr[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int j = threadIdx.x + r * THREADS;
        int id = j < nown + nh ? (ids[r] & 0x7fffffff) : n0;      // (bit 31: a surface node, for EN1)
#ifdef DES_EXP_EN3_FAKE_STAGE
        id = n0 + (j & 63);              // timing experiment only (wrong results): the staging gathers hit lines already on their way
#endif
#if DES_EXP_EN3 & 4
        xr[r].x = 1.0 * j; xr[r].y = 2.0 * (j & 7); xr[r].z = 3.0 * (j & 3) + 0.5 * j; xr[r].w = 300.0; ntr[r] = 1.0 * id;   // timing experiment only
#else
        xr[r] = xt[id];
        ntr[r] = nmd ? ntmp[id] : 0.0;
#endif
    }
    // (a) this lane's first patch element: stress / volume / dpressure (holding a second one in registers as well costs a wave
    //     of occupancy and more than it hides)
    load_elem_rec(rec0, E0);
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int j = threadIdx.x + r * THREADS;
        if (j < nown + nh) { lxt[j] = xr[r]; if (nmd) lnt[j] = ntr[r]; }
    }
    int r0 = sums ? s_a - s_k : 0, r1 = sums ? s_b - s_k : 0;
    const unsigned flag = has_node ? flag_raw : 0u;
    __syncthreads();
    DES_STAMP0(1, 1);

    // the force terms of one patch element into the LDS slots of its incidences in this block
    auto do_elem = [&](const Elem &E) {
        const int e = E.ew & 0x3fffffff;
#if DES_EXP_EN3 & 1
        // timing experiment only (wrong results): the element phase without LDS reads and arithmetic (loads and LDS stores stay)
        {
            const double fake = E.s[0] + E.s[1] + E.s[2] + E.s[3] + E.s[4] + E.s[5] + E.vol + E.dpo + E.mono;
            const int slot_[4] = {E.sl.x, E.sl.y, E.sl.z, E.sl.w};
            for (int k = 0; k < 4; ++k) if (slot_[k] >= 0) { lf[0][slot_[k]] = fake; lf[1][slot_[k]] = fake; lf[2][slot_[k]] = fake; }
            return;
        }
#endif
        d4 c[4];
        c[0] = lxt[E.ln.x]; c[1] = lxt[E.ln.y]; c[2] = lxt[E.ln.z]; c[3] = lxt[E.ln.w];
        double s[6];
        for (int k = 0; k < 6; ++k) s[k] = E.s[k];
        if (nmd) {                                          // is_using_mixed_stress, outside the isostasy loop
            double dp = 0;
            dp += lnt[E.ln.x]; dp += lnt[E.ln.y]; dp += lnt[E.ln.z]; dp += lnt[E.ln.w];
            double dp_el = dp / 4;
            double dp_orig = E.dpo;
            double ddp = (-dp_orig + dp_el) / 3;
            for (int k = 0; k < 3; ++k) s[k] += ddp;
            if (E.ew & 0x40000000) ddp_out[e] = ddp;        // this block owns the element
        }
        const double vol = E.vol;
        double sx[4], sy[4], sz[4];
        desk::shape_fn(c, vol, sx, sy, sz);
        double buoy = 0;
        if (grav) {
            double T = 0;
            T += c[0].w; T += c[1].w; T += c[2].w; T += c[3].w;
            T /= 4;
            const desk::Mix mx = mix_from_mono(md, nmat, e, E.mono);
            const double rho = desk::mat_rho(p, mx, T);
            const double phi = load_props(p, md, mx, ne, e).phi;
            buoy = (rho * (1 - phi) + 1000.0 * phi) * gravity / 4;
        }
        const int slot[4] = {E.sl.x, E.sl.y, E.sl.z, E.sl.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (slot[k] < 0) continue;
            lf[0][slot[k]] = (s[0]*sx[k] + s[3]*sy[k] + s[4]*sz[k]) * vol;
            lf[1][slot[k]] = (s[3]*sx[k] + s[1]*sy[k] + s[5]*sz[k]) * vol;
            lf[2][slot[k]] = (s[4]*sx[k] + s[5]*sy[k] + s[2]*sz[k] + buoy) * vol;
        }
    };
    if (i0 < e_end) do_elem(E0);
    for (int i = i1; i < e_end; i += THREADS) {             // the rest of the patch (one more round, seldom two)
        Elem E;
        load_elem(i, E);
        do_elem(E);
    }
    DES_STAMP0(1, 2);
    __syncthreads();
    DES_STAMP0(1, 3);

    // the block's nodes: force sums in CSR order, then the rest of the nodal update
    double l2 = 0.0;
#if DES_EN3_SPLIT
    if (split) {
        // The three components of a node's force are independent sums over the same CSR segment: wavefront c forms
        // component c for the block's 64 nodes -- each sum in list order, the reference's association (fields.cxx:667-675),
        // the same bits -- instead of one wavefront of eight walking all three.  The sum goes back into the FIRST slot of
        // the segment it was formed from (no other lane reads that segment; the last slot still holds the term
        // force_residual is assigned, fields.cxx:673); a segment of one slot is left alone and finished below.
        d4 m4;
        if (has_node) m4 = vm[n];                           // {vx,vy,vz,mass}: requested behind the element phase (see below)
        if (sums) {
            double *L = lf[part];
            double f = 0;
            int k = r0;
            for (; k + 8 <= r1; k += 8) {                   // eight slots requested from LDS before the first is used
                double t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) t[u] = L[k + u];
#pragma unroll
                for (int u = 0; u < 8; ++u) f -= t[u];
            }
            for (; k + 4 <= r1; k += 4) {
                double t[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) t[u] = L[k + u];
#pragma unroll
                for (int u = 0; u < 4; ++u) f -= t[u];
            }
            for (; k < r1; ++k) f -= L[k];
            if (r1 - r0 >= 2) L[r0] = f;
        }
        __syncthreads();
        DES_STAMP0(1, 4);
        if (has_node) {
            double f[3] = {0, 0, 0}, fr[3] = {0, 0, 0};
            if (r1 - r0 >= 2) {
#pragma unroll
                for (int c = 0; c < 3; ++c) { f[c] = lf[c][r0]; fr[c] = lf[c][r1 - 1]; }
            } else if (r1 - r0 == 1) {
#pragma unroll
                for (int c = 0; c < 3; ++c) { const double tv = lf[c][r0]; f[c] -= tv; fr[c] = tv; }
            }
            l2 = n3_finish_node(p, clk, n, nn, o0, nn_own_end, nn_global, f, fr, bcflag, bc_mask, bcn_idx, bcn_ent, bcf_tmp,
                                coord0, ymass, bnormals, edge_vec, edge_slot, flag, lxt[threadIdx.x], m4, xt_out, true, vm, force, fres, outs != 0);
        }
    } else
#endif
#if DES_EXP_EN3 & 2
    if (r1 > r0) r1 = r0 + 1;          // timing experiment only (wrong results): one term per force sum
#endif
    if (has_node) {
        // {vx,vy,vz,mass}: requested here, behind the force sums, not ahead of the element phase -- held across it
        // the record went to scratch (at 80 VGPRs), 24 MB of stores per launch for 2901 workgroups
        const d4 m4 = vm[n];
        double f[3] = {0, 0, 0}, fr[3] = {0, 0, 0};
        int k = r0;
        for (; k + 8 <= r1; k += 8) {                       // eight slots requested from LDS before the first is used
            double t[3][8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { t[0][u] = lf[0][k + u]; t[1][u] = lf[1][k + u]; t[2][u] = lf[2][k + u]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) { f[0] -= t[0][u]; f[1] -= t[1][u]; f[2] -= t[2][u]; }
            fr[0] = t[0][7]; fr[1] = t[1][7]; fr[2] = t[2][7];
        }
        for (; k + 4 <= r1; k += 4) {
            double t[3][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { t[0][u] = lf[0][k + u]; t[1][u] = lf[1][k + u]; t[2][u] = lf[2][k + u]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) { f[0] -= t[0][u]; f[1] -= t[1][u]; f[2] -= t[2][u]; }
            fr[0] = t[0][3]; fr[1] = t[1][3]; fr[2] = t[2][3];
        }
        for (; k < r1; ++k) {
            const double t0v = lf[0][k], t1v = lf[1][k], t2v = lf[2][k];
            f[0] -= t0v; f[1] -= t1v; f[2] -= t2v;
            fr[0] = t0v; fr[1] = t1v; fr[2] = t2v;          // assignment: fields.cxx:673
        }
        l2 = n3_finish_node(p, clk, n, nn, o0, nn_own_end, nn_global, f, fr, bcflag, bc_mask, bcn_idx, bcn_ent, bcf_tmp,
                            coord0, ymass, bnormals, edge_vec, edge_slot, flag, lxt[threadIdx.x], m4, xt_out, true, vm, force, fres, outs != 0);
    }
    DES_STAMP0(1, 5);
    // per-block partial of the residual; the partials are added in block order afterwards
    l2 = desk::wave_sum(l2);
    if (npb <= 64) {
        // the block's nodes are all lanes of the first wavefront: the other wavefronts' shares are +0.0, and t + 0.0 = t for
        // the t >= +0.0 of a sum of squares -- the same bits without the barrier and the trip through LDS
        if (threadIdx.x == 0) res_part[lb] = l2;
        DES_STAMP0(1, 6);
        return;
    }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = l2;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = red[0];
        for (int i = 1; i < THREADS / 64; ++i) t += red[i];
        res_part[lb] = t;
    }
    DES_STAMP0(1, 6);
}
