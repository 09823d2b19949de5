// engine/launch.hpp -- Host side: allocation helpers and the launch of every pass, in step order.
// Part of the single translation unit des_dev.hip (included inside namespace des_hip, after
// DevClock / struct des_dev); not a stand-alone header.

// =====================================================================================
// host side of the engine
// =====================================================================================
#define DES_ALLOC_SLACK 8192
// the over-read of the pipelined stress update's last tile: up to TILE - 1 = 255 records past the end of an array, the widest
// being a connectivity record (passes/e2.hpp: dma)
static_assert(DES_ALLOC_SLACK >= 255 * sizeof(int4), "DES_ALLOC_SLACK no longer covers the last tile of E2_update_stress_pipe");
template <typename T>
int dev_alloc(T *&ptr, size_t count)
{
    ptr = nullptr;
    if (count == 0) count = 1;
    // DES_ALLOC_SLACK bytes behind every array: the pipelined stress update (passes/e2.hpp) moves whole 64-element pieces of
    // the element arrays and the connectivity, so its last tile reads up to 255 records past the end (never used, never written)
    hipError_t e = hipMalloc((void **)&ptr, count * sizeof(T) + DES_ALLOC_SLACK);
    if (e != hipSuccess) { g_last_error = std::string("hipMalloc: ") + hipGetErrorString(e); return DES_ERR_RESOURCE; }
    return DES_OK;
}

template <typename T>
int dev_upload(T *dst, const T *src, size_t count, hipStream_t s)
{
    if (count == 0) return DES_OK;
    HIP_OK(hipMemcpyAsync(dst, src, count * sizeof(T), hipMemcpyHostToDevice, s));
    HIP_OK(hipStreamSynchronize(s));
    return DES_OK;
}

inline int nblk(long long n) { return (int)((n + DES_BLOCK - 1) / DES_BLOCK); }
// LDS array lengths of the patch passes for a mesh's largest block (multiples of 8 entries: 16-byte aligned sub-arrays)
inline int patch_cap(int n) { return (n + 7) & ~7; }
// grids are rounded up to a multiple of 8 so the XCD-aware block map covers every chunk
inline int nblk8(long long n) { int b = nblk(n); return (b + 7) / 8 * 8; }

struct Launch {
    des_dev *h; int k; ProfRec rec; bool on; hipStream_t s;
    Launch(des_dev *h_, int k_, hipStream_t s_ = nullptr) : h(h_), k(k_), on(h_->prof), s(s_ ? s_ : h_->stream) {
        // (timing-only events: no system-scope fence when they complete -- hip_runtime_api.h: hipEventDisableSystemFence, "for events
        //  that are only being used to measure timing ... avoiding the cost of cache writeback and invalidation"; with the default
        //  flags the end event of a pass that has just written ~100 MB also times the write-back of the dirty L2 lines, which a
        //  dependent kernel of the same stream never waits for: the 2-D stress update read 79 us by events against 71 by rocprofv3)
        if (on) { hipEventCreateWithFlags(&rec.a, hipEventDisableSystemFence); hipEventCreateWithFlags(&rec.b, hipEventDisableSystemFence); rec.k = k; hipEventRecord(rec.a, s); }
    }
    ~Launch() { if (on) { hipEventRecord(rec.b, s); h->prof_recs.push_back(rec); } }
};

// Timing experiments only (tools/launch_cost.py): DES_EXP_SKIP=e2r,s2,s3,dt leaves those launches
// out, which makes the results WRONG wherever they had work.  Compiled into experiment builds alone
// (make HIPFLAGS+=-DDES_EXPERIMENTS): the shipped library has no such switch.
#ifdef DES_EXPERIMENTS
inline bool exp_skip(const char *what)
{
    static const char *env = des_env::get("DES_EXP_SKIP");
    return env && std::strstr(env, what) != nullptr;
}
#else
inline bool exp_skip(const char *) { return false; }
#endif

inline MatData mat_data(const des_dev *h) { return MatData{ h->markers, h->mono, h->props, h->ptab, h->pptab }; }

void refresh_props(des_dev *h)
{
    if (!h->markers_dirty) return;
    Launch l(h, K_MISC);
    hipLaunchKernelGGL(k_props, dim3(nblk(h->ne)), dim3(DES_BLOCK), 0, h->stream, h->d_p, h->markers, h->props, h->mono, h->ne);
    h->markers_dirty = false;
}

// which elements an E1 launch covers: everything, the interior ones (every node owned), or the two
// groups that touch the ghost nodes (first and last in the engine's order, engine/order.hpp)
enum { E1_ALL = 0, E1_INTERIOR = 1, E1_GHOST_SIDE = 2 };

template <int MODE>
void launch_e1(des_dev *h, int part = E1_ALL)
{
    int b0 = 0, c0 = h->ne, b1 = 0, c1 = 0;
    if (part == E1_INTERIOR)   { b0 = h->e_int0; c0 = h->e_int1 - h->e_int0; }
    if (part == E1_GHOST_SIDE) { b0 = 0; c0 = h->e_int0; b1 = h->e_int1; c1 = h->ne - h->e_int1; }
    if (c0 + c1 == 0) return;
    Launch l(h, K_E1);
    // MODE_DT: one slot of compute_dt partials per workgroup, after those of an earlier part of the same pass
    if ((MODE & MODE_DT) && h->dt_parts_used + nblk8(c0 + c1) > h->dt_part_cap) {
        hipLaunchKernelGGL(k_dt_fold, dim3(1), dim3(DES_BLOCK), 0, h->stream, h->d_clk, h->dt_part, h->dt_part_cap, h->dt_parts_used);
        h->dt_parts_used = 0;
    }
    const int dt_base = h->dt_parts_used;
    if (MODE & MODE_DT) h->dt_parts_used += nblk8(c0 + c1);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(E1_geom_rotate_strainrate<MODE>), dim3(nblk8(c0 + c1)), dim3(DES_BLOCK), 0, h->stream,
                       h->d_p, h->d_clk, h->ne, nblk(c0 + c1), b0, c0, b1, c1, h->conn, h->xt, h->vm, mat_data(h), h->radiogenic,
                       h->topflag, h->stress, (h->patch && h->ddp_live) ? h->ddp : nullptr, h->strain, h->plstrain, h->volume, h->volume_old,
                       h->strain_rate, h->mrec, h->ttmp, h->spin, h->dt_part, h->dt_part_cap, dt_base);
    if (MODE & MODE_DEFER) { h->rot_pending = true; h->rot_prev_dt = (MODE & MODE_DT) != 0; }
    // a C pass folds EN3's pending NMD increments into the stress (passes/e1.hpp): they are spent once its last part is
    // launched -- a second end-of-step pass without a force pass in between then gets no ddp pointer and adds nothing
    // (with MODE_DEFER the elements off the top surface are folded by the next stress update, launch_e2)
    if ((MODE & MODE_C) && !(MODE & (MODE_INIT | MODE_DEFER)) && part != E1_INTERIOR) h->ddp_live = false;
}

// A fused end-of-step pass may leave rotate_stress to the next stress update when nothing reads the
// stress in between: rotation on (an elastic part in the rheology), a plain time step (not the
// isostasy / pseudo-transient loops), no per-step averaging pass, no hipGraph replay of the classic passes (that E2 node
// is captured once, without the pending pointers; the fused step has graphs of its own, des_dev_step).
// (geo: the deferral into E2<GEO>, which folds Output::average_fields in; the E1<MODE_DEFER> route does not: the
//  averaging pass behind it would see the stress before its rotation)
inline bool defer_rot_ok(const des_dev *h, bool geo = false)
{
    return h->defer_rot && h->spin && (h->p.rheol_type & DES_RH_ELASTIC) && !h->iso && !h->p.has_PT
           && (geo || !h->p.is_outputting_averaged_fields) && !(h->use_graph && !h->patch);
}

// Output::average_fields (output.cxx:327-370) on the end-of-step fields, i.e. after the C part
// of E1.  A kernel of its own: fused into E1 it cost that kernel a wave of occupancy (186 VGPRs;
// E1 117 us instead of 74 + 30 for this pure stream of 168 B per element).
__global__ void __launch_bounds__(DES_BLOCK)
k_average_fields(const des_params *__restrict__ p, DevClock *__restrict__ clk, int ne,
                 const double *__restrict__ stress, const double *__restrict__ strain,
                 const double *__restrict__ delta_plstrain, double *__restrict__ stress_avg,
                 double *__restrict__ dplstrain_avg, double *__restrict__ strain0)
{
    const int e = blockIdx.x * DES_BLOCK + threadIdx.x;
    if (e >= ne) return;
    if (clk->steps % p->quality_check_step_interval == 1) {
        if (e == 0) clk->avg_time0 = clk->time;
        for (int i = 0; i < 6; ++i) {
            stress_avg[(size_t)i*ne + e] = stress[(size_t)i*ne + e];
            strain0[(size_t)i*ne + e] = strain[(size_t)i*ne + e];
        }
        dplstrain_avg[e] = delta_plstrain[e];
    } else {
        for (int i = 0; i < 6; ++i) stress_avg[(size_t)i*ne + e] += stress[(size_t)i*ne + e];
        dplstrain_avg[e] += delta_plstrain[e];
    }
}

// EN1 (passes/en1.hpp) can stand in for N1 when the element terms are functions of the CURRENT nodal
// records alone: moving mesh (volumes and masses recomputed every step), not the isostasy loop,
// no damping option 4 (its Young's-modulus mass is a gather of its own), and not the overlapped
// multi-GPU schedule (which cuts the end-of-step pass in two)
inline bool en1_ok(const des_dev *h)
{
    // (nor with the pseudo-transient loop: its iterations re-enter the passes without a clock or temperature update)
    return h->patch && h->patch_n1 && h->p.has_moving_mesh && !h->iso && h->p.damping_option != 4 && !h->p.has_PT;
}

// The next stress update can do the end-of-step pass of this step itself: E2<GEO> (passes/e2.hpp) forms
// volume, strain rate and spin from the nodal records and finishes the top elements in registers -- no
// end-of-step launch at all between two plain steps of a call (every 10th step: the compute_dt reduction
// alone, E1<DT | VOLX>).
inline bool e2geo_ok(const des_dev *h)
{
    static const char *env = des_env::get("DES_E2GEO");
    return !(env && env[0] == '0') && en1_ok(h) && defer_rot_ok(h, true) && h->topflag;
}

// The first step of a call can start like an interior one -- EN1, then E2<GEO> with RotPending::fresh -- instead of
// E1<A> + N1 + E2 when the state is the one the last des_dev_step call left (h->finished: no upload, clock change or
// other entry point in between) and the fused step is available: EN1 and E2<GEO> form from the nodal records what E1<A>
// would store for N1 / E2 to read back.  (A decomposed mesh: the last step of a call ends in order, ghost region refreshed,
// end-of-step pass on every local element -- the same holds.)
inline bool fresh_ok(const des_dev *h)
{
    return h->finished && h->fresh_on && e2geo_ok(h) && !h->use_graph && !h->p.is_outputting_averaged_fields;
}

// The surface step (surface_processes, bc.cxx:1709-1872: S2 + S3 here) of a step can be left to the passes of the
// NEXT step: its two launches do O(surface) work and cost ~5 us each whatever the mesh size -- 12 of a 57-us step on
// a 137k-tet shard, 14 of 201 at 1M tets.  EN1 redoes the diffusion for the surface nodes of each patch while it
// stages their records (passes/en1.hpp: SurfPending), writes the committed heights with its own nodes' records, and
// the next stress update's tail workgroups add the edvacc_surf terms.  When: the next step starts with EN1 and runs
// E2<GEO> (so nothing else reads coordinates in between), the step is a plain one -- no compute_dt (its finalize
// reads max_surf_vel, and EN1 would see the new dt), no quality-check step (the dhacc reset follows the surface
// step), no first step of an averaging interval (its coordinate snapshot wants the committed surface) -- and nobody
// needs the step's scalars (l2 residual, max_surf_vel: the last step of a call never defers).  DES_S2_DEFER=0: off.
inline bool s2_defer_ok(const des_dev *h, bool with_next, long long step_no)
{
    if (!h->s2_defer || !with_next || !e2geo_ok(h) || h->use_graph || !h->tfan) return false;
    if (!(h->p.has_moving_mesh && h->ntop > 0)) return false;
    const long long qcsi = h->p.quality_check_step_interval;
    if (step_no % 10 == 0 || step_no % qcsi == 0) return false;
    if (h->p.is_outputting_averaged_fields && step_no % qcsi == 1) return false;
    return true;
}

// Overlapped multi-GPU schedule on the fused step (DES_OVERLAP=1): a plain step whose surface step is left to the next
// step's passes ends with EN3 -- nothing of it reads the ghost region after the exchange has been issued.  So the
// transfer + unpack run on the side stream while the NEXT step's EN1 and E2<GEO> work through the node blocks /
// elements deep inside the slab (engine/order.hpp: nothing they read is written by the unpack); step_front joins, then
// runs the two passes on the rest.  Needs a slab thick enough to have a deep part.
inline bool deep_split_ok(const des_dev *h)
{
    if (!h->patch || h->e_deep1 <= h->e_deep0) return false;
    const int npb = h->patch_npb, d0 = (h->n_deep0 + npb - 1) / npb, d1 = h->n_deep1 / npb;
    return d1 > d0 && npb <= DES_DEEP_MARGIN && !h->p.is_outputting_averaged_fields;
}

// end-of-step E1 (C part) of step `step_no`, optionally fused with the A part of the next step
void launch_e1_end(des_dev *h, long long step_no, bool with_next, int part = E1_ALL)
{
    const bool do_dt = (step_no % 10 == 0);
    if (with_next && part == E1_ALL && e2geo_ok(h)) {
        // nothing to launch, the next E2 does it -- but for the compute_dt reduction of every 10th step, which
        // then runs alone (on the new volume, formed from the coordinates); the rotation of this step uses
        // the dt k_dt_finalize is about to replace
        if (do_dt) launch_e1<MODE_DT | MODE_VOLX>(h, part);
        h->e2geo_next = true;
        h->rot_prev_dt = do_dt;
        return;
    }

    const int sel = (with_next ? 1 : 0) | (do_dt ? 2 : 0);
    const bool norec = with_next && en1_ok(h);       // the next step's N1 is EN1: no mrec / ttmp needed
    const bool dfr = with_next && defer_rot_ok(h);
    switch (sel) {
    case 0: launch_e1<MODE_C>(h, part); break;
    case 1:
        if (dfr) { if (norec) launch_e1<MODE_C | MODE_A | MODE_NOREC | MODE_DEFER>(h, part); else launch_e1<MODE_C | MODE_A | MODE_DEFER>(h, part); }
        else     { if (norec) launch_e1<MODE_C | MODE_A | MODE_NOREC>(h, part); else launch_e1<MODE_C | MODE_A>(h, part); }
        break;
    case 2: launch_e1<MODE_C | MODE_DT>(h, part); break;
    case 3:
        if (dfr) { if (norec) launch_e1<MODE_C | MODE_A | MODE_DT | MODE_NOREC | MODE_DEFER>(h, part); else launch_e1<MODE_C | MODE_A | MODE_DT | MODE_DEFER>(h, part); }
        else     { if (norec) launch_e1<MODE_C | MODE_A | MODE_DT | MODE_NOREC>(h, part); else launch_e1<MODE_C | MODE_A | MODE_DT>(h, part); }
        break;
    }
    if (part == E1_INTERIOR) return;               // the averaging pass follows the last part
    if (h->p.is_outputting_averaged_fields) {
        Launch l(h, K_MISC);
        hipLaunchKernelGGL(k_average_fields, dim3(nblk(h->ne)), dim3(DES_BLOCK), 0, h->stream, h->d_p, h->d_clk, h->ne,
                           h->stress, h->strain, h->delta_plstrain, h->stress_avg, h->dplstrain_avg, h->strain0);
    }
}

// coordinates at the first step of an averaging interval (output.cxx:334-338)
__global__ void k_avg_coord0(int nn, const d4 *__restrict__ xt, double *__restrict__ coord_avg0)
{
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= nn) return;
    const d4 c = xt[n];
    coord_avg0[n] = c.x; coord_avg0[(size_t)nn + n] = c.y; coord_avg0[(size_t)2*nn + n] = c.z;
}

void launch_avg_coord0(des_dev *h, long long step_no)
{
    if (h->p.is_outputting_averaged_fields && step_no % h->p.quality_check_step_interval == 1)
        hipLaunchKernelGGL(k_avg_coord0, dim3(nblk(h->nn)), dim3(DES_BLOCK), 0, h->stream, h->nn, h->xt, h->coord_avg0);
}

// node workgroups: ceil(owned nodes / nodes per workgroup), and the grid rounded up to the 8 XCDs
// The node kernels run over EVERY local node: on a decomposed mesh the ghost region is computed
// redundantly (des_halo); only the reductions are restricted to the owned range [o0, o1).
inline int node_blocks(const des_dev *h) { return (h->nn + h->npb - 1) / h->npb; }
inline int node_grid(const des_dev *h) { return (node_blocks(h) + 7) / 8 * 8; }
// residual partials: one per N3 workgroup or one per EN3 patch block, whichever pass runs
inline int res_part_size(const des_dev *h) { return std::max(node_grid(h), h->patch ? (h->patch_nb + 7) / 8 * 8 : 0); }
// EN3's NMD increments that E1 has not folded into the stress yet (between the force pass and the
// end-of-step pass of a step outside the isostasy loop): what the ghost-region pack must add
inline double *pending_ddp(const des_dev *h) { return (h->patch && h->ddp_live && h->p.is_using_mixed_stress && !h->iso) ? h->ddp : nullptr; }

// nodes per node-kernel workgroup: 256, or 64 while that leaves fewer than two workgroups per CU
// (a 137k-tet mesh has 31k nodes = 120 workgroups of 256 on 256 CUs, each walking 4-5 incidence
// tiles one after the other)
void choose_npb(des_dev *h)
{
    const char *env = des_env::get("DES_NPB");
    const int nown = h->nn;
    h->npb = (nown < 512 * DES_BLOCK) ? 64 : DES_BLOCK;
    if (env && (std::atoi(env) == 64 || std::atoi(env) == 128 || std::atoi(env) == 256)) h->npb = std::atoi(env);
}

// compute_mass gather alone (N1 without the temperature / dvoldt parts)
void launch_mass_gather(des_dev *h)
{
    hipLaunchKernelGGL(HIP_KERNEL_NAME(N1_mass_temperature_dvoldt<0, 0>), dim3(node_grid(h)), dim3(DES_BLOCK), 0, h->stream,
                       h->d_p, h->d_clk, 0, h->nn, node_blocks(h), h->npb, h->sup_idx, h->sup_pack, h->bcflag, h->mrec, h->ttmp,
                       mat_data(h), h->ne, h->xt, h->vm, h->volume_n, h->tmass, h->ymass, h->ntmp);
}

void launch_dt_finalize(des_dev *h, const double *red)
{
    if (exp_skip("dt")) return;
    Launch l(h, K_DTFIN);
    // (after the cross-rank reduction k_dt_pack has already folded the partials in: dt_parts_used is 0 then)
    hipLaunchKernelGGL(k_dt_finalize, dim3(1), dim3(DES_BLOCK), 0, h->stream, h->d_p, h->d_clk, red, h->dt_part, h->dt_part_cap,
                       h->dt_parts_used);
    h->dt_parts_used = 0;
}

inline bool surface_diffusion_on(const des_dev *h)
{
    return (h->p.has_moving_mesh || h->iso) && h->p.surface_process_option == 1 && h->ntop > 0;
}

// ---- passes of one step, in launch order -----------------------------------------
void launch_n1(des_dev *h)
{
    Launch l(h, K_N1);
    const int nbn = node_blocks(h);
    if (h->const_mass)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(N1_mass_temperature_dvoldt<1, 1>), dim3(node_grid(h)), dim3(DES_BLOCK), 0, h->stream,
                           h->d_p, h->d_clk, 0, h->nn, nbn, h->npb, h->sup_idx, h->sup_pack, h->bcflag, h->mrec, h->ttmp, mat_data(h),
                           h->ne, h->xt, h->vm, h->volume_n, h->tmass, h->ymass, h->ntmp);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(N1_mass_temperature_dvoldt<1, 0>), dim3(node_grid(h)), dim3(DES_BLOCK), 0, h->stream,
                           h->d_p, h->d_clk, 0, h->nn, nbn, h->npb, h->sup_idx, h->sup_pack, h->bcflag, h->mrec, h->ttmp, mat_data(h),
                           h->ne, h->xt, h->vm, h->volume_n, h->tmass, h->ymass, h->ntmp);
}

// update_stress in two passes when the rheology has a yield surface: the first pass (3 waves
// per SIMD) sets the few elements that need the return mapping aside, the second one (the same
// code with the return mapping, 2 waves per SIMD) works that list off.  The list is sparse, so
// the second pass pays ~6x per element; above DES_E2_DEFER_MAX of the mesh one pass is cheaper.
// Both give the same bits.  DES_E2_DEFER=0 / 1 pins the mode; default: choose_e2_mode().
// With E2<GEO> (two waves per SIMD either way, and the return mapping inlined without a scratch frame) one pass
// is never slower -- 200.9 against 201.6 us per step with nothing set aside (1M tets), 115 against 128 us for
// the pass at 2.5 % --, so the fused step runs ONE pass and the second launch is gone from it.
#ifndef DES_E2R_GRID
#define DES_E2R_GRID 512          // workgroups of the second pass (grid-stride loop): two per CU, all resident
#endif
#ifndef DES_E2_DEFER_MAX
#define DES_E2_DEFER_MAX 0.01      // (the classic first pass runs three waves per SIMD: 70 + 5 us against 81 in one pass with nothing set aside)
#endif
// called whenever the host copy of the clock is fresh (end of des_dev_step / des_dev_phase calls)
void choose_e2_mode(des_dev *h)
{
    if (h->e2_defer == 2) h->e2_two_pass = e2geo_ok(h) ? false : h->h_clk->n_defer <= DES_E2_DEFER_MAX * h->ne;
}

// which elements / node blocks a launch of the fused step covers (overlapped multi-GPU schedule, step_front): everything,
// the part deep inside the slab (nothing it reads is written by the ghost-region exchange, engine/order.hpp), or the
// rest around it
enum { PART_ALL = 0, PART_DEEP = 1, PART_REST = 2 };

// Launch shape of the one-pass E2<GEO> by size: the kernel runs at two waves per SIMD (2 x 4 x CUs wavefronts resident) and
// has an instantiation held to three (passes/e2.hpp: W3).  Where the launch is a few rounds of workgroups at most, the
// number of ROUNDS decides -- a shard whose wavefronts just overflow the two-wave residency pays a whole second round --
// so take three waves when that saves a round; on large meshes (many rounds) two waves are as fast and spill nothing.
// DES_E2_W3 = 0 / 1 (read when the engine is created) pins the choice.
inline bool e2_three_waves(const des_dev *h, long long nelem)
{
    if (h->e2_w3_mode >= 0) return h->e2_w3_mode == 1;
    const long long waves = (nelem + 63) / 64, res2 = 2LL * 4 * h->n_cu, res3 = 3LL * 4 * h->n_cu;
    const long long r2 = (waves + res2 - 1) / res2, r3 = (waves + res3 - 1) / res3;
    return r3 < r2 && r2 <= 4;
}

// The pipelined E2<GEO> (passes/e2.hpp: E2_update_stress_pipe): 2 x CUs resident workgroups walking the tiles, the next tile's
// own-index data arriving in LDS by DMA under the current tile's arithmetic.  Worth it where a workgroup has several tiles to
// walk (>= 4 per resident workgroup); a shard of a few rounds keeps the plain kernel (and its three-wave shape).  Needs the
// whole mesh as one range and an even plane stride (16-byte DMA pieces).  DES_E2_PIPE = 0 / 1 (read when the engine is created) pins the choice.
inline bool e2_pipelined(const des_dev *h, int part, long long nelem)
{
    if (part != PART_ALL || (h->ne & 1) || !h->topflag) return false;
    if (h->e2_pipe_mode >= 0) return h->e2_pipe_mode == 1;
    const long long tiles = (nelem + DES_BLOCK - 1) / DES_BLOCK;
    return tiles >= 4LL * 2 * h->n_cu;
}

void launch_e2(des_dev *h, int part = PART_ALL)
{
    int e_begin = 0, e_count = h->ne, e_begin2 = 0, e_count2 = 0;
    if (part == PART_DEEP) { e_begin = h->e_deep0; e_count = h->e_deep1 - h->e_deep0; }
    if (part == PART_REST) { e_count = h->e_deep0; e_begin2 = h->e_deep1; e_count2 = h->ne - h->e_deep1; }
    const bool whole = part != PART_DEEP;                  // the launch that carries the facet workgroups and ends the pass
    if (e_count + e_count2 == 0 && !whole) return;
    // (PART_DEEP runs between the two EN1 launches of the step, before the coordinate buffers swap: the records of
    //  the deep nodes are the ones EN1 has just written to the other buffer)
    const d4 *const xt_now = part == PART_DEEP ? h->xt_alt : h->xt;
    // (the first step of a call in the fused flow is a classic stress update: three waves per SIMD in the first of two
    //  passes against two in one pass -- it keeps the classic rule when the mode is not pinned)
    const bool fresh = h->e2_fresh;                        // first step of a call on a finished state (fresh_ok): E2<GEO>, nothing pending
    bool two_pass = h->e2_two_pass;
    if (h->e2_defer == 2 && !h->e2geo_next && !fresh && !h->use_graph) two_pass = h->h_clk->n_defer <= DES_E2_DEFER_MAX * h->ne;
    // The two parts of the overlapped schedule run ONE pass, whatever DES_E2_DEFER pins: both would append to the same
    // defer_list / DevClock::n_defer (only EN1's clock lane resets the count, once per step), so the return-mapping launch
    // behind PART_REST would walk the deep part's entries a second time -- and e2_element<RM = 1> is not idempotent (it
    // re-reads the stress it has stored).  One pass gives the same bits and keeps the count (n_return_mapping) right.
    if (part != PART_ALL) two_pass = false;
    const bool defer = two_pass && (h->p.rheol_type == DES_RH_EP || h->p.rheol_type == DES_RH_EVP);
    int *count = &h->d_clk->n_defer;
    // the rotation (and NMD increment) the fused E1<MODE_DEFER> of the step before left for this pass
    RotPending rp = {nullptr, nullptr, nullptr, 0, nullptr, 1, nullptr, nullptr, nullptr, 0, 1, 0, nullptr, nullptr, nullptr, fresh ? 1 : 0};
    if (h->p.is_outputting_averaged_fields && e2geo_ok(h) && part == PART_ALL) {
        rp.dplstrain_avg = h->dplstrain_avg;
        rp.avg_dpl = h->e2_not_last ? 1 : 0;            // the last step of a call ends with E1 + k_average_fields
        rp.qcsi = (int)h->p.quality_check_step_interval;
    }
    const bool geo = h->e2geo_next || fresh;
    if (h->rot_pending || geo) {
        rp.spin = h->spin; rp.topflag = h->topflag; rp.prev_dt = h->rot_prev_dt ? 1 : 0; rp.vm = h->vm;
        rp.ddp = pending_ddp(h);
        rp.outputs = (geo && h->e2_elide) ? 0 : 1;
        if (geo && !fresh && h->p.is_outputting_averaged_fields) { rp.stress_avg = h->stress_avg; rp.strain0 = h->strain0; }
    }
    {
        Launch l(h, geo ? K_E2G : K_E2);
        auto k = h->portable_libm
            ? (geo ? (defer ? E2_update_stress<desk::MathPortable, 1, 1> : E2_update_stress<desk::MathPortable, 0, 1>)
                   : (defer ? E2_update_stress<desk::MathPortable, 1, 0> : E2_update_stress<desk::MathPortable, 0, 0>))
            : (geo ? (defer ? E2_update_stress<desk::MathOcml, 1, 1> : E2_update_stress<desk::MathOcml, 0, 1>)
                   : (defer ? E2_update_stress<desk::MathOcml, 1, 0> : E2_update_stress<desk::MathOcml, 0, 0>));
        // the headline rheology has kernels of its own (the law known at compile time: passes/e2.hpp)
        if (geo && h->portable_libm && h->p.rheol_type == DES_RH_EVP && h->p.is_using_mixed_stress && !h->p.is_outputting_averaged_fields)
            k = defer ? E2_update_stress<desk::MathPortable, 1, 1, DES_RH_EVP> : E2_update_stress<desk::MathPortable, 0, 1, DES_RH_EVP>;
        // ... and the one-pass E2<GEO> a three-wave shape for launches that it saves a round of workgroups (e2_three_waves)
        if (geo && !defer && h->portable_libm && e2_three_waves(h, e_count + e_count2)) {
            if (h->p.rheol_type == DES_RH_EVP && h->p.is_using_mixed_stress && !h->p.is_outputting_averaged_fields)
                k = E2_update_stress<desk::MathPortable, 0, 1, DES_RH_EVP, 1>;
            else
                k = E2_update_stress<desk::MathPortable, 0, 1, 0, 1>;
        }
        // with EN3 the stress-bc facet workgroups ride here (with the classic pair: in E3's launch)
        const int nbf = (h->patch && whole) ? nblk(h->nbcf) : 0;
        // ... and so does the edvacc_surf update of a surface step EN1 has just done for the step before (s2_defer_ok)
        int nsf = 0;
        if (h->edv_pending && h->patch && whole) {
            rp.edv_etop = h->etop; rp.edv_conn_surf = h->conn_surf; rp.edv_dh_n = h->dh_n; rp.edv_edvacc = h->edvacc;
            nsf = nblk(h->etop);
            h->edv_pending = false;
        }
        const int e_all = e_count + e_count2;
        if (geo && !defer && h->portable_libm && e2_pipelined(h, part, e_all)) {
            // (a variant with ONE resident workgroup of twelve wavefronts per CU -- three waves per SIMD at 168 VGPRs -- spills 350 B
            //  per lane: the tile loop with its gathers in flight wants ~250 registers; not kept)
            constexpr int nw = 4;
            const int tile = nw * 64;
            const int ntiles = (e_all + tile - 1) / tile;
            const int npers = std::min((2 * h->n_cu + 7) / 8 * 8, (ntiles + 7) / 8 * 8);
            const bool evp = h->p.rheol_type == DES_RH_EVP && h->p.is_using_mixed_stress && !h->p.is_outputting_averaged_fields;
            auto kp = evp ? E2_update_stress_pipe<desk::MathPortable, DES_RH_EVP, 4> : E2_update_stress_pipe<desk::MathPortable, 0, 4>;
            const int nbf_p = nbf, nsf_p = nsf;
            if (h->verbose && !h->said_pipe) {
                std::fprintf(stderr, "E2<GEO>: pipelined launch, %d resident workgroups over %d tiles of %d elements%s\n", npers, ntiles, tile, evp ? " (evp instantiation)" : "");
                h->said_pipe = true;
            }
            hipLaunchKernelGGL(kp, dim3(npers + nbf_p + nsf_p), dim3(tile), 0, h->stream, h->d_p, h->d_vt, h->d_clk, h->ne,
                               ntiles, npers, h->conn, xt_now, h->ntmp, mat_data(h), h->volume, h->volume_old,
                               h->stress, h->strain, h->strain_rate, h->plstrain, h->delta_plstrain, h->viscosity, h->dpressure,
                               h->etmp2, count, (nbf || nsf) ? h->nbcf : 0, h->bcf_elem, h->bcf_facet, h->bcf_kind, h->bcf_val, h->bcf_tmp, rp);
        } else
        hipLaunchKernelGGL(k, dim3(nblk8(e_all) + nbf + nsf), dim3(DES_BLOCK), 0, h->stream, h->d_p, h->d_vt, h->d_clk, h->ne,
                           e_begin, e_count, e_begin2, e_count2, nblk(e_all), h->conn, xt_now, h->ntmp, mat_data(h), h->volume, h->volume_old, h->stress,
                           h->strain, h->strain_rate, h->plstrain, h->delta_plstrain, h->viscosity, h->dpressure,
                           h->etmp2, h->defer_list, count,
                           nblk8(e_all), (nbf || nsf) ? h->nbcf : 0, h->bcf_elem, h->bcf_facet, h->bcf_kind, h->bcf_val, h->bcf_tmp, rp);
    }
    if (defer && !exp_skip("e2r")) {
        Launch l(h, K_E2R);
        auto k = h->portable_libm ? (geo ? E2_return_mapping<desk::MathPortable, 1> : E2_return_mapping<desk::MathPortable, 0>)
                                  : (geo ? E2_return_mapping<desk::MathOcml, 1> : E2_return_mapping<desk::MathOcml, 0>);
        hipLaunchKernelGGL(k, dim3(std::min(nblk(e_count + e_count2), DES_E2R_GRID)), dim3(DES_BLOCK), 0, h->stream, h->d_p, h->d_vt, h->d_clk, h->ne,
                           h->conn, xt_now, h->ntmp, mat_data(h), h->volume, h->volume_old, h->stress,
                           h->strain, h->strain_rate, h->plstrain, h->delta_plstrain, h->viscosity, h->dpressure,
                           h->etmp2, h->defer_list, count, rp);
    }
    if (whole) {                                           // (PART_DEEP is followed by PART_REST)
        if (h->rot_pending || geo) h->ddp_live = false;    // ... and has folded the pending NMD increments in
        h->rot_pending = false; h->e2geo_next = false; h->e2_fresh = false;
    }
}

void launch_n2(des_dev *h)
{
    static const char *en2 = des_env::get("DES_PATCH_N2");
    if (h->patch && !(en2 && en2[0] == '0') && h->patch_max_pe <= DES_PATCH_PE) {
        // the gather over node-block patches (passes/en2.hpp): one etmp2 fetch per patch element
        Launch l(h, K_EN2);
        const dim3 grid((h->patch_nb + 7) / 8 * 8);
        if (h->patch_max_inc <= 1664 && h->patch_max_pe <= 896)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(EN2_nmd_gather<1664, 896>), grid, dim3(256), 0, h->stream, h->nn, h->patch_nb, h->patch_npb,
                               h->pe_ptr, h->pe_pack, h->sup_idx, h->etmp2, h->volume_n, h->ntmp);
        else
            hipLaunchKernelGGL(HIP_KERNEL_NAME(EN2_nmd_gather<DES_PATCH_INC, DES_PATCH_PE>), grid, dim3(256), 0, h->stream, h->nn, h->patch_nb,
                               h->patch_npb, h->pe_ptr, h->pe_pack, h->sup_idx, h->etmp2, h->volume_n, h->ntmp);
        return;
    }
    Launch l(h, K_N2);
    // one double per incidence: the lightest gather, best with at most 128 nodes per workgroup
    // even on large meshes (1.1M tets: 22.7 us at 256, 17.6 at 128)
    const int npb2 = std::min(h->npb, 128), nb2 = (h->nn + npb2 - 1) / npb2;
    hipLaunchKernelGGL(N2_nmd_gather, dim3((nb2 + 7) / 8 * 8), dim3(DES_BLOCK), 0, h->stream, 0, h->nn, nb2, npb2, h->sup_idx,
                       h->sup_pack, h->etmp2, h->volume_n, h->ntmp);
}

// `facets`: this launch also carries the stress-bc facet workgroups (once per step)
void launch_e3(des_dev *h, int e_begin = 0, int e_count = -1, bool facets = true)
{
    if (e_count < 0) e_count = h->ne;
    const int nbe8 = nblk8(e_count), nbf = facets ? nblk(h->nbcf) : 0;
    if (nbe8 + nbf == 0) return;
    Launch l(h, K_E3);
    hipLaunchKernelGGL(E3_nmd_force, dim3(nbe8 + nbf), dim3(DES_BLOCK), 0, h->stream, h->d_p,
                       (int)(h->p.is_using_mixed_stress && !h->iso && !h->in_pt), h->ne, e_begin, e_count,
                       nblk(e_count), nbe8,
                       h->conn, h->xt, h->ntmp, mat_data(h), h->volume, h->dpressure, h->stress, h->ftmp,
                       facets ? h->nbcf : 0, h->bcf_elem, h->bcf_facet, h->bcf_kind, h->bcf_val, h->bcf_tmp);
}

void launch_n3(des_dev *h)
{
    h->res_nb = node_blocks(h);
    Launch l(h, K_N3);
    const int nown = h->o1 - h->o0;
    hipLaunchKernelGGL(N3_force_velocity_coord, dim3(node_grid(h)), dim3(DES_BLOCK), 0, h->stream, h->d_p, h->d_clk, h->o0, h->o1,
                       h->nn, h->nn_global, node_blocks(h), h->npb, h->sup_idx, h->sup_pack, h->bcflag, h->ftmp, h->bc_mask, h->bcn_idx,
                       h->bcn_ent, h->bcf_tmp, h->coord0, h->ymass, h->bnormals, h->edge_vec, h->edge_slot, h->xt, h->vm,
                       h->force, h->fres, h->res_part);
}

// N1 as a pass over node-block patches (passes/en1.hpp): the element terms are recomputed from the
// nodal records instead of being stored by E1 and gathered
void launch_en1(des_dev *h, int part = PART_ALL)
{
    // node blocks of this launch: all, the ones made of deep nodes only, or the rest around them
    const int nb = h->patch_nb, npb = h->patch_npb;
    int b0 = 0, c0 = nb, b1 = 0, c1 = 0;
    if (part != PART_ALL) {
        const int d0 = std::min(nb, (h->n_deep0 + npb - 1) / npb), d1 = std::max(d0, h->n_deep1 / npb);
        if (part == PART_DEEP) { b0 = d0; c0 = d1 - d0; }
        else                   { c0 = d0; b1 = d1; c1 = nb - d1; }
    }
    if (c0 + c1 > 0) {
        Launch l(h, K_EN1);
        void (*k)(const des_params *, DevClock *, int, int, int, int, int, int, int, int, int, int, int, const int *, const ulonglong2 *,
                  const int *, const int *, const int *, const unsigned *, const MatData, const double *, const d4 *, d4 *, d4 *,
                  double *, double *, double *, const SurfPending, const int *, const int *);
        // the surface step of the step before, if its S2 / S3 launches were left out (s2_defer_ok)
        SurfPending sp = {nullptr, nullptr, nullptr, nullptr, nullptr};
        if (h->s2_pending) {
            sp.tfan = h->tfan; sp.ssup_nodes = h->ssup_nodes;
            sp.dh = h->dh; sp.dhacc = h->dhacc; sp.dh_n = h->dh_n;
            if (part != PART_DEEP) {                // (PART_DEEP is followed by PART_REST, which needs it too)
                h->s2_pending = false;
                h->edv_pending = true;              // the edvacc_surf part rides in the next stress update
            }
        }
        const bool cm = h->const_mass;
        static const char *tenv = des_env::get("DES_EN1_THREADS");
        const int T = (tenv && std::atoi(tenv) == 512) ? 512 : 256;
        const bool th = h->p.has_thermal_diffusion != 0;       // the common launch has kernels of its own (passes/en1.hpp)
#define DES_EN1_PICK(TT) (th ? (cm ? EN1_mass_temperature_dvoldt<TT, 1, 1> : EN1_mass_temperature_dvoldt<TT, 0, 1>) \
                             : (cm ? EN1_mass_temperature_dvoldt<TT, 1, 0> : EN1_mass_temperature_dvoldt<TT, 0, 0>))
        k = T == 512 ? DES_EN1_PICK(512) : DES_EN1_PICK(256);
#undef DES_EN1_PICK
        // LDS for the mesh's largest block (dynamic: the workgroups per CU follow from the mesh and the block size)
        const int ci = patch_cap(h->patch_max_inc), cn = patch_cap(h->patch_max_pn), ce = patch_cap(h->patch_max_pe);
        const size_t lds = en1_lds_bytes(ci, cn, ce, cm);
        if (lds > 65536) hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k, dim3((c0 + c1 + 7) / 8 * 8), dim3(T), lds, h->stream, h->d_p, h->d_clk, h->nn, h->ne, b0, c0, b1, c1,
                           (int)(part != PART_REST), h->patch_npb, ci, cn, ce, h->pe_ptr, h->pe_pack, h->pn_ptr, h->pn_id, h->sup_idx, h->bcflag,
                           mat_data(h), h->radiogenic_zero ? (const double *)nullptr : h->radiogenic, h->xt, h->xt_alt, h->vm, h->volume_n,
                           h->tmass, h->ntmp, sp, h->pb_top, (b0 == 0 && c0 == h->patch_nb && c1 == 0) ? h->bperm : (const int *)nullptr);
    }
    if (part != PART_DEEP) std::swap(h->xt, h->xt_alt);      // EN1 wrote the records with the new temperatures there
}

// E3 + N3 as one pass over node-block patches (passes/en3.hpp); the stress-bc facet terms, which
// rode in E3's launch, ride in E2's instead (launch_e2)
void launch_en3(des_dev *h)
{
    h->res_nb = h->patch_nb;
    {
        Launch l(h, K_EN3);
        // LDS for the mesh's largest block (dynamic) -> as many workgroups per CU as the mesh allows
        void (*k)(const des_params *, const DevClock *, int, int, int, int, int, int, int, int, int, int, const int *, const ulonglong2 *, const int *, const int *, const int *, const unsigned *, const double *, const MatData, const double *,
                  const double *, const double *, double *, unsigned, const int *, const int *, const double *, const double *, const double *,
                  const double *, const double *, const int *, const d4 *, d4 *, d4 *, double *, double *, double *, int);
        const int T = h->patch_threads;
        const bool nmd = h->p.is_using_mixed_stress && !h->iso && !h->in_pt;
        const bool known = nmd && h->p.gravity != 0;           // the common launch has kernels of its own (passes/en3.hpp)
        k = known ? (T == 512 ? EN3_force_nodes<512, 1> : EN3_force_nodes<256, 1>) : (T == 512 ? EN3_force_nodes<512, 0> : EN3_force_nodes<256, 0>);
        const int ci = patch_cap(h->patch_max_inc), cn = patch_cap(h->patch_max_pn);
        const size_t lds = en3_lds_bytes(ci, cn);
        if (lds > 65536) hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k, dim3((h->patch_nb + 7) / 8 * 8), dim3(T), lds, h->stream, h->d_p, h->d_clk,
                           (int)nmd, h->o0, h->o1, h->nn, h->nn_global, h->ne, h->patch_nb, h->patch_npb, ci, cn,
                           h->pe_ptr, h->pe_pack, h->pn_ptr, h->pn_id, h->sup_idx, h->bcflag, h->ntmp, mat_data(h),
                           h->volume, h->dpressure, h->stress, h->ddp, h->bc_mask, h->bcn_idx, h->bcn_ent, h->bcf_tmp, h->coord0, h->ymass,
                           h->bnormals, h->edge_vec, h->edge_slot, h->xt, h->xt_alt, h->vm, h->force, h->fres, h->res_part, (h->e2_elide && !h->p.has_PT && !h->in_pt && !h->iso) ? 0 : 1);
    }
    std::swap(h->xt, h->xt_alt);               // the records EN3 wrote are the current ones from here on
    if (h->p.is_using_mixed_stress && !h->iso && !h->in_pt) h->ddp_live = true;      // ddp[] now holds this step's NMD increments
}

// update_force + everything nodal that follows it in a step
void launch_force_pass(des_dev *h)
{
    if (h->patch) launch_en3(h);
    else { launch_e3(h); launch_n3(h); }
}

// surface_processes (bc.cxx:1709-1872) as far as the device state is concerned, first part:
// diffusion of the owned surface nodes
void launch_s2(des_dev *h, long long step_no)
{
    if (!(h->p.has_moving_mesh || h->iso)) return;          // surface_processes is part of update_mesh
    if (h->ntop > 0 && !exp_skip("s2")) {
        Launch l(h, K_S2);
        hipLaunchKernelGGL(k_s2, dim3(nblk(h->ntop)), dim3(DES_BLOCK), 0, h->stream, h->d_p, h->d_clk, h->ntop,
                           (int)(h->p.surface_process_option == 1), h->top_nodes, h->ssup_idx, h->ssup_nodes, h->conn_surf,
                           h->etop, h->xt, h->o0, h->o1, h->dh, h->dhacc, h->znew, h->dh_n);
    }
    if (h->ntop > 0 && step_no != 0 && step_no % h->p.quality_check_step_interval == 0)
        hipLaunchKernelGGL(k_dhacc_reset, dim3(nblk(h->ntop)), dim3(DES_BLOCK), 0, h->stream, h->ntop,
                           h->top_nodes, h->dhacc);
}

// commit of the new surface heights / edvacc_surf / end-of-step scalars (k_s3_finalize)
void launch_s3(des_dev *h, bool commit, bool edvacc, bool finalize)
{
    if (exp_skip("s3")) return;
    Launch l(h, K_S3);
    const bool surf = (h->p.has_moving_mesh || h->iso) && h->ntop > 0;
    const int nsb = (edvacc && surface_diffusion_on(h)) ? nblk(h->etop) : 0;
    const int nzb = (commit && surf) ? nblk(h->ntop) : 0;
    const int nown = h->o1 - h->o0;
    hipLaunchKernelGGL(k_s3_finalize, dim3(nsb + nzb + 1), dim3(DES_BLOCK), 0, h->stream, h->d_clk, h->etop, nsb,
                       h->conn_surf, h->xt, h->dh_n, h->edvacc, h->res_part, h->res_nb, h->ntop, nzb, h->top_nodes,
                       h->znew, h->o0, h->o1, finalize ? ((h->p.has_moving_mesh || h->iso) ? 1 : 2) : 0);
}
