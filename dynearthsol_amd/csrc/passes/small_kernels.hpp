// passes/small_kernels.hpp -- Ghost-region pack / unpack and the small service kernels (dt, NaN count, mesh quality, libm check).
// Part of the single translation unit des_dev.hip (included inside namespace des_hip, after
// DevClock / struct des_dev); not a stand-alone header.

// ---- ghost-region exchange (des_halo, des_params.h) ------------------------------------
// State of the listed nodes {x,y,z,vx,vy,vz,T,dh} and elements {stress, strain, plstrain} to /
// from a message buffer; off[i] = position (in doubles) of item i's record in the buffer, so one
// launch fills the messages of all neighbours (a message = node records, then element records).
__global__ void __launch_bounds__(DES_BLOCK)
k_state_pack(int nnodes, const int *__restrict__ nidx, const int *__restrict__ noff,
             int nelems, const int *__restrict__ eidx, const int *__restrict__ eoff,
             const d4 *__restrict__ xt, const d4 *__restrict__ vm, const double *__restrict__ dh_n,
             const double *__restrict__ stress, const double *__restrict__ ddp, const double *__restrict__ strain,
             const double *__restrict__ plstrain, int ne, double *__restrict__ buf)
{
    const int i = blockIdx.x * DES_BLOCK + threadIdx.x;
    if (i < nnodes) {
        const int k = nidx[i];
        const d4 x = xt[k], v = vm[k];
        double *b = buf + noff[i];
        b[0] = x.x; b[1] = x.y; b[2] = x.z; b[3] = v.x; b[4] = v.y; b[5] = v.z; b[6] = x.w; b[7] = dh_n[k];
    } else if (i < nnodes + nelems) {
        const int j = i - nnodes, e = eidx[j];
        double *b = buf + eoff[j];
        for (int c = 0; c < 6; ++c) { b[c] = stress[(size_t)c*ne + e]; b[6 + c] = strain[(size_t)c*ne + e]; }
        // EN3 leaves the NMD increment of the diagonal for the next E1 (passes/en3.hpp): what is sent
        // is the stress that E1 will see
        if (ddp) { const double d = ddp[e]; if (d != 0.0) { b[0] += d; b[1] += d; b[2] += d; } }
        b[12] = plstrain[e];
    }
}

__global__ void __launch_bounds__(DES_BLOCK)
k_state_unpack(int nnodes, const int *__restrict__ nidx, const int *__restrict__ noff,
               int nelems, const int *__restrict__ eidx, const int *__restrict__ eoff,
               d4 *__restrict__ xt, d4 *__restrict__ vm, double *__restrict__ dh_n,
               double *__restrict__ stress, double *__restrict__ ddp, double *__restrict__ strain, double *__restrict__ plstrain,
               int ne, const double *__restrict__ buf)
{
    const int i = blockIdx.x * DES_BLOCK + threadIdx.x;
    if (i < nnodes) {
        const int k = nidx[i];
        const double *b = buf + noff[i];
        d4 x, v = vm[k];                                   // the nodal mass stays this rank's own
        x.x = b[0]; x.y = b[1]; x.z = b[2]; x.w = b[6];
        v.x = b[3]; v.y = b[4]; v.z = b[5];
        xt[k] = x; vm[k] = v; dh_n[k] = b[7];
    } else if (i < nnodes + nelems) {
        const int j = i - nnodes, e = eidx[j];
        const double *b = buf + eoff[j];
        for (int c = 0; c < 6; ++c) { stress[(size_t)c*ne + e] = b[c]; strain[(size_t)c*ne + e] = b[6 + c]; }
        if (ddp) ddp[e] = 0.0;                             // the received stress already holds its increment
        plstrain[e] = b[12];
    }
}

// compute_dt partials of this rank, all arranged for a MIN reduction across ranks
// (the partial slots are about to run out: fold what is there into clk->r_*)
__global__ void __launch_bounds__(DES_BLOCK)
k_dt_fold(DevClock *clk, const double *__restrict__ dt_part, int dt_cap, int dt_count)
{
    dt_reduce_partials(clk, dt_part, dt_cap, dt_count);
}

__global__ void __launch_bounds__(DES_BLOCK)
k_dt_pack(DevClock *clk, double *red, const double *__restrict__ dt_part, int dt_cap, int dt_count)
{
    dt_reduce_partials(clk, dt_part, dt_cap, dt_count);
    if (threadIdx.x != 0) return;
    red[0] = clk->r_minl; red[1] = clk->r_dt_maxwell; red[2] = clk->r_dt_diffusion;
    red[3] = clk->r_global_dt_min; red[4] = -clk->r_max_vem; red[5] = -clk->max_surf_vel;
}

__global__ void k_dhacc_reset(int ntop, const int *__restrict__ top_nodes, double *__restrict__ dhacc)
{
    const int i = blockIdx.x * DES_BLOCK + threadIdx.x;
    if (i < ntop) dhacc[top_nodes[i]] = 0.;
}

// check_nan (utils.hpp:323-394)
// des_dev_libm_eval: one portable-libm function over an array (diagnostic entry)
__global__ void k_libm_eval(int fn, long long n, const double *__restrict__ x, const double *__restrict__ y,
                            double *__restrict__ out)
{
    deslibm::lds_stage();
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double a = x[i], b = y ? y[i] : 0.0;
    double r;
    switch (fn) {
    case DES_LIBM_POW:   r = deslibm::pow(a, b); break;
    case DES_LIBM_EXP:   r = deslibm::exp(a); break;
    case DES_LIBM_SIN:   r = deslibm::sin(a); break;
    case DES_LIBM_COS:   r = deslibm::cos(a); break;
    case DES_LIBM_TAN:   r = deslibm::tan(a); break;
    case DES_LIBM_SINCOS_S: { double c_; deslibm::sincos(a, &r, &c_); break; }
    case DES_LIBM_SINCOS_C: { double s_; deslibm::sincos(a, &s_, &r); break; }
    default:             r = deslibm::atan2(a, b); break;
    }
    out[i] = r;
}

// des_dev_eigen_eval: the device build of the reference's 3x3 solvers over an array of tensors
// {A00, A11, A22, A01, A02, A12}.  fn 0 dsyevc3, 1 dsyevh3, 2 dsyevq3; q row-major, eigenvectors
// in columns as in 3x3-C; branch[i] = 1 where dsyevh3 handed over to dsyevq3 (dsyevh3.c:152, 177).
template <class M>
__global__ void k_eigen_eval(int fn, long long n, const double *__restrict__ a, double *__restrict__ w,
                             double *__restrict__ q, int *__restrict__ branch)
{
    M::stage_begin();
    M::stage_end();
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double A[6], W[3] = {0, 0, 0}, Q[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (int k = 0; k < 6; ++k) A[k] = a[i*6 + k];
    int fb = 0;
    if (fn == 0)      desk::dsyevc3<M>(A, W);
    else if (fn == 1) desk::dsyevh3<M>(A, Q, W, &fb);
    else              fb = desk::dsyevq3(A, Q, W);
    for (int k = 0; k < 3; ++k) w[i*3 + k] = W[k];
    if (fn != 0) for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) q[i*9 + r*3 + c] = Q[r][c];
    branch[i] = fb;
}

// des_dev_elasto_plastic_eval: one elasto_plastic call (rheology.cxx:312-484) per item, the very
// function E2 runs.  props[i] = {bulkm, shearm, amc, anphi, anpsi, hardn, ten_max}.
template <class M>
__global__ void k_elasto_plastic_eval(long long n, const double *__restrict__ props, const double *__restrict__ de,
                                      double *__restrict__ s, double *__restrict__ depls, int *__restrict__ mode)
{
    M::stage_begin();
    M::stage_end();
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double *pp = props + i*7;
    double d[6], t[6];
    for (int k = 0; k < 6; ++k) { d[k] = de[i*6 + k]; t[k] = s[i*6 + k]; }
    bool past = false;
    int fm = 0;
    const double dp = desk::elasto_plastic<M, 0>(pp[0], pp[1], pp[2], pp[3], pp[4], pp[5], pp[6], d, t, &past, &fm);
    for (int k = 0; k < 6; ++k) s[i*6 + k] = t[k];
    depls[i] = dp;
    mode[i] = fm + (past ? 1000 : 0);
}

// Streaming device copy, 16 bytes per lane with UNROLL loads in flight per lane before the stores,
// grid-stride: the achievable-HBM yardstick SURVEY.md 8(d) asks for next to the 8 TB/s nominal
// peak.  NT: non-temporal loads and stores (no reuse: keeps the copy out of L2 / MALL).
template <int UNROLL, bool NT>
__global__ void __launch_bounds__(256) k_copy16(const double2 *__restrict__ src, double2 *__restrict__ dst, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n; i += UNROLL * stride) {
        double2 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if (NT) { v[u].x = __builtin_nontemporal_load(&src[i + u * stride].x); v[u].y = __builtin_nontemporal_load(&src[i + u * stride].y); }
            else v[u] = src[i + u * stride];
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if (NT) { __builtin_nontemporal_store(v[u].x, &dst[i + u * stride].x); __builtin_nontemporal_store(v[u].y, &dst[i + u * stride].y); }
            else dst[i + u * stride] = v[u];
        }
    }
    for (; i < n; i += stride) dst[i] = src[i];
}

// The same copy with ONE 16-byte item per lane and as many workgroups as it takes (no grid-stride loop): the launch
// shape that reaches the guide's 6.2-6.3 TB/s from HBM on this machine (tools/plane_stream_bench.hip measured 6.19 with it
// where the grid-stride shapes above stop at 4.6-4.9).
__global__ void __launch_bounds__(256) k_copy16_flat(const double2 *__restrict__ src, double2 *__restrict__ dst, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

// des_dev_access_bench: the engine's memory access shapes on a KNOWN byte count, to calibrate the
// rocprofv3 FETCH_SIZE / WRITE_SIZE counters (MI355X_MICROARCH.md calibrates them for 16 B per lane
// streaming only).  PATTERN 0: 16 B/lane stream copy; 1: 8 B/lane stream copy (one SoA plane);
// 2: 32-B record gather through a permutation (every record once; the {x,y,z,T} node records),
// 8 B/lane result stream; 3: 8-B gather through a permutation (one double per incidence), 8 B/lane result.
template <int PATTERN>
__global__ void __launch_bounds__(256) k_access(const double *__restrict__ src, double *__restrict__ dst,
                                                const int *__restrict__ perm, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (PATTERN == 0)      ((double2 *)dst)[i] = ((const double2 *)src)[i];
    else if (PATTERN == 1) dst[i] = src[i];
    else if (PATTERN == 2) { const d4 r = ((const d4 *)src)[perm[i]]; dst[i] = (r.x + r.y) + (r.z + r.w); }
    else                   dst[i] = src[perm[i]];
}

// nr planes read + nw planes written per element, 8 B per lane and plane, one element per lane, no arithmetic but the sum
// that ties the loads to the stores: what a pass with the stress update's MEMORY SHAPE reaches when it only moves its bytes
// (tools/plane_stream_bench.hip is the stand-alone form).  Shapes: 18 + 15 (E2<GEO>, interior step: 141 B read, 120 B written)
// and 12 + 9 (the 2-D k2_stress<M, 2>: 96 + 72 B).
template <int NR, int NW>
__global__ void __launch_bounds__(256) k_plane_stream(const double *__restrict__ src, double *__restrict__ dst, long long ne)
{
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= ne) return;
    double v[NR];
#pragma unroll
    for (int k = 0; k < NR; ++k) v[k] = src[(size_t)k * ne + e];
    double s = 0;
#pragma unroll
    for (int k = 0; k < NR; ++k) s += v[k];
#pragma unroll
    for (int k = 0; k < NW; ++k) dst[(size_t)k * ne + e] = s + v[k % NR];
}

__global__ void k_count_nan(const double *a, long long n, unsigned long long *count)
{
    long long i = (long long)blockIdx.x * DES_BLOCK + threadIdx.x;
    unsigned long long c = 0;
    for (; i < n; i += (long long)gridDim.x * DES_BLOCK) c += isnan(a[i]) ? 1 : 0;
    if (c) atomicAdd(count, c);
}

// bad_mesh_quality reductions (remeshing.cxx:2752-2866).  slots: [0] min quality (double bits),
// then ints: [2] first tiny element, [3] first distorted bottom node, [4] first worst element
__device__ __forceinline__ double elem_quality3(const int4 cn, const d4 *__restrict__ xt, double vol)
{
    const d4 a = xt[cn.x], b = xt[cn.y], c = xt[cn.z], d = xt[cn.w];
    const double normalization_factor = 216 * sqrt(3.0);
    const double area_sum = (desk::tri_area(a, b, c) + desk::tri_area(a, b, d) +
                             desk::tri_area(c, d, a) + desk::tri_area(c, d, b));
    return normalization_factor * vol * vol / (area_sum * area_sum * area_sum);
}

__global__ void k_quality_a(int ne, int nn, const int4 *__restrict__ conn, const d4 *__restrict__ xt,
                            const double *__restrict__ volume, const unsigned *__restrict__ bcflag,
                            double smallest_vol, double bottom, double bottom_dist, double *qmin, int *islot,
                            const int *__restrict__ n_id, const int *__restrict__ e_id)
{
    // n_id / e_id: the caller's index of a device index ("first" means first in the caller's order)
    const int i = blockIdx.x * DES_BLOCK + threadIdx.x;
    double q = 1.0;
    if (i < ne) {
        const double vol = volume[i];
        if (vol < smallest_vol) atomicMin(&islot[0], e_id ? e_id[i] : i);
        q = fmin(q, elem_quality3(conn[i], xt, vol));
    }
    if (i < nn && bottom_dist >= 0 && (bcflag[i] & (1u << 4)))            // is_bottom: BOUNDZ0
        if (fabs(xt[i].z - bottom) > bottom_dist) atomicMin(&islot[1], n_id ? n_id[i] : i);
    q = desk::wave_min(q);
    if ((threadIdx.x & 63) == 0 && q < 1.0) desk::atomic_min_double(qmin, q);
}

__global__ void k_quality_b(int ne, const int4 *__restrict__ conn, const d4 *__restrict__ xt,
                            const double *__restrict__ volume, const double *qmin, int *islot,
                            const int *__restrict__ e_id)
{
    const int e = blockIdx.x * DES_BLOCK + threadIdx.x;
    if (e >= ne) return;
    const double q = elem_quality3(conn[e], xt, volume[e]);
    if (q < 1.0 && q == *qmin) atomicMin(&islot[2], e_id ? e_id[e] : e);
}
