// passes/surface.hpp -- Surface kernels S2 / S3.
// Part of the single translation unit des_dev.hip (included inside namespace des_hip, after
// DevClock / struct des_dev); not a stand-alone header.

// ---- surface processes -----------------------------------------------------------
// simple_diffusion (bc.cxx:954-1107) + coordinate/dhacc update (bc.cxx:1770-1777), one thread
// per surface node.  The facet quantities of bc.cxx:954-1039 (projected area, slope term of the
// facet's local node) are recomputed by every node that touches the facet -- ~6x redundant
// work on O(surface) data, in exchange for one launch and no facet temporaries; the values are
// the same deterministic expressions, so the sums are bit-identical.
#ifndef DES_S2_BATCH
#define DES_S2_BATCH 4
#endif
// dh of surface node `n` (top-list position i): -diffusivity * dt * (sum of slope terms) / (sum of projected areas) over
// the node's facet fan, bc.cxx:954-1107.  One source for k_s2 and for EN1's deferred surface step (passes/en1.hpp): the
// same expressions in the same order, so the same bits.
// the facets [jb, je) of the node's fan; the node ids of 2 x BATCH facets are requested together, their records
// in two rounds of BATCH facets (3 x BATCH records of 32 B in registers at a time)
template <int BATCH>
__device__ __forceinline__ double s2_node_dh_range(int n, int jb, int je, const int *__restrict__ ssup_nodes,
                                                   const d4 *__restrict__ xt_in, double surface_diffusivity, double dt)
{
    double d = 0.;
    double total_dx = 0., total_slope = 0.;
    // all facet node ids of a double batch, then all node records of each half are requested before the first is
    // used, so a double batch costs three memory latencies instead of two per facet; the sums below still run in list order
    for (int j0 = jb; j0 < je; j0 += 2 * BATCH) {
        int nd[2 * BATCH][3];
        // ssup_nodes[3k + m] = conn_surf[m*etop + ssup_arr[k]], flattened once at create
        // (one dependent look-up less on this latency-bound kernel)
#pragma unroll
        for (int u = 0; u < 2 * BATCH; ++u)
            for (int m = 0; m < 3; ++m) nd[u][m] = (j0 + u < je) ? ssup_nodes[3 * (size_t)(j0 + u) + m] : n;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if (half == 1 && j0 + BATCH >= je) break;
            d4 cf[BATCH][3];
#pragma unroll
            for (int u = 0; u < BATCH; ++u)
                for (int m = 0; m < 3; ++m) cf[u][m] = xt_in[nd[half * BATCH + u][m]];
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                if (j0 + half * BATCH + u >= je) continue;
                const d4 *c = cf[u];
                const int *ndu = nd[half * BATCH + u];
                double x01 = c[1].x - c[0].x, y01 = c[1].y - c[0].y;
                double x02 = c[2].x - c[0].x, y02 = c[2].y - c[0].y;
                double projected_area = 0.5 * (x01*y02 - y01*x02);
                total_dx += projected_area;
                double shp2dx[3], shp2dy[3];
                double iv = 1 / (2 * projected_area);
                shp2dx[0] = iv * (c[1].y - c[2].y);
                shp2dx[1] = iv * (c[2].y - c[0].y);
                shp2dx[2] = iv * (c[0].y - c[1].y);
                shp2dy[0] = iv * (c[2].x - c[1].x);
                shp2dy[1] = iv * (c[0].x - c[2].x);
                shp2dy[2] = iv * (c[1].x - c[0].x);
                const double zz[3] = {c[0].z, c[1].z, c[2].z};
                for (int m = 0; m < 3; ++m) {
                    if (ndu[m] == n) {
                        double slope = 0;
                        for (int q = 0; q < 3; q++)
                            slope += (shp2dx[m] * shp2dx[q] + shp2dy[m] * shp2dy[q]) * zz[q];
                        total_slope += slope * projected_area;
                        break;
                    }
                }
            }
        }
    }
    double conv = surface_diffusivity * dt * total_slope / total_dx;
    d -= conv;
    return d;
}

template <int BATCH>
__device__ __forceinline__ double s2_node_dh(int i, int n, const int *__restrict__ ssup_idx, const int *__restrict__ ssup_nodes,
                                             const d4 *__restrict__ xt_in, double surface_diffusivity, double dt)
{
    return s2_node_dh_range<BATCH>(n, ssup_idx[i], ssup_idx[i+1], ssup_nodes, xt_in, surface_diffusivity, dt);
}

__global__ void __launch_bounds__(DES_BLOCK)
k_s2(const des_params *__restrict__ p, DevClock *__restrict__ clk, int ntop, int diffuse,
     const int *__restrict__ top_nodes, const int *__restrict__ ssup_idx, const int *__restrict__ ssup_nodes,
     const int *__restrict__ conn_surf, int etop, const d4 *__restrict__ xt_in, int o0, int o1,
     double *__restrict__ dh, double *__restrict__ dhacc, double *__restrict__ znew, double *__restrict__ dh_n)
{
    const int i = blockIdx.x * DES_BLOCK + threadIdx.x;
    double d = 0.;
    const int n = (i < ntop) ? top_nodes[i] : -1;
    if (n >= 0) {                               // every local surface node; [o0, o1) = the owned ones
        // (requested with the first batch of facets, not behind the last one)
        const double z_old = xt_in[n].z, dhacc_old = dhacc[n];
        if (diffuse) d = s2_node_dh<DES_S2_BATCH>(i, n, ssup_idx, ssup_nodes, xt_in, p->surface_diffusivity, clk->dt);
        dh[i] = d;
        // neighbours still need this node's OLD height: the new one goes to a side buffer and
        // is committed by the next launch (k_s3_finalize)
        znew[i] = z_old + d;
        dhacc[n] = dhacc_old + d;
        dh_n[n] = d;
    }
    // max |dh| (bc.cxx:1811-1821); max is order-independent
    __shared__ double red[DES_BLOCK / 64];
    double m = desk::wave_max((n >= o0 && n < o1) ? fabs(d) : 0.0);      // owned nodes only: ghosts may be stale
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < DES_BLOCK / 64; ++k) m = fmax(m, red[k]);
        if (m > 0) desk::atomic_max_double(&clk->maxdh, m);
    }
}

// edvacc_surf of surface facet i (bc.cxx:1784-1794): only x and y of its nodes enter, so it does not matter whether
// the new heights are committed yet
__device__ __forceinline__ void edvacc_facet(int i, int etop, const int *__restrict__ conn_surf, const d4 *__restrict__ xt,
                                             const double *__restrict__ dh_n, double *__restrict__ edvacc)
{
    const int na = conn_surf[i], nb = conn_surf[(size_t)etop + i], nc = conn_surf[(size_t)2*etop + i];
    double dh_e = 0.;
    dh_e += dh_n[na]; dh_e += dh_n[nb]; dh_e += dh_n[nc];
    const d4 a = xt[na], b = xt[nb], c = xt[nc];
    double ab0 = b.x - a.x, ab1 = b.y - a.y, ac0 = c.x - a.x, ac1 = c.y - a.y;
    double base = fabs(ab0*ac1 - ab1*ac0) / 2;           // triangle_area2d, geometry.cxx:59-73
    edvacc[i] += dh_e * base / 3;
}

// edvacc_surf update (bc.cxx:1784-1794), commit of the surface heights k_s2 computed
// (bc.cxx:1775), and -- in the last workgroup -- the end-of-step scalars: l2_residual
// (fields.cxx:721) and max_surf_vel (bc.cxx:1825).  The three kinds of workgroup do not depend on
// each other (the facet-area term only reads x and y); a decomposed run launches the commit
// before the surface halo exchange and the rest after it.
__global__ void __launch_bounds__(DES_BLOCK)
k_s3_finalize(DevClock *__restrict__ clk, int etop, int nsurf_blocks, const int *__restrict__ conn_surf,
              d4 *__restrict__ xt, const double *__restrict__ dh_n, double *__restrict__ edvacc,
              const double *__restrict__ res_part, int nres, int ntop, int nz_blocks,
              const int *__restrict__ top_nodes, const double *__restrict__ znew, int o0, int o1, int do_finalize)
{
    if ((int)blockIdx.x >= nsurf_blocks && (int)blockIdx.x < nsurf_blocks + nz_blocks) {
        const int i = ((int)blockIdx.x - nsurf_blocks) * DES_BLOCK + threadIdx.x;
        if (i < ntop) {
            xt[top_nodes[i]].z = znew[i];
        }
        return;
    }
    if ((int)blockIdx.x < nsurf_blocks) {
        const int i = blockIdx.x * DES_BLOCK + threadIdx.x;
        if (i < etop) edvacc_facet(i, etop, conn_surf, xt, dh_n, edvacc);
        return;
    }
    if (!do_finalize) return;
    __shared__ double red[DES_BLOCK];
    double t = 0;
    for (int i = threadIdx.x; i < nres; i += DES_BLOCK) t += res_part[i];
    red[threadIdx.x] = t;
    __syncthreads();
    for (int off = DES_BLOCK / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        clk->l2_sum = red[0];
        clk->l2_residual = sqrt(red[0]);
        if (do_finalize == 1) clk->max_surf_vel = clk->maxdh / clk->dt;     // part of surface_processes: moving mesh only
    }
}
