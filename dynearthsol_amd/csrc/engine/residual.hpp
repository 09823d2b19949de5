// engine/residual.hpp -- The partition-independent residual of the pseudo-transient loop (des_params.h: DES_RES_BLOCK).
// Part of the single translation unit des_dev.hip (included inside namespace des_hip, after engine/launch.hpp).
//
// calculate_residual_force (fields.cxx:700-722) is an OpenMP reduction in the reference: the order of the sum is open.  The
// per-step scalar des_scalars::l2_residual is a by-product of the force pass (one partial per node block, passes/n3.hpp) --
// good to the rounding of a tree sum.  Where a DECISION hangs on the value -- the convergence test of the pseudo-transient
// loop (dynearthsol.cxx:829-833), of initial_body_force_adjustment (:571-575) -- every rank of a decomposed run must take
// the same one, and a run on N ranks the one a single engine takes.  So there the sum has ONE association: per block of
// B = des_res_block(nn_global) consecutive GLOBAL node ids the nodes' terms one after the other in ascending id; the blocks
// in a fixed shape over the global block array (k_residual_final).  Slabs are cut at multiples of B, so a block has one
// owner: ranks fill their part of the array, the parts are put together (RCCL: a SUM over zero-filled arrays is exact,
// x + 0 = x; a group of engines: copies), and every rank reduces the same array.  The oracle does the same arithmetic
// (oracle/des_oracle.cpp: residual_blocks_local / residual_final): the same bits.

// one wavefront per block: lane i takes node B b + i's term, lane 0's running sum walks them in order
__global__ void __launch_bounds__(DES_BLOCK)
k_residual_blocks(int nown, int B, int nblocks, int nn, int nn_global, const int *__restrict__ gidx, const double *__restrict__ fres,
                  double *__restrict__ out)
{
    const int b = (int)(blockIdx.x * (DES_BLOCK / 64) + (threadIdx.x >> 6)), lane = (int)(threadIdx.x & 63);
    if (b >= nblocks) return;
    const double num = (double)nn_global * 3;
    const int i = b * B + lane;
    double t = 0.0;
    if (lane < B && i < nown) {
        const int n = gidx[i];
        const double f0 = fres[n], f1 = fres[(size_t)nn + n], f2 = fres[(size_t)2 * nn + n];
        t = f0 * f0 / num;
        t += f1 * f1 / num;
        t += f2 * f2 / num;
    }
    double s = 0.0;
    for (int k = 0; k < B; ++k) s += __shfl(t, k);          // (terms past the block's / the mesh's end are +0.0: x + 0 = x)
    if (lane == 0) out[b] = s;
}

// the fixed shape over the global block array: 256 strided serial sums, then a pairwise tree
__global__ void __launch_bounds__(DES_BLOCK)
k_residual_final(DevClock *__restrict__ clk, const double *__restrict__ blocks, int nb)
{
    __shared__ double red[DES_BLOCK];
    double t = 0;
    for (int i = threadIdx.x; i < nb; i += DES_BLOCK) t += blocks[i];
    red[threadIdx.x] = t;
    __syncthreads();
    for (int off = DES_BLOCK / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) { clk->l2_sum = red[0]; clk->l2_residual = sqrt(red[0]); }
}

// (re)built whenever the owned range is set: des_dev_create (the whole mesh) and des_dev_set_halo
int build_residual_blocks(des_dev *h)
{
    const int B = des_res_block(h->nn_global), nown = h->o1 - h->o0;
    h->res_b0 = h->g0 / B;
    h->res_nb_own = (nown + B - 1) / B;
    h->res_nb_global = (h->nn_global + B - 1) / B;
    if (h->g0 % B != 0 || h->res_b0 + h->res_nb_own > h->res_nb_global) {
        g_last_error = "des_halo::owned_global_begin is not a multiple of the residual's block size (des_params.h: des_res_block)";
        return DES_ERR_INTERNAL;
    }
    if (h->res_gidx) hipFree(h->res_gidx);
    if (h->res_blocks) hipFree(h->res_blocks);
    h->res_gidx = nullptr; h->res_blocks = nullptr;
    std::vector<int> gidx((size_t)nown);
    for (int i = 0; i < nown; ++i) gidx[i] = h->n_old2new.empty() ? h->o0 + i : h->n_old2new[(size_t)h->o0 + i];
    int rc;
    if ((rc = dev_alloc(h->res_gidx, (size_t)nown)) || (rc = dev_upload(h->res_gidx, gidx.data(), (size_t)nown, h->stream))) return rc;
    if ((rc = dev_alloc(h->res_blocks, (size_t)h->res_nb_global))) return rc;
    HIP_OK(hipMemsetAsync(h->res_blocks, 0, (size_t)h->res_nb_global * sizeof(double), h->stream));
    return DES_OK;
}

// this rank's block partials into their places of the (zero-filled) global array
void launch_residual_blocks(des_dev *h)
{
    const int B = des_res_block(h->nn_global);
    if (h->res_nb_own < h->res_nb_global) hipMemsetAsync(h->res_blocks, 0, (size_t)h->res_nb_global * sizeof(double), h->stream);
    hipLaunchKernelGGL(k_residual_blocks, dim3((h->res_nb_own + DES_BLOCK / 64 - 1) / (DES_BLOCK / 64)), dim3(DES_BLOCK), 0, h->stream,
                       h->o1 - h->o0, B, h->res_nb_own, h->nn, h->nn_global, h->res_gidx, h->fres, h->res_blocks + h->res_b0);
}

void launch_residual_final(des_dev *h)
{
    hipLaunchKernelGGL(k_residual_final, dim3(1), dim3(DES_BLOCK), 0, h->stream, h->d_clk, h->res_blocks, h->res_nb_global);
}

// the residual of the force_residual the engine holds, on the engine's stream: one engine, or a rank on RCCL
int residual_global(des_dev *h)
{
    launch_residual_blocks(h);
    if (h->res_nb_own < h->res_nb_global) {
        if (!h->comm || h->comm_size <= 1) { g_last_error = "the pseudo-transient loop on a decomposed mesh needs the ranks' residual partials: des_dev_step on RCCL, des_dev_step_group, or des_dev_phase + des_dev_residual_blocks / _set"; return DES_ERR_UNSUPPORTED; }
        const ncclResult_t r = ncclAllReduce(h->res_blocks, h->res_blocks, (size_t)h->res_nb_global, ncclDouble, ncclSum, h->comm, h->stream);
        if (r != ncclSuccess) { g_last_error = std::string("RCCL: ") + ncclGetErrorString(r); return DES_ERR_RESOURCE; }
    }
    launch_residual_final(h);
    return DES_OK;
}

// ... of the engines of a group (des_dev_step_group): the parts change hands through the host (this path serves tests and
// rehearsals, and the loop joins the streams once per iteration for its decision anyway)
int residual_global_group(des_dev **g, int n)
{
    std::vector<double> all((size_t)g[0]->res_nb_global, 0.0);
    for (int k = 0; k < n; ++k) {
        des_dev *h = g[k];
        hipSetDevice(h->device);
        launch_residual_blocks(h);
        HIP_OK(hipMemcpyAsync(all.data() + h->res_b0, h->res_blocks + h->res_b0, (size_t)h->res_nb_own * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIP_OK(hipStreamSynchronize(h->stream));
    }
    for (int k = 0; k < n; ++k) {
        des_dev *h = g[k];
        hipSetDevice(h->device);
        HIP_OK(hipMemcpyAsync(h->res_blocks, all.data(), all.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
        HIP_OK(hipStreamSynchronize(h->stream));            // (all[] is a local buffer)
        launch_residual_final(h);
    }
    return DES_OK;
}
