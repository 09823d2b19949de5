// engine/exchange.hpp -- The per-step ghost-region exchange through RCCL.
// Part of the single translation unit des_dev.hip (included inside namespace des_hip, after
// DevClock / struct des_dev); not a stand-alone header.

// ---- halo exchange through RCCL on the engine's stream ---------------------------

// The exchange of a step: one grouped send/recv per neighbour carrying the state of the whole
// ghost region, between a pack and an unpack launch, all on the engine's stream.
// `xs`: the stream the transfer and the unpack are issued on -- the engine's own, or (overlapped
// schedule) the side stream; the pack always runs on the engine's stream.
int exchange(des_dev *h, hipStream_t xs)
{
    if (h->nnbr == 0) return DES_OK;
    if (!h->comm) { g_last_error = "decomposed engine without a communicator: call des_dev_comm_init"; return DES_ERR_INTERNAL; }
    const int ns = h->send_ptr[h->nnbr], nes = h->esend_ptr[h->nnbr];
    const int nr = h->recv_ptr[h->nnbr], ner = h->erecv_ptr[h->nnbr];
    const bool side = xs != h->stream;
    {
        Launch l(h, K_EXCH);
        hipLaunchKernelGGL(k_state_pack, dim3(nblk(ns + nes)), dim3(DES_BLOCK), 0, h->stream, ns, h->d_send_idx, h->d_send_noff,
                           nes, h->d_esend_idx, h->d_send_eoff, h->xt, h->vm, h->dh_n, h->stress, pending_ddp(h), h->strain, h->plstrain,
                           h->ne, h->d_sendbuf);
        if (side) {
            // fork: everything up to and including the pack precedes the transfer.  (The pack stays on
            // the engine's stream: it reads the stress of elements the interior pass is about to rotate.)
            HIP_OK(hipEventRecord(h->ev_fork, h->stream));
            HIP_OK(hipStreamWaitEvent(xs, h->ev_fork, 0));
        }
    }
    Launch l(h, K_EXCH, xs);
    ncclGroupStart();
    for (int q = 0; q < h->nnbr; ++q) {
        ncclSend(h->d_sendbuf + h->send_off[q], (size_t)(h->send_off[q+1] - h->send_off[q]), ncclDouble,
                 h->nbr_rank[q], h->comm, xs);
        ncclRecv(h->d_recvbuf + h->recv_off[q], (size_t)(h->recv_off[q+1] - h->recv_off[q]), ncclDouble,
                 h->nbr_rank[q], h->comm, xs);
    }
    ncclResult_t r = ncclGroupEnd();
    if (r != ncclSuccess) { g_last_error = std::string("RCCL: ") + ncclGetErrorString(r); return DES_ERR_RESOURCE; }
    hipLaunchKernelGGL(k_state_unpack, dim3(nblk(nr + ner)), dim3(DES_BLOCK), 0, xs, nr, h->d_recv_idx, h->d_recv_noff,
                       ner, h->d_erecv_idx, h->d_recv_eoff, h->xt, h->vm, h->dh_n, h->stress, pending_ddp(h), h->strain, h->plstrain,
                       h->ne, h->d_recvbuf);
    if (side) HIP_OK(hipEventRecord(h->ev_join, xs));
    return DES_OK;
}
int exchange(des_dev *h) { return exchange(h, h->stream); }

// Overlapped schedule (opt-in, DES_OVERLAP=1): transfer + unpack run on a side stream while the
// engine's stream works through the end-of-step pass of the INTERIOR elements -- those whose four
// nodes are owned: they read owned nodes only, and what they write (their own stress / strain /
// volume / temporaries) is neither received nor -- once the pack is done -- sent.  The unpack
// writes ghost nodes and ghost elements only.  exchange_begin() forks after the surface heights
// are committed and the messages packed; exchange_join() makes the engine's stream wait for the
// unpack before anything touches the ghost region.
int exchange_begin(des_dev *h) { return exchange(h, h->comm_stream); }
int exchange_join(des_dev *h)
{
    HIP_OK(hipStreamWaitEvent(h->stream, h->ev_join, 0));
    return DES_OK;
}

// ---- the same exchange between engines of ONE process (des_dev_step_group) ---------------------
// No communicator: a neighbour's message buffer is plain device memory of this process, so the
// transfer is one device-to-device copy per neighbour, ordered by events instead of RCCL's
// rendezvous -- ev_packed (my messages are complete) and ev_taken (I have copied my neighbours'
// messages out: they may pack the next step's).  Same lists, same pack / unpack kernels, same
// place in the step as the RCCL path; the group issues every engine's pack before any take.
int exchange_local_pack(des_dev *h)
{
    if (h->nnbr == 0) return DES_OK;
    const int ns = h->send_ptr[h->nnbr], nes = h->esend_ptr[h->nnbr];
    for (int q = 0; q < h->nnbr; ++q)                      // (first step: never recorded, no wait)
        HIP_OK(hipStreamWaitEvent(h->stream, h->group[h->nbr_rank[q]]->ev_taken, 0));
    Launch l(h, K_EXCH);
    hipLaunchKernelGGL(k_state_pack, dim3(nblk(ns + nes)), dim3(DES_BLOCK), 0, h->stream, ns, h->d_send_idx, h->d_send_noff,
                       nes, h->d_esend_idx, h->d_send_eoff, h->xt, h->vm, h->dh_n, h->stress, pending_ddp(h), h->strain, h->plstrain,
                       h->ne, h->d_sendbuf);
    HIP_OK(hipEventRecord(h->ev_packed, h->stream));
    return DES_OK;
}

// `xs`: the engine's stream, or its side stream (overlapped schedule: exchange_join follows)
int exchange_local_take(des_dev *h, hipStream_t xs)
{
    if (h->nnbr == 0) return DES_OK;
    const int nr = h->recv_ptr[h->nnbr], ner = h->erecv_ptr[h->nnbr];
    const bool side = xs != h->stream;
    if (side) HIP_OK(hipStreamWaitEvent(xs, h->ev_packed, 0));          // fork: behind everything this engine has issued
    Launch l(h, K_EXCH, xs);
    for (int q = 0; q < h->nnbr; ++q) {
        des_dev *o = h->group[h->nbr_rank[q]];
        int qo = 0;
        while (o->nbr_rank[qo] != h->group_rank) ++qo;                  // (checked by des_dev_group_attach)
        HIP_OK(hipStreamWaitEvent(xs, o->ev_packed, 0));
        HIP_OK(hipMemcpyAsync(h->d_recvbuf + h->recv_off[q], o->d_sendbuf + o->send_off[qo],
                              (size_t)(h->recv_off[q+1] - h->recv_off[q]) * sizeof(double), hipMemcpyDeviceToDevice, xs));
    }
    hipLaunchKernelGGL(k_state_unpack, dim3(nblk(nr + ner)), dim3(DES_BLOCK), 0, xs, nr, h->d_recv_idx, h->d_recv_noff,
                       ner, h->d_erecv_idx, h->d_recv_eoff, h->xt, h->vm, h->dh_n, h->stress, pending_ddp(h), h->strain, h->plstrain,
                       h->ne, h->d_recvbuf);
    HIP_OK(hipEventRecord(h->ev_taken, xs));
    if (side) HIP_OK(hipEventRecord(h->ev_join, xs));
    return DES_OK;
}

// compute_dt across ranks: pack the six partials, MIN-allreduce, finalize
int reduce_dt(des_dev *h)
{
    if (h->comm_size <= 1) { launch_dt_finalize(h, nullptr); return DES_OK; }
    hipLaunchKernelGGL(k_dt_pack, dim3(1), dim3(DES_BLOCK), 0, h->stream, h->d_clk, h->d_red, h->dt_part, h->dt_part_cap,
                       h->dt_parts_used);
    h->dt_parts_used = 0;
    ncclResult_t r = ncclAllReduce(h->d_red, h->d_red, 6, ncclDouble, ncclMin, h->comm, h->stream);
    if (r != ncclSuccess) { g_last_error = std::string("RCCL: ") + ncclGetErrorString(r); return DES_ERR_RESOURCE; }
    launch_dt_finalize(h, h->d_red);
    return DES_OK;
}

int sync_clock(des_dev *h)
{
    HIP_OK(hipMemcpyAsync(h->h_clk, h->d_clk, sizeof(DevClock), hipMemcpyDeviceToHost, h->stream));
    HIP_OK(hipStreamSynchronize(h->stream));
    return DES_OK;
}

// The pseudo-transient loop of a step (dynearthsol.cxx:803-864), entered after the force pass has
// left velocities and the residual partials (N3 stops there when has_PT): per iteration
// apply_vbcs (boundaries at rest) + update_coordinate, update_mesh without surface processes,
// strain rate, dvoldt / edvoldt, update_stress, update_force, update_velocity, residual -- i.e. the
// passes of a step under DevClock::pt (no clock, no temperature update, no NMD, no rotate_stress) --
// until the relative change of the residual drops below the tolerance.  The host joins the stream
// once per iteration to take that decision, as the reference's loop does; afterwards the real
// apply_vbcs + update_coordinate of the step follow.
int set_pt(des_dev *h, int on)
{
    static const int vals[2] = {0, 1};
    h->in_pt = on != 0;
    HIP_OK(hipMemcpyAsync(&h->d_clk->pt, &vals[on ? 1 : 0], sizeof(int), hipMemcpyHostToDevice, h->stream));
    return DES_OK;
}

void launch_vbcs_coord(des_dev *h)
{
    hipLaunchKernelGGL(k_apply_vbcs, dim3(nblk(h->nn)), dim3(DES_BLOCK), 0, h->stream, h->d_p, h->d_clk, h->nn,
                       h->bcflag, h->bnormals, h->edge_vec, h->edge_slot, h->vm, h->xt);
}

// The passes of one iteration (the ghost region of a decomposed mesh refreshed just before)
void pt_iteration(des_dev *h)
{
    launch_vbcs_coord(h);
    launch_e1<MODE_C | MODE_A>(h);
    launch_n1(h);
    launch_e2(h);
    launch_force_pass(h);                              // E3 + N3, or EN3 (the facet terms then ride in E2's launch)
}

// the reference's test (dynearthsol.cxx:829-833) on the residual every rank holds
inline bool pt_converged(const des_dev *h, double l2, double residual_old)
{
    return std::fabs((l2 - residual_old) / residual_old) < h->p.PT_relative_tolerance;
}

// in_step = false: initial_body_force_adjustment's loop (dynearthsol.cxx:546-591) -- the same iterations on the state
// as it stands, no step around them (so no apply_vbcs + update_coordinate of a step behind the loop).
// A decomposed mesh (round 4): every iteration starts with a refresh of the ghost region -- an iteration consumes two of
// its four layers (the dvoldt gather, the force gather) -- and the residual is the partition-independent one of
// engine/residual.hpp, its block partials put together over RCCL: every rank holds the same value and takes the same
// decision, the one a single engine takes.
int pt_loop(des_dev *h, bool in_step = true)
{
    int rc;
    const bool multi = h->nnbr > 0;
    if ((rc = residual_global(h)) || (rc = sync_clock(h))) return rc;     // l2 of the step's own update_force
    double residual_old = h->h_clk->l2_residual;
    if ((rc = set_pt(h, 1))) return rc;
    rc = [&]() -> int {
        int r;
        for (int pt_step = 0; pt_step < h->p.PT_max_iter; ++pt_step) {
            if (multi && (r = exchange(h))) return r;
            pt_iteration(h);
            if ((r = residual_global(h)) || (r = sync_clock(h))) return r;
            ++h->n_pt_iterations;
            const double l2 = h->h_clk->l2_residual;
            if (pt_converged(h, l2, residual_old)) break;
            residual_old = l2;
        }
        return DES_OK;
    }();
    // DevClock::pt goes back to 0 on EVERY exit of the loop: an exchange / RCCL / sync error inside it must not leave an
    // engine that is "steppable again" running its later steps with the boundaries at rest and no clock
    const int rc_off = set_pt(h, 0);
    if (rc || (rc = rc_off)) return rc;
    if (in_step) launch_vbcs_coord(h);                     // apply_vbcs + update_coordinate of the step itself
    return DES_OK;
}

// ... for the engines of a group in lockstep (des_dev_step_group, des_dev_body_force_adjustment_group)
int pt_loop_group(des_dev **g, int n, bool in_step)
{
    int rc;
    auto residual = [&](double *l2) -> int {
        int r;
        if ((r = residual_global_group(g, n))) return r;
        for (int k = 0; k < n; ++k) { hipSetDevice(g[k]->device); if ((r = sync_clock(g[k]))) return r; }
        *l2 = g[0]->h_clk->l2_residual;
        for (int k = 1; k < n; ++k)
            if (g[k]->h_clk->l2_residual != *l2) { g_last_error = "pseudo-transient loop: the ranks of a group hold different residuals"; return DES_ERR_INTERNAL; }
        return DES_OK;
    };
    double residual_old = 0, l2 = 0;
    if ((rc = residual(&residual_old))) return rc;
    auto pt_all = [&](int on) -> int {
        int first = DES_OK;
        for (int k = 0; k < n; ++k) { hipSetDevice(g[k]->device); const int r = set_pt(g[k], on); if (r && !first) first = r; }
        return first;
    };
    if ((rc = pt_all(1))) { pt_all(0); return rc; }
    rc = [&]() -> int {
        int r;
        for (int pt_step = 0; pt_step < g[0]->p.PT_max_iter; ++pt_step) {
            for (int k = 0; k < n; ++k) { hipSetDevice(g[k]->device); if (g[k]->nnbr > 0 && (r = exchange_local_pack(g[k]))) return r; }
            for (int k = 0; k < n; ++k) {
                hipSetDevice(g[k]->device);
                if (g[k]->nnbr > 0 && (r = exchange_local_take(g[k], g[k]->stream))) return r;
                pt_iteration(g[k]);
            }
            if ((r = residual(&l2))) return r;
            for (int k = 0; k < n; ++k) ++g[k]->n_pt_iterations;
            if (pt_converged(g[0], l2, residual_old)) break;
            residual_old = l2;
        }
        return DES_OK;
    }();
    const int rc_off = pt_all(0);                          // on every exit of the loop (see pt_loop)
    if (rc || (rc = rc_off)) return rc;
    if (in_step) for (int k = 0; k < n; ++k) { hipSetDevice(g[k]->device); launch_vbcs_coord(g[k]); }
    return DES_OK;
}

// compute_dt across the engines of a group: the six partials of each, their minimum on the host (this
// path serves tests and rehearsals; every 10th step), finalize on each
int reduce_dt_group(des_dev **g, int n)
{
    double red[6], mine[6];
    for (int k = 0; k < n; ++k) {
        des_dev *h = g[k];
        hipSetDevice(h->device);
        hipLaunchKernelGGL(k_dt_pack, dim3(1), dim3(DES_BLOCK), 0, h->stream, h->d_clk, h->d_red, h->dt_part, h->dt_part_cap,
                           h->dt_parts_used);
        h->dt_parts_used = 0;
        HIP_OK(hipMemcpyAsync(mine, h->d_red, 48, hipMemcpyDeviceToHost, h->stream));
        HIP_OK(hipStreamSynchronize(h->stream));
        for (int j = 0; j < 6; ++j) red[j] = k == 0 ? mine[j] : std::min(red[j], mine[j]);
    }
    for (int k = 0; k < n; ++k) {
        des_dev *h = g[k];
        hipSetDevice(h->device);
        HIP_OK(hipMemcpyAsync(h->d_red, red, 48, hipMemcpyHostToDevice, h->stream));
        HIP_OK(hipStreamSynchronize(h->stream));             // (red[] is a stack buffer)
        launch_dt_finalize(h, h->d_red);
    }
    return DES_OK;
}
