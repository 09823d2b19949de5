// engine/exchange.hpp -- The per-step ghost-region exchange through RCCL.
// Part of the single translation unit des_dev.hip (included inside namespace des_hip, after
// DevClock / struct des_dev); not a stand-alone header.

// ---- halo exchange through RCCL on the engine's stream ---------------------------

// The exchange of a step: one grouped send/recv per neighbour carrying the state of the whole
// ghost region, between a pack and an unpack launch, all on the engine's stream.
int exchange(des_dev *h)
{
    if (h->nnbr == 0) return DES_OK;
    if (!h->comm) { g_last_error = "decomposed engine without a communicator: call des_dev_comm_init"; return DES_ERR_INTERNAL; }
    const int ns = h->send_ptr[h->nnbr], nes = h->esend_ptr[h->nnbr];
    const int nr = h->recv_ptr[h->nnbr], ner = h->erecv_ptr[h->nnbr];
    hipLaunchKernelGGL(k_state_pack, dim3(nblk(ns + nes)), dim3(DES_BLOCK), 0, h->stream, ns, h->d_send_idx, h->d_send_noff,
                       nes, h->d_esend_idx, h->d_send_eoff, h->xt, h->vm, h->dh_n, h->stress, h->strain, h->plstrain,
                       h->ne, h->d_sendbuf);
    ncclGroupStart();
    for (int q = 0; q < h->nnbr; ++q) {
        ncclSend(h->d_sendbuf + h->send_off[q], (size_t)(h->send_off[q+1] - h->send_off[q]), ncclDouble,
                 h->nbr_rank[q], h->comm, h->stream);
        ncclRecv(h->d_recvbuf + h->recv_off[q], (size_t)(h->recv_off[q+1] - h->recv_off[q]), ncclDouble,
                 h->nbr_rank[q], h->comm, h->stream);
    }
    ncclResult_t r = ncclGroupEnd();
    if (r != ncclSuccess) { g_last_error = std::string("RCCL: ") + ncclGetErrorString(r); return DES_ERR_RESOURCE; }
    hipLaunchKernelGGL(k_state_unpack, dim3(nblk(nr + ner)), dim3(DES_BLOCK), 0, h->stream, nr, h->d_recv_idx, h->d_recv_noff,
                       ner, h->d_erecv_idx, h->d_recv_eoff, h->xt, h->vm, h->dh_n, h->stress, h->strain, h->plstrain,
                       h->ne, h->d_recvbuf);
    return DES_OK;
}

// compute_dt across ranks: pack the six partials, MIN-allreduce, finalize
int reduce_dt(des_dev *h)
{
    if (h->comm_size <= 1) { launch_dt_finalize(h, nullptr); return DES_OK; }
    hipLaunchKernelGGL(k_dt_pack, dim3(1), dim3(1), 0, h->stream, h->d_clk, h->d_red);
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
