// engine/selfcheck.hpp -- Start-up self-check of an engine's RCCL communicator (des_dev_comm_selfcheck).
// Shared by the 3-D engine (des_dev.hip) and the 2-D engine (des_dev2d.hip); plain functions of the buffers both keep.
//
// What it proves before the first step of a multi-GPU run is trusted (nothing in the build container has two GPUs, so
// the first run on real xGMI must be able to tell "the wires carry my bytes" from "the step is wrong"):
//   1. ncclCommCount == the world size the caller expects and ncclCommUserRank == its rank;
//   2. every neighbour message of the ghost-region exchange -- the SAME buffers, offsets, lengths and grouped
//      ncclSend / ncclRecv calls des_dev_step issues -- arrives whole and from the right rank: message q of rank r to
//      neighbour s is filled with the integers (r * 4096 + s) * 2^24 + (i mod 2^24) (exact in a double), so the
//      receiver knows every double it must hold without a second message; a truncated, swapped or stale message fails;
//   3. ncclAllReduce(SUM) of rank + 1 and ncclAllReduce(MIN / MAX) of values that differ per rank give the closed forms
//      (the three reductions a run uses: l2_residual, compute_dt, the 2-D wall extent).
// The message buffers are scratch between steps (every exchange rewrites them), so nothing has to be restored.
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <string>
#include <vector>

namespace des_selfcheck {

inline double pattern(int from, int to, long long i)
{
    return (double)((long long)from * 4096 + to) * 16777216.0 + (double)(i % 16777216);
}

// returns an empty string when everything holds, else what failed
inline std::string run(ncclComm_t comm, hipStream_t stream, int expect_world, int expect_rank, int nnbr, const int *nbr_rank,
                       const long long *send_off, const long long *recv_off, double *d_sendbuf, double *d_recvbuf, double *d_red /* >= 3 */)
{
    if (!comm) return "no communicator attached (des_dev_comm_init)";
    int count = 0, urank = -1;
    if (ncclCommCount(comm, &count) != ncclSuccess || ncclCommUserRank(comm, &urank) != ncclSuccess) return "ncclCommCount / ncclCommUserRank failed";
    if (count != expect_world) return "ncclCommCount = " + std::to_string(count) + ", expected " + std::to_string(expect_world);
    if (urank != expect_rank) return "ncclCommUserRank = " + std::to_string(urank) + ", expected " + std::to_string(expect_rank);
    auto hip_ok = [](hipError_t e) { return e == hipSuccess; };
    // 2. the exchange's own messages
    if (nnbr > 0) {
        const long long ns = send_off[nnbr], nr = recv_off[nnbr];
        std::vector<double> hs((size_t)ns), hr((size_t)nr, -1.0);
        for (int q = 0; q < nnbr; ++q)
            for (long long i = send_off[q]; i < send_off[q + 1]; ++i) hs[(size_t)i] = pattern(urank, nbr_rank[q], i - send_off[q]);
        if (!hip_ok(hipMemcpyAsync(d_sendbuf, hs.data(), (size_t)ns * sizeof(double), hipMemcpyHostToDevice, stream))
            || !hip_ok(hipMemcpyAsync(d_recvbuf, hr.data(), (size_t)nr * sizeof(double), hipMemcpyHostToDevice, stream)))
            return "hipMemcpy of the message buffers failed";
        if (ncclGroupStart() != ncclSuccess) return "ncclGroupStart failed";
        for (int q = 0; q < nnbr; ++q) {
            ncclSend(d_sendbuf + send_off[q], (size_t)(send_off[q + 1] - send_off[q]), ncclDouble, nbr_rank[q], comm, stream);
            ncclRecv(d_recvbuf + recv_off[q], (size_t)(recv_off[q + 1] - recv_off[q]), ncclDouble, nbr_rank[q], comm, stream);
        }
        const ncclResult_t r = ncclGroupEnd();
        if (r != ncclSuccess) return std::string("grouped ncclSend / ncclRecv: ") + ncclGetErrorString(r);
        if (!hip_ok(hipMemcpyAsync(hr.data(), d_recvbuf, (size_t)nr * sizeof(double), hipMemcpyDeviceToHost, stream))
            || !hip_ok(hipStreamSynchronize(stream)))
            return "reading the received messages back failed";
        for (int q = 0; q < nnbr; ++q)
            for (long long i = recv_off[q]; i < recv_off[q + 1]; ++i)
                if (hr[(size_t)i] != pattern(nbr_rank[q], urank, i - recv_off[q]))
                    return "message from rank " + std::to_string(nbr_rank[q]) + ": double " + std::to_string(i - recv_off[q]) + " of "
                           + std::to_string(recv_off[q + 1] - recv_off[q]) + " is " + std::to_string(hr[(size_t)i]) + ", expected "
                           + std::to_string(pattern(nbr_rank[q], urank, i - recv_off[q]));
    }
    // 3. the reductions
    const double mine[3] = {(double)(urank + 1), (double)(urank + 1), (double)(urank + 1)};
    double got[3] = {0, 0, 0};
    if (!hip_ok(hipMemcpyAsync(d_red, mine, sizeof(mine), hipMemcpyHostToDevice, stream))) return "hipMemcpy failed";
    if (ncclAllReduce(d_red, d_red, 1, ncclDouble, ncclSum, comm, stream) != ncclSuccess
        || ncclAllReduce(d_red + 1, d_red + 1, 1, ncclDouble, ncclMin, comm, stream) != ncclSuccess
        || ncclAllReduce(d_red + 2, d_red + 2, 1, ncclDouble, ncclMax, comm, stream) != ncclSuccess)
        return "ncclAllReduce failed";
    if (!hip_ok(hipMemcpyAsync(got, d_red, sizeof(got), hipMemcpyDeviceToHost, stream)) || !hip_ok(hipStreamSynchronize(stream)))
        return "reading the reductions back failed";
    const double want_sum = 0.5 * count * (count + 1);
    if (got[0] != want_sum || got[1] != 1.0 || got[2] != (double)count)
        return "ncclAllReduce: sum " + std::to_string(got[0]) + " (expected " + std::to_string(want_sum) + "), min " + std::to_string(got[1])
               + " (1), max " + std::to_string(got[2]) + " (" + std::to_string(count) + ")";
    return std::string();
}

} // namespace des_selfcheck
