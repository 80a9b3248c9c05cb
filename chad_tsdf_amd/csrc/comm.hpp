// comm.hpp -- the transport of the multi-GPU exchange (the reference is single-process, there is nothing
// to cite): two collectives over the ranks of a job, all-gather and all-reduce(sum), on DEVICE buffers of the map's stream.
//   * RCCL (the product path): ncclAllGather / ncclAllReduce over xGMI.  librccl.so.1 is opened at run time (dlopen; a process
//     that already holds one -- PyTorch ships its own -- gets that same instance), so the library itself loads without RCCL
//     and single-GPU users never touch it.
//   * caller-supplied collectives on HOST buffers (tests and rehearsals: gloo / MPI / anything): the buffers are staged
//     through host memory around each call.  RCCL refuses several ranks on one GPU; this transport is how the exchange's
//     device-side logic is exercised on a one-GPU box.
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/chad_hip.h"
#include "exchange_plan.hpp"

namespace chadhip {

// the few RCCL entry points the exchange needs, by their documented C signatures (rccl.h: ncclGetUniqueId :187,
// ncclCommInitRank :220, ncclCommDestroy :260, ncclGetErrorString :339, ncclAllReduce :611, ncclAllGather :678)
struct RcclApi {
    struct UniqueId { char internal[128]; };
    typedef int (*GetUniqueId)(UniqueId*);
    typedef int (*CommInitRank)(void** comm, int nranks, UniqueId id, int rank);
    typedef int (*CommDestroy)(void* comm);
    typedef int (*CommCount)(const void* comm, int* count);
    typedef int (*CommAbort)(void* comm);
    typedef int (*CommGetAsyncError)(void* comm, int* async_error);
    typedef const char* (*GetErrorString)(int);
    typedef int (*AllReduce)(const void* send, void* recv, size_t count, int datatype, int op, void* comm, hipStream_t stream);
    typedef int (*AllGather)(const void* send, void* recv, size_t sendcount, int datatype, void* comm, hipStream_t stream);
    typedef int (*Send)(const void* send, size_t count, int datatype, int peer, void* comm, hipStream_t stream);
    typedef int (*Recv)(void* recv, size_t count, int datatype, int peer, void* comm, hipStream_t stream);
    typedef int (*GroupCall)(void);
    enum { kInt8 = 0, kUint32 = 3, kInt64 = 4, kUint64 = 5, kSum = 0 };   // ncclDataType_t / ncclRedOp_t values (rccl.h:448-472)
    void* handle = nullptr;
    GetUniqueId get_unique_id = nullptr;
    CommInitRank comm_init_rank = nullptr;
    CommDestroy comm_destroy = nullptr;
    CommCount comm_count = nullptr;             // (optional: what the transport itself says about its size)
    CommAbort comm_abort = nullptr;             // (optional)
    CommGetAsyncError comm_async_error = nullptr;   // (optional)
    GetErrorString get_error_string = nullptr;
    AllReduce all_reduce = nullptr;
    AllGather all_gather = nullptr;
    Send send = nullptr; Recv recv = nullptr;   // (optional: the sparse exchange's point-to-point step; without them it reports an error)
    GroupCall group_start = nullptr, group_end = nullptr;

    // empty string on success
    std::string load() {
        if (handle) return "";
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (handle) break;
        }
        if (!handle) return std::string("cannot load librccl.so.1: ") + dlerror();
        get_unique_id = reinterpret_cast<GetUniqueId>(dlsym(handle, "ncclGetUniqueId"));
        comm_init_rank = reinterpret_cast<CommInitRank>(dlsym(handle, "ncclCommInitRank"));
        comm_destroy = reinterpret_cast<CommDestroy>(dlsym(handle, "ncclCommDestroy"));
        comm_count = reinterpret_cast<CommCount>(dlsym(handle, "ncclCommCount"));
        comm_abort = reinterpret_cast<CommAbort>(dlsym(handle, "ncclCommAbort"));
        comm_async_error = reinterpret_cast<CommGetAsyncError>(dlsym(handle, "ncclCommGetAsyncError"));
        get_error_string = reinterpret_cast<GetErrorString>(dlsym(handle, "ncclGetErrorString"));
        all_reduce = reinterpret_cast<AllReduce>(dlsym(handle, "ncclAllReduce"));
        all_gather = reinterpret_cast<AllGather>(dlsym(handle, "ncclAllGather"));
        send = reinterpret_cast<Send>(dlsym(handle, "ncclSend"));
        recv = reinterpret_cast<Recv>(dlsym(handle, "ncclRecv"));
        group_start = reinterpret_cast<GroupCall>(dlsym(handle, "ncclGroupStart"));
        group_end = reinterpret_cast<GroupCall>(dlsym(handle, "ncclGroupEnd"));
        if (!get_unique_id || !comm_init_rank || !comm_destroy || !get_error_string || !all_reduce || !all_gather) { handle = nullptr; return "librccl.so.1 lacks an expected symbol"; }
        return "";
    }
};
inline RcclApi& rccl() { static RcclApi api; return api; }

}  // namespace chadhip

struct chad_comm {
    static constexpr size_t kCallbackStagingLimit = size_t(16) << 30;   // bytes of host memory the callback transport may stage at once
    int32_t rank = 0, world = 1;
    int device = 0;
    void* nccl = nullptr;                         // ncclComm_t (RCCL transport)
    chad_allgather_fn gather = nullptr;           // callback transport
    chad_allreduce_fn reduce = nullptr;
    void* user = nullptr;
    std::vector<unsigned char> host;              // callback transport: staging
    std::string error;
    bool no_shortcut = false;                     // CHAD_DEBUG_COMM_NO_SHORTCUT: a one-rank communicator goes through its transport as well
                                                  // (tests: the RCCL calls themselves on a one-GPU box)
    bool exchanges() const { return world > 1 || no_shortcut; }
    uint64_t p2p_calls = 0;                       // exchange_slices calls that went through ncclSend / ncclRecv (chad_pipeline_times: tests read it)
    bool dead = false;                            // abort() has been called: every later collective fails at once
    void abort() {
        if (nccl && !dead && chadhip::rccl().comm_abort) { (void)chadhip::rccl().comm_abort(nccl); nccl = nullptr; }
        dead = true; error = "the communicator was aborted after a failed exchange";
    }
    // the size of the job as the TRANSPORT sees it (RCCL: ncclCommCount; a one-rank communicator that never opened RCCL: 1; callbacks: what the caller said)
    int32_t transport_world() const {
        int n = world;
        if (nccl && chadhip::rccl().comm_count && chadhip::rccl().comm_count(nccl, &n) != 0) n = -1;
        return n;
    }

    // recv[world * bytes] <- every rank's send[bytes], in rank order.  Device buffers; ordered on `stream`.
    bool all_gather(const void* send, void* recv, size_t bytes, hipStream_t stream) {
        if (dead) return false;
        if (!exchanges()) return hipMemcpyAsync(recv, send, bytes, hipMemcpyDeviceToDevice, stream) == hipSuccess;
        if (nccl) {
            if (bytes % 8) { error = "all_gather: size not a multiple of 8"; return false; }
            const int r = chadhip::rccl().all_gather(send, recv, bytes / 8, chadhip::RcclApi::kUint64, nccl, stream);
            if (r) { error = std::string("ncclAllGather: ") + chadhip::rccl().get_error_string(r); return false; }
            return true;
        }
        host.resize(bytes * size_t(world + 1));
        if (hipMemcpyAsync(host.data(), send, bytes, hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) { error = "all_gather: staging copy failed"; return false; }
        if (gather(user, host.data(), host.data() + bytes, bytes) != 0) { error = "all_gather callback failed"; return false; }
        if (hipMemcpyAsync(recv, host.data() + bytes, bytes * size_t(world), hipMemcpyHostToDevice, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) { error = "all_gather: staging copy failed"; return false; }
        return true;
    }
    // Every rank hands every other rank one slice of `send` and receives one into `recv` (sizes and places: the plan, the same picture on
    // every rank).  RCCL: one group of ncclSend / ncclRecv pairs -- point-to-point over xGMI, every link carrying only what its two
    // ends exchange.  Callbacks: whole send buffers gathered through the caller's all_gather, the slices picked out on the host.
    bool exchange_slices(const void* send, void* recv, const chadhip::ExchangePlan& p, hipStream_t stream) {
        if (dead) return false;
        const char* src = static_cast<const char*>(send);
        char* dst = static_cast<char*>(recv);
        // a one-rank RCCL communicator that is not short-cut (CHAD_DEBUG_COMM_NO_SHORTCUT) sends its own slice to ITSELF through the grouped
        // ncclSend / ncclRecv below -- RCCL allows a rank to be its own peer inside a group -- so that the product transport's symbols, dtype
        // and grouping run on the one GPU a test box has
        const bool self_through_rccl = nccl && no_shortcut && world == 1;
        if (!self_through_rccl && p.send_bytes[rank] && hipMemcpyAsync(dst + p.recv_off[rank], src + p.send_off[rank], p.send_bytes[rank], hipMemcpyDeviceToDevice, stream) != hipSuccess) { error = "exchange_slices: local copy failed"; return false; }
        if (world == 1 && !self_through_rccl) return true;
        if (nccl) {
            chadhip::RcclApi& api = chadhip::rccl();
            if (!api.send || !api.recv || !api.group_start || !api.group_end) { error = "librccl.so.1 lacks ncclSend / ncclRecv / ncclGroupStart / ncclGroupEnd"; return false; }
            int r = api.group_start();
            for (int peer = 0; peer < world && !r; peer++) {
                if (peer == rank && !self_through_rccl) continue;
                const size_t to_recv = peer == rank ? p.send_bytes[rank] : p.recv_bytes[peer];   // (a rank's slice for itself has one size)
                if (p.send_bytes[peer]) r = api.send(src + p.send_off[peer], p.send_bytes[peer], chadhip::RcclApi::kInt8, peer, nccl, stream);
                if (!r && to_recv) r = api.recv(dst + p.recv_off[peer], to_recv, chadhip::RcclApi::kInt8, peer, nccl, stream);
            }
            const int e = api.group_end();
            if (r || e) { error = std::string("ncclSend / ncclRecv: ") + api.get_error_string(r ? r : e); return false; }
            p2p_calls++;
            return true;
        }
        const size_t pad = (std::max<size_t>(p.longest_send, 8) + 7) & ~size_t(7);
        size_t mine = 0;
        for (int r = 0; r < world; r++) mine = std::max(mine, p.send_off[r] + p.send_bytes[r]);
        if (mine > pad) { error = "exchange_slices: a send buffer longer than the plan's longest"; return false; }
        if (pad > kCallbackStagingLimit / size_t(world + 1)) { error = "exchange_slices: the callback transport stages whole send buffers through host memory and this exchange is too large for that (a rehearsal path: use RCCL)"; return false; }
        host.resize(pad * size_t(world + 1));
        if (mine && (hipMemcpyAsync(host.data(), send, mine, hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess)) { error = "exchange_slices: staging copy failed"; return false; }
        if (gather(user, host.data(), host.data() + pad, pad) != 0) { error = "exchange_slices: all_gather callback failed"; return false; }
        for (int r = 0; r < world; r++) {
            if (r == rank || !p.recv_bytes[r]) continue;
            if (p.peer_off[r] + p.recv_bytes[r] > pad) { error = "exchange_slices: a slice outside its sender's buffer"; return false; }
            if (hipMemcpyAsync(dst + p.recv_off[r], host.data() + pad * size_t(r + 1) + p.peer_off[r], p.recv_bytes[r], hipMemcpyHostToDevice, stream) != hipSuccess) { error = "exchange_slices: staging copy failed"; return false; }
        }
        if (hipStreamSynchronize(stream) != hipSuccess) { error = "exchange_slices: staging copy failed"; return false; }
        return true;
    }
    // buf[count] <- sum over the ranks, in place.  dtype: CHAD_REDUCE_U32 / CHAD_REDUCE_I64.
    bool all_reduce(void* buf, size_t count, int32_t dtype, hipStream_t stream) {
        if (dead) return false;
        if (!exchanges() || count == 0) return true;
        const size_t bytes = count * (dtype == CHAD_REDUCE_I64 ? 8 : 4);
        if (nccl) {
            const int r = chadhip::rccl().all_reduce(buf, buf, count, dtype == CHAD_REDUCE_I64 ? chadhip::RcclApi::kInt64 : chadhip::RcclApi::kUint32, chadhip::RcclApi::kSum, nccl, stream);
            if (r) { error = std::string("ncclAllReduce: ") + chadhip::rccl().get_error_string(r); return false; }
            return true;
        }
        host.resize(bytes);
        if (hipMemcpyAsync(host.data(), buf, bytes, hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) { error = "all_reduce: staging copy failed"; return false; }
        if (reduce(user, host.data(), count, dtype) != 0) { error = "all_reduce callback failed"; return false; }
        if (hipMemcpyAsync(buf, host.data(), bytes, hipMemcpyHostToDevice, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) { error = "all_reduce: staging copy failed"; return false; }
        return true;
    }
};
