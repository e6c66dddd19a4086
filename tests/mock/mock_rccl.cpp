// mock_rccl.cpp -- TEST INFRASTRUCTURE: an in-process stand-in for the ten RCCL entry points libmirhi.so resolves with dlsym
// (mirhi_api.hip, rccl_load), so that the C ABI's band exchange (mirhi_comm_all_gather_bands: band arithmetic, the grouped
// send / receive pattern, the broadcast alternative, buffers and byte counts per peer) can run with N "ranks" = N mirhi devices
// of ONE process on ONE GPU, where the real library needs N GPUs.  Loaded through MIRHI_RCCL_LIBRARY by tests/test_gpu_api.py in a
// child process.  Semantics kept: point-to-point operations match in posting order per (sender, receiver) pair; a matched pair is
// a device-to-device copy on the receiver's stream, ordered behind what the sender's stream held when it posted, and the sender's
// stream continues behind the copy.  Not kept: a group's operations start only when the peer has posted (one thread plays all
// ranks, so nothing may block) -- the test synchronises the device before it looks at the frames.  Consequence for the PACKED exchange of
// interleaved tile rows (the library scatters the received slots into the frame right behind the group): a receive whose peer posts later
// delivers behind that scatter, so the test runs every exchange twice on unchanged frames -- the second scatter finds the first round's data
// (real RCCL blocks the stream at the receive; a stand-in that blocked HIP streams on one another deadlocks on the runtime's four hardware
// queues, which is how this was found).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <cstring>
#include <deque>
#include <mutex>
#include <vector>

struct ncclComm { int rank, world; };
namespace {
struct Op { bool send; int rank, peer; void* buf; size_t bytes; hipStream_t stream; };
std::mutex g_mu;
std::deque<Op> g_pending;
std::vector<Op> g_group;
int g_depth = 0;
size_t g_matched = 0, g_bytes = 0;
int g_fail_send_in = 0;          // test hook: the n-th ncclSend from now fails (0 = none)

size_t type_size(ncclDataType_t t) { return (t == ncclInt8 || t == ncclUint8) ? 1 : (t == ncclFloat16 || t == ncclBfloat16) ? 2 : (t == ncclFloat64 || t == ncclInt64 || t == ncclUint64) ? 8 : 4; }

ncclResult_t match_all() {
    bool progress = true;
    while (progress) {
        progress = false;
        for (size_t i = 0; i < g_pending.size() && !progress; i++) {
            if (!g_pending[i].send) continue;
            const Op s = g_pending[i];
            for (size_t j = 0; j < g_pending.size(); j++) {
                const Op r = g_pending[j];
                if (r.send || r.rank != s.peer || r.peer != s.rank) continue;
                bool earlier_send = false;      // the first still-pending send of this pair goes with the first pending receive
                for (size_t k = 0; k < i; k++) earlier_send |= g_pending[k].send && g_pending[k].rank == s.rank && g_pending[k].peer == s.peer;
                if (earlier_send) break;
                if (r.bytes != s.bytes) return ncclInvalidArgument;          // a real exchange would hang or corrupt: the test must see it
                hipEvent_t posted, copied;
                if (hipEventCreateWithFlags(&posted, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&copied, hipEventDisableTiming) != hipSuccess) return ncclUnhandledCudaError;
                if (hipEventRecord(posted, s.stream) != hipSuccess || hipStreamWaitEvent(r.stream, posted, 0) != hipSuccess) return ncclUnhandledCudaError;
                if (hipMemcpyAsync(r.buf, s.buf, s.bytes, hipMemcpyDeviceToDevice, r.stream) != hipSuccess) return ncclUnhandledCudaError;
                if (hipEventRecord(copied, r.stream) != hipSuccess || hipStreamWaitEvent(s.stream, copied, 0) != hipSuccess) return ncclUnhandledCudaError;
                (void)hipEventDestroy(posted); (void)hipEventDestroy(copied);
                g_matched++; g_bytes += s.bytes;
                g_pending.erase(g_pending.begin() + (long)(i > j ? i : j));
                g_pending.erase(g_pending.begin() + (long)(i > j ? j : i));
                progress = true;
                break;
            }
        }
    }
    return ncclSuccess;
}
ncclResult_t post(const Op& op) {
    std::lock_guard<std::mutex> lock(g_mu);
    if (g_depth > 0) { g_group.push_back(op); return ncclSuccess; }
    g_pending.push_back(op);
    return match_all();
}
}  // namespace

extern "C" {
ncclResult_t ncclGetUniqueId(ncclUniqueId* id) { memset(id, 0, sizeof *id); memcpy(id->internal, "mirhi-mock-rccl", 16); return ncclSuccess; }
ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
    if (memcmp(id.internal, "mirhi-mock-rccl", 16) != 0 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    *comm = new ncclComm{rank, nranks};
    return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t comm) { delete comm; return ncclSuccess; }
ncclResult_t ncclCommCount(const ncclComm_t comm, int* count) { *count = comm->world; return ncclSuccess; }
ncclResult_t ncclGroupStart() { std::lock_guard<std::mutex> lock(g_mu); g_depth++; return ncclSuccess; }
ncclResult_t ncclGroupEnd() {
    std::lock_guard<std::mutex> lock(g_mu);
    if (g_depth <= 0) return ncclInvalidUsage;
    if (--g_depth > 0) return ncclSuccess;
    for (const Op& op : g_group) g_pending.push_back(op);
    g_group.clear();
    return match_all();
}
ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream) {
    if (peer < 0 || peer >= comm->world || peer == comm->rank) return ncclInvalidArgument;
    { std::lock_guard<std::mutex> lock(g_mu); if (g_fail_send_in > 0 && --g_fail_send_in == 0) return ncclSystemError; }
    return post(Op{true, comm->rank, peer, const_cast<void*>(buf), count * type_size(type), stream});
}
ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream) {
    if (peer < 0 || peer >= comm->world || peer == comm->rank) return ncclInvalidArgument;
    return post(Op{false, comm->rank, peer, buf, count * type_size(type), stream});
}
ncclResult_t ncclBroadcast(const void* sendbuf, void* recvbuf, size_t count, ncclDataType_t type, int root, ncclComm_t comm, hipStream_t stream) {
    if (root < 0 || root >= comm->world) return ncclInvalidArgument;
    if (comm->rank != root) return post(Op{false, comm->rank, root, recvbuf, count * type_size(type), stream});
    for (int p = 0; p < comm->world; p++)
        if (p != root) { const ncclResult_t r = post(Op{true, root, p, const_cast<void*>(sendbuf), count * type_size(type), stream}); if (r != ncclSuccess) return r; }
    return ncclSuccess;
}
const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : r == ncclInvalidArgument ? "mock: invalid argument (peer, count or id)" : "mock: failure"; }
// test hooks: operations still waiting for their peer (must be 0 after every rank has posted), pairs and bytes moved so far
size_t mock_rccl_pending() { std::lock_guard<std::mutex> lock(g_mu); return g_pending.size() + g_group.size(); }
size_t mock_rccl_matched() { std::lock_guard<std::mutex> lock(g_mu); return g_matched; }
size_t mock_rccl_bytes() { std::lock_guard<std::mutex> lock(g_mu); return g_bytes; }
// the n-th ncclSend from now returns ncclSystemError; the thread's group depth (must be 0 again after a failed exchange); forget everything posted
void mock_rccl_fail_send_in(int n) { std::lock_guard<std::mutex> lock(g_mu); g_fail_send_in = n; }
int mock_rccl_group_depth() { std::lock_guard<std::mutex> lock(g_mu); return g_depth; }
void mock_rccl_reset() { std::lock_guard<std::mutex> lock(g_mu); g_pending.clear(); g_group.clear(); g_fail_send_in = 0; }
}
