// A stand-in for librccl used by ONE test (tests/test_multiproc.py::test_gather_features_rank_order_on_a_mock_backend): every
// "rank" lives in the calling process, ncclSend parks the sender's device pointer and ncclRecv copies from it device to device,
// so that ss_gather_features' multi-rank code paths (the peers' send, the root's grouped receives, the offsets and the rank
// order of its output) can run on a one-GPU box.  RCCL itself refuses two ranks on one device, and no multi-GPU box was
// available: this checks the library's logic, not RCCL and not xGMI.  Test infrastructure only.
//   hipcc -shared -fPIC tests/mock_rccl/mock_rccl.cpp -o <dir>/libmock_rccl.so
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdio>
#include <map>
#include <mutex>
#include <vector>

namespace {
struct Parked {
    const void *ptr;
    size_t count;
    int dtype;
};
std::mutex mu;
std::map<int, Parked> parked;     // sender rank -> what it sent (the comm handle carries the caller's rank, see below)
std::vector<int> log_;            // call log: 1 group start, 2 group end, 100 + peer: recv from peer, 200 + peer: send to peer
int in_group = 0;
}  // namespace

// The test passes a pointer to an int holding the calling "rank" as the communicator.
extern "C" {

int ncclGroupStart()
{
    std::lock_guard<std::mutex> l(mu);
    ++in_group;
    log_.push_back(1);
    return 0;
}

int ncclGroupEnd()
{
    std::lock_guard<std::mutex> l(mu);
    --in_group;
    log_.push_back(2);
    return 0;
}

int ncclSend(const void *sendbuff, size_t count, int datatype, int peer, void *comm, hipStream_t)
{
    std::lock_guard<std::mutex> l(mu);
    const int me = *static_cast<const int *>(comm);
    parked[me] = Parked{sendbuff, count, datatype};
    log_.push_back(200 + peer);
    return 0;
}

int ncclRecv(void *recvbuff, size_t count, int datatype, int peer, void *comm, hipStream_t stream)
{
    std::lock_guard<std::mutex> l(mu);
    (void)comm;
    log_.push_back(100 + peer);
    if (!in_group) return 5;  // the root must receive inside a group (all peers progress together)
    auto it = parked.find(peer);
    if (it == parked.end() || it->second.count != count || it->second.dtype != datatype || datatype != 7 /* ncclFloat32 */) return 4;
    return hipMemcpyAsync(recvbuff, it->second.ptr, count * sizeof(float), hipMemcpyDeviceToDevice, stream) == hipSuccess ? 0 : 1;
}

int ncclAllGather(const void *sendbuff, void *recvbuff, size_t sendcount, int datatype, void *comm, hipStream_t stream)
{
    (void)datatype;
    const int me = *static_cast<const int *>(comm);
    return hipMemcpyAsync(static_cast<float *>(recvbuff) + static_cast<size_t>(me) * sendcount, sendbuff, sendcount * sizeof(float),
                          hipMemcpyDeviceToDevice, stream) == hipSuccess ? 0 : 1;
}

// test hooks
int mock_log_size() { return static_cast<int>(log_.size()); }
int mock_log_at(int i) { return log_[static_cast<size_t>(i)]; }
void mock_reset()
{
    std::lock_guard<std::mutex> l(mu);
    parked.clear();
    log_.clear();
}

}  // extern "C"
