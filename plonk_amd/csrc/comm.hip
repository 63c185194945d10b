// The two all-gather implementations behind comm.hpp.
//
// RCCL is bound at run time (dlopen of librccl.so.1, which resolves to the copy already loaded in the
// process if there is one): a single-GPU user of the library never loads it, and the library has no
// link-time dependency on it.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <mutex>
#include <vector>

#include "comm.hpp"

namespace plonk {

Exchange::~Exchange() {
    for (auto& pair : ev)
        for (auto& e : pair)
            if (e) (void)hipEventDestroy(e);
}

void Exchange::retire_oldest() {
    const int i = (head + RING - pending) % RING;
    float ms = 0.f;
    HIP_CHECK(hipEventSynchronize(ev[i][1]));
    HIP_CHECK(hipEventElapsedTime(&ms, ev[i][0], ev[i][1]));
    elapsed_ms += ms;
    --pending;
}

void Exchange::mark_begin(hipStream_t stream) {
    if (!timed) return;
    if (pending == RING) retire_oldest();
    for (auto& e : ev[head])
        if (!e) HIP_CHECK(hipEventCreate(&e));
    HIP_CHECK(hipEventRecord(ev[head][0], stream));
}

void Exchange::mark_end(hipStream_t stream) {
    if (!timed) return;
    HIP_CHECK(hipEventRecord(ev[head][1], stream));
    head = (head + 1) % RING;
    ++pending;
}

double Exchange::take_elapsed_ms() {
    while (pending) retire_oldest();
    const double ms = elapsed_ms;
    elapsed_ms = 0.0;
    return ms;
}

namespace {

struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

const RcclApi& rccl() {
    static RcclApi api;
    static std::once_flag once;
    static std::string failure;
    std::call_once(once, [] {
        void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (!h) {
            const char* e = dlerror();
            failure = std::string("cannot load RCCL: ") + (e ? e : "unknown error");
            return;
        }
        api.handle = h;
        api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
        api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
        api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
        api.AllGather = reinterpret_cast<decltype(api.AllGather)>(dlsym(h, "ncclAllGather"));
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
        if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllGather || !api.GetErrorString) {
            failure = "RCCL library lacks an expected entry point";
            api.handle = nullptr;
        }
    });
    if (!api.handle) throw Error(PLONK_E_HIP, failure);
    return api;
}

void rccl_check(ncclResult_t rc, const char* what) {
    if (rc != ncclSuccess) throw Error(PLONK_E_HIP, std::string(what) + ": " + rccl().GetErrorString(rc));
}

struct RcclExchange : Exchange {
    ncclComm_t comm = nullptr;
    ~RcclExchange() override {
        if (comm) {
            (void)hipSetDevice(device);
            (void)rccl().CommDestroy(comm);
        }
    }
    void allgather(const void* send_dev, void* recv_dev, size_t nbytes, hipStream_t stream) override {
        mark_begin(stream);
        rccl_check(rccl().AllGather(send_dev, recv_dev, nbytes, ncclUint8, comm, stream), "ncclAllGather");
        mark_end(stream);
        ++calls;
        bytes += nbytes;
    }
    const char* kind() const override { return "rccl"; }
};

struct CallbackExchange : Exchange {
    plonk_allgather_fn fn = nullptr;
    void* user = nullptr;
    std::vector<uint8_t> send, recv;
    void allgather(const void* send_dev, void* recv_dev, size_t nbytes, hipStream_t stream) override {
        send.resize(nbytes);
        recv.resize(nbytes * world);
        mark_begin(stream);
        HIP_CHECK(hipMemcpyAsync(send.data(), send_dev, nbytes, hipMemcpyDeviceToHost, stream));
        HIP_CHECK(hipStreamSynchronize(stream));
        if (fn(user, send.data(), nbytes, recv.data()) != 0) throw Error(PLONK_E_INTERNAL, "all-gather callback failed");
        HIP_CHECK(hipMemcpyAsync(recv_dev, recv.data(), nbytes * world, hipMemcpyHostToDevice, stream));
        mark_end(stream);
        HIP_CHECK(hipStreamSynchronize(stream));   // `recv` is reused by the next call
        ++calls;
        bytes += nbytes;
    }
    const char* kind() const override { return "callback"; }
};

// Measurement aid: rank `rank` of a `world`-rank proof running alone.  Every rank's slot of an all-gather is
// filled with this rank's own record by device-to-device copies, except the 48-byte records of the coverage
// check (first point and count of the rank's plain slice and of its slices for rounds of 2 and 3 commitments),
// which are made up as the balanced splits of `total_points`.  The proofs that come out are not valid; the
// device work of the rank (its shares of the MSMs, its class of round 3) is real.
struct LoopbackExchange : Exchange {
    size_t total_points = 0;
    static void split(size_t total, uint32_t index, uint32_t parts, uint64_t* first, uint64_t* count) {
        const size_t base = total / parts, extra = total % parts;
        *first = index * base + std::min<size_t>(index, extra);
        *count = base + (index < extra ? 1 : 0);
    }
    void allgather(const void* send_dev, void* recv_dev, size_t nbytes, hipStream_t stream) override {
        if (nbytes == 6 * sizeof(uint64_t)) {
            uint64_t mine[6];
            HIP_CHECK(hipMemcpyAsync(mine, send_dev, sizeof(mine), hipMemcpyDeviceToHost, stream));
            HIP_CHECK(hipStreamSynchronize(stream));
            std::vector<uint64_t> rec(6 * (size_t)world, 0);
            for (uint32_t r = 0; r < world; ++r) {
                split(total_points, r, world, &rec[6 * r], &rec[6 * r + 1]);
                for (uint32_t g = 0; g < 2; ++g) {
                    if (!mine[3 + 2 * g]) continue;   // this rank brings no slice for rounds of g + 2 commitments
                    const uint32_t b = g + 2, k = r % b, members = (world - k + b - 1) / b;
                    split(total_points, r / b, members, &rec[6 * r + 2 + 2 * g], &rec[6 * r + 3 + 2 * g]);
                }
            }
            HIP_CHECK(hipMemcpyAsync(recv_dev, rec.data(), rec.size() * sizeof(uint64_t), hipMemcpyHostToDevice, stream));
            HIP_CHECK(hipStreamSynchronize(stream));
        } else {
            for (uint32_t r = 0; r < world; ++r)
                HIP_CHECK(hipMemcpyAsync(static_cast<char*>(recv_dev) + (size_t)r * nbytes, send_dev, nbytes,
                                         hipMemcpyDeviceToDevice, stream));
        }
        ++calls;
        bytes += nbytes;
    }
    const char* kind() const override { return "loopback"; }
};

}  // namespace

Exchange* comm_create_loopback(int device, uint32_t rank, uint32_t world, size_t total_points) {
    PLONK_REQUIRE(world >= 1 && rank < world, "comm: rank must be below world");
    std::unique_ptr<LoopbackExchange> ex(new LoopbackExchange());
    ex->rank = rank;
    ex->world = world;
    ex->device = device;
    ex->total_points = total_points;
    return ex.release();
}

void comm_unique_id(uint8_t out[COMM_ID_BYTES]) {
    static_assert(sizeof(ncclUniqueId) == COMM_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId id;
    rccl_check(rccl().GetUniqueId(&id), "ncclGetUniqueId");
    memcpy(out, &id, COMM_ID_BYTES);
}

Exchange* comm_create_rccl(int device, const uint8_t id_bytes[COMM_ID_BYTES], uint32_t rank, uint32_t world) {
    PLONK_REQUIRE(world >= 1 && rank < world, "comm: rank must be below world");
    std::unique_ptr<RcclExchange> ex(new RcclExchange());
    ex->rank = rank;
    ex->world = world;
    ex->device = device;
    ncclUniqueId id;
    memcpy(&id, id_bytes, COMM_ID_BYTES);
    rccl_check(rccl().CommInitRank(&ex->comm, (int)world, id, (int)rank), "ncclCommInitRank");
    return ex.release();
}

Exchange* comm_create_callback(int device, uint32_t rank, uint32_t world, plonk_allgather_fn fn, void* user) {
    PLONK_REQUIRE(world >= 1 && rank < world, "comm: rank must be below world");
    PLONK_REQUIRE(fn != nullptr, "comm: null all-gather callback");
    std::unique_ptr<CallbackExchange> ex(new CallbackExchange());
    ex->rank = rank;
    ex->world = world;
    ex->device = device;
    ex->fn = fn;
    ex->user = user;
    return ex.release();
}

}  // namespace plonk
