// Exchange between the ranks of a multi-GPU proof (one process per GPU).
//
// The prover needs one collective only: an all-gather of equal-sized records that live in device
// memory (per-rank partial sums of a sharded MSM; per-rank slices of the quotient polynomial).
// Two implementations: RCCL over xGMI (ncclAllGather on the caller's stream, device buffers end to
// end) and a host callback (the record is staged through host memory and handed to a function the
// launcher provides -- for rehearsing the multi-rank path where RCCL cannot run, e.g. several ranks
// sharing one GPU).  No counterpart in the single-process reference.
#pragma once
#include <memory>

#include "common.hpp"

namespace plonk {

struct Exchange {
    uint32_t rank = 0, world = 1;
    int device = 0;
    uint64_t calls = 0, bytes = 0;   // all-gathers issued and bytes contributed by this rank
    virtual ~Exchange();
    // recv_dev[r * bytes .. (r + 1) * bytes) = send_dev[0 .. bytes) of rank r, for every rank r;
    // ordered on `stream` like a kernel launch.
    virtual void allgather(const void* send_dev, void* recv_dev, size_t bytes, hipStream_t stream) = 0;
    virtual const char* kind() const = 0;
    // Optional timing: a pair of events around every all-gather on its stream; the time between them is what
    // the exchange costs this rank, the wait for the slowest peer included.
    void set_timing(bool on) { timed = on; }
    double take_elapsed_ms();   // stream time of the all-gathers since the last call (their streams must have drained)

  protected:
    void mark_begin(hipStream_t stream);
    void mark_end(hipStream_t stream);

  private:
    static constexpr int RING = 64;
    bool timed = false;
    hipEvent_t ev[RING][2] = {};
    int head = 0, pending = 0;
    double elapsed_ms = 0.0;
    void retire_oldest();
};

constexpr size_t COMM_ID_BYTES = 128;   // ncclUniqueId
void comm_unique_id(uint8_t out[COMM_ID_BYTES]);
Exchange* comm_create_rccl(int device, const uint8_t id[COMM_ID_BYTES], uint32_t rank, uint32_t world);
Exchange* comm_create_loopback(int device, uint32_t rank, uint32_t world, size_t total_points);
Exchange* comm_create_callback(int device, uint32_t rank, uint32_t world, plonk_allgather_fn fn, void* user);

}  // namespace plonk
