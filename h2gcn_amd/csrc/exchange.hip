// exchange.hip -- all-gather of the row-sharded embedding between the GPUs of one node through IPC-exported
// device buffers (h2gcn_xchg_* in include/h2gcn_hip.h).  No counterpart in the reference (single process, single
// device); it serves the row partition SURVEY.md 8(e) adds in front of GCNLayer.call
// (reference h2gcn/models/_layers.py:78-81): every rank needs all of X before its local fused SpMM.
//
// Protocol (per channel, sequence number q = 1, 2, ...; every rank runs the same code):
//   begin:  [stream]      wait for this channel's previous pulls      (so slot q&1, last used by q-2, is free: a
//                                                                      peer can only have announced q-1 after its
//                                                                      own pulls of q-2 finished -- no ack needed)
//           [stream]      stage_kernel: src -> my slot[q&1]  (+ my own block of `full`)
//           [stream]      signal_kernel: system-scope release, then store q into flag[channel][my rank] of EVERY
//                         peer (their memory, over xGMI): "my shard q is readable"
//           [peer stream] wait until flag[channel][peer] >= q (spin on LOCAL fine-grained memory, bounded by a
//                         wall-clock limit), then copy peer's slot[q&1] -> full[peer block]
//   end:    [stream]      wait for the pull events of the channel.
// Pulls are either copy-engine transfers (hipMemcpyAsync on one stream per peer -- on a fully connected xGMI node
// every peer is its own link, so P-1 transfers run in parallel and no CU is taken from the SpMM) or ONE copy
// kernel whose workgroups are split over the peers (CU-driven loads over xGMI).
//
// hipGraph replay (copy-kernel mode): nothing the device code needs is baked into a launch -- the sequence number of a
// channel lives in DEVICE memory (post_seq: bumped by the signal kernel; pull_seq: bumped by the last workgroup of the
// pull kernel), and every address that depends on its parity (which of the two slots) is computed on the device.  A
// captured train/eval step therefore replays correctly: each replay advances the counters exactly as an eager call
// would.  Copy-engine mode keeps host-side sequence numbers (hipMemcpyAsync needs its source address on the host) and
// cannot be captured.
//
// Visibility across devices -- why a peer that sees flag >= q reads the bytes of step q, and what a one-GPU test box cannot show.
//   The slots are ordinary (coarse-grained) hipMalloc memory: while stage_kernel runs, its stores may sit dirty in the L2 of
//   whichever of the 8 XCDs ran the workgroup -- and the 8 L2s are NOT coherent with each other, let alone with a remote device,
//   which reads this memory over xGMI at its home (the memory side: Infinity Cache / HBM), never through the owner's L2s.
//   (1) WRITER, release.  stage_kernel and signal_kernel are two dispatches on ONE stream.  A dispatch boundary on a multi-XCD
//       part ends with a release of at least agent scope: every XCD's L2 writes its dirty lines back to the memory side
//       (without that, kernel B on XCD 1 could not read what kernel A wrote on XCD 0 -- ordinary same-stream producer/consumer
//       code relies on it).  So when signal_kernel STARTS, the whole slot is at its home, on every XCD.  signal_kernel's own
//       __threadfence_system() is not what publishes the slot (it runs on one XCD and can only write back that one L2); it
//       orders signal_kernel's OWN flag stores behind everything this device did before, and the flag stores are system-scope
//       release stores into the peer's UNCACHED flag words.  The slot is therefore at its home before any flag says q.
//   (2) hipGraph replay keeps (1): a captured stream dependency stage -> signal becomes a graph edge; kernel nodes joined by an
//       edge are dispatched with the barrier bit and the same end-of-kernel release as on a stream.  Nothing here relies on
//       host-side ordering.
//   (3) READER, acquire.  One wave per pull workgroup spins on its LOCAL uncached flag with system-scope acquire loads; after
//       the workgroup barrier EVERY wave executes a system-scope acquire fence (buffer_inv sc0 sc1: drops the non-local lines
//       of its CU's L1 and its XCD's L2), so a line of the peer's slot cached two steps ago (same slot parity) cannot be
//       served; the pull loads are non-temporal on top.  Every XCD that runs a pull workgroup invalidates for itself.
//       Copy-engine mode: the wait kernel's acquire precedes an SDMA copy, which reads the peer's memory without the CU caches.
//   (4) The pulled bytes land in `full` (local memory) and reach the SpMM through a stream / event / graph edge: (1) again.
//   With every rank on ONE device (the test box) all of this traffic has the same home and the same L2s, so (1) and (3) are
//   exercised but can never FAIL there: a missing write-back or invalidate would be invisible.  That is the one property of
//   this file no test on this box can pin.  It is therefore gated at run time instead: setting up a row-partitioned run
//   (partition.ShardedHops) sends FOUR test patterns -- both slots of a channel, each used twice: the re-use is where a stale
//   line could be served -- through an L2-sized IPC exchange of the requested mode AND through ncclAllGather and compares the
//   results on every rank (eager, before anything decides on hipGraph replay); a mismatch -- H2GCN_XCHG_INJECT_STALE=1 fakes one
//   by not updating the slot from the second step on -- switches the run to the `allgather` exchange with a warning.  bench.py
//   gates every IPC candidate the same way (against the regenerated embedding, two inputs) before it is timed.
//
// Deadlock freedom: signal(q) is enqueued before any wait(q) of the same rank, and everything enqueued before
// signal(q) depends only on signals < q of the peers -- induction over (step, channel) order, independent of how
// HIP maps streams onto hardware queues.  A peer that dies leaves waits that give up after `timeout_ms`.
#include <hip/hip_runtime.h>
#include <unistd.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <type_traits>
#include <vector>

#include "capi_internal.h"
#include "h2gcn_hip.h"

namespace {

using h2gcn::fail;

constexpr int kMaxWorld = 64;
constexpr int kMaxChannels = 64;
constexpr int kPullBlocksPerPeer = 16;  // copy-kernel mode: workgroups per peer
constexpr int kPullThreads = 256;
using u32x4 = unsigned int __attribute__((ext_vector_type(4)));

struct Blob {  // what h2gcn_xchg_export writes (H2GCN_XCHG_BLOB_BYTES, POD)
    hipIpcMemHandle_t data;
    hipIpcMemHandle_t flags;
    uint64_t slot_bytes;
    int32_t world, rank, n_channels, device;
    int64_t pid;
    uint32_t magic;
    uint32_t pad[7];
};
static_assert(sizeof(Blob) <= H2GCN_XCHG_BLOB_BYTES, "blob too large");
constexpr uint32_t kMagic = 0x48324758u;  // "H2GX"

struct PeerTable {  // device-resident copy of the peer pointers (kernel argument by value)
    char* data[kMaxWorld];
    uint32_t* flags[kMaxWorld];
};

// ---- device code ------------------------------------------------------------------------------------------

// src [rows, width] (stride ld_src) -> slot [rows_per_rank, width] and own block of `full`; rows beyond `rows`
// are written as zeros (short last shard).  width % 4 == 0 and 16-B alignment take the float4 path.
// Sequence number of the step being issued on a channel: copy-kernel mode reads it from device memory (`dev` = the
// channel's post counter, which still holds the previous step's number until the signal kernel bumps it), copy-engine
// mode gets it from the host.
struct SeqRef {
    const uint32_t* dev;
    uint32_t host;
    __device__ __forceinline__ uint32_t next() const { return dev ? __hip_atomic_load(dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u : host; }
};

template <bool VEC4>
__global__ void stage_kernel(const float* __restrict__ src, int64_t ld_src, int64_t rows, int64_t rows_per_rank,
                             int width, char* data, size_t slot_bytes, int channel, SeqRef seq, float* __restrict__ own, int skip_slot) {
    using T = typename std::conditional<VEC4, float4, float>::type;
    float* __restrict__ slot = reinterpret_cast<float*>(data + ((size_t)channel * 2 + (seq.next() & 1u)) * slot_bytes);
    const int wv = VEC4 ? width / 4 : width;
    const int64_t total = rows_per_rank * wv;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / wv;
        const int c = (int)(i - r * wv);
        T v;
        if constexpr (VEC4) v = make_float4(0.f, 0.f, 0.f, 0.f); else v = 0.f;
        if (r < rows) v = *reinterpret_cast<const T*>(src + r * ld_src + (int64_t)c * (VEC4 ? 4 : 1));
        if (!skip_slot) reinterpret_cast<T*>(slot)[i] = v;   // (skip_slot: failure injection, see H2GCN_XCHG_INJECT_STALE)
        reinterpret_cast<T*>(own)[i] = v;
    }
}

// whole matrix -> this step's slot (reduce-scatter staging in copy-kernel mode; bytes a multiple of 4)
__global__ void copy_to_slot_kernel(const float* __restrict__ src, size_t n_floats, char* data, size_t slot_bytes, int channel, SeqRef seq) {
    float* __restrict__ slot = reinterpret_cast<float*>(data + ((size_t)channel * 2 + (seq.next() & 1u)) * slot_bytes);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_floats; i += (size_t)gridDim.x * blockDim.x) slot[i] = src[i];
}

// "my shard `seq` of `channel` is readable": release everything this device wrote so far to system scope, then
// store the sequence number into the peers' flag words.
__global__ void signal_kernel(PeerTable peers, int world, int rank, int channel, SeqRef seq_ref, uint32_t* post_seq_dev) {
    const int q = threadIdx.x;
    const uint32_t seq = seq_ref.next();
    __threadfence_system();
    if (q < world && q != rank)
        __hip_atomic_store(peers.flags[q] + channel * kMaxWorld + rank, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    __syncthreads();   // every lane has read the old counter
    if (q == 0 && post_seq_dev) __hip_atomic_store(post_seq_dev, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// spin until *flag >= seq (wrap-safe signed difference) or the wall clock (100 MHz) runs out
__device__ __forceinline__ bool wait_flag(const uint32_t* flag, uint32_t seq, long long timeout_ticks, int* err) {
    const long long t0 = wall_clock64();
    while (true) {
        const uint32_t v = __hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((int32_t)(v - seq) >= 0) return true;
        if (wall_clock64() - t0 > timeout_ticks) {
            *err = 1;
            __threadfence_system();
            return false;
        }
        __builtin_amdgcn_s_sleep(32);
    }
}

__global__ void wait_kernel(const uint32_t* flag, uint32_t seq, long long timeout_ticks, int* err) {
    if (threadIdx.x == 0) wait_flag(flag, seq, timeout_ticks, err);
}

// A wait that gave up must not leave plausible numbers behind: once the error word is set, every block that was (or
// would have been) filled from a peer is overwritten with NaN, so that whatever consumes it fails loudly (a NaN loss)
// even if nobody asks h2gcn_xchg_status.  Launched after each copy-engine pull; does nothing while *err == 0.
__global__ void poison_kernel(const int* err, float* dst, size_t n_floats) {
    if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0) return;
    const float nan = __int_as_float(0x7fc00000);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_floats; i += (size_t)gridDim.x * blockDim.x) dst[i] = nan;
}

// out[i] = sum over ranks q = 0 .. world-1 (ascending: a fixed order, the result does not depend on arrival order) of
// block q, where block `rank` is this rank's own contribution and the others arrived in recv[q]
__global__ void sum_blocks_kernel(const float* __restrict__ recv, const float* own_host, char* data, size_t slot_bytes,
                                  int channel, const uint32_t* post_seq_dev, int world, int rank, size_t block_elems,
                                  float* __restrict__ out) {
    // this rank's own block sits in the slot of the step that was just posted (copy-kernel mode: parity from the device
    // counter, already bumped by the signal kernel that precedes this launch on the stream)
    const float* __restrict__ own = own_host;
    if (post_seq_dev) {
        const uint32_t seq = __hip_atomic_load(post_seq_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        own = reinterpret_cast<const float*>(data + ((size_t)channel * 2 + (seq & 1u)) * slot_bytes) + (size_t)rank * block_elems;
    }
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < block_elems; i += (size_t)gridDim.x * blockDim.x) {
        float acc = 0.f;
        for (int q = 0; q < world; ++q) acc += (q == rank) ? own[i] : recv[(size_t)q * block_elems + i];
        out[i] = acc;
    }
}

// Copy-kernel pulls: block b serves peer slot b / kPullBlocksPerPeer (the peers other than `rank`, in order).
__global__ __launch_bounds__(kPullThreads) void pull_kernel(PeerTable peers, const uint32_t* my_flags, int world, int rank,
                                                            int channel, uint32_t* pull_seq_dev, unsigned int* pull_done,
                                                            size_t slot_bytes, size_t extra_off, size_t bytes,
                                                            char* full, long long timeout_ticks, int* err) {
    const int pi = blockIdx.x / kPullBlocksPerPeer;         // 0 .. world-2
    const int sub = blockIdx.x - pi * kPullBlocksPerPeer;
    const int q = pi < rank ? pi : pi + 1;
    // the step this launch serves = pulls completed so far on the channel + 1 (device counter: a replayed graph advances it)
    const uint32_t seq = __hip_atomic_load(pull_seq_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    const size_t slot_off = ((size_t)channel * 2 + (seq & 1u)) * slot_bytes + extra_off;
    __shared__ int ok;
    if (threadIdx.x == 0) ok = wait_flag(my_flags + channel * kMaxWorld + q, seq, timeout_ticks, err) ? 1 : 0;
    __syncthreads();
    if (!ok) {  // timed out: poison this peer's block (see poison_kernel)
        float* bad = reinterpret_cast<float*>(full + (size_t)q * bytes);
        const float nan = __int_as_float(0x7fc00000);
        for (size_t j = (size_t)sub * kPullThreads + threadIdx.x; j < bytes / 4; j += (size_t)kPullBlocksPerPeer * kPullThreads) bad[j] = nan;
    } else {
        // the acquire above ran on one wave; make sure no stale line of the peer's slot (read two steps ago) is
        // served from this XCD's caches to the other waves
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
        const u32x4* __restrict__ s = reinterpret_cast<const u32x4*>(peers.data[q] + slot_off);
        u32x4* __restrict__ d = reinterpret_cast<u32x4*>(full + (size_t)q * bytes);
        const size_t n16 = bytes / 16;
        const size_t stride = (size_t)kPullBlocksPerPeer * kPullThreads;
        size_t i = (size_t)sub * kPullThreads + threadIdx.x;
        for (; i + 3 * stride < n16; i += 4 * stride) {  // 4 independent 16-B loads in flight per lane
            const u32x4 a = __builtin_nontemporal_load(s + i);
            const u32x4 b = __builtin_nontemporal_load(s + i + stride);
            const u32x4 c = __builtin_nontemporal_load(s + i + 2 * stride);
            const u32x4 e = __builtin_nontemporal_load(s + i + 3 * stride);
            d[i] = a;
            d[i + stride] = b;
            d[i + 2 * stride] = c;
            d[i + 3 * stride] = e;
        }
        for (; i < n16; i += stride) d[i] = __builtin_nontemporal_load(s + i);
        // tail bytes (shards are multiples of 4 bytes)
        if (sub == 0) {
            const size_t done = n16 * 16;
            for (size_t j = done + threadIdx.x * 4; j < bytes; j += kPullThreads * 4)
                *reinterpret_cast<uint32_t*>(full + (size_t)q * bytes + j) =
                    *reinterpret_cast<const uint32_t*>(peers.data[q] + slot_off + j);
        }
    }
    // the LAST workgroup to finish bumps the channel's pull counter (every workgroup read it on entry) and re-arms the ticket
    __syncthreads();
    if (threadIdx.x == 0 && atomicAdd(pull_done, 1u) == gridDim.x - 1) {
        __hip_atomic_store(pull_done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(pull_seq_dev, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}


// Halo pull (copy-kernel mode): instead of a peer's whole shard, only the rows of it this rank's hop matrices really name
// (`halo.rows[q]`: ascending LOCAL row ids of peer q, `halo.count[q]` of them).  On the synthetic shapes and on any
// products-like 2-hop ring that is every row and the dense pull is used; on graphs with locality -- a row block of Cora names
// 71 % of the remote rows at P = 8, its 1-hop ring alone 32 % -- it is what actually has to cross the links.  Same protocol as
// pull_kernel (flag wait, device-side sequence counters, last-workgroup ticket); rows not listed are left untouched in `full`
// (nothing reads them).  width % 4 == 0 (16-byte pieces).
struct HaloTable {
    const int32_t* rows[kMaxWorld];
    int64_t count[kMaxWorld];
};

__global__ __launch_bounds__(kPullThreads) void pull_rows_kernel(PeerTable peers, HaloTable halo, const uint32_t* my_flags, int world, int rank,
                                                                 int channel, uint32_t* pull_seq_dev, unsigned int* pull_done,
                                                                 size_t slot_bytes, int64_t rows_per_rank, int width, char* full,
                                                                 long long timeout_ticks, int* err) {
    const int pi = blockIdx.x / kPullBlocksPerPeer;
    const int sub = blockIdx.x - pi * kPullBlocksPerPeer;
    const int q = pi < rank ? pi : pi + 1;
    const uint32_t seq = __hip_atomic_load(pull_seq_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    const size_t slot_off = ((size_t)channel * 2 + (seq & 1u)) * slot_bytes;
    const int32_t* __restrict__ rows = halo.rows[q];
    const int64_t w4 = width / 4, units = halo.count[q] * w4;
    const int64_t stride = (int64_t)kPullBlocksPerPeer * kPullThreads;
    __shared__ int ok;
    if (threadIdx.x == 0) ok = wait_flag(my_flags + channel * kMaxWorld + q, seq, timeout_ticks, err) ? 1 : 0;
    __syncthreads();
    u32x4* __restrict__ d = reinterpret_cast<u32x4*>(full + (size_t)q * (size_t)rows_per_rank * (size_t)width * 4);
    if (!ok) {  // timed out: poison the rows that were to arrive
        const u32x4 nan = {0x7fc00000u, 0x7fc00000u, 0x7fc00000u, 0x7fc00000u};
        for (int64_t u = (int64_t)sub * kPullThreads + threadIdx.x; u < units; u += stride) d[(int64_t)rows[u / w4] * w4 + u % w4] = nan;
    } else {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
        const u32x4* __restrict__ s = reinterpret_cast<const u32x4*>(peers.data[q] + slot_off);
        int64_t u = (int64_t)sub * kPullThreads + threadIdx.x;
        for (; u + 3 * stride < units; u += 4 * stride) {  // 4 independent 16-B loads in flight per lane
            int64_t o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int64_t uk = u + k * stride;
                o[k] = (int64_t)rows[uk / w4] * w4 + uk % w4;
            }
            const u32x4 a = __builtin_nontemporal_load(s + o[0]);
            const u32x4 b = __builtin_nontemporal_load(s + o[1]);
            const u32x4 c = __builtin_nontemporal_load(s + o[2]);
            const u32x4 e = __builtin_nontemporal_load(s + o[3]);
            d[o[0]] = a;
            d[o[1]] = b;
            d[o[2]] = c;
            d[o[3]] = e;
        }
        for (; u < units; u += stride) {
            const int64_t o = (int64_t)rows[u / w4] * w4 + u % w4;
            d[o] = __builtin_nontemporal_load(s + o);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0 && atomicAdd(pull_done, 1u) == gridDim.x - 1) {
        __hip_atomic_store(pull_done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(pull_seq_dev, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

}  // namespace

// ---- host side ---------------------------------------------------------------------------------------------

struct h2gcn_xchg {
    int world = 1, rank = 0, n_channels = 1, mode = 0, device = 0;
    size_t slot_bytes = 0;
    long long timeout_ticks = 0;
    char* data = nullptr;       // exported: n_channels * 2 slots
    uint32_t* flags = nullptr;  // exported: [kMaxChannels][kMaxWorld] arrival counters, written by the peers
    int* err = nullptr;         // host-mapped: set by a wait that gave up
    uint32_t* seq_dev = nullptr;  // copy-kernel mode: [3][kMaxChannels] device counters: post_seq, pull_seq, pull_done
    bool connected = false;
    bool inject_stale = false;  // H2GCN_XCHG_INJECT_STALE=1 (tests): see allgather_impl
    PeerTable peers;
    std::vector<void*> opened;          // pointers to close with hipIpcCloseMemHandle
    std::vector<hipStream_t> streams;   // [world]; entry `rank` is the copy-kernel stream
    std::vector<hipEvent_t> fence;      // [n_channels]
    std::vector<hipEvent_t> pulled;     // [n_channels * world]
    std::vector<char> pulled_valid;     // event has been recorded at least once
    std::vector<hipEvent_t> summed;     // [n_channels]: end of the reduce-scatter's fixed-order sum (it reads this rank's own
    std::vector<char> summed_valid;     // slot, so the next begin on the channel -- on whatever stream -- must wait for it)
    std::vector<uint32_t> seq;          // [n_channels]
    std::vector<char> open_channel;     // begin without end (1 = all-gather, 2 = reduce-scatter)
    // reduce-scatter state per channel: receive buffer (world blocks) and what `end` has to sum
    std::vector<float*> rs_recv;
    std::vector<size_t> rs_recv_bytes;
    struct RsPending { float* out; const float* own; size_t block_elems; };
    std::vector<RsPending> rs_pending;
};

namespace {

void release(h2gcn_xchg* x) {
    if (!x) return;
    for (hipStream_t s : x->streams)
        if (s) (void)hipStreamSynchronize(s);
    for (void* p : x->opened) (void)hipIpcCloseMemHandle(p);
    for (hipEvent_t e : x->fence)
        if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : x->pulled)
        if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : x->summed)
        if (e) (void)hipEventDestroy(e);
    for (hipStream_t s : x->streams)
        if (s) (void)hipStreamDestroy(s);
    for (float* r : x->rs_recv)
        if (r) (void)hipFree(r);
    if (x->data) (void)hipFree(x->data);
    if (x->flags) (void)hipFree(x->flags);
    if (x->err) (void)hipHostFree(x->err);
    if (x->seq_dev) (void)hipFree(x->seq_dev);
    delete x;
}

// Pull `bytes` from every peer q -- its exported memory at offset `src_off` -- into dst + q * bytes, after the peer has
// announced sequence number `seq` on `channel`; records this channel's pull events.
int issue_pulls(h2gcn_xchg* x, int channel, uint32_t seq, size_t extra_off, size_t bytes, char* dst, const HaloTable* halo = nullptr,
                int64_t rows_per_rank = 0, int width = 0) {
    if (x->mode == H2GCN_XCHG_COPY_KERNEL && halo) {
        hipStream_t cs = x->streams[x->rank];
        H2GCN_HIP_TRY(hipStreamWaitEvent(cs, x->fence[channel], 0));
        hipLaunchKernelGGL(pull_rows_kernel, dim3((x->world - 1) * kPullBlocksPerPeer), dim3(kPullThreads), 0, cs, x->peers, *halo,
                           (const uint32_t*)x->flags, x->world, x->rank, channel, x->seq_dev + kMaxChannels + channel,
                           (unsigned int*)(x->seq_dev + 2 * kMaxChannels + channel), x->slot_bytes, rows_per_rank, width, dst,
                           x->timeout_ticks, x->err);
        H2GCN_HIP_TRY(hipGetLastError());
        const size_t ei = (size_t)channel * x->world + x->rank;
        H2GCN_HIP_TRY(hipEventRecord(x->pulled[ei], cs));
        x->pulled_valid[ei] = 1;
    } else if (x->mode == H2GCN_XCHG_COPY_KERNEL) {
        hipStream_t cs = x->streams[x->rank];
        H2GCN_HIP_TRY(hipStreamWaitEvent(cs, x->fence[channel], 0));
        hipLaunchKernelGGL(pull_kernel, dim3((x->world - 1) * kPullBlocksPerPeer), dim3(kPullThreads), 0, cs, x->peers,
                           (const uint32_t*)x->flags, x->world, x->rank, channel, x->seq_dev + kMaxChannels + channel,
                           (unsigned int*)(x->seq_dev + 2 * kMaxChannels + channel), x->slot_bytes, extra_off, bytes, dst,
                           x->timeout_ticks, x->err);
        H2GCN_HIP_TRY(hipGetLastError());
        const size_t ei = (size_t)channel * x->world + x->rank;
        H2GCN_HIP_TRY(hipEventRecord(x->pulled[ei], cs));
        x->pulled_valid[ei] = 1;
    } else {
        const size_t src_off = ((size_t)channel * 2 + (seq & 1u)) * x->slot_bytes + extra_off;
        for (int shift = 1; shift < x->world; ++shift) {
            const int q = (x->rank + shift) % x->world;  // start with a different peer on every rank
            hipStream_t ps = x->streams[q];
            H2GCN_HIP_TRY(hipStreamWaitEvent(ps, x->fence[channel], 0));
            hipLaunchKernelGGL(wait_kernel, dim3(1), dim3(64), 0, ps, (const uint32_t*)(x->flags + channel * kMaxWorld + q), seq,
                               x->timeout_ticks, x->err);
            H2GCN_HIP_TRY(hipGetLastError());
            H2GCN_HIP_TRY(hipMemcpyAsync(dst + (size_t)q * bytes, x->peers.data[q] + src_off, bytes, hipMemcpyDeviceToDevice, ps));
            hipLaunchKernelGGL(poison_kernel, dim3(64), dim3(256), 0, ps, (const int*)x->err, (float*)(dst + (size_t)q * bytes), bytes / 4);
            H2GCN_HIP_TRY(hipGetLastError());
            const size_t ei = (size_t)channel * x->world + q;
            H2GCN_HIP_TRY(hipEventRecord(x->pulled[ei], ps));
            x->pulled_valid[ei] = 1;
        }
    }
    return H2GCN_OK;
}

// sequence reference handed to the kernels of a step (see SeqRef)
SeqRef seq_ref(const h2gcn_xchg* x, int channel, uint32_t host_seq) {
    return SeqRef{x->mode == H2GCN_XCHG_COPY_KERNEL ? x->seq_dev + channel : nullptr, host_seq};
}

}  // namespace

extern "C" {

int h2gcn_xchg_create(int world, int rank, int n_channels, size_t slot_bytes, int mode, int timeout_ms,
                      h2gcn_xchg_t** out) {
    try {
        if (!out) return fail(H2GCN_ERR_INVALID_ARGUMENT, "out is NULL");
        *out = nullptr;
        if (world < 1 || world > kMaxWorld || rank < 0 || rank >= world)
            return fail(H2GCN_ERR_INVALID_ARGUMENT, "bad rank %d / world %d (max %d)", rank, world, kMaxWorld);
        if (n_channels < 1 || n_channels > kMaxChannels)
            return fail(H2GCN_ERR_INVALID_ARGUMENT, "n_channels = %d, supported 1..%d", n_channels, kMaxChannels);
        if (mode != H2GCN_XCHG_COPY_ENGINE && mode != H2GCN_XCHG_COPY_KERNEL)
            return fail(H2GCN_ERR_INVALID_ARGUMENT, "unknown exchange mode %d", mode);
        if (slot_bytes == 0 || slot_bytes % 4 != 0)
            return fail(H2GCN_ERR_INVALID_ARGUMENT, "slot_bytes = %zu must be a positive multiple of 4", slot_bytes);
        h2gcn_xchg* x = new h2gcn_xchg();
        struct Guard {
            h2gcn_xchg* p;
            ~Guard() { release(p); }
        } guard{x};
        x->world = world;
        x->rank = rank;
        x->n_channels = n_channels;
        x->mode = mode;
        x->slot_bytes = (slot_bytes + 255) / 256 * 256;  // keep every slot 256-B aligned
        x->timeout_ticks = (long long)(timeout_ms > 0 ? timeout_ms : 10000) * 100000LL;  // wall_clock64: 100 MHz
        {
            const char* inj = getenv("H2GCN_XCHG_INJECT_STALE");
            x->inject_stale = inj && inj[0] == '1';
        }
        memset(&x->peers, 0, sizeof(x->peers));
        H2GCN_HIP_TRY(hipGetDevice(&x->device));
        H2GCN_HIP_TRY(hipMalloc((void**)&x->data, x->slot_bytes * 2 * (size_t)n_channels));
        H2GCN_HIP_TRY(hipMemset(x->data, 0, x->slot_bytes * 2 * (size_t)n_channels));
        // arrival counters: fine-grained (uncached) device memory so that a store arriving over xGMI is seen by
        // a spinning wave without any cache maintenance
        const size_t flag_bytes = sizeof(uint32_t) * kMaxChannels * kMaxWorld;
        const char* flag_kind = "uncached";
        hipError_t fe = hipExtMallocWithFlags((void**)&x->flags, flag_bytes, hipDeviceMallocUncached);
        if (fe != hipSuccess) {
            (void)hipGetLastError();
            flag_kind = "fine-grained";
            fe = hipExtMallocWithFlags((void**)&x->flags, flag_bytes, hipDeviceMallocFinegrained);
        }
        if (fe != hipSuccess) {
            (void)hipGetLastError();
            flag_kind = "coarse-grained (hipMalloc)";
            H2GCN_HIP_TRY(hipMalloc((void**)&x->flags, flag_bytes));
        }
        if (getenv("H2GCN_XCHG_DEBUG"))
            fprintf(stderr, "[h2gcn_xchg] rank %d/%d: %d channels x 2 slots of %zu bytes, mode %s, flag memory: %s\n", rank, world,
                    n_channels, x->slot_bytes, mode == H2GCN_XCHG_COPY_ENGINE ? "copy-engine" : "copy-kernel", flag_kind);
        H2GCN_HIP_TRY(hipMemset(x->flags, 0, flag_bytes));
        H2GCN_HIP_TRY(hipHostMalloc((void**)&x->err, sizeof(int), hipHostMallocMapped));
        *x->err = 0;
        H2GCN_HIP_TRY(hipMalloc((void**)&x->seq_dev, sizeof(uint32_t) * 3 * kMaxChannels));
        H2GCN_HIP_TRY(hipMemset(x->seq_dev, 0, sizeof(uint32_t) * 3 * kMaxChannels));
        H2GCN_HIP_TRY(hipDeviceSynchronize());
        int lo = 0, hi = 0;
        H2GCN_HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));  // hi = numerically lowest = highest priority
        x->streams.assign(world, nullptr);
        for (int q = 0; q < world; ++q) {
            if (mode == H2GCN_XCHG_COPY_KERNEL && q != rank) continue;
            if (mode == H2GCN_XCHG_COPY_ENGINE && q == rank) continue;
            H2GCN_HIP_TRY(hipStreamCreateWithPriority(&x->streams[q], hipStreamNonBlocking, hi));
        }
        x->fence.assign(n_channels, nullptr);
        for (auto& e : x->fence) H2GCN_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        x->pulled.assign((size_t)n_channels * world, nullptr);
        for (auto& e : x->pulled) H2GCN_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        x->pulled_valid.assign((size_t)n_channels * world, 0);
        x->summed.assign(n_channels, nullptr);
        for (auto& e : x->summed) H2GCN_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        x->summed_valid.assign(n_channels, 0);
        x->seq.assign(n_channels, 0);
        x->open_channel.assign(n_channels, 0);
        x->rs_recv.assign(n_channels, nullptr);
        x->rs_recv_bytes.assign(n_channels, 0);
        x->rs_pending.assign(n_channels, h2gcn_xchg::RsPending{nullptr, nullptr, 0});
        x->peers.data[rank] = x->data;
        x->peers.flags[rank] = x->flags;
        x->connected = (world == 1);
        guard.p = nullptr;
        *out = x;
        return H2GCN_OK;
    } catch (const std::bad_alloc&) {
        return fail(H2GCN_ERR_OUT_OF_MEMORY, "host allocation failed in xchg_create");
    } catch (...) {
        return fail(H2GCN_ERR_INTERNAL, "unexpected exception in xchg_create");
    }
}

int h2gcn_xchg_export(const h2gcn_xchg_t* x, void* blob_out) {
    if (!x || !blob_out) return fail(H2GCN_ERR_INVALID_ARGUMENT, "NULL argument");
    Blob b;
    memset(&b, 0, sizeof(b));
    H2GCN_HIP_TRY(hipIpcGetMemHandle(&b.data, x->data));
    H2GCN_HIP_TRY(hipIpcGetMemHandle(&b.flags, x->flags));
    b.slot_bytes = x->slot_bytes;
    b.world = x->world;
    b.rank = x->rank;
    b.n_channels = x->n_channels;
    b.device = x->device;
    b.pid = (int64_t)getpid();
    b.magic = kMagic;
    memset(blob_out, 0, H2GCN_XCHG_BLOB_BYTES);
    memcpy(blob_out, &b, sizeof(b));
    return H2GCN_OK;
}

int h2gcn_xchg_connect(h2gcn_xchg_t* x, const void* blobs) {
    try {
        if (!x || !blobs) return fail(H2GCN_ERR_INVALID_ARGUMENT, "NULL argument");
        if (x->connected) return x->world == 1 ? H2GCN_OK : fail(H2GCN_ERR_INVALID_ARGUMENT, "already connected");
        for (int q = 0; q < x->world; ++q) {
            if (q == x->rank) continue;
            Blob b;
            memcpy(&b, (const char*)blobs + (size_t)q * H2GCN_XCHG_BLOB_BYTES, sizeof(b));
            if (b.magic != kMagic || b.rank != q || b.world != x->world || b.n_channels != x->n_channels ||
                b.slot_bytes != x->slot_bytes)
                return fail(H2GCN_ERR_INVALID_ARGUMENT, "blob %d does not describe a matching exchange (rank %d, world %d, "
                            "channels %d, slot %llu)", q, b.rank, b.world, b.n_channels, (unsigned long long)b.slot_bytes);
            if (b.pid == (int64_t)getpid())
                return fail(H2GCN_ERR_INVALID_ARGUMENT, "rank %d lives in this process: IPC handles can only be opened by "
                            "other processes (one rank per process)", q);
            void* pd = nullptr;
            void* pf = nullptr;
            H2GCN_HIP_TRY(hipIpcOpenMemHandle(&pd, b.data, hipIpcMemLazyEnablePeerAccess));
            x->opened.push_back(pd);
            H2GCN_HIP_TRY(hipIpcOpenMemHandle(&pf, b.flags, hipIpcMemLazyEnablePeerAccess));
            x->opened.push_back(pf);
            x->peers.data[q] = (char*)pd;
            x->peers.flags[q] = (uint32_t*)pf;
        }
        x->connected = true;
        return H2GCN_OK;
    } catch (...) {
        return fail(H2GCN_ERR_INTERNAL, "unexpected exception in xchg_connect");
    }
}

static int allgather_impl(h2gcn_xchg_t* x, int channel, const float* src, int64_t ld_src, int64_t rows,
                          int64_t rows_per_rank, int32_t width, float* full, void* stream_v, bool do_post, bool do_pull,
                          const HaloTable* halo = nullptr) {
    try {
        if (!x) return fail(H2GCN_ERR_INVALID_ARGUMENT, "exchange is NULL");
        if (!x->connected) return fail(H2GCN_ERR_INVALID_ARGUMENT, "exchange is not connected");
        if (channel < 0 || channel >= x->n_channels) return fail(H2GCN_ERR_INVALID_ARGUMENT, "channel %d outside 0..%d", channel, x->n_channels - 1);
        if (x->open_channel[channel]) return fail(H2GCN_ERR_INVALID_ARGUMENT, "channel %d: begin without a matching end", channel);
        if (width < 1 || rows < 0 || rows > rows_per_rank || ld_src < width || (!src && rows > 0) || !full)
            return fail(H2GCN_ERR_INVALID_ARGUMENT, "bad shard (rows %lld of %lld, width %d, ld %lld)", (long long)rows,
                        (long long)rows_per_rank, width, (long long)ld_src);
        const size_t bytes = (size_t)rows_per_rank * (size_t)width * 4;
        if (bytes > x->slot_bytes) return fail(H2GCN_ERR_INVALID_ARGUMENT, "shard of %zu bytes exceeds the slot (%zu)", bytes, x->slot_bytes);
        int cur = -1;
        H2GCN_HIP_TRY(hipGetDevice(&cur));
        if (cur != x->device) return fail(H2GCN_ERR_INVALID_ARGUMENT, "exchange lives on device %d, current device is %d", x->device, cur);
        hipStream_t stream = (hipStream_t)stream_v;
        if (!do_post) {  // second half of a split begin: the pulls of what allgather_post staged and announced
            if (x->world == 1 || bytes == 0) return H2GCN_OK;
            return issue_pulls(x, channel, x->seq[channel], 0, bytes, (char*)full, halo, rows_per_rank, width);
        }
        const uint32_t seq = ++x->seq[channel];   // host mirror (copy-engine mode; copy-kernel mode counts on the device)
        const SeqRef sr = seq_ref(x, channel, seq);
        float* own = full + (size_t)x->rank * (size_t)rows_per_rank * width;

        // everything enqueued on `stream` so far (e.g. the SpMM still reading `full`) precedes the pulls
        H2GCN_HIP_TRY(hipEventRecord(x->fence[channel], stream));
        // the slot is free once this channel's previous pulls are done (see the protocol note above)
        for (int q = 0; q < x->world; ++q)
            if (x->pulled_valid[(size_t)channel * x->world + q])
                H2GCN_HIP_TRY(hipStreamWaitEvent(stream, x->pulled[(size_t)channel * x->world + q], 0));
        if (x->summed_valid[channel]) H2GCN_HIP_TRY(hipStreamWaitEvent(stream, x->summed[channel], 0));
        if (rows_per_rank > 0) {
            const bool vec4 = width % 4 == 0 && ld_src % 4 == 0 && ((uintptr_t)src & 15u) == 0 && ((uintptr_t)own & 15u) == 0;
            const int64_t total = rows_per_rank * (vec4 ? width / 4 : width);
            const unsigned blocks = (unsigned)std::min<int64_t>((total + 255) / 256, 2048);
            // failure injection for the run-time gate of the IPC exchange (see "Visibility across devices"): from the second step
            // of a channel on, the slot is NOT updated -- the peers pull what an earlier step (or nobody) left there
            const int skip_slot = (x->inject_stale && seq >= 2) ? 1 : 0;
            if (vec4)
                hipLaunchKernelGGL(stage_kernel<true>, dim3(blocks), dim3(256), 0, stream, src, ld_src, rows, rows_per_rank,
                                   (int)width, x->data, x->slot_bytes, channel, sr, own, skip_slot);
            else
                hipLaunchKernelGGL(stage_kernel<false>, dim3(blocks), dim3(256), 0, stream, src, ld_src, rows, rows_per_rank,
                                   (int)width, x->data, x->slot_bytes, channel, sr, own, skip_slot);
            H2GCN_HIP_TRY(hipGetLastError());
        }
        if (x->world > 1) {
            hipLaunchKernelGGL(signal_kernel, dim3(1), dim3(64), 0, stream, x->peers, x->world, x->rank, channel, sr,
                               sr.dev ? x->seq_dev + channel : nullptr);
            H2GCN_HIP_TRY(hipGetLastError());
        }
        x->open_channel[channel] = do_pull ? 1 : 3;  // 3 = posted, pulls not issued yet
        if (x->world == 1 || bytes == 0 || !do_pull) return H2GCN_OK;

        return issue_pulls(x, channel, seq, 0, bytes, (char*)full);
    } catch (...) {
        return fail(H2GCN_ERR_INTERNAL, "unexpected exception in xchg_allgather_begin");
    }
}

int h2gcn_xchg_allgather_begin(h2gcn_xchg_t* x, int channel, const float* src, int64_t ld_src, int64_t rows,
                               int64_t rows_per_rank, int32_t width, float* full, void* stream) {
    return allgather_impl(x, channel, src, ld_src, rows, rows_per_rank, width, full, stream, true, true);
}

int h2gcn_xchg_allgather_post(h2gcn_xchg_t* x, int channel, const float* src, int64_t ld_src, int64_t rows,
                              int64_t rows_per_rank, int32_t width, float* full, void* stream) {
    return allgather_impl(x, channel, src, ld_src, rows, rows_per_rank, width, full, stream, true, false);
}

int h2gcn_xchg_allgather_pull(h2gcn_xchg_t* x, int channel, int64_t rows_per_rank, int32_t width, float* full) {
    if (!x) return fail(H2GCN_ERR_INVALID_ARGUMENT, "exchange is NULL");
    if (channel < 0 || channel >= x->n_channels) return fail(H2GCN_ERR_INVALID_ARGUMENT, "channel %d outside 0..%d", channel, x->n_channels - 1);
    if (x->open_channel[channel] != 3) return fail(H2GCN_ERR_INVALID_ARGUMENT, "channel %d: allgather_pull without allgather_post", channel);
    x->open_channel[channel] = 0;  // allgather_impl re-checks "not open"
    const int st = allgather_impl(x, channel, nullptr, width, 0, rows_per_rank, width, full, nullptr, false, true);
    x->open_channel[channel] = st == H2GCN_OK ? 1 : 3;  // a failed pull leaves the channel "posted": it can be retried
    return st;
}

int h2gcn_xchg_allgather_pull_rows(h2gcn_xchg_t* x, int channel, int64_t rows_per_rank, int32_t width, float* full,
                                   const int32_t* const* rows_dev, const int64_t* counts) {
    if (!x) return fail(H2GCN_ERR_INVALID_ARGUMENT, "exchange is NULL");
    if (channel < 0 || channel >= x->n_channels) return fail(H2GCN_ERR_INVALID_ARGUMENT, "channel %d outside 0..%d", channel, x->n_channels - 1);
    if (x->mode != H2GCN_XCHG_COPY_KERNEL) return fail(H2GCN_ERR_INVALID_ARGUMENT, "halo pulls exist in copy-kernel mode only");
    if (width < 4 || width % 4 != 0 || (reinterpret_cast<uintptr_t>(full) & 15u)) return fail(H2GCN_ERR_INVALID_ARGUMENT, "halo pulls move 16-byte pieces: width %d, full %p", width, (void*)full);
    if (!rows_dev || !counts) return fail(H2GCN_ERR_INVALID_ARGUMENT, "NULL row table");
    if (x->open_channel[channel] != 3) return fail(H2GCN_ERR_INVALID_ARGUMENT, "channel %d: allgather_pull_rows without allgather_post", channel);
    HaloTable halo;
    memset(&halo, 0, sizeof(halo));
    for (int q = 0; q < x->world; ++q) {
        if (q == x->rank) continue;
        if (counts[q] < 0 || counts[q] > rows_per_rank || (counts[q] > 0 && !rows_dev[q]))
            return fail(H2GCN_ERR_INVALID_ARGUMENT, "halo of peer %d: %lld rows (shard height %lld) or a NULL list", q, (long long)counts[q], (long long)rows_per_rank);
        halo.rows[q] = rows_dev[q];
        halo.count[q] = counts[q];
    }
    x->open_channel[channel] = 0;  // allgather_impl re-checks "not open"
    const int st = allgather_impl(x, channel, nullptr, width, 0, rows_per_rank, width, full, nullptr, false, true, &halo);
    x->open_channel[channel] = st == H2GCN_OK ? 1 : 3;
    return st;
}

int h2gcn_xchg_allgather_end(h2gcn_xchg_t* x, int channel, void* stream_v) {
    if (!x) return fail(H2GCN_ERR_INVALID_ARGUMENT, "exchange is NULL");
    if (channel < 0 || channel >= x->n_channels) return fail(H2GCN_ERR_INVALID_ARGUMENT, "channel %d outside 0..%d", channel, x->n_channels - 1);
    if (x->open_channel[channel] != 1) return fail(H2GCN_ERR_INVALID_ARGUMENT, "channel %d: allgather_end without its begin", channel);
    hipStream_t stream = (hipStream_t)stream_v;
    for (int q = 0; q < x->world; ++q)
        if (x->pulled_valid[(size_t)channel * x->world + q])
            H2GCN_HIP_TRY(hipStreamWaitEvent(stream, x->pulled[(size_t)channel * x->world + q], 0));
    x->open_channel[channel] = 0;
    return H2GCN_OK;
}

int h2gcn_xchg_reduce_scatter_begin(h2gcn_xchg_t* x, int channel, const float* src, int64_t rows_per_rank, int32_t width,
                                    float* out, void* stream_v) {
    try {
        if (!x) return fail(H2GCN_ERR_INVALID_ARGUMENT, "exchange is NULL");
        if (!x->connected) return fail(H2GCN_ERR_INVALID_ARGUMENT, "exchange is not connected");
        if (channel < 0 || channel >= x->n_channels) return fail(H2GCN_ERR_INVALID_ARGUMENT, "channel %d outside 0..%d", channel, x->n_channels - 1);
        if (x->open_channel[channel]) return fail(H2GCN_ERR_INVALID_ARGUMENT, "channel %d: begin without a matching end", channel);
        if (width < 1 || rows_per_rank < 0 || (!src && rows_per_rank > 0) || !out)
            return fail(H2GCN_ERR_INVALID_ARGUMENT, "bad reduce-scatter operands (rows per rank %lld, width %d)", (long long)rows_per_rank, width);
        const size_t block = (size_t)rows_per_rank * (size_t)width * 4;
        const size_t bytes = block * (size_t)x->world;
        if (bytes > x->slot_bytes) return fail(H2GCN_ERR_INVALID_ARGUMENT, "matrix of %zu bytes exceeds the slot (%zu)", bytes, x->slot_bytes);
        int cur = -1;
        H2GCN_HIP_TRY(hipGetDevice(&cur));
        if (cur != x->device) return fail(H2GCN_ERR_INVALID_ARGUMENT, "exchange lives on device %d, current device is %d", x->device, cur);
        hipStream_t stream = (hipStream_t)stream_v;
        if (x->rs_recv_bytes[channel] < bytes) {  // receive buffer of this channel (first use / growth)
            if (x->rs_recv[channel]) {
                H2GCN_HIP_TRY(hipDeviceSynchronize());
                (void)hipFree(x->rs_recv[channel]);
                x->rs_recv[channel] = nullptr;
                x->rs_recv_bytes[channel] = 0;
            }
            H2GCN_HIP_TRY(hipMalloc((void**)&x->rs_recv[channel], bytes ? bytes : 16));
            x->rs_recv_bytes[channel] = bytes;
        }
        const uint32_t seq = ++x->seq[channel];
        const SeqRef sr = seq_ref(x, channel, seq);
        const size_t slot_off = ((size_t)channel * 2 + (seq & 1u)) * x->slot_bytes;   // copy-engine mode only
        H2GCN_HIP_TRY(hipEventRecord(x->fence[channel], stream));
        for (int q = 0; q < x->world; ++q)
            if (x->pulled_valid[(size_t)channel * x->world + q])
                H2GCN_HIP_TRY(hipStreamWaitEvent(stream, x->pulled[(size_t)channel * x->world + q], 0));
        if (x->summed_valid[channel]) H2GCN_HIP_TRY(hipStreamWaitEvent(stream, x->summed[channel], 0));
        if (bytes > 0) {
            if (sr.dev) {   // the slot's parity is only known on the device
                const unsigned blocks = (unsigned)std::min<size_t>((bytes / 4 + 255) / 256, 4096);
                hipLaunchKernelGGL(copy_to_slot_kernel, dim3(blocks), dim3(256), 0, stream, src, bytes / 4, x->data, x->slot_bytes, channel, sr);
                H2GCN_HIP_TRY(hipGetLastError());
            } else {
                H2GCN_HIP_TRY(hipMemcpyAsync(x->data + slot_off, src, bytes, hipMemcpyDeviceToDevice, stream));
            }
        }
        if (x->world > 1) {
            hipLaunchKernelGGL(signal_kernel, dim3(1), dim3(64), 0, stream, x->peers, x->world, x->rank, channel, sr,
                               sr.dev ? x->seq_dev + channel : nullptr);
            H2GCN_HIP_TRY(hipGetLastError());
        } else if (sr.dev) {   // world 1: nobody to tell, but the device counter still names the slot the sum reads
            hipLaunchKernelGGL(signal_kernel, dim3(1), dim3(64), 0, stream, x->peers, 1, 0, channel, sr, x->seq_dev + channel);
            H2GCN_HIP_TRY(hipGetLastError());
        }
        x->open_channel[channel] = 2;
        x->rs_pending[channel] = h2gcn_xchg::RsPending{out, (const float*)(x->data + slot_off + (size_t)x->rank * block), block / 4};
        if (x->world == 1 || block == 0) return H2GCN_OK;
        // every peer's slot holds its whole matrix; this rank needs block `rank` of each
        return issue_pulls(x, channel, seq, (size_t)x->rank * block, block, (char*)x->rs_recv[channel]);
    } catch (...) {
        return fail(H2GCN_ERR_INTERNAL, "unexpected exception in xchg_reduce_scatter_begin");
    }
}

int h2gcn_xchg_reduce_scatter_end(h2gcn_xchg_t* x, int channel, void* stream_v) {
    if (!x) return fail(H2GCN_ERR_INVALID_ARGUMENT, "exchange is NULL");
    if (channel < 0 || channel >= x->n_channels) return fail(H2GCN_ERR_INVALID_ARGUMENT, "channel %d outside 0..%d", channel, x->n_channels - 1);
    if (x->open_channel[channel] != 2) return fail(H2GCN_ERR_INVALID_ARGUMENT, "channel %d: reduce_scatter_end without its begin", channel);
    hipStream_t stream = (hipStream_t)stream_v;
    for (int q = 0; q < x->world; ++q)
        if (x->pulled_valid[(size_t)channel * x->world + q])
            H2GCN_HIP_TRY(hipStreamWaitEvent(stream, x->pulled[(size_t)channel * x->world + q], 0));
    const h2gcn_xchg::RsPending& pd = x->rs_pending[channel];
    if (pd.block_elems > 0) {
        const unsigned blocks = (unsigned)std::min<size_t>((pd.block_elems + 255) / 256, 4096);
        hipLaunchKernelGGL(sum_blocks_kernel, dim3(blocks), dim3(256), 0, stream, (const float*)x->rs_recv[channel], pd.own, x->data,
                           x->slot_bytes, channel, x->mode == H2GCN_XCHG_COPY_KERNEL ? (const uint32_t*)(x->seq_dev + channel) : nullptr,
                           x->world, x->rank, pd.block_elems, pd.out);
        H2GCN_HIP_TRY(hipGetLastError());
        // the sum reads this rank's own slot: it must be finished before the slot is staged again -> the next begin
        // on this channel (any mode, any stream) waits for this event
        H2GCN_HIP_TRY(hipEventRecord(x->summed[channel], stream));
        x->summed_valid[channel] = 1;
    }
    x->open_channel[channel] = 0;
    return H2GCN_OK;
}

int h2gcn_xchg_reset_dependencies(h2gcn_xchg_t* x) {
    if (!x) return fail(H2GCN_ERR_INVALID_ARGUMENT, "exchange is NULL");
    for (char c : x->open_channel)
        if (c) return fail(H2GCN_ERR_INVALID_ARGUMENT, "a channel is between begin and end");
    std::fill(x->pulled_valid.begin(), x->pulled_valid.end(), 0);
    std::fill(x->summed_valid.begin(), x->summed_valid.end(), 0);
    return H2GCN_OK;
}

int h2gcn_xchg_status(const h2gcn_xchg_t* x) {
    if (!x) return fail(H2GCN_ERR_INVALID_ARGUMENT, "exchange is NULL");
    if (*(volatile int*)x->err)
        return fail(H2GCN_ERR_EXCHANGE_TIMEOUT, "rank %d: a peer's shard did not arrive in time (peer missing, or ranks "
                    "issuing different exchange sequences)", x->rank);
    return H2GCN_OK;
}

void h2gcn_xchg_destroy(h2gcn_xchg_t* x) { release(x); }

}  // extern "C"
