// scopa_p2p.h -- device side of the peer-memory exchange (protocol and host side: scopa_p2p.hip).
//
// Inbox of a rank (fine-grained device memory, mapped by every peer): lines of 64 bytes, one per (parity, sender, infoset
// row): float64 d[5] (4 regret deltas + visit count) | uint64 sequence number | 16 bytes unused.  A sender stores the 40
// data bytes of its row into that line on EVERY peer, fences at system scope, then release-stores the sequence number
// into the same lines; a receiver polls the `world` sequence words of its row, fences (acquire), and adds the rows in
// RANK ORDER.  Rows are independent: no kernel-wide or grid-wide rendezvous is needed, so the exchange can sit inside the
// reduce+apply kernel, between "this rank's delta of my 4 rows is known" and "apply it".
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "scopa_rules.h"

namespace scopa {

constexpr int kP2PMaxWorld = 16;
constexpr int kP2PLineDoubles = 8;

struct P2PArgs {
    double *inbox[kP2PMaxWorld];   // every rank's inbox as mapped in this process ([rank] = the local one)
    unsigned int *err;             // waits that gave up (device word; non-zero makes all later waits fall through)
    unsigned long long seq;        // sequence number of this exchange (1, 2, ...); parity selects the line set
    unsigned long long budget;     // wait budget in 100 MHz wall-clock ticks
    int rank, world;
};

__device__ __forceinline__ double *p2p_line(double *inbox, int par, int world, int sender, int row) {
    return inbox + (((size_t)par * world + sender) * kDecision + row) * kP2PLineDoubles;
}

// d[5] of infoset row `row` on this rank -> sum over ranks, added in rank order (identical bits on every rank).
__device__ __forceinline__ void p2p_exchange_row(const P2PArgs &a, int row, double (&d)[5]) {
    const int par = (int)(a.seq & 1ull);
    for (int p = 0; p < a.world; p++) {
        double *ln = p2p_line(a.inbox[p], par, a.world, a.rank, row);
        reinterpret_cast<double2 *>(ln)[0] = make_double2(d[0], d[1]);
        reinterpret_cast<double2 *>(ln)[1] = make_double2(d[2], d[3]);
        ln[4] = d[4];
    }
    __threadfence_system();   // the row is visible system-wide before any peer can see its sequence number
    for (int p = 0; p < a.world; p++)
        __hip_atomic_store(reinterpret_cast<unsigned long long *>(p2p_line(a.inbox[p], par, a.world, a.rank, row) + 5), a.seq,
                           __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    double *mine = a.inbox[a.rank];
    const unsigned long long t0 = wall_clock64();
    for (;;) {
        bool all = true;
        for (int q = 0; q < a.world; q++)   // the loads are independent: one memory round trip per poll, not `world`
            all = all && __hip_atomic_load(reinterpret_cast<unsigned long long *>(p2p_line(mine, par, a.world, q, row) + 5),
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) >= a.seq;
        if (all) break;
        if (__hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;   // an earlier wait already gave up
        if (wall_clock64() - t0 > a.budget) { atomicAdd(a.err, 1u); break; }
        __builtin_amdgcn_s_sleep(2);
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);   // system scope: the rows as the peers published them
    double2 lo[kP2PMaxWorld], hi[kP2PMaxWorld];
    double cnt[kP2PMaxWorld];
    for (int q = 0; q < a.world; q++) {
        const double *ln = p2p_line(mine, par, a.world, q, row);
        lo[q] = reinterpret_cast<const double2 *>(ln)[0]; hi[q] = reinterpret_cast<const double2 *>(ln)[1]; cnt[q] = ln[4];
    }
    d[0] = lo[0].x; d[1] = lo[0].y; d[2] = hi[0].x; d[3] = hi[0].y; d[4] = cnt[0];
    for (int q = 1; q < a.world; q++) { d[0] += lo[q].x; d[1] += lo[q].y; d[2] += hi[q].x; d[3] += hi[q].y; d[4] += cnt[q]; }
}

// host side (scopa_p2p.hip): arguments of the NEXT exchange (increments the sequence number); false if not connected
bool p2p_next_args(struct ::scopa_ctx *ctx, P2PArgs *out);

}  // namespace scopa
