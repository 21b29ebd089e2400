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
    double *const *inbox;          // [world] device-resident table: every rank's inbox as mapped in this process ([rank] = the local one); a table in
                                   // memory, not an array in the kernel arguments: indexing those by lane would put them in scratch
    unsigned int *err;             // waits that gave up (device word; non-zero makes all later waits fall through)
    unsigned int *err_host;        // the same verdict in pinned host memory: the host reads it after a stream sync, without a copy
    int light;                     // 0: plain accesses + system-scope fences; 1: sc0 sc1 accesses ordered by s_waitcnt (see below)
    unsigned long long seq;        // sequence number of this exchange (1, 2, ...); parity selects the line set
    unsigned long long budget;     // wait budget in 100 MHz wall-clock ticks
    int rank, world;
};

__device__ __forceinline__ double *p2p_line(double *inbox, int par, int world, int sender, int row) {
    return inbox + (((size_t)par * world + sender) * kDecision + row) * kP2PLineDoubles;
}

// One WAVEFRONT exchanges up to four infoset rows: lane = 16*rl + q serves row (row0 + rl) towards peer q and from sender
// q, so the `world` remote stores, the `world` polls and the `world` line loads of a row each cost one memory round trip,
// not `world`.  d = this rank's delta of the lane's row (the same values in the 16 lanes of a row); xch = LDS scratch
// [kP2PMaxWorld][5] of the lane's row.  On return d is the sum over ranks, added in rank order (identical bits on every
// rank), in all 16 lanes of the row.  Must be called by all 64 lanes of the wavefront.
//
// Two forms of the same protocol, chosen per exchange by P2PArgs.light (wave-uniform).  light = 0 is the textbook one and
// the library's default: plain stores, __threadfence_system() (L2 write-back + invalidate at system scope), release-store of
// the sequence word; poll, system-scope acquire fence, plain loads.  light = 1 makes every access to the line a system-scope
// (sc0 sc1) access -- inbox memory is fine-grained, such accesses go to memory, not through this XCD's L2 -- ordered by
// s_waitcnt alone: the sequence word is issued only after the data stores of the same line were acknowledged (vmcnt(0)),
// and the data loads are issued only after the sequence word was seen.  No cache maintenance, so the kernel's L2 contents
// (group tables, solver tables) are left alone; it relies on the fabric acknowledging a remote store only once it is visible at its
// destination, which is why callers switch it on only after validating it on their topology (scopa_p2p_set_form).

typedef double p2p_v2f64 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void p2p_exchange_wave4(const P2PArgs &a, int row, bool row_valid, int q, double (&d)[5], double (*xch)[5]) {
    const int par = (int)(a.seq & 1ull);
    if (row_valid && q < a.world) {
        double *out = p2p_line(a.inbox[q], par, a.world, a.rank, row);        // my row, in peer q's inbox
        const double *in = p2p_line(a.inbox[a.rank], par, a.world, q, row);   // sender q's row, in my inbox
        double v[5];
        if (a.light) {
            for (int k = 0; k < 5; k++) __hip_atomic_store(out + k, d[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // the write-through stores are acknowledged (gfx9: vmcnt counts stores)
            __hip_atomic_store(reinterpret_cast<unsigned long long *>(out + 5), a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        } else {
            reinterpret_cast<double2 *>(out)[0] = make_double2(d[0], d[1]);
            reinterpret_cast<double2 *>(out)[1] = make_double2(d[2], d[3]);
            out[4] = d[4];
            __threadfence_system();
            __hip_atomic_store(reinterpret_cast<unsigned long long *>(out + 5), a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        const unsigned long long t0 = wall_clock64();
        while (__hip_atomic_load(reinterpret_cast<const unsigned long long *>(in + 5), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < a.seq) {
            if (__hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;   // an earlier wait already gave up
            if (wall_clock64() - t0 > a.budget) {
                atomicAdd(a.err, 1u);
                __hip_atomic_store(a.err_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        if (a.light) {
            p2p_v2f64 lo, hi;
            double last;
            asm volatile("global_load_dwordx4 %0, %3, off sc0 sc1\n\t"
                         "global_load_dwordx4 %1, %3, off offset:16 sc0 sc1\n\t"
                         "global_load_dwordx2 %2, %3, off offset:32 sc0 sc1\n\t"
                         "s_waitcnt vmcnt(0)"
                         : "=&v"(lo), "=&v"(hi), "=&v"(last) : "v"(in) : "memory");
            v[0] = lo.x; v[1] = lo.y; v[2] = hi.x; v[3] = hi.y; v[4] = last;
        } else {
            __atomic_thread_fence(__ATOMIC_ACQUIRE);   // system scope: the row as the peer published it
            const double2 lo = reinterpret_cast<const double2 *>(in)[0], hi = reinterpret_cast<const double2 *>(in)[1];
            v[0] = lo.x; v[1] = lo.y; v[2] = hi.x; v[3] = hi.y; v[4] = in[4];
        }
        for (int k = 0; k < 5; k++) xch[q][k] = v[k];
    }
    // the wavefront's own LDS hand-off: writes above, broadcast reads below
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (row_valid) {
        for (int k = 0; k < 5; k++) d[k] = xch[0][k];
        for (int s = 1; s < a.world; s++)
            for (int k = 0; k < 5; k++) d[k] += xch[s][k];   // rank order, everywhere
    }
}

// host side (scopa_p2p.hip): arguments of the NEXT exchange (increments the sequence number); false if not connected
bool p2p_next_args(struct ::scopa_ctx *ctx, P2PArgs *out);
// host side, after a stream sync: SCOPA_ETIMEOUT (and ctx->err) if any wait of the context's exchange ever gave up
int32_t p2p_check(struct ::scopa_ctx *ctx, const char *where);

}  // namespace scopa
