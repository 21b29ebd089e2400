// scopa_p2p.hip -- one-shot all-reduce of the MCCFR delta over xGMI peer memory, for the N > 1 path.
//
// The per-iteration exchange is 29.5 KB (n_infosets x 5 float64): far below the size at which a ring collective's
// bandwidth matters and dominated by its latency.  With one process per GPU on one node every rank can map every peer's
// buffer (hipIpc*), and the exchange is done row by row (scopa_p2p.h: p2p_exchange_wave4): the owner of an infoset row stores
// its 40 bytes into a 64-byte line of every peer's inbox, fences at system scope, release-stores the sequence number into
// the same lines, waits (bounded) for the `world` sequence words of its row and adds the rows IN RANK ORDER -- the same
// order on every rank, so the replicas' tables stay bit-identical and the sum is reproducible run to run (a ring's order
// is not ours to fix).  Rows are independent, so the exchange sits INSIDE the apply kernel
// (k_mccfr_exchange_apply, scopa_mccfr.hip): an N > 1 iteration is the same two launches as a single-GPU one.
// k_p2p_rows is the stand-alone form on the delta buffer (scopa_p2p_allreduce_delta: validation and the split path).
// Inboxes are fine-grained device memory (coherent at system scope inside a kernel).  Lines are double-buffered by
// sequence parity: a peer can only be one exchange ahead (its exchange s+1 of a row needs this rank's row of s+1, which is
// launched after this rank's kernel of exchange s has finished), so a line of parity s & 1 is never overwritten while
// it is being read.
// Every wait is bounded by a wall-clock budget; a timeout is counted in an error word, sets a pinned host word that makes
// scopa_p2p_allreduce_delta / scopa_mccfr_iterate_sharded fail with SCOPA_ETIMEOUT (p2p_check), and makes all later waits fall
// through, so a dead peer can never hang the GPU nor go unnoticed.
// The reference has no distributed code; torch.distributed (RCCL) remains the portable path and the check
// (scopa_amd/distributed.py validates this exchange against it before using it).
#include <string.h>

#include <cstdio>
#include <cstdlib>

#include "scopa_ctx.h"
#include "scopa_p2p.h"

using scopa::fail;
using scopa::kP2PMaxWorld;
using scopa::P2PArgs;

struct scopa_p2p {
    int rank = 0, world = 0;
    void *local = nullptr;
    void *peer[kP2PMaxWorld] = {nullptr};
    bool opened[kP2PMaxWorld] = {false};
    size_t bytes = 0;
    unsigned long long seq = 0;
    unsigned int *d_err = nullptr;
    unsigned int *h_err = nullptr;          // pinned host word, set by the first wait that gives up ...
    unsigned int *h_err_dev = nullptr;      // ... and its device-side address
    double **d_inbox_tab = nullptr;         // peer[] on the device (P2PArgs.inbox)
    unsigned long long budget = 500000000ull;  // 5 s of the 100 MHz wall clock
    int light = 0;
    bool connected = false;
};

// stand-alone exchange of the delta buffer: one wavefront per 4 infoset rows (lane = 16 * row + peer)
__global__ void __launch_bounds__(64) k_p2p_rows(P2PArgs a, double *__restrict__ delta, int n_rows) {
    __shared__ double xch[4][kP2PMaxWorld][5];
    const int rl = threadIdx.x >> 4, q = threadIdx.x & 15;
    const int r = blockIdx.x * 4 + rl;
    const bool valid = r < n_rows;
    double d[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    if (valid) for (int k = 0; k < 5; k++) d[k] = delta[r * 5 + k];
    scopa::p2p_exchange_wave4(a, r, valid, q, d, xch[rl]);
    if (valid && q == 0) for (int k = 0; k < 5; k++) delta[r * 5 + k] = d[k];
}

namespace scopa {
void p2p_release(scopa_ctx *ctx) {
    scopa_p2p *p = ctx->p2p;
    if (!p) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (int r = 0; r < p->world; r++) if (p->opened[r] && p->peer[r]) (void)hipIpcCloseMemHandle(p->peer[r]);
    if (p->local) (void)hipFree(p->local);
    if (p->d_err) (void)hipFree(p->d_err);
    if (p->h_err) (void)hipHostFree(p->h_err);
    if (p->d_inbox_tab) (void)hipFree(p->d_inbox_tab);
    delete p;
    ctx->p2p = nullptr;
}

// after the stream has been synchronised: did any wait of this exchange object give up?
int32_t p2p_check(scopa_ctx *ctx, const char *where) {
    scopa_p2p *p = ctx->p2p;
    if (p && p->h_err && *static_cast<volatile unsigned int *>(p->h_err) != 0u) {
        snprintf(ctx->err, sizeof ctx->err, "%s: a peer did not answer within %.3f s; sums were applied without it -- the tables of this "
                 "context are invalid (re-create the exchange and restore the tables)", where, (double)p->budget * 1e-8);
        return SCOPA_ETIMEOUT;
    }
    return SCOPA_OK;
}

bool p2p_next_args(scopa_ctx *ctx, P2PArgs *out) {
    scopa_p2p *p = ctx->p2p;
    if (!p || !p->connected) return false;
    out->inbox = p->d_inbox_tab;
    out->err = p->d_err;
    out->err_host = p->h_err_dev;
    out->light = p->light;
    out->seq = ++p->seq;
    out->budget = p->budget;
    out->rank = p->rank; out->world = p->world;
    return true;
}
}  // namespace scopa

extern "C" {

int32_t scopa_p2p_create(scopa_ctx *ctx, int32_t rank, int32_t world, uint8_t handle_out[64]) {
    if (!ctx || !handle_out || world < 1 || world > kP2PMaxWorld || rank < 0 || rank >= world) return SCOPA_EINVAL;
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is expected to be 64 bytes");
    SC_HIP(ctx, hipSetDevice(ctx->device));
    scopa::p2p_release(ctx);
    scopa_p2p *p = new (std::nothrow) scopa_p2p();
    if (!p) return SCOPA_ENOMEM;
    p->rank = rank; p->world = world;
    p->bytes = (size_t)2 * world * scopa::kDecision * scopa::kP2PLineDoubles * sizeof(double);   // 211 KB per rank of the group
    ctx->p2p = p;
    hipError_t e = hipExtMallocWithFlags(&p->local, p->bytes, hipDeviceMallocFinegrained);
    if (e != hipSuccess) { scopa::p2p_release(ctx); return fail(ctx, SCOPA_EHIP, "scopa_p2p_create: hipExtMallocWithFlags(fine-grained)", e); }
    if ((e = hipMalloc(&p->d_err, 64)) != hipSuccess) { scopa::p2p_release(ctx); return fail(ctx, SCOPA_EHIP, "scopa_p2p_create: hipMalloc", e); }
    if ((e = hipHostMalloc(reinterpret_cast<void **>(&p->h_err), 64, hipHostMallocMapped)) != hipSuccess) { scopa::p2p_release(ctx); return fail(ctx, SCOPA_EHIP, "scopa_p2p_create: hipHostMalloc", e); }
    memset(p->h_err, 0, 64);
    if ((e = hipHostGetDevicePointer(reinterpret_cast<void **>(&p->h_err_dev), p->h_err, 0)) != hipSuccess) { scopa::p2p_release(ctx); return fail(ctx, SCOPA_EHIP, "scopa_p2p_create: hipHostGetDevicePointer", e); }
    if ((e = hipMemsetAsync(p->local, 0, p->bytes, ctx->stream)) != hipSuccess || (e = hipMemsetAsync(p->d_err, 0, 64, ctx->stream)) != hipSuccess ||
        (e = hipStreamSynchronize(ctx->stream)) != hipSuccess) { scopa::p2p_release(ctx); return fail(ctx, SCOPA_EHIP, "scopa_p2p_create: clearing the inbox", e); }
    hipIpcMemHandle_t h;
    if ((e = hipIpcGetMemHandle(&h, p->local)) != hipSuccess) { scopa::p2p_release(ctx); return fail(ctx, SCOPA_EHIP, "scopa_p2p_create: hipIpcGetMemHandle", e); }
    memcpy(handle_out, &h, 64);
    p->peer[rank] = p->local;
    return SCOPA_OK;
}

int32_t scopa_p2p_connect(scopa_ctx *ctx, const uint8_t *handles) {
    if (!ctx || !handles) return SCOPA_EINVAL;
    scopa_p2p *p = ctx->p2p;
    SC_REQUIRE(ctx, p != nullptr, SCOPA_ESTATE, "scopa_p2p_connect: call scopa_p2p_create first");
    SC_HIP(ctx, hipSetDevice(ctx->device));
    for (int r = 0; r < p->world; r++) {
        if (r == p->rank || p->opened[r]) continue;
        hipIpcMemHandle_t h;
        memcpy(&h, handles + (size_t)r * 64, 64);
        void *ptr = nullptr;
        const hipError_t e = hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) {   // first contact with another device's memory happens unattended (the driver's multi-GPU run): name the call and the pair
            char what[192];
            int peer_access = -1;
            (void)hipGetLastError();
            snprintf(what, sizeof what, "scopa_p2p_connect: hipIpcOpenMemHandle(rank %d's inbox) on rank %d / device %d (HSA_ENABLE_IPC_MODE_LEGACY=%s; same-process peer access to device %d: %s)",
                     r, p->rank, ctx->device, getenv("HSA_ENABLE_IPC_MODE_LEGACY") ? getenv("HSA_ENABLE_IPC_MODE_LEGACY") : "unset", r,
                     (r < 64 && hipDeviceCanAccessPeer(&peer_access, ctx->device, r) == hipSuccess) ? (peer_access ? "possible" : "not possible") : "unknown");
            return fail(ctx, SCOPA_EHIP, what, e);
        }
        p->peer[r] = ptr; p->opened[r] = true;
    }
    if (!p->d_inbox_tab) SC_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&p->d_inbox_tab), kP2PMaxWorld * sizeof(double *)));
    double *tab[kP2PMaxWorld] = {nullptr};
    for (int r = 0; r < p->world; r++) tab[r] = static_cast<double *>(p->peer[r]);
    SC_HIP(ctx, hipMemcpyAsync(p->d_inbox_tab, tab, sizeof tab, hipMemcpyHostToDevice, ctx->stream));
    SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    p->connected = true;
    return SCOPA_OK;
}

int32_t scopa_p2p_allreduce_delta(scopa_ctx *ctx) {
    if (!ctx) return SCOPA_EINVAL;
    SC_REQUIRE(ctx, ctx->p2p && ctx->p2p->connected, SCOPA_ESTATE, "scopa_p2p_allreduce_delta: exchange not connected");
    SC_REQUIRE(ctx, ctx->has_deal, SCOPA_ESTATE, "scopa_p2p_allreduce_delta: no deal set");
    SC_HIP(ctx, hipSetDevice(ctx->device));
    P2PArgs a;
    scopa::p2p_next_args(ctx, &a);
    hipLaunchKernelGGL(k_p2p_rows, dim3((ctx->n_infosets + 3) / 4), dim3(64), 0, ctx->stream, a, ctx->d_delta, ctx->n_infosets);
    SC_HIP(ctx, hipGetLastError());
    SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return scopa::p2p_check(ctx, "scopa_p2p_allreduce_delta");
}

int32_t scopa_p2p_set_form(scopa_ctx *ctx, int32_t light) {
    if (!ctx || (light != 0 && light != 1)) return SCOPA_EINVAL;
    SC_REQUIRE(ctx, ctx->p2p != nullptr, SCOPA_ESTATE, "scopa_p2p_set_form: no exchange");
    ctx->p2p->light = light;
    return SCOPA_OK;
}

int32_t scopa_p2p_set_budget(scopa_ctx *ctx, double seconds) {
    if (!ctx || !(seconds >= 1e-3 && seconds <= 60.0)) return SCOPA_EINVAL;
    SC_REQUIRE(ctx, ctx->p2p != nullptr, SCOPA_ESTATE, "scopa_p2p_set_budget: no exchange");
    ctx->p2p->budget = (unsigned long long)(seconds * 1e8);
    return SCOPA_OK;
}

int32_t scopa_p2p_status(scopa_ctx *ctx, int32_t *timeouts, uint64_t *exchanges) {
    if (!ctx) return SCOPA_EINVAL;
    scopa_p2p *p = ctx->p2p;
    SC_REQUIRE(ctx, p != nullptr, SCOPA_ESTATE, "scopa_p2p_status: no exchange");
    unsigned int e = 0;
    SC_HIP(ctx, hipMemcpyAsync(&e, p->d_err, 4, hipMemcpyDeviceToHost, ctx->stream));
    SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (timeouts) *timeouts = (int32_t)e;
    if (exchanges) *exchanges = p->seq;
    return SCOPA_OK;
}

int32_t scopa_p2p_destroy(scopa_ctx *ctx) {
    if (!ctx) return SCOPA_EINVAL;
    scopa::p2p_release(ctx);
    return SCOPA_OK;
}

}  // extern "C"
