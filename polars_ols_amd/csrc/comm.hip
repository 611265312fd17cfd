// comm.hip -- multi-GPU behind the C-ABI: group partitioning and the one exchange step of the path (re-assembling an output
// column) over RCCL / xGMI.  See the pols_comm_* block of include/pols_mi355x.h.
//
// Groups are independent in the reference (every plugin call sees one group's rows; no cross-group state anywhere in
// src/least_squares.rs), so the data path has no collective: a GPU owns a contiguous range of groups and runs the same kernels
// on it.  What remains is what Polars does on the host when it concatenates the per-group outputs: gathering the per-group
// coefficient table (small) or, on request, a per-row column.  RCCL is bound at run time (dlopen of librccl.so.1 -- the
// instance the process already holds, e.g. PyTorch's, is reused), so the library itself loads on hosts without RCCL and
// single-GPU callers never touch it.
#include <dlfcn.h>

#include <algorithm>
#include <mutex>

#include "common.hpp"

namespace pols {

// the slice of the NCCL / RCCL API this file uses (rccl.h: ncclResult_t, ncclUniqueId, ncclDataType_t::ncclChar == 0)
struct NcclUniqueId { char internal[128]; };
typedef void *NcclComm;
struct Rccl {
    void *handle = nullptr;
    int (*GetUniqueId)(NcclUniqueId *) = nullptr;
    int (*CommInitRank)(NcclComm *, int, NcclUniqueId, int) = nullptr;
    int (*CommInitAll)(NcclComm *, int, const int *) = nullptr;
    int (*CommDestroy)(NcclComm) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, NcclComm, hipStream_t) = nullptr;
    int (*Broadcast)(const void *, void *, size_t, int, int, NcclComm, hipStream_t) = nullptr;
    int (*Send)(const void *, size_t, int, int, NcclComm, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, NcclComm, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    int (*CommCount)(NcclComm, int *) = nullptr;
    int (*CommUserRank)(NcclComm, int *) = nullptr;
    int (*CommCuDevice)(NcclComm, int *) = nullptr;
    int (*GetVersion)(int *) = nullptr;
};

static Rccl g_rccl;
static std::once_flag g_rccl_once;
static char g_rccl_err[256] = "";

static void rccl_bind() {
    // an instance already mapped into the process first (RTLD_NOLOAD), then the ROCm installation's
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names)
        if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
    for (int i = 0; !h && i < 3; ++i) h = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
    if (!h) { std::snprintf(g_rccl_err, sizeof(g_rccl_err), "cannot load librccl.so.1: %s", dlerror()); return; }
    Rccl r;
    r.handle = h;
#define BIND(field, sym)                                                                                   \
    r.field = reinterpret_cast<decltype(r.field)>(dlsym(h, sym));                                          \
    if (!r.field) { std::snprintf(g_rccl_err, sizeof(g_rccl_err), "librccl lacks %s", sym); return; }
    BIND(GetUniqueId, "ncclGetUniqueId") BIND(CommInitRank, "ncclCommInitRank") BIND(CommInitAll, "ncclCommInitAll")
    BIND(CommDestroy, "ncclCommDestroy") BIND(AllGather, "ncclAllGather") BIND(Broadcast, "ncclBroadcast")
    BIND(Send, "ncclSend") BIND(Recv, "ncclRecv") BIND(GroupStart, "ncclGroupStart") BIND(GroupEnd, "ncclGroupEnd")
    BIND(GetErrorString, "ncclGetErrorString")
    BIND(CommCount, "ncclCommCount") BIND(CommUserRank, "ncclCommUserRank") BIND(CommCuDevice, "ncclCommCuDevice") BIND(GetVersion, "ncclGetVersion")
#undef BIND
    g_rccl = r;
}

static int rccl_ready() {
    std::call_once(g_rccl_once, rccl_bind);
    if (!g_rccl.handle) return fail(POLS_ERR_UNSUPPORTED, "RCCL is not available: %s", g_rccl_err);
    return POLS_OK;
}

#define POLS_NCCL(call)                                                                                              \
    do {                                                                                                             \
        const int r_ = (call);                                                                                       \
        if (r_ != 0) return ::pols::fail(POLS_ERR_HIP, "%s failed: %s (%s:%d)", #call, g_rccl.GetErrorString(r_), __FILE__, __LINE__); \
    } while (0)

}  // namespace pols

using namespace pols;

struct pols_comm {
    pols_ctx *ctx = nullptr;
    NcclComm comm = nullptr;
    int world = 1, rank = 0;
};

extern "C" {

int pols_partition_groups(const int64_t *group_offsets, int64_t n_groups, int world_size, int64_t *bounds_out) {
    if (!group_offsets || !bounds_out || n_groups < 0 || world_size < 1) return fail(POLS_ERR_INVALID, "bad partition arguments");
    // contiguous ranges with near-equal ROW counts: boundary r is the first group boundary whose cumulative row count reaches
    // r / world of the rows -- a pure function of the offsets, so every rank (or every device thread) computes the same partition
    const int64_t total = group_offsets[n_groups];
    bounds_out[0] = 0;
    for (int r = 1; r < world_size; ++r) {
        const double target = (double)total * (double)r / (double)world_size;
        const int64_t *it = std::lower_bound(group_offsets, group_offsets + n_groups + 1, target,
                                             [](int64_t v, double t) { return (double)v < t; });
        int64_t g = it - group_offsets;
        g = std::min(std::max(g, bounds_out[r - 1]), n_groups);
        bounds_out[r] = g;
    }
    bounds_out[world_size] = n_groups;
    return POLS_OK;
}

int pols_comm_unique_id(void *id_out) {
    if (!id_out) return fail(POLS_ERR_INVALID, "id_out is NULL");
    int rc = rccl_ready();
    if (rc) return rc;
    NcclUniqueId id;
    POLS_NCCL(g_rccl.GetUniqueId(&id));
    std::memcpy(id_out, &id, sizeof(id));
    return POLS_OK;
}

int pols_comm_create(pols_ctx *ctx, const void *id, int world_size, int rank, pols_comm **out) {
    if (!ctx || !id || !out) return fail(POLS_ERR_INVALID, "ctx / id / out is NULL");
    if (world_size < 1 || rank < 0 || rank >= world_size) return fail(POLS_ERR_INVALID, "rank %d outside [0, %d)", rank, world_size);
    *out = nullptr;
    int rc = rccl_ready();
    if (rc) return rc;
    POLS_HIP(hipSetDevice(ctx->device));
    NcclUniqueId uid;
    std::memcpy(&uid, id, sizeof(uid));
    NcclComm c = nullptr;
    POLS_NCCL(g_rccl.CommInitRank(&c, world_size, uid, rank));
    pols_comm *pc = new pols_comm();
    pc->ctx = ctx; pc->comm = c; pc->world = world_size; pc->rank = rank;
    *out = pc;
    return POLS_OK;
}

int pols_comm_create_all(pols_ctx *const *ctxs, int n, pols_comm **out) {
    if (!ctxs || !out || n < 1 || n > 64) return fail(POLS_ERR_INVALID, "bad arguments");
    int rc = rccl_ready();
    if (rc) return rc;
    std::vector<int> devs(n);
    for (int i = 0; i < n; ++i) {
        if (!ctxs[i]) return fail(POLS_ERR_INVALID, "ctxs[%d] is NULL", i);
        devs[i] = ctxs[i]->device;
        out[i] = nullptr;
    }
    std::vector<NcclComm> comms(n, nullptr);
    POLS_NCCL(g_rccl.CommInitAll(comms.data(), n, devs.data()));
    for (int i = 0; i < n; ++i) {
        pols_comm *pc = new pols_comm();
        pc->ctx = ctxs[i]; pc->comm = comms[i]; pc->world = n; pc->rank = i;
        out[i] = pc;
    }
    return POLS_OK;
}

void pols_comm_destroy(pols_comm *comm) {
    if (!comm) return;
    if (comm->comm && g_rccl.CommDestroy) {
        hipSetDevice(comm->ctx->device);
        g_rccl.CommDestroy(comm->comm);
    }
    delete comm;
}

int pols_comm_world_size(const pols_comm *comm) { return comm ? comm->world : -1; }
int pols_comm_rank(const pols_comm *comm) { return comm ? comm->rank : -1; }

int pols_comm_query(const pols_comm *comm, pols_comm_info *out) {
    if (!comm || !out) return fail(POLS_ERR_INVALID, "comm / out is NULL");
    std::memset(out, 0, sizeof(*out));
    int rc = rccl_ready();
    if (rc) return rc;
    int v = 0;
    POLS_NCCL(g_rccl.CommCount(comm->comm, &v)); out->nranks_seen = v;
    POLS_NCCL(g_rccl.CommUserRank(comm->comm, &v)); out->rank_seen = v;
    POLS_NCCL(g_rccl.CommCuDevice(comm->comm, &v)); out->device = v;
    POLS_NCCL(g_rccl.GetVersion(&v)); out->rccl_version = v;
    POLS_HIP(hipDeviceGetPCIBusId(out->pci_bus_id, (int)sizeof(out->pci_bus_id), out->device));
    return POLS_OK;
}

int pols_comm_group_begin(void) {
    int rc = rccl_ready();
    if (rc) return rc;
    POLS_NCCL(g_rccl.GroupStart());
    return POLS_OK;
}

int pols_comm_group_end(void) {
    int rc = rccl_ready();
    if (rc) return rc;
    POLS_NCCL(g_rccl.GroupEnd());
    return POLS_OK;
}

static int comm_check(pols_comm *comm, const int64_t *counts, int64_t row_bytes) {
    if (!comm || !counts) return fail(POLS_ERR_INVALID, "comm / counts is NULL");
    if (row_bytes < 1) return fail(POLS_ERR_INVALID, "row_bytes must be positive");
    for (int r = 0; r < comm->world; ++r)
        if (counts[r] < 0) return fail(POLS_ERR_INVALID, "counts[%d] is negative", r);
    POLS_HIP(hipSetDevice(comm->ctx->device));
    return POLS_OK;
}

int pols_comm_allgather_rows(pols_comm *comm, const void *local, const int64_t *counts, int64_t row_bytes, void *out) {
    int rc = comm_check(comm, counts, row_bytes);
    if (rc) return rc;
    if (!out || (!local && counts[comm->rank])) return fail(POLS_ERR_INVALID, "local / out is NULL");
    hipStream_t st = comm->ctx->stream;
    bool equal = true;
    for (int r = 1; r < comm->world; ++r) equal = equal && counts[r] == counts[0];
    if (equal) {   // one ring all-gather: every shard the same size (the benchmark's weak-scaling case)
        if (counts[0] == 0) return POLS_OK;
        POLS_NCCL(g_rccl.AllGather(local, out, (size_t)(counts[0] * row_bytes), /*ncclChar*/ 0, comm->comm, st));
        return POLS_OK;
    }
    // uneven shards (balanced by rows, not by groups): one broadcast per owner inside ONE group call -- an all-gatherv
    POLS_NCCL(g_rccl.GroupStart());
    int64_t off = 0;
    for (int r = 0; r < comm->world; ++r) {
        const size_t bytes = (size_t)(counts[r] * row_bytes);
        char *dst = static_cast<char *>(out) + off * row_bytes;
        if (bytes) {
            const int nrc = g_rccl.Broadcast(r == comm->rank ? local : dst, dst, bytes, 0, r, comm->comm, st);
            if (nrc != 0) { g_rccl.GroupEnd(); return fail(POLS_ERR_HIP, "ncclBroadcast failed: %s", g_rccl.GetErrorString(nrc)); }
        }
        off += counts[r];
    }
    POLS_NCCL(g_rccl.GroupEnd());
    return POLS_OK;
}

int pols_comm_gather_rows(pols_comm *comm, const void *local, const int64_t *counts, int64_t row_bytes, int root, void *out_on_root) {
    int rc = comm_check(comm, counts, row_bytes);
    if (rc) return rc;
    if (root < 0 || root >= comm->world) return fail(POLS_ERR_INVALID, "root %d outside [0, %d)", root, comm->world);
    if (comm->rank == root && !out_on_root) return fail(POLS_ERR_INVALID, "out_on_root is NULL on the root");
    hipStream_t st = comm->ctx->stream;
    // gather-to-root over point-to-point xGMI: the root receives from its 7 peers over 7 distinct links at once
    POLS_NCCL(g_rccl.GroupStart());
    int nrc = 0;
    if (comm->rank == root) {
        int64_t off = 0;
        for (int r = 0; r < comm->world && nrc == 0; ++r) {
            const size_t bytes = (size_t)(counts[r] * row_bytes);
            char *dst = static_cast<char *>(out_on_root) + off * row_bytes;
            if (bytes && r != root) nrc = g_rccl.Recv(dst, bytes, 0, r, comm->comm, st);
            off += counts[r];
        }
    } else if (counts[comm->rank]) {
        nrc = g_rccl.Send(local, (size_t)(counts[comm->rank] * row_bytes), 0, root, comm->comm, st);
    }
    if (nrc != 0) { g_rccl.GroupEnd(); return fail(POLS_ERR_HIP, "ncclSend / ncclRecv failed: %s", g_rccl.GetErrorString(nrc)); }
    POLS_NCCL(g_rccl.GroupEnd());
    if (comm->rank == root && counts[root]) {                // the root's own shard: a device copy behind the receives
        int64_t off = 0;
        for (int r = 0; r < root; ++r) off += counts[r];
        POLS_HIP(hipMemcpyAsync(static_cast<char *>(out_on_root) + off * row_bytes, local, (size_t)(counts[root] * row_bytes),
                                hipMemcpyDeviceToDevice, st));
    }
    return POLS_OK;
}

}  // extern "C"
