// Collectives of the C ABI: RCCL over xGMI, bound directly (qrec_comm_init, qrec_allreduce,
// qrec_alltoall_rows).  One communicator per process, one process per GPU; every call only ENQUEUES on the caller's
// HIP stream, so a training step is kernels and collectives back to back on one stream with no host in between.
//
// librccl is opened at run time (single-GPU runs never touch it), and from the directory of the HIP runtime this
// library is bound to: a process that imported torch first runs on torch's bundled libamdhip64 + librccl, any other on
// /opt/rocm's -- a communicator must live on the same runtime as the streams and buffers it is given.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "common.h"

using namespace qrec;

namespace {

struct Rccl {
    void *handle = nullptr;
    std::string path;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclReduceScatter) ReduceScatter = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclCommUserRank) CommUserRank = nullptr;
    decltype(&ncclCommCuDevice) CommCuDevice = nullptr;
};
Rccl g_rccl;

struct Comm {
    ncclComm_t nccl = nullptr;
    int world = 1, rank = 0;
};

bool try_open(const std::string &p) {
    void *h = dlopen(p.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!h) return false;
    g_rccl.handle = h;
    g_rccl.path = p;
    return true;
}

int load_rccl() {
    if (g_rccl.handle) return QREC_OK;
    const char *env = getenv("QREC_RCCL_LIB");
    if (env && *env) {
        if (!try_open(env)) {
            set_error("QREC_RCCL_LIB=%s cannot be opened: %s", env, dlerror());
            return QREC_ERR_UNSUPPORTED;
        }
    } else {
        Dl_info info;
        std::string dir;
        if (dladdr(reinterpret_cast<void *>(&hipGetDeviceCount), &info) && info.dli_fname) {
            dir = info.dli_fname;
            size_t slash = dir.rfind('/');
            dir = slash == std::string::npos ? std::string() : dir.substr(0, slash + 1);
        }
        const char *names[] = {"librccl.so.1", "librccl.so"};
        bool ok = false;
        for (const char *n : names)
            if (!dir.empty() && try_open(dir + n)) { ok = true; break; }
        for (int k = 0; !ok && k < 2; ++k) ok = try_open(names[k]);
        if (!ok) {
            set_error("librccl not found next to the HIP runtime (%s) nor on the loader path: %s", dir.c_str(), dlerror());
            return QREC_ERR_UNSUPPORTED;
        }
    }
#define QREC_SYM(field, name)                                                             \
    g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(g_rccl.handle, name)); \
    if (!g_rccl.field) {                                                                  \
        set_error("%s has no symbol %s", g_rccl.path.c_str(), name);                      \
        dlclose(g_rccl.handle);                                                           \
        g_rccl.handle = nullptr;                                                          \
        return QREC_ERR_UNSUPPORTED;                                                      \
    }
    QREC_SYM(GetUniqueId, "ncclGetUniqueId")
    QREC_SYM(CommInitRank, "ncclCommInitRank")
    QREC_SYM(CommDestroy, "ncclCommDestroy")
    QREC_SYM(AllReduce, "ncclAllReduce")
    QREC_SYM(AllGather, "ncclAllGather")
    QREC_SYM(ReduceScatter, "ncclReduceScatter")
    QREC_SYM(Send, "ncclSend")
    QREC_SYM(Recv, "ncclRecv")
    QREC_SYM(GroupStart, "ncclGroupStart")
    QREC_SYM(GroupEnd, "ncclGroupEnd")
    QREC_SYM(GetErrorString, "ncclGetErrorString")
    QREC_SYM(GetVersion, "ncclGetVersion")
    QREC_SYM(CommCount, "ncclCommCount")
    QREC_SYM(CommUserRank, "ncclCommUserRank")
    QREC_SYM(CommCuDevice, "ncclCommCuDevice")
#undef QREC_SYM
    return QREC_OK;
}

#define QREC_NCCL_CHECK(expr)                                                                              \
    do {                                                                                                   \
        ncclResult_t _r = (expr);                                                                          \
        if (_r != ncclSuccess) {                                                                           \
            set_error("%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(_r), __FILE__, __LINE__);      \
            return QREC_ERR_HIP;                                                                           \
        }                                                                                                  \
    } while (0)

bool nccl_type(int dtype, ncclDataType_t *t, size_t *size) {
    switch (dtype) {
        case QREC_F32: *t = ncclFloat32; *size = 4; return true;
        case QREC_F64: *t = ncclFloat64; *size = 8; return true;
        case QREC_I32: *t = ncclInt32; *size = 4; return true;
        default: return false;
    }
}

}  // namespace

extern "C" {

int qrec_comm_library(char *path_out, int path_len, int *version) {
    int rc = load_rccl();
    if (rc != QREC_OK) return rc;
    if (path_out && path_len > 0) { strncpy(path_out, g_rccl.path.c_str(), path_len - 1); path_out[path_len - 1] = 0; }
    if (version) QREC_NCCL_CHECK(g_rccl.GetVersion(version));
    return QREC_OK;
}

int qrec_comm_unique_id(uint8_t *h_uid) {
    QREC_REQUIRE(h_uid, "qrec_comm_unique_id: null output");
    static_assert(sizeof(ncclUniqueId) == QREC_COMM_UID_BYTES, "ncclUniqueId size");
    int rc = load_rccl();
    if (rc != QREC_OK) return rc;
    ncclUniqueId id;
    QREC_NCCL_CHECK(g_rccl.GetUniqueId(&id));
    memcpy(h_uid, &id, sizeof(id));
    return QREC_OK;
}

int qrec_comm_init(int32_t world, int32_t rank, const uint8_t *h_uid, void **comm) {
    QREC_REQUIRE(comm && h_uid, "qrec_comm_init: null argument");
    QREC_REQUIRE(world >= 1 && rank >= 0 && rank < world, "qrec_comm_init: rank %d outside world %d", rank, world);
    *comm = nullptr;
    int rc = load_rccl();
    if (rc != QREC_OK) return rc;
    ncclUniqueId id;
    memcpy(&id, h_uid, sizeof(id));
    Comm *c = new Comm;
    c->world = world;
    c->rank = rank;
    ncclResult_t r = g_rccl.CommInitRank(&c->nccl, world, id, rank);   // binds the calling thread's current device
    if (r != ncclSuccess) {
        set_error("ncclCommInitRank(world=%d, rank=%d) failed: %s", world, rank, g_rccl.GetErrorString(r));
        delete c;
        return QREC_ERR_HIP;
    }
    // RCCL's own set-up may leave a (harmless, handled) HIP error as the thread's "last error" -- seen under rocprofv3:
    // "invalid device symbol" -- which this library's launch checks (hipGetLastError after a launch) would then report
    // as theirs.  Read it away.
    (void)hipGetLastError();
    *comm = c;
    return QREC_OK;
}

int qrec_comm_destroy(void *comm) {
    if (!comm) return QREC_OK;
    Comm *c = static_cast<Comm *>(comm);
    ncclResult_t r = c->nccl ? g_rccl.CommDestroy(c->nccl) : ncclSuccess;
    delete c;
    if (r != ncclSuccess) { set_error("ncclCommDestroy failed: %s", g_rccl.GetErrorString(r)); return QREC_ERR_HIP; }
    return QREC_OK;
}

int qrec_comm_info(void *comm, int32_t *world, int32_t *rank) {
    QREC_REQUIRE(comm, "qrec_comm_info: null communicator");
    Comm *c = static_cast<Comm *>(comm);
    if (world) *world = c->world;
    if (rank) *rank = c->rank;
    return QREC_OK;
}

int qrec_comm_query(void *comm, int32_t *rccl_ranks, int32_t *rccl_rank, int32_t *device) {
    QREC_REQUIRE(comm, "qrec_comm_query: null communicator");
    Comm *c = static_cast<Comm *>(comm);
    int v = 0;
    if (rccl_ranks) { QREC_NCCL_CHECK(g_rccl.CommCount(c->nccl, &v)); *rccl_ranks = v; }
    if (rccl_rank) { QREC_NCCL_CHECK(g_rccl.CommUserRank(c->nccl, &v)); *rccl_rank = v; }
    if (device) { QREC_NCCL_CHECK(g_rccl.CommCuDevice(c->nccl, &v)); *device = v; }
    return QREC_OK;
}

int qrec_allreduce(void *comm, void *d_buf, int64_t count, int dtype, void *stream) {
    QREC_REQUIRE(comm, "qrec_allreduce: null communicator");
    QREC_REQUIRE(count >= 0 && (d_buf || count == 0), "qrec_allreduce: bad buffer");
    ncclDataType_t t; size_t sz;
    QREC_REQUIRE(nccl_type(dtype, &t, &sz), "qrec_allreduce: bad dtype %d", dtype);
    if (count == 0) return QREC_OK;
    Comm *c = static_cast<Comm *>(comm);
    QREC_NCCL_CHECK(g_rccl.AllReduce(d_buf, d_buf, (size_t)count, t, ncclSum, c->nccl, as_stream(stream)));
    return QREC_OK;
}

int qrec_allreduce_pair(void *comm, void *d_a, int64_t count_a, int dtype_a, void *d_b, int64_t count_b, int dtype_b,
                        void *stream) {
    QREC_REQUIRE(comm, "qrec_allreduce_pair: null communicator");
    QREC_REQUIRE(count_a >= 0 && count_b >= 0 && (d_a || !count_a) && (d_b || !count_b), "qrec_allreduce_pair: bad buffer");
    ncclDataType_t ta, tb; size_t sa, sb;
    QREC_REQUIRE(nccl_type(dtype_a, &ta, &sa) && nccl_type(dtype_b, &tb, &sb), "qrec_allreduce_pair: bad dtype");
    Comm *c = static_cast<Comm *>(comm);
    QREC_NCCL_CHECK(g_rccl.GroupStart());
    ncclResult_t r1 = count_a ? g_rccl.AllReduce(d_a, d_a, (size_t)count_a, ta, ncclSum, c->nccl, as_stream(stream)) : ncclSuccess;
    ncclResult_t r2 = count_b ? g_rccl.AllReduce(d_b, d_b, (size_t)count_b, tb, ncclSum, c->nccl, as_stream(stream)) : ncclSuccess;
    ncclResult_t r3 = g_rccl.GroupEnd();
    QREC_NCCL_CHECK(r1);
    QREC_NCCL_CHECK(r2);
    QREC_NCCL_CHECK(r3);
    return QREC_OK;
}

int qrec_allgather(void *comm, const void *d_send, void *d_recv, int64_t count, int dtype, void *stream) {
    QREC_REQUIRE(comm, "qrec_allgather: null communicator");
    QREC_REQUIRE(count >= 0 && ((d_send && d_recv) || count == 0), "qrec_allgather: bad buffer");
    ncclDataType_t t; size_t sz;
    QREC_REQUIRE(nccl_type(dtype, &t, &sz), "qrec_allgather: bad dtype %d", dtype);
    if (count == 0) return QREC_OK;
    Comm *c = static_cast<Comm *>(comm);
    QREC_NCCL_CHECK(g_rccl.AllGather(d_send, d_recv, (size_t)count, t, c->nccl, as_stream(stream)));
    return QREC_OK;
}

int qrec_reduce_scatter(void *comm, const void *d_send, void *d_recv, int64_t count, int dtype, void *stream) {
    QREC_REQUIRE(comm, "qrec_reduce_scatter: null communicator");
    QREC_REQUIRE(count >= 0 && ((d_send && d_recv) || count == 0), "qrec_reduce_scatter: bad buffer");
    ncclDataType_t t; size_t sz;
    QREC_REQUIRE(nccl_type(dtype, &t, &sz), "qrec_reduce_scatter: bad dtype %d", dtype);
    if (count == 0) return QREC_OK;
    Comm *c = static_cast<Comm *>(comm);
    QREC_NCCL_CHECK(g_rccl.ReduceScatter(d_send, d_recv, (size_t)count, t, ncclSum, c->nccl, as_stream(stream)));
    return QREC_OK;
}

int qrec_alltoall_rows(void *comm, const void *d_send, const int64_t *h_send_rows, void *d_recv,
                       const int64_t *h_recv_rows, int64_t row_bytes, void *stream) {
    QREC_REQUIRE(comm && h_send_rows && h_recv_rows, "qrec_alltoall_rows: null argument");
    QREC_REQUIRE(row_bytes >= 1, "qrec_alltoall_rows: row_bytes must be positive");
    Comm *c = static_cast<Comm *>(comm);
    int64_t s_total = 0, r_total = 0;
    for (int p = 0; p < c->world; ++p) {
        QREC_REQUIRE(h_send_rows[p] >= 0 && h_recv_rows[p] >= 0, "qrec_alltoall_rows: negative row count for peer %d", p);
        s_total += h_send_rows[p];
        r_total += h_recv_rows[p];
    }
    QREC_REQUIRE((d_send || !s_total) && (d_recv || !r_total), "qrec_alltoall_rows: null buffer");
    if (s_total == 0 && r_total == 0) return QREC_OK;
    // one group = one fused launch: segment p of d_send goes to peer p, segment p of d_recv comes from peer p
    // (segments in rank order, back to back).  Rows travel as bytes.
    const char *s = static_cast<const char *>(d_send);
    char *r = static_cast<char *>(d_recv);
    ncclResult_t first_bad = ncclSuccess;
    QREC_NCCL_CHECK(g_rccl.GroupStart());
    for (int p = 0; p < c->world; ++p) {
        if (h_send_rows[p]) {
            ncclResult_t e = g_rccl.Send(s, (size_t)(h_send_rows[p] * row_bytes), ncclInt8, p, c->nccl, as_stream(stream));
            if (e != ncclSuccess && first_bad == ncclSuccess) first_bad = e;
        }
        if (h_recv_rows[p]) {
            ncclResult_t e = g_rccl.Recv(r, (size_t)(h_recv_rows[p] * row_bytes), ncclInt8, p, c->nccl, as_stream(stream));
            if (e != ncclSuccess && first_bad == ncclSuccess) first_bad = e;
        }
        s += h_send_rows[p] * row_bytes;
        r += h_recv_rows[p] * row_bytes;
    }
    ncclResult_t e = g_rccl.GroupEnd();
    QREC_NCCL_CHECK(first_bad);
    QREC_NCCL_CHECK(e);
    return QREC_OK;
}

int qrec_sendrecv_segments(void *comm, const void *d_send, const int32_t *h_send_peer, const int64_t *h_send_off,
                           const int64_t *h_send_bytes, int32_t n_send, void *d_recv, const int32_t *h_recv_peer,
                           const int64_t *h_recv_off, const int64_t *h_recv_bytes, int32_t n_recv, void *stream) {
    QREC_REQUIRE(comm && n_send >= 0 && n_recv >= 0, "qrec_sendrecv_segments: bad arguments");
    QREC_REQUIRE((n_send == 0 || (h_send_peer && h_send_off && h_send_bytes)) && (n_recv == 0 || (h_recv_peer && h_recv_off && h_recv_bytes)),
                 "qrec_sendrecv_segments: null segment list");
    Comm *c = static_cast<Comm *>(comm);
    int64_t s_total = 0, r_total = 0;
    for (int k = 0; k < n_send; ++k) {
        QREC_REQUIRE(h_send_peer[k] >= 0 && h_send_peer[k] < c->world && h_send_off[k] >= 0 && h_send_bytes[k] >= 0,
                     "qrec_sendrecv_segments: bad send segment %d", k);
        s_total += h_send_bytes[k];
    }
    for (int k = 0; k < n_recv; ++k) {
        QREC_REQUIRE(h_recv_peer[k] >= 0 && h_recv_peer[k] < c->world && h_recv_off[k] >= 0 && h_recv_bytes[k] >= 0,
                     "qrec_sendrecv_segments: bad recv segment %d", k);
        r_total += h_recv_bytes[k];
    }
    QREC_REQUIRE((d_send || !s_total) && (d_recv || !r_total), "qrec_sendrecv_segments: null buffer");
    if (s_total == 0 && r_total == 0) return QREC_OK;
    // ONE group = one fused launch.  Between two ranks sends and receives are matched in the order they are listed, so both
    // sides list their segments for a pair in the same order (dist.py: batch by batch).
    ncclResult_t first_bad = ncclSuccess;
    QREC_NCCL_CHECK(g_rccl.GroupStart());
    for (int k = 0; k < n_send; ++k)
        if (h_send_bytes[k]) {
            ncclResult_t e = g_rccl.Send(static_cast<const char *>(d_send) + h_send_off[k], (size_t)h_send_bytes[k], ncclInt8, h_send_peer[k],
                                         c->nccl, as_stream(stream));
            if (e != ncclSuccess && first_bad == ncclSuccess) first_bad = e;
        }
    for (int k = 0; k < n_recv; ++k)
        if (h_recv_bytes[k]) {
            ncclResult_t e = g_rccl.Recv(static_cast<char *>(d_recv) + h_recv_off[k], (size_t)h_recv_bytes[k], ncclInt8, h_recv_peer[k],
                                         c->nccl, as_stream(stream));
            if (e != ncclSuccess && first_bad == ncclSuccess) first_bad = e;
        }
    ncclResult_t e = g_rccl.GroupEnd();
    QREC_NCCL_CHECK(first_bad);
    QREC_NCCL_CHECK(e);
    return QREC_OK;
}

}  // extern "C"
