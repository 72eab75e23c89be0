// qgym_comm.cpp -- the multi-GPU hand-over behind the C ABI (include/qgym.h "Multi-GPU hand-over").
//
// The reference is one process on CPU threads and has no communication backend; BASELINE.json's north_star adds exactly one
// exchange: the observation (with reward and flags) of every shard handed back to the learner.  This file offers it two ways:
//   * RCCL: ncclGetUniqueId / ncclCommInitRank / ncclAllGather / ncclCommDestroy of librccl.so.1, resolved with dlopen on the
//     first qg_comm_init, so that a single-GPU host never loads the library;
//   * direct write: hipIpc-shared windows and the push / wait / release kernels of kernels_comm.hip.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <memory>
#include <mutex>

#include "qgym_comm.hpp"
#include "qgym_host.hpp"

using namespace qg;

namespace {

// ---- librccl.so.1, resolved at run time ----------------------------------------------------------------------------------------
struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // the soname: inside a process that already holds an RCCL (e.g. the one PyTorch ships) this resolves to that copy, which
        // shares the process's HIP runtime
        for (const char *name : {"librccl.so.1", "/opt/rocm/lib/librccl.so.1"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.lib) break;
        }
        if (!r.lib) {
            const char *e = dlerror();
            r.error = std::string("cannot load librccl.so.1: ") + (e ? e : "?");
            return;
        }
        auto sym = [&](const char *n) {
            void *p = dlsym(r.lib, n);
            if (!p && r.error.empty()) r.error = std::string("librccl.so.1 has no symbol ") + n;
            return p;
        };
        r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
        r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
        r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
        r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    });
    return r;
}

#define NCCL_TRY(expr)                                                                                                  \
    do {                                                                                                                \
        ncclResult_t _r = (expr);                                                                                       \
        if (_r != ncclSuccess) return set_error(QG_ERR_DEVICE, "%s failed: %s", #expr, rccl().GetErrorString(_r));      \
    } while (0)

uint64_t round_up(uint64_t x, uint64_t m) { return (x + m - 1) / m * m; }

}  // namespace

struct qg_comm {
    int rank = 0, world = 1, device = 0;
    ncclComm_t nccl = nullptr;
    hipStream_t side = nullptr;  // the collective's own stream (overlapped all-gather, handle exchange)

    // all-gather: staged shard(s) and, for the overlapped form, the gathered buffers
    void *snap[2] = {nullptr, nullptr};
    void *inline_snap = nullptr;  // qg_vec_gather_learner_shard's own staging (the overlapped form may have snap[] in flight)
    void *out[2] = {nullptr, nullptr};
    uint64_t shard_bytes = 0;  // what snap / out are sized for
    hipEvent_t ready[2] = {nullptr, nullptr}, done[2] = {nullptr, nullptr};
    uint64_t submitted = 0;
    int pending = -1;
    int latest = -1;

    // direct write
    void *window = nullptr;  // own window (uncached)
    uint64_t p2p_stride = 0;
    void *peer[COMM_MAX_WORLD] = {};  // mapped windows, peer[rank] = window
    bool opened[COMM_MAX_WORLD] = {};
    bool connected = false;
    void *stage = nullptr;  // packed shard before the copy
    uint32_t *ticket = nullptr, *error = nullptr;
    uint32_t push_epoch = 0, wait_epoch = 0;
    uint64_t timeout_ticks = 200000000ull;  // 2 s of the 100 MHz wall clock
};

namespace {

int comm_alloc_common(qg_comm *c) {
    HIP_TRY(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
    for (int b = 0; b < 2; ++b) {
        HIP_TRY(hipEventCreateWithFlags(&c->ready[b], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&c->done[b], hipEventDisableTiming));
    }
    return QG_OK;
}

int ensure_gather_buffers(qg_comm *c, uint64_t shard_bytes, bool outputs) {
    if (c->shard_bytes != shard_bytes) {
        HIP_TRY(hipDeviceSynchronize());
        for (int b = 0; b < 2; ++b) {
            if (c->snap[b]) HIP_TRY(hipFree(c->snap[b]));
            if (c->out[b]) HIP_TRY(hipFree(c->out[b]));
            c->snap[b] = c->out[b] = nullptr;
        }
        if (c->inline_snap) HIP_TRY(hipFree(c->inline_snap));
        c->inline_snap = nullptr;
        c->shard_bytes = shard_bytes;
        c->pending = c->latest = -1;
        c->submitted = 0;
    }
    if (!outputs) {
        if (!c->inline_snap) HIP_TRY(hipMalloc(&c->inline_snap, shard_bytes));
        return QG_OK;
    }
    for (int b = 0; b < 2; ++b) {
        if (!c->snap[b]) HIP_TRY(hipMalloc(&c->snap[b], shard_bytes));
        if (!c->out[b]) HIP_TRY(hipMalloc(&c->out[b], shard_bytes * (uint64_t)c->world));
    }
    return QG_OK;
}

int check_pair(const qg_vec *v, const qg_comm *c) {
    if (!v || !c) return set_error(QG_ERR_INVALID, "null argument");
    if (v->device != c->device) return set_error(QG_ERR_INVALID, "the handle lives on GPU %d, the communicator on GPU %d", v->device, c->device);
    return QG_OK;
}

int hand_over(qg_comm *c, int b) {
    HIP_TRY(hipEventSynchronize(c->ready[b]));  // host-mediated: see qg_comm_gather_submit in qgym.h
    NCCL_TRY(rccl().AllGather(c->snap[b], c->out[b], c->shard_bytes, ncclInt8, c->nccl, c->side));
    HIP_TRY(hipEventRecord(c->done[b], c->side));
    c->latest = b;
    return QG_OK;
}

}  // namespace

extern "C" {

int qg_vec_learner_shard_layout(const qg_vec *v, qg_shard_layout *out) {
    if (!v || !out) return set_error(QG_ERR_INVALID, "null argument");
    qg_vec_info info;
    qg_vec_get_info(v, &info);
    out->batch = v->B;
    out->obs_offset = 0;
    out->obs_bytes = v->B * (uint64_t)info.packed_words_per_env * info.packed_word_bytes;
    out->reward_offset = round_up(out->obs_bytes, 4);
    out->final_offset = out->reward_offset + 4 * v->B;
    out->success_offset = out->final_offset + round_up(v->B, 4);
    out->bytes = round_up(out->success_offset + v->B, 16);
    return QG_OK;
}

int qg_vec_pack_learner_shard(qg_vec *v, void *shard_dev, void *stream) {
    if (!v || !shard_dev) return set_error(QG_ERR_INVALID, "null argument");
    if ((uintptr_t)shard_dev & 15u) return set_error(QG_ERR_INVALID, "the shard buffer must be 16-byte aligned");
    qg_shard_layout l;
    qg_vec_learner_shard_layout(v, &l);
    if (int rc = qg_vec_observe_packed(v, shard_dev, stream)) return rc;
    QG_ON_DEVICE(v);
    const ShardLayout lay{l.batch, l.bytes, l.obs_bytes, l.reward_offset, l.final_offset, l.success_offset};
    HIP_TRY(shard_scalars(v->reward, v->done, v->success, shard_dev, lay, (hipStream_t)stream));
    return QG_OK;
}

int qg_comm_unique_id(uint8_t id_out[QG_COMM_ID_BYTES]) {
    if (!id_out) return set_error(QG_ERR_INVALID, "null argument");
    Rccl &r = rccl();
    if (!r.error.empty()) return set_error(QG_ERR_DEVICE, "%s", r.error.c_str());
    static_assert(sizeof(ncclUniqueId) == QG_COMM_ID_BYTES, "QG_COMM_ID_BYTES must equal NCCL_UNIQUE_ID_BYTES");
    ncclUniqueId id;
    NCCL_TRY(r.GetUniqueId(&id));
    memcpy(id_out, &id, sizeof id);
    return QG_OK;
}

static int comm_new(int rank, int world, int device, std::unique_ptr<qg_comm> &c) {
    if (world < 1 || world > (int)COMM_MAX_WORLD) return set_error(QG_ERR_INVALID, "world size %d outside 1..%u", world, COMM_MAX_WORLD);
    if (rank < 0 || rank >= world) return set_error(QG_ERR_INVALID, "rank %d outside 0..%d", rank, world - 1);
    const int ndev = qg_device_count();
    if (ndev <= 0) return set_error(QG_ERR_DEVICE, "no HIP device visible");
    if (device < 0 || device >= ndev) return set_error(QG_ERR_INVALID, "device %d out of range (%d visible)", device, ndev);
    c.reset(new qg_comm());
    c->rank = rank;
    c->world = world;
    c->device = device;
    return QG_OK;
}

int qg_comm_init(const uint8_t id_in[QG_COMM_ID_BYTES], int rank, int world, int device, qg_comm **out) {
    if (!id_in || !out) return set_error(QG_ERR_INVALID, "null argument");
    *out = nullptr;
    Rccl &r = rccl();
    if (!r.error.empty()) return set_error(QG_ERR_DEVICE, "%s", r.error.c_str());
    std::unique_ptr<qg_comm> c;
    if (int rc = comm_new(rank, world, device, c)) return rc;
    QG_ON_DEVICE(c);
    ncclUniqueId id;
    memcpy(&id, id_in, sizeof id);
    NCCL_TRY(r.CommInitRank(&c->nccl, world, id, rank));  // binds to the current device
    if (int rc = comm_alloc_common(c.get())) {
        (void)r.CommDestroy(c->nccl);
        return rc;
    }
    *out = c.release();
    return QG_OK;
}

int qg_comm_init_local(int rank, int world, int device, qg_comm **out) {
    if (!out) return set_error(QG_ERR_INVALID, "null argument");
    *out = nullptr;
    std::unique_ptr<qg_comm> c;
    if (int rc = comm_new(rank, world, device, c)) return rc;
    QG_ON_DEVICE(c);
    if (int rc = comm_alloc_common(c.get())) return rc;
    *out = c.release();
    return QG_OK;
}

void qg_comm_destroy(qg_comm *c) {
    if (!c) return;
    {
        qg::DeviceGuard guard(c->device);
        (void)hipDeviceSynchronize();
        for (int p = 0; p < c->world; ++p)
            if (c->opened[p] && c->peer[p]) (void)hipIpcCloseMemHandle(c->peer[p]);
        for (void *p : {c->snap[0], c->snap[1], c->inline_snap, c->out[0], c->out[1], c->window, c->stage, (void *)c->ticket, (void *)c->error})
            if (p) (void)hipFree(p);
        for (int b = 0; b < 2; ++b) {
            if (c->ready[b]) (void)hipEventDestroy(c->ready[b]);
            if (c->done[b]) (void)hipEventDestroy(c->done[b]);
        }
        if (c->nccl) (void)rccl().CommDestroy(c->nccl);
        if (c->side) (void)hipStreamDestroy(c->side);
    }
    delete c;
}

int qg_comm_rank(const qg_comm *c) { return c ? c->rank : -1; }
int qg_comm_world(const qg_comm *c) { return c ? c->world : -1; }

// ---- RCCL all-gather -------------------------------------------------------------------------------------------------------------
int qg_vec_gather_learner_shard(qg_vec *v, qg_comm *c, void *out_dev, void *stream) {
    if (int rc = check_pair(v, c)) return rc;
    if (!out_dev) return set_error(QG_ERR_INVALID, "null argument");
    if (!c->nccl) return set_error(QG_ERR_INVALID, "this communicator has no RCCL (qg_comm_init_local): use the direct-write entry points");
    QG_ON_DEVICE(c);
    qg_shard_layout l;
    qg_vec_learner_shard_layout(v, &l);
    if (int rc = ensure_gather_buffers(c, l.bytes, false)) return rc;
    if (int rc = qg_vec_pack_learner_shard(v, c->inline_snap, stream)) return rc;
    NCCL_TRY(rccl().AllGather(c->inline_snap, out_dev, l.bytes, ncclInt8, c->nccl, (hipStream_t)stream));
    return QG_OK;
}

int qg_comm_gather_submit(qg_comm *c, qg_vec *v, void *stream) {
    if (int rc = check_pair(v, c)) return rc;
    if (!c->nccl) return set_error(QG_ERR_INVALID, "this communicator has no RCCL (qg_comm_init_local): use the direct-write entry points");
    QG_ON_DEVICE(c);
    qg_shard_layout l;
    qg_vec_learner_shard_layout(v, &l);
    if (int rc = ensure_gather_buffers(c, l.bytes, true)) return rc;
    const int b = (int)(c->submitted & 1);
    if (c->submitted >= 2) HIP_TRY(hipEventSynchronize(c->done[b]));  // the collective that read snap[b] is over
    if (int rc = qg_vec_pack_learner_shard(v, c->snap[b], stream)) return rc;
    HIP_TRY(hipEventRecord(c->ready[b], (hipStream_t)stream));
    if (c->pending >= 0)
        if (int rc = hand_over(c, c->pending)) return rc;
    c->pending = b;
    c->submitted += 1;
    return QG_OK;
}

int qg_comm_gather_flush(qg_comm *c) {
    if (!c) return set_error(QG_ERR_INVALID, "null argument");
    QG_ON_DEVICE(c);
    if (c->pending >= 0) {
        if (int rc = hand_over(c, c->pending)) return rc;
        c->pending = -1;
    }
    return QG_OK;
}

int qg_comm_gather_latest(qg_comm *c, const void **gathered_dev) {
    if (!c || !gathered_dev) return set_error(QG_ERR_INVALID, "null argument");
    *gathered_dev = nullptr;
    if (c->latest < 0) return QG_OK;
    QG_ON_DEVICE(c);
    HIP_TRY(hipEventSynchronize(c->done[c->latest]));
    *gathered_dev = c->out[c->latest];
    return QG_OK;
}

// ---- direct write ------------------------------------------------------------------------------------------------------------------
static uint64_t window_bytes(const qg_comm *c) { return COMM_HEADER_BYTES + 2ull * c->world * c->p2p_stride; }

int qg_comm_p2p_export(qg_comm *c, uint64_t shard_bytes, uint8_t handle_out[QG_P2P_HANDLE_BYTES]) {
    if (!c || !handle_out) return set_error(QG_ERR_INVALID, "null argument");
    if (!shard_bytes || (shard_bytes & 15u)) return set_error(QG_ERR_INVALID, "shard_bytes must be a positive multiple of 16 (qg_shard_layout.bytes)");
    if (c->window) return set_error(QG_ERR_INVALID, "this communicator already has a window");
    QG_ON_DEVICE(c);
    c->p2p_stride = shard_bytes;
    // uncached on the owner: peers write it behind this GPU's L2, so the owner must not keep lines of it
    hipError_t e = hipExtMallocWithFlags(&c->window, window_bytes(c), hipDeviceMallocUncached);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        e = hipExtMallocWithFlags(&c->window, window_bytes(c), hipDeviceMallocFinegrained);
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        c->window = nullptr;
        return set_error(QG_ERR_DEVICE, "cannot allocate the %llu-byte window: %s", (unsigned long long)window_bytes(c), hipGetErrorString(e));
    }
    // a failure below leaves the communicator as it was before the call (no window: a retry starts over)
    auto undo = [&](hipError_t err, const char *what) {
        (void)hipGetLastError();
        for (void **p : {(void **)&c->window, (void **)&c->stage, (void **)&c->ticket, (void **)&c->error}) {
            if (*p) (void)hipFree(*p);
            *p = nullptr;
        }
        c->peer[c->rank] = nullptr;
        return set_error(QG_ERR_DEVICE, "%s failed: %s", what, hipGetErrorString(err));
    };
#define P2P_TRY(expr)                                    \
    do {                                                 \
        hipError_t _e = (expr);                          \
        if (_e != hipSuccess) return undo(_e, #expr);    \
    } while (0)
    P2P_TRY(hipMemset(c->window, 0, COMM_HEADER_BYTES));
    P2P_TRY(hipMalloc(&c->stage, shard_bytes));
    P2P_TRY(hipMalloc(&c->ticket, sizeof(uint32_t) * 2 * COMM_MAX_WORLD));  // per peer: block ticket, "a block skipped" flag
    P2P_TRY(hipMalloc(&c->error, sizeof(uint32_t)));
    P2P_TRY(hipMemset(c->ticket, 0, sizeof(uint32_t) * 2 * COMM_MAX_WORLD));
    P2P_TRY(hipMemset(c->error, 0, sizeof(uint32_t)));
    P2P_TRY(hipDeviceSynchronize());
    c->peer[c->rank] = c->window;
    static_assert(sizeof(hipIpcMemHandle_t) == QG_P2P_HANDLE_BYTES, "QG_P2P_HANDLE_BYTES must equal HIP_IPC_HANDLE_SIZE");
    hipIpcMemHandle_t h;
    memset(&h, 0, sizeof h);
    if (c->world > 1) P2P_TRY(hipIpcGetMemHandle(&h, c->window));
#undef P2P_TRY
    memcpy(handle_out, &h, sizeof h);
    return QG_OK;
}

int qg_comm_p2p_open(qg_comm *c, const uint8_t *handles) {
    if (!c || (!handles && c->world > 1)) return set_error(QG_ERR_INVALID, "null argument");
    if (!c->window) return set_error(QG_ERR_INVALID, "qg_comm_p2p_export first");
    if (c->connected) return set_error(QG_ERR_INVALID, "already connected");
    QG_ON_DEVICE(c);
    for (int p = 0; p < c->world; ++p) {
        if (p == c->rank) continue;
        hipIpcMemHandle_t h;
        memcpy(&h, handles + (size_t)p * QG_P2P_HANDLE_BYTES, sizeof h);
        hipError_t e = hipIpcOpenMemHandle(&c->peer[p], h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            return set_error(QG_ERR_DEVICE, "hipIpcOpenMemHandle of rank %d's window failed: %s", p, hipGetErrorString(e));
        }
        c->opened[p] = true;
    }
    c->connected = true;
    return QG_OK;
}

int qg_comm_p2p_connect(qg_comm *c, uint64_t shard_bytes) {
    if (!c) return set_error(QG_ERR_INVALID, "null argument");
    if (!c->nccl && c->world > 1) return set_error(QG_ERR_INVALID, "no RCCL on this communicator: exchange the handles yourself (qg_comm_p2p_export / _open)");
    uint8_t mine[QG_P2P_HANDLE_BYTES];
    if (int rc = qg_comm_p2p_export(c, shard_bytes, mine)) return rc;
    if (c->world == 1) return qg_comm_p2p_open(c, mine);
    QG_ON_DEVICE(c);
    // the handles travel through the communicator itself: 64 bytes per rank
    uint8_t *dev = nullptr;
    HIP_TRY(hipMalloc(&dev, (size_t)QG_P2P_HANDLE_BYTES * (c->world + 1)));
    std::unique_ptr<uint8_t, void (*)(uint8_t *)> guard(dev, [](uint8_t *p) { (void)hipFree(p); });
    HIP_TRY(hipMemcpyAsync(dev, mine, QG_P2P_HANDLE_BYTES, hipMemcpyHostToDevice, c->side));
    NCCL_TRY(rccl().AllGather(dev, dev + QG_P2P_HANDLE_BYTES, QG_P2P_HANDLE_BYTES, ncclInt8, c->nccl, c->side));
    std::vector<uint8_t> all((size_t)QG_P2P_HANDLE_BYTES * c->world);
    HIP_TRY(hipMemcpyAsync(all.data(), dev + QG_P2P_HANDLE_BYTES, all.size(), hipMemcpyDeviceToHost, c->side));
    HIP_TRY(hipStreamSynchronize(c->side));
    return qg_comm_p2p_open(c, all.data());
}

int qg_vec_push_learner_shard(qg_vec *v, qg_comm *c, void *stream) {
    if (int rc = check_pair(v, c)) return rc;
    if (!c->connected) return set_error(QG_ERR_INVALID, "qg_comm_p2p_connect (or export + open) first");
    qg_shard_layout l;
    qg_vec_learner_shard_layout(v, &l);
    if (l.bytes != c->p2p_stride)
        return set_error(QG_ERR_INVALID, "the handle's shard is %llu bytes, the windows were sized for %llu", (unsigned long long)l.bytes,
                         (unsigned long long)c->p2p_stride);
    QG_ON_DEVICE(c);
    if (int rc = qg_vec_pack_learner_shard(v, c->stage, stream)) return rc;
    const uint32_t epoch = c->push_epoch + 1;
    PushArgs a;
    memset(&a, 0, sizeof a);
    a.src = (const uint4 *)c->stage;
    a.n16 = l.bytes / 16;
    for (int p = 0; p < c->world; ++p) {
        uint8_t *w = (uint8_t *)c->peer[p];
        a.dst[p] = (uint4 *)(w + COMM_HEADER_BYTES + ((uint64_t)(epoch & 1u) * c->world + c->rank) * c->p2p_stride);
        a.arrive[p] = (uint32_t *)w;
    }
    a.local_ack = (const uint32_t *)((uint8_t *)c->window + COMM_ACK_OFFSET);
    a.ticket = c->ticket;
    a.error = c->error;
    a.timeout_ticks = c->timeout_ticks;
    a.rank = (uint32_t)c->rank;
    a.world = (uint32_t)c->world;
    a.epoch = epoch;
    HIP_TRY(push_shard(a, (hipStream_t)stream));
    c->push_epoch = epoch;
    return QG_OK;
}

int qg_comm_p2p_wait(qg_comm *c, const void **gathered_dev, void *stream) {
    if (!c || !gathered_dev) return set_error(QG_ERR_INVALID, "null argument");
    if (!c->connected) return set_error(QG_ERR_INVALID, "qg_comm_p2p_connect (or export + open) first");
    QG_ON_DEVICE(c);
    const uint32_t epoch = c->wait_epoch + 1;
    HIP_TRY(wait_arrivals((const uint32_t *)c->window, (uint32_t)c->world, epoch, c->timeout_ticks, c->error, (hipStream_t)stream));
    c->wait_epoch = epoch;
    *gathered_dev = (uint8_t *)c->window + COMM_HEADER_BYTES + (uint64_t)(epoch & 1u) * c->world * c->p2p_stride;
    return QG_OK;
}

int qg_comm_p2p_release(qg_comm *c, void *stream) {
    if (!c) return set_error(QG_ERR_INVALID, "null argument");
    if (!c->connected || !c->wait_epoch) return set_error(QG_ERR_INVALID, "nothing to release: qg_comm_p2p_wait first");
    QG_ON_DEVICE(c);
    AckArgs a;
    memset(&a, 0, sizeof a);
    for (int p = 0; p < c->world; ++p) a.ack[p] = (uint32_t *)((uint8_t *)c->peer[p] + COMM_ACK_OFFSET);
    a.rank = (uint32_t)c->rank;
    a.world = (uint32_t)c->world;
    a.epoch = c->wait_epoch;
    HIP_TRY(release_window(a, (hipStream_t)stream));
    return QG_OK;
}

int qg_comm_p2p_reset(qg_comm *c, void *stream) {
    if (!c) return set_error(QG_ERR_INVALID, "null argument");
    if (!c->window) return QG_OK;
    QG_ON_DEVICE(c);
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    HIP_TRY(hipMemset(c->window, 0, COMM_HEADER_BYTES));  // this rank's arrival flags and the releases stored here
    HIP_TRY(hipMemset(c->ticket, 0, sizeof(uint32_t) * 2 * COMM_MAX_WORLD));
    HIP_TRY(hipMemset(c->error, 0, sizeof(uint32_t)));
    HIP_TRY(hipDeviceSynchronize());
    c->push_epoch = 0;
    c->wait_epoch = 0;
    return QG_OK;
}

int qg_comm_p2p_check(qg_comm *c, void *stream) {
    if (!c) return set_error(QG_ERR_INVALID, "null argument");
    if (!c->error) return QG_OK;
    QG_ON_DEVICE(c);
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    uint32_t err = 0;
    HIP_TRY(hipMemcpy(&err, c->error, sizeof err, hipMemcpyDeviceToHost));
    if (err)
        return set_error(QG_ERR_DEVICE, "direct-write hand-over: %s%s within the deadline (error bits 0x%x)",
                         (err & QG_COMM_ERR_ARRIVE_TIMEOUT) ? "a peer's shard did not arrive" : "",
                         (err & QG_COMM_ERR_ACK_TIMEOUT) ? ((err & QG_COMM_ERR_ARRIVE_TIMEOUT) ? " and a peer did not release its window" : "a peer did not release its window") : "",
                         err);
    return QG_OK;
}

}  // extern "C"
