// libtsearch.so - C ABI (include/tsearch.h), part 4: merging partial top-k lists and the row-sharded search over RCCL.
#include "host.h"
#include "kernels_select.h"

extern "C" int ts_merge_topk(int device, const float* scores, const int64_t* idx, int32_t nparts, int32_t nq, int32_t k_in,
                             int32_t k_out, float* out_scores, int64_t* out_idx, int on_device, void* stream) {
    if (!scores || !idx || !out_scores || !out_idx) return fail(TS_ERR_INVALID, "NULL argument");
    if (nparts < 1 || nq < 0 || k_in < 1 || k_out < 1 || k_out > TS_MAX_K)
        return fail(TS_ERR_INVALID, "bad merge shape");
    if ((int64_t)nparts * k_in > kMergeMax)
        return fail(TS_ERR_UNSUPPORTED, "nparts * k_in = %lld exceeds %d", (long long)nparts * k_in, kMergeMax);
    if (nq == 0) return TS_OK;
    TS_TRY(check_device(device));
    HIP_TRY(hipSetDevice(device));
    hipStream_t st = (hipStream_t)stream;
    MergeArgs a;
    a.nparts = nparts; a.nq = nq; a.k_in = k_in; a.k_out = k_out;
    a.part_stride = a.part_stride_idx = (int64_t)nq * k_in;
    const size_t nin = (size_t)nparts * nq * k_in, nout = (size_t)nq * k_out;
    if (on_device) {
        a.scores = scores; a.idx = idx; a.out_scores = out_scores; a.out_idx = out_idx;
        launch_merge(a, st);
        HIP_TRY(hipGetLastError());
        return TS_OK;
    }
    // one temporary block (freed on every return path): ids in | ids out | scores in | scores out
    DevBuf tmp;
    HIP_TRY(tmp.alloc(nin * 12 + nout * 12));
    int64_t* di = tmp.as<int64_t>();
    int64_t* doi = di + nin;
    float* ds = (float*)(doi + nout);
    float* dos = ds + nin;
    HIP_TRY(hipMemcpyAsync(ds, scores, nin * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(di, idx, nin * 8, hipMemcpyHostToDevice, st));
    a.scores = ds; a.idx = di; a.out_scores = dos; a.out_idx = doi;
    launch_merge(a, st);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out_scores, dos, nout * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(out_idx, doi, nout * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return TS_OK;
}

extern "C" int ts_merge_topk_packed(int device, const void* packed, int64_t part_stride_bytes, int64_t idx_offset_bytes,
                                    int32_t nparts, int32_t nq, int32_t k_in, int32_t k_out, float* out_scores,
                                    int64_t* out_idx, void* stream) {
    if (!packed || !out_scores || !out_idx) return fail(TS_ERR_INVALID, "NULL argument");
    if (nparts < 1 || nq < 0 || k_in < 1 || k_out < 1 || k_out > TS_MAX_K) return fail(TS_ERR_INVALID, "bad merge shape");
    if ((int64_t)nparts * k_in > kMergeMax)
        return fail(TS_ERR_UNSUPPORTED, "nparts * k_in = %lld exceeds %d", (long long)nparts * k_in, kMergeMax);
    if (part_stride_bytes % 8 || idx_offset_bytes % 8 || idx_offset_bytes < (int64_t)nq * k_in * 4 ||
        part_stride_bytes < idx_offset_bytes + (int64_t)nq * k_in * 8)
        return fail(TS_ERR_INVALID, "bad packed layout (stride %lld, idx offset %lld)", (long long)part_stride_bytes,
                    (long long)idx_offset_bytes);
    if (nq == 0) return TS_OK;
    TS_TRY(check_device(device));
    HIP_TRY(hipSetDevice(device));
    MergeArgs a;
    a.nparts = nparts; a.nq = nq; a.k_in = k_in; a.k_out = k_out;
    a.scores = (const float*)packed;
    a.idx = (const int64_t*)((const char*)packed + idx_offset_bytes);
    a.part_stride = part_stride_bytes / 4;
    a.part_stride_idx = part_stride_bytes / 8;
    a.out_scores = out_scores;
    a.out_idx = out_idx;
    launch_merge(a, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return TS_OK;
}

// Row-sharded search behind the C ABI (SURVEY.md section 8b "ts_shards_*", section 8e): the exchange of the per-shard
// top-k runs over RCCL inside this library, so the search path needs neither torch.distributed nor MPI.
//
//   ts_comm_*    one communicator member per PROCESS (the `torch.distributed.run` model of bench.py: one process per GPU);
//                the launcher only carries the 128-byte unique id from rank 0 to the other ranks.
//   ts_shards_*  one PROCESS driving all the GPUs of the node (ncclCommInitAll, one stream per device) - what a serving
//                process (the Streamlit app) would hold.
//
// RCCL is bound with dlopen at the
// first use: libtsearch.so itself has no link-time dependency on it, and a process that already carries a copy of RCCL
// (PyTorch-ROCm bundles one) reuses that copy instead of loading a second one.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <condition_variable>
#include <functional>
#include <memory>
#include <thread>

namespace {

struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    char why[256] = "";
};

RcclApi* rccl_api() {
    static std::mutex mu;
    static RcclApi api;
    std::lock_guard<std::mutex> lock(mu);
    if (api.handle) return &api;
    const char* env = getenv("TS_RCCL_LIB");
    const char* names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    // a copy that is already mapped (by any name the process used) wins: never two RCCLs in one process
    for (const char* n : names)
        if (n && *n && (h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL))) break;
    for (int i = 0; !h && i < 4; ++i)
        if (names[i] && *names[i]) h = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
    if (!h) {
        snprintf(api.why, sizeof(api.why), "RCCL not loadable: %s", dlerror());
        return nullptr;
    }
#define TS_SYM(field, name)                                                  \
    api.field = (decltype(api.field))dlsym(h, name);                         \
    if (!api.field) {                                                        \
        snprintf(api.why, sizeof(api.why), "RCCL lacks symbol %s", name);    \
        dlclose(h);                                                          \
        return nullptr;                                                      \
    }
    TS_SYM(GetUniqueId, "ncclGetUniqueId")
    TS_SYM(CommInitRank, "ncclCommInitRank")
    TS_SYM(CommInitAll, "ncclCommInitAll")
    TS_SYM(CommDestroy, "ncclCommDestroy")
    TS_SYM(AllGather, "ncclAllGather")
    TS_SYM(GroupStart, "ncclGroupStart")
    TS_SYM(GroupEnd, "ncclGroupEnd")
    TS_SYM(GetErrorString, "ncclGetErrorString")
#undef TS_SYM
    api.handle = h;
    return &api;
}

#define NCCL_TRY(api, expr)                                                                                     \
    do {                                                                                                        \
        ncclResult_t r_ = (expr);                                                                               \
        if (r_ != ncclSuccess) return fail(TS_ERR_HIP, "%s failed: %s (%s:%d)", #expr, (api)->GetErrorString(r_), __FILE__, __LINE__); \
    } while (0)

// Packed result block of one shard for one query batch: scores [nq x k] f32 at 0, ids [nq x k] i64 at idx_off.
inline int64_t packed_idx_off(int nq, int k) { return ((int64_t)nq * k * 4 + 7) / 8 * 8; }
inline int64_t packed_bytes(int nq, int k) { return packed_idx_off(nq, k) + (int64_t)nq * k * 8; }

int launch_merge_packed(const void* packed, int64_t part_stride_bytes, int64_t idx_off, int nparts, int nq, int k_in, int k_out,
                        float* out_scores, int64_t* out_idx, hipStream_t st) {
    if ((int64_t)nparts * k_in > kMergeMax)
        return fail(TS_ERR_UNSUPPORTED, "nparts * k_in = %lld exceeds %d", (long long)nparts * k_in, kMergeMax);
    MergeArgs a;
    a.nparts = nparts; a.nq = nq; a.k_in = k_in; a.k_out = k_out;
    a.scores = (const float*)packed;
    a.idx = (const int64_t*)((const char*)packed + idx_off);
    a.part_stride = part_stride_bytes / 4;
    a.part_stride_idx = part_stride_bytes / 8;
    a.out_scores = out_scores;
    a.out_idx = out_idx;
    launch_merge(a, st);
    HIP_TRY(hipGetLastError());
    return TS_OK;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// ts_comm: one member per process
// ---------------------------------------------------------------------------------------------
struct ts_comm {
    int device = 0, world = 1, rank = 0;
    ncclComm_t comm = nullptr;
    std::mutex mu;
    // exchange buffers of ts_comm_search, two of each (a caller may keep one batch in flight while the next starts)
    void* mine[2] = {nullptr, nullptr};
    void* all[2] = {nullptr, nullptr};
    size_t mine_bytes[2] = {0, 0}, all_bytes[2] = {0, 0};
    int parity = 0;
};

extern "C" int ts_comm_unique_id(void* id_out, int32_t id_bytes) {
    if (!id_out || id_bytes < (int32_t)sizeof(ncclUniqueId)) return fail(TS_ERR_INVALID, "id buffer must hold %zu bytes", sizeof(ncclUniqueId));
    RcclApi* api = rccl_api();
    if (!api) return fail(TS_ERR_UNSUPPORTED, "%s", "RCCL is not available in this process");
    ncclUniqueId id;
    NCCL_TRY(api, api->GetUniqueId(&id));
    memset(id_out, 0, (size_t)id_bytes);
    memcpy(id_out, &id, sizeof(id));
    return TS_OK;
}

extern "C" int ts_comm_create(int device, int32_t world, int32_t rank, const void* id, int32_t id_bytes, ts_comm** out) {
    if (!out) return fail(TS_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) return fail(TS_ERR_INVALID, "rank %d of %d", rank, world);
    if (!id || id_bytes < (int32_t)sizeof(ncclUniqueId)) return fail(TS_ERR_INVALID, "id must hold %zu bytes (ts_comm_unique_id)", sizeof(ncclUniqueId));
    TS_TRY(check_device(device));
    RcclApi* api = rccl_api();
    if (!api) return fail(TS_ERR_UNSUPPORTED, "%s", "RCCL is not available in this process");
    HIP_TRY(hipSetDevice(device));
    ts_comm* c = new (std::nothrow) ts_comm();
    if (!c) return fail(TS_ERR_NOMEM, "host allocation failed");
    c->device = device;
    c->world = world;
    c->rank = rank;
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    ncclResult_t r = api->CommInitRank(&c->comm, world, uid, rank);
    if (r != ncclSuccess) {
        delete c;
        return fail(TS_ERR_HIP, "ncclCommInitRank(rank %d of %d) failed: %s", rank, world, api->GetErrorString(r));
    }
    *out = c;
    return TS_OK;
}

extern "C" int ts_comm_destroy(ts_comm* c) {
    if (!c) return TS_OK;
    hipSetDevice(c->device);
    hipDeviceSynchronize();
    RcclApi* api = rccl_api();
    if (api && c->comm) api->CommDestroy(c->comm);
    for (int b = 0; b < 2; ++b) {
        if (c->mine[b]) hipFree(c->mine[b]);
        if (c->all[b]) hipFree(c->all[b]);
    }
    delete c;
    return TS_OK;
}

extern "C" int ts_comm_info(const ts_comm* c, int32_t* world, int32_t* rank, int32_t* device) {
    if (!c) return fail(TS_ERR_INVALID, "comm is NULL");
    if (world) *world = c->world;
    if (rank) *rank = c->rank;
    if (device) *device = c->device;
    return TS_OK;
}

extern "C" int ts_comm_allgather(ts_comm* c, const void* send_dev, void* recv_dev, int64_t bytes, void* stream) {
    if (!c || !send_dev || !recv_dev || bytes < 0) return fail(TS_ERR_INVALID, "bad argument");
    if (bytes == 0) return TS_OK;
    RcclApi* api = rccl_api();
    if (!api) return fail(TS_ERR_UNSUPPORTED, "%s", "RCCL is not available in this process");
    std::lock_guard<std::mutex> lock(c->mu);         // one thread at a time per communicator
    HIP_TRY(hipSetDevice(c->device));
    NCCL_TRY(api, api->AllGather(send_dev, recv_dev, (size_t)bytes, ncclUint8, c->comm, (hipStream_t)stream));
    return TS_OK;
}

// The whole sharded search of one rank, enqueued on `stream`: search of the local shard into a packed block, ONE
// all-gather of the blocks (12 * nq * k bytes per rank), merge of the world * k candidates per query.  Device or host
// queries / results as in ts_search.  Every rank ends up with the global answer.
extern "C" int ts_comm_search(ts_comm* c, ts_index* shard, const void* queries, int q_dtype, int q_on_device, int32_t nq, int32_t k,
                              float* out_scores, int64_t* out_idx, int out_on_device, void* stream) {
    if (!c || !shard || !queries || !out_scores || !out_idx) return fail(TS_ERR_INVALID, "NULL argument");
    if (shard->device != c->device) return fail(TS_ERR_INVALID, "the shard lives on device %d, the communicator on %d", shard->device, c->device);
    if (nq < 0 || k < 1 || k > TS_MAX_K) return fail(TS_ERR_INVALID, "bad shape");
    if (nq == 0) return TS_OK;
    RcclApi* api = rccl_api();
    if (!api) return fail(TS_ERR_UNSUPPORTED, "%s", "RCCL is not available in this process");
    std::lock_guard<std::mutex> lock(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    void* ix_stream = nullptr;
    TS_TRY(ts_index_stream(shard, &ix_stream));
    hipStream_t st = stream ? (hipStream_t)stream : (hipStream_t)ix_stream;
    const int64_t blk = packed_bytes(nq, k), idx_off = packed_idx_off(nq, k);
    const int b = c->parity;
    c->parity ^= 1;
    TS_TRY(ensure(&c->mine[b], &c->mine_bytes[b], (size_t)blk));
    TS_TRY(ensure(&c->all[b], &c->all_bytes[b], (size_t)blk * c->world));
    TS_TRY(ts_search(shard, queries, q_dtype, q_on_device, nq, k, (float*)c->mine[b], (int64_t*)((char*)c->mine[b] + idx_off), 1, st));
    NCCL_TRY(api, api->AllGather(c->mine[b], c->all[b], (size_t)blk, ncclUint8, c->comm, st));
    float* ds = out_scores;
    int64_t* di = out_idx;
    DevBuf tmp;
    if (!out_on_device) {
        HIP_TRY(tmp.alloc((size_t)nq * k * 12));
        di = tmp.as<int64_t>();
        ds = (float*)(di + (size_t)nq * k);
    }
    TS_TRY(launch_merge_packed(c->all[b], blk, idx_off, c->world, nq, k, k, ds, di, st));
    if (!out_on_device) {
        HIP_TRY(hipMemcpyAsync(out_scores, ds, (size_t)nq * k * 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(out_idx, di, (size_t)nq * k * 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    return TS_OK;
}

// ---------------------------------------------------------------------------------------------
// ts_shards: one process, all devices
// ---------------------------------------------------------------------------------------------
// One host thread per shard.  A search of a 1.25M-row shard is five launches and 0.44 ms of device time; enqueueing the
// query copy and those launches for eight devices from ONE thread, device after device, starts device 7 when device 0 is
// nearly done.  Each shard therefore has a persistent worker that owns its device's enqueue calls: the caller stages the
// queries in pinned host memory once, posts the same job to every worker, waits until all have ENQUEUED (not finished),
// and runs the exchange from its own thread (RCCL's grouped calls want one thread).
struct ShardWorker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::function<int()> job;
    bool has_job = false, quit = false, done = true;
    int rc = TS_OK;
    char err[512] = "";
    int device = 0;

    void run() {
        (void)hipSetDevice(device);
        std::unique_lock<std::mutex> lock(mu);
        for (;;) {
            cv.wait(lock, [&] { return has_job || quit; });
            if (quit) return;
            std::function<int()> fn = std::move(job);
            has_job = false;
            lock.unlock();
            g_err[0] = 0;
            const int r = fn();
            lock.lock();
            rc = r;
            snprintf(err, sizeof(err), "%s", g_err);       // the worker's thread-local message, for the caller's thread
            done = true;
            cv.notify_all();
        }
    }
    void post(std::function<int()> fn) {
        std::lock_guard<std::mutex> lock(mu);
        job = std::move(fn);
        has_job = true;
        done = false;
        cv.notify_all();
    }
    int wait() {
        std::unique_lock<std::mutex> lock(mu);
        cv.wait(lock, [&] { return done; });
        if (rc != TS_OK) snprintf(g_err, sizeof(g_err), "%s", err);
        return rc;
    }
    void stop() {
        {
            std::lock_guard<std::mutex> lock(mu);
            quit = true;
            cv.notify_all();
        }
        if (th.joinable()) th.join();
    }
};

struct ts_shards {
    int ngpu = 0;
    int64_t n_total = 0;
    int d = 0, dtype = 0, metric = 0;
    bool use_rccl = false;            // false: several shards share a device (rehearsal on one GPU) -> copies instead
    std::vector<int> device;
    std::vector<int64_t> lo;          // shard g holds global rows [lo[g], lo[g + 1])
    std::vector<ts_index*> shard;
    std::vector<ncclComm_t> comm;
    std::vector<void*> qdev, mine, all;
    std::vector<size_t> qdev_bytes, mine_bytes, all_bytes;
    std::vector<hipEvent_t> done;
    void* fin = nullptr;
    size_t fin_bytes = 0;
    std::vector<std::unique_ptr<ShardWorker>> worker;   // one per shard when ngpu > 1
    void* qpin = nullptr;                               // the query batch in pinned host memory (every device copies from it)
    size_t qpin_bytes = 0;
    std::mutex mu;
};

extern "C" int ts_shards_destroy(ts_shards* s) {
    if (!s) return TS_OK;
    for (auto& w : s->worker) w->stop();
    if (s->qpin) hipHostFree(s->qpin);
    RcclApi* api = s->use_rccl ? rccl_api() : nullptr;
    for (int g = 0; g < (int)s->shard.size(); ++g) {
        hipSetDevice(s->device[g]);
        hipDeviceSynchronize();
        if (api && g < (int)s->comm.size() && s->comm[g]) api->CommDestroy(s->comm[g]);
        if (g < (int)s->qdev.size() && s->qdev[g]) hipFree(s->qdev[g]);
        if (g < (int)s->mine.size() && s->mine[g]) hipFree(s->mine[g]);
        if (g < (int)s->all.size() && s->all[g]) hipFree(s->all[g]);
        ts_index_destroy(s->shard[g]);
    }
    for (hipEvent_t e : s->done)
        if (e) hipEventDestroy(e);
    if (s->fin) {
        hipSetDevice(s->device[0]);
        hipFree(s->fin);
    }
    delete s;
    return TS_OK;
}

// devices = NULL: devices 0 .. ngpu-1.  A device id may repeat (several shards on one GPU: the exchange then uses
// device copies instead of RCCL, which refuses two ranks on one GPU) - for rehearsing the sharded path on a one-GPU box.
extern "C" int ts_shards_create(int32_t ngpu, const int32_t* devices, int64_t n_total, int32_t d, int dtype, int metric,
                                ts_shards** out) {
    if (!out) return fail(TS_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (ngpu < 1 || ngpu > 64 || n_total < 0) return fail(TS_ERR_INVALID, "bad shape");
    int ndev = 0;
    ts_device_count(&ndev);
    if (ndev <= 0) return fail(TS_ERR_NODEVICE, "no HIP device visible (libtsearch has no CPU path)");
    ts_shards* s = new (std::nothrow) ts_shards();
    if (!s) return fail(TS_ERR_NOMEM, "host allocation failed");
    s->ngpu = ngpu;
    s->n_total = n_total;
    s->d = d;
    s->dtype = dtype;
    s->metric = metric;
    bool distinct = true;
    for (int g = 0; g < ngpu; ++g) {
        const int dev = devices ? devices[g] : g;
        if (dev < 0 || dev >= ndev) {
            delete s;
            return fail(TS_ERR_INVALID, "device %d out of range [0, %d)", dev, ndev);
        }
        for (int o : s->device) distinct = distinct && o != dev;
        s->device.push_back(dev);
    }
    for (int g = 0; g <= ngpu; ++g) s->lo.push_back(n_total * g / ngpu);
    s->qdev.assign(ngpu, nullptr); s->mine.assign(ngpu, nullptr); s->all.assign(ngpu, nullptr);
    s->qdev_bytes.assign(ngpu, 0); s->mine_bytes.assign(ngpu, 0); s->all_bytes.assign(ngpu, 0);
    s->done.assign(ngpu, nullptr);
    for (int g = 0; g < ngpu; ++g) {
        ts_index* ix = nullptr;
        int rc = ts_index_create(s->device[g], s->lo[g + 1] - s->lo[g], d, dtype, metric, &ix);
        if (rc == TS_OK) rc = ts_index_set_row_offset(ix, s->lo[g]);
        if (rc == TS_OK && hipEventCreateWithFlags(&s->done[g], hipEventDisableTiming) != hipSuccess) rc = fail(TS_ERR_HIP, "event creation failed");
        if (rc != TS_OK) {
            if (ix) ts_index_destroy(ix);
            ts_shards_destroy(s);
            return rc;
        }
        s->shard.push_back(ix);
    }
    s->use_rccl = distinct && ngpu > 1;
    if (s->use_rccl) {
        RcclApi* api = rccl_api();
        if (!api) {
            ts_shards_destroy(s);
            return fail(TS_ERR_UNSUPPORTED, "%s", "RCCL is not available in this process");
        }
        s->comm.assign(ngpu, nullptr);
        ncclResult_t r = api->CommInitAll(s->comm.data(), ngpu, s->device.data());
        if (r != ncclSuccess) {
            s->comm.clear();
            ts_shards_destroy(s);
            return fail(TS_ERR_HIP, "ncclCommInitAll over %d devices failed: %s", ngpu, api->GetErrorString(r));
        }
    }
    // TS_SHARDS_THREADS=0: every device's enqueue from the caller's thread, in order (the A/B of tools/shards_call.py)
    const char* thr_env = getenv("TS_SHARDS_THREADS");
    if (ngpu > 1 && !(thr_env && atoi(thr_env) == 0))
        for (int g = 0; g < ngpu; ++g) {
            std::unique_ptr<ShardWorker> w(new (std::nothrow) ShardWorker());
            if (!w) {
                ts_shards_destroy(s);
                return fail(TS_ERR_NOMEM, "host allocation failed");
            }
            w->device = s->device[g];
            ShardWorker* raw = w.get();
            s->worker.push_back(std::move(w));
            raw->th = std::thread([raw] { raw->run(); });
        }
    *out = s;
    return TS_OK;
}

extern "C" int ts_shards_info(const ts_shards* s, int32_t* ngpu, int64_t* n_total, int32_t* uses_rccl) {
    if (!s) return fail(TS_ERR_INVALID, "shards is NULL");
    if (ngpu) *ngpu = s->ngpu;
    if (n_total) *n_total = s->n_total;
    if (uses_rccl) *uses_rccl = s->use_rccl ? 1 : 0;
    return TS_OK;
}

// Borrowed handle of shard g (rows [lo, hi) of the corpus, *lo_out = lo): for uploads from device memory, options, profiling.
extern "C" int ts_shards_shard(ts_shards* s, int32_t g, ts_index** ix, int64_t* lo_out, int64_t* hi_out) {
    if (!s || g < 0 || g >= s->ngpu) return fail(TS_ERR_INVALID, "bad shard number");
    if (ix) *ix = s->shard[g];
    if (lo_out) *lo_out = s->lo[g];
    if (hi_out) *hi_out = s->lo[g + 1];
    return TS_OK;
}

// Global rows [row0, row0 + nrows) from host memory, routed to the shards that own them.
extern "C" int ts_shards_upload(ts_shards* s, const void* host_rows, int src_dtype, int64_t row0, int64_t nrows) {
    if (!s || (!host_rows && nrows > 0)) return fail(TS_ERR_INVALID, "NULL argument");
    if (src_dtype != TS_F32 && src_dtype != TS_BF16) return fail(TS_ERR_INVALID, "src_dtype %d", src_dtype);
    if (row0 < 0 || nrows < 0 || row0 + nrows > s->n_total) return fail(TS_ERR_INVALID, "rows outside the corpus");
    const size_t src_row = (size_t)s->d * (src_dtype == TS_BF16 ? 2 : 4);
    for (int g = 0; g < s->ngpu; ++g) {
        const int64_t a = std::max(row0, s->lo[g]), b = std::min(row0 + nrows, s->lo[g + 1]);
        if (a >= b) continue;
        TS_TRY(ts_index_upload(s->shard[g], (const char*)host_rows + (size_t)(a - row0) * src_row, src_dtype, a - s->lo[g], b - a));
    }
    return TS_OK;
}

// Host queries in, host results out (the serving call): every device searches its shard on its own stream, the packed
// per-shard results are exchanged with one ncclAllGather per device (grouped), device 0 merges and returns the answer.
extern "C" int ts_shards_search(ts_shards* s, const void* queries, int q_dtype, int32_t nq, int32_t k, float* out_scores,
                                int64_t* out_idx) {
    if (!s || !queries || !out_scores || !out_idx) return fail(TS_ERR_INVALID, "NULL argument");
    if (q_dtype != TS_F32 && q_dtype != TS_BF16) return fail(TS_ERR_INVALID, "q_dtype %d", q_dtype);
    if (nq < 0 || k < 1 || k > TS_MAX_K) return fail(TS_ERR_INVALID, "bad shape");
    if ((int64_t)s->ngpu * k > kMergeMax) return fail(TS_ERR_UNSUPPORTED, "ngpu * k exceeds %d", kMergeMax);
    if (nq == 0) return TS_OK;
    std::lock_guard<std::mutex> lock(s->mu);
    const int G = s->ngpu;
    const int64_t blk = packed_bytes(nq, k), idx_off = packed_idx_off(nq, k);
    const size_t qbytes = (size_t)nq * s->d * (q_dtype == TS_BF16 ? 2 : 4);
    std::vector<hipStream_t> st(G);
    for (int g = 0; g < G; ++g) {
        void* p = nullptr;
        TS_TRY(ts_index_stream(s->shard[g], &p));
        st[g] = (hipStream_t)p;
    }
    // every device's part: its buffers, the query copy, the search of its shard into its packed block - enqueued by the
    // shard's own thread, all of them at once
    const void* qsrc = queries;
    if (G > 1) {
        if (s->qpin_bytes < qbytes) {
            if (s->qpin) HIP_TRY(hipHostFree(s->qpin));
            s->qpin = nullptr;
            s->qpin_bytes = 0;
            HIP_TRY(hipHostMalloc(&s->qpin, qbytes, hipHostMallocPortable));
            s->qpin_bytes = qbytes;
        }
        memcpy(s->qpin, queries, qbytes);     // the previous call's copies are complete: every call ends synchronised
        qsrc = s->qpin;
    }
    auto enqueue = [s, qsrc, qbytes, blk, idx_off, G, q_dtype, nq, k, &st](int g) -> int {
        HIP_TRY(hipSetDevice(s->device[g]));
        TS_TRY(ensure(&s->qdev[g], &s->qdev_bytes[g], qbytes));
        TS_TRY(ensure(&s->mine[g], &s->mine_bytes[g], (size_t)blk));
        TS_TRY(ensure(&s->all[g], &s->all_bytes[g], (size_t)blk * G));
        HIP_TRY(hipMemcpyAsync(s->qdev[g], qsrc, qbytes, hipMemcpyHostToDevice, st[g]));
        TS_TRY(ts_search(s->shard[g], s->qdev[g], q_dtype, 1, nq, k, (float*)s->mine[g], (int64_t*)((char*)s->mine[g] + idx_off), 1, st[g]));
        if (!s->use_rccl) HIP_TRY(hipEventRecord(s->done[g], st[g]));
        return TS_OK;
    };
    if (s->worker.empty()) {
        for (int g = 0; g < G; ++g) TS_TRY(enqueue(g));
    } else {
        for (int g = 0; g < G; ++g) s->worker[g]->post([enqueue, g] { return enqueue(g); });
        int rc_all = TS_OK;
        char first_err[512] = "";
        for (int g = 0; g < G; ++g) {             // every worker is heard before anything returns (they hold `st` by reference)
            const int rc = s->worker[g]->wait();
            if (rc != TS_OK && rc_all == TS_OK) {
                rc_all = rc;
                snprintf(first_err, sizeof(first_err), "%s", g_err);
            }
        }
        if (rc_all != TS_OK) {
            for (int g = 0; g < G; ++g) {
                (void)hipSetDevice(s->device[g]);
                (void)hipStreamSynchronize(st[g]);
            }
            return fail(rc_all, "%s", first_err);
        }
    }
    if (s->use_rccl) {
        RcclApi* api = rccl_api();
        NCCL_TRY(api, api->GroupStart());
        for (int g = 0; g < G; ++g) {
            ncclResult_t r = api->AllGather(s->mine[g], s->all[g], (size_t)blk, ncclUint8, s->comm[g], st[g]);
            if (r != ncclSuccess) {
                api->GroupEnd();
                return fail(TS_ERR_HIP, "ncclAllGather on device %d failed: %s", s->device[g], api->GetErrorString(r));
            }
        }
        NCCL_TRY(api, api->GroupEnd());
    } else {
        // shards sharing a device (or a single shard): the root gathers the blocks with device copies
        HIP_TRY(hipSetDevice(s->device[0]));
        for (int g = 0; g < G; ++g) {
            if (g) HIP_TRY(hipStreamWaitEvent(st[0], s->done[g], 0));
            HIP_TRY(hipMemcpyPeerAsync((char*)s->all[0] + (size_t)g * blk, s->device[0], s->mine[g], s->device[g], (size_t)blk, st[0]));
        }
    }
    HIP_TRY(hipSetDevice(s->device[0]));
    TS_TRY(ensure(&s->fin, &s->fin_bytes, (size_t)nq * k * 12));
    int64_t* di = (int64_t*)s->fin;
    float* ds = (float*)(di + (size_t)nq * k);
    TS_TRY(launch_merge_packed(s->all[0], blk, idx_off, G, nq, k, k, ds, di, st[0]));
    HIP_TRY(hipMemcpyAsync(out_scores, ds, (size_t)nq * k * 4, hipMemcpyDeviceToHost, st[0]));
    HIP_TRY(hipMemcpyAsync(out_idx, di, (size_t)nq * k * 8, hipMemcpyDeviceToHost, st[0]));
    HIP_TRY(hipStreamSynchronize(st[0]));
    // the other devices' streams hold only their search + their all-gather: both are complete once the root's merge is
    // (the collective ends everywhere together); with the copy exchange their blocks have been read by the root
    if (s->use_rccl)
        for (int g = 1; g < G; ++g) {
            HIP_TRY(hipSetDevice(s->device[g]));
            HIP_TRY(hipStreamSynchronize(st[g]));
        }
    return TS_OK;
}
