// libtsearch.so - C ABI (include/tsearch.h) over the gfx950 kernels.  Host side only in this file:
// argument checking, device memory, launch sequencing.  No CPU compute path exists: without a
// HIP device every compute entry point fails with TS_ERR_NODEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cctype>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <mutex>
#include <new>
#include <type_traits>
#include <vector>

#include "../../include/tsearch.h"
#include "common.h"
#include "kernels_mfma.h"
#include "kernels_mfma16.h"
#include "kernels_mfma_f32.h"
#include "kernels_attention.h"
#include "kernels_prep.h"
#include "kernels_sample.h"
#include "kernels_scan.h"
#include "kernels_select.h"

using namespace ts;

// ---------------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return fail(e_ == hipErrorOutOfMemory ? TS_ERR_NOMEM : TS_ERR_HIP, "%s failed: %s (%s:%d)", \
                        #expr, hipGetErrorString(e_), __FILE__, __LINE__);                              \
    } while (0)

#define TS_TRY(expr)            \
    do {                        \
        int rc_ = (expr);       \
        if (rc_ != TS_OK) return rc_; \
    } while (0)

// Tuning / diagnostic knobs.  Read from the environment ONCE per handle (ts_index_create / ts_index_view /
// ts_index_subset), changed afterwards only through ts_index_set_option: no getenv on the search path.
enum Knob {
    K_MFMA_MIN_RANK, K_MFMA_VARIANT, K_MFMA_GROUPS, K_MFMA_GRID, K_MFMA_STAT, K_MFMA_STAT_CANDS, K_MFMA_TAIL_FIT,
    K_MFMA_NO_IDLE, K_MFMA_AHEAD, K_MFMA_TARGET_CANDS, K_MFMA_FIRST_ROWS, K_MFMA_TARGET_SPARSE, K_MFMA_RUN,
    K_MFMA_MIN_ROWS, K_MFMA_SHAPE, K_MFMA_F32, K_SCAN_GENERIC, K_SCAN_MAX_QUERIES, K_MFMA_BALANCE, K_PROBE_SPREAD, K_MFMA_SAMPLE,
    K_COUNT
};
static const char* const kKnobNames[K_COUNT] = {
    "TS_MFMA_MIN_RANK", "TS_MFMA_VARIANT", "TS_MFMA_GROUPS", "TS_MFMA_GRID", "TS_MFMA_STAT", "TS_MFMA_STAT_CANDS",
    "TS_MFMA_TAIL_FIT", "TS_MFMA_NO_IDLE", "TS_MFMA_AHEAD", "TS_MFMA_TARGET_CANDS", "TS_MFMA_FIRST_ROWS",
    "TS_MFMA_TARGET_SPARSE", "TS_MFMA_RUN", "TS_MFMA_MIN_ROWS", "TS_MFMA_SHAPE", "TS_MFMA_F32", "TS_SCAN_GENERIC",
    "TS_SCAN_MAX_QUERIES", "TS_MFMA_BALANCE", "TS_PROBE_SPREAD", "TS_MFMA_SAMPLE"};
struct Knobs {
    int v[K_COUNT];
    bool set[K_COUNT];
    Knobs() {
        for (int i = 0; i < K_COUNT; ++i) {
            const char* e = getenv(kKnobNames[i]);
            set[i] = e && *e;
            v[i] = set[i] ? atoi(e) : 0;
        }
#ifndef TS_DIAG
        set[K_MFMA_VARIANT] = false;      // the timing-only kernel variants exist in the diagnostic build only (make diag)
        v[K_MFMA_VARIANT] = 0;
#endif
    }
    int get(Knob k, int dflt) const { return set[k] ? v[k] : dflt; }
};

// ---------------------------------------------------------------------------------------------
// handles
// ---------------------------------------------------------------------------------------------
constexpr int kQBlock = 256;          // queries per pass of the search driver
constexpr int kCandCap = 8192;        // candidate slots per query (MFMA path)
constexpr int kScanGridPerCU = 4;
constexpr size_t kStageBytes = (size_t)256 << 20;

struct ts_index {
    int device = 0;
    int64_t n = 0, n_pad = 0, ld = 0, row_offset = 0;
    int d = 0, dtype = 0, metric = 0;
    int cu_count = 256;
    void* rows = nullptr;
    hipStream_t stream = nullptr;
    std::mutex mu;
    // scratch (lazily sized)
    void* stage = nullptr;      size_t stage_bytes = 0;      // host->device staging
    void* qstore = nullptr;     float* qf32 = nullptr;       // prepared queries [256 x ld]
    u64* cand = nullptr;        u32* count = nullptr;        float* thr = nullptr;
    u64* priv = nullptr;        u32* pcount = nullptr;       int priv_writers = 0;  // MFMA path: lane-private candidate lists
    float* sample = nullptr;                                 // MFMA path: dense [256 x 8192] score matrix of the threshold sample
    bool rebalance_pending = false; int rebalance_grid = 0;  // the exact re-run's launch also moves the full pass's tile boundaries
    int* fb_list = nullptr;     int* fb_count = nullptr;     unsigned long long* stat = nullptr;
    u64* partial = nullptr;     u64* partial2 = nullptr;     size_t partial_bytes = 0;
    float* res_scores = nullptr; int64_t* res_idx = nullptr; size_t res_cap = 0;  // device result buffers (entries)
    u32* mask_dev = nullptr;    size_t mask_bytes = 0;       // filtered search: device copy of a host bitmask
    float* bias_dev = nullptr;  size_t bias_bytes = 0;       // biased search: device copy of a host bias array
    const float* active_bias = nullptr; float active_bias_w = 0.f;   // per-row additive term of the search in progress (under `mu`)
    int64_t* id_map = nullptr;                               // subset index: local row -> global id
    bool borrowed = false;                                   // a view: rows / id_map belong to another handle
    bool attached = false;                                   // rows adopted from the caller (ts_index_attach_device): never freed here
    ts_index* parent = nullptr;                              // a view: the handle that owns the rows
    std::atomic<int> nviews{0};                              // live views of this handle (it cannot grow meanwhile)
    void* rank_buf = nullptr;   size_t rank_bytes = 0;       // ts_rank_of: targets | counts | target scores, one query block
    const u32* active_mask = nullptr;                        // bitmask of the search in progress (under `mu`)
    int64_t active_allowed = 0;                              // rows that bitmask allows (host masks: counted; else n)
    bool attr_done = false;
    bool attr_done_hist = false;
    Knobs knobs;                                             // env at creation, then ts_index_set_option
    hipStream_t last_stream = nullptr;                       // stream the previous call ran on: compared, never used (it may be gone)
    hipEvent_t order_ev = nullptr;                           // recorded at the end of every call on that call's stream: orders the
    bool ordered = false;                                    //   next call behind it (`ordered`: recorded at least once)
    int64_t* part = nullptr;    unsigned* wg_ticks = nullptr;    // full pass of the 16x16 kernel: tile boundaries per workgroup, their times
    int part_g = 0;             int64_t part_ntiles = -1;        // ... the grid and tile count the table was made for
    unsigned long long* dbg = nullptr;                       // TS_MFMA_VARIANT=3: per-wave cycle sums / clock probe
    double probe_ghz = 0.0, probe_cycles_per_unit = 0.0, probe_units = 0.0;   // last clock probe (16x16 shape, VARIANT 3)
    // optional event brackets around the dominant kernel (ts_index_profile_*)
    bool profiling = false;
    std::vector<hipEvent_t> ev_pool;   // pairs: [2i] start, [2i+1] stop
    size_t ev_used = 0;                // events handed out since the last read
    int64_t prof_rows = 0;
    size_t elem() const { return dtype == TS_BF16 ? 2 : 4; }
};

struct ts_timer {
    int device = 0;
    hipEvent_t a = nullptr, b = nullptr;
};

static int ensure(void** p, size_t* have, size_t want) {
    if (*have >= want && *p) return TS_OK;
    if (*p) HIP_TRY(hipFree(*p));
    *p = nullptr;
    *have = 0;
    HIP_TRY(hipMalloc(p, want));
    *have = want;
    return TS_OK;
}

// Event bracket around one launch: prof_begin records the start event and returns the stop event
// (NULL when profiling is off); the caller records it with prof_end after the launch.
static hipEvent_t prof_begin(ts_index* ix, hipStream_t st, int64_t rows) {
    if (!ix->profiling) return nullptr;
    if (ix->ev_used + 2 > ix->ev_pool.size()) {
        if (ix->ev_pool.size() >= 16384) return nullptr;
        hipEvent_t a = nullptr, b = nullptr;
        if (hipEventCreate(&a) != hipSuccess) return nullptr;
        if (hipEventCreate(&b) != hipSuccess) { hipEventDestroy(a); return nullptr; }
        ix->ev_pool.push_back(a);
        ix->ev_pool.push_back(b);
    }
    hipEvent_t start = ix->ev_pool[ix->ev_used], stop = ix->ev_pool[ix->ev_used + 1];
    ix->ev_used += 2;
    ix->prof_rows = rows;
    hipEventRecord(start, st);
    return stop;
}
static void prof_end(hipEvent_t stop, hipStream_t st) {
    if (stop) hipEventRecord(stop, st);
}

// Stream of this call (NULL = the index's own).  The per-handle scratch buffers are shared by all calls: a call that
// arrives on ANOTHER stream than the previous one is ordered behind it, so that it never overwrites scratch the first
// one still reads.  The order event is recorded at the END of every entry point, on the stream of that call, while that
// stream is known to be alive (StreamScope's destructor, on every return path); the next call only waits on the event,
// and ts_index_synchronize / ts_index_destroy only synchronise on it: a caller's stream handle is never touched after
// the call that was given it has returned, so the caller may destroy the stream whenever its own work on it is done.
// Called under ix->mu.
// The index's own stream is a BLOCKING stream: it orders with the legacy null stream, which is what a torch
// default stream's handle (0 = NULL here) means - encoder kernels before an upload / search, torch ops after it.
struct StreamScope {
    ts_index* ix = nullptr;
    hipStream_t st = nullptr;
    ~StreamScope() {
        if (!ix || !ix->order_ev) return;
        if (hipEventRecord(ix->order_ev, st) == hipSuccess) {
            ix->last_stream = st;
            ix->ordered = true;
        } else {
            (void)hipGetLastError();
        }
    }
};
static int enter_stream(ts_index* ix, void* stream, hipStream_t* out, StreamScope* scope) {
    hipStream_t st = stream ? (hipStream_t)stream : ix->stream;
    if (!ix->order_ev) HIP_TRY(hipEventCreateWithFlags(&ix->order_ev, hipEventDisableTiming));
    if (ix->ordered && ix->last_stream != st) HIP_TRY(hipStreamWaitEvent(st, ix->order_ev, 0));
    scope->ix = ix;
    scope->st = st;
    *out = st;
    return TS_OK;
}

// Device buffer freed on every return path.
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
    template <class T> T* as() const { return (T*)p; }
};

// ---------------------------------------------------------------------------------------------
// library / device
// ---------------------------------------------------------------------------------------------
extern "C" int ts_version(void) { return TS_VERSION; }
extern "C" const char* ts_last_error(void) { return g_err; }

extern "C" int ts_device_count(int* count) {
    if (!count) return fail(TS_ERR_INVALID, "count is NULL");
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) c = 0;
    (void)hipGetLastError();
    *count = c;
    return TS_OK;
}

static int check_device(int device) {
    int c = 0;
    ts_device_count(&c);
    if (c <= 0) return fail(TS_ERR_NODEVICE, "no HIP device visible (libtsearch has no CPU path)");
    if (device < 0 || device >= c) return fail(TS_ERR_INVALID, "device %d out of range [0, %d)", device, c);
    return TS_OK;
}

extern "C" int ts_device_info(int device, char* name, int name_len, int64_t* total_mem, int32_t* cus) {
    TS_TRY(check_device(device));
    hipDeviceProp_t p;
    HIP_TRY(hipGetDeviceProperties(&p, device));
    if (name && name_len > 0) snprintf(name, name_len, "%s (%s)", p.name, p.gcnArchName);
    if (total_mem) *total_mem = (int64_t)p.totalGlobalMem;
    if (cus) *cus = p.multiProcessorCount;
    return TS_OK;
}

extern "C" int ts_device_synchronize(int device) {
    TS_TRY(check_device(device));
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipDeviceSynchronize());
    return TS_OK;
}

// ---------------------------------------------------------------------------------------------
// index
// ---------------------------------------------------------------------------------------------
extern "C" int ts_index_create(int device, int64_t n, int32_t d, int dtype, int metric, ts_index** out) {
    if (!out) return fail(TS_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (n < 0 || n > 0xFFFFFFF0ll) return fail(TS_ERR_INVALID, "n = %lld out of range", (long long)n);
    if (d <= 0 || d > 16384) return fail(TS_ERR_INVALID, "d = %d out of range [1, 16384]", d);
    if (dtype != TS_F32 && dtype != TS_BF16) return fail(TS_ERR_INVALID, "dtype %d", dtype);
    if (metric != TS_METRIC_IP && metric != TS_METRIC_COS) return fail(TS_ERR_INVALID, "metric %d", metric);
    TS_TRY(check_device(device));
    HIP_TRY(hipSetDevice(device));
    ts_index* ix = new (std::nothrow) ts_index();
    if (!ix) return fail(TS_ERR_NOMEM, "host allocation failed");
    ix->device = device;
    ix->n = n;
    ix->n_pad = std::max<int64_t>(kRowPad, (n + kRowPad - 1) / kRowPad * kRowPad);
    ix->d = d;
    ix->ld = (d + kLdPad - 1) / kLdPad * kLdPad;
    ix->dtype = dtype;
    ix->metric = metric;
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, device) == hipSuccess) ix->cu_count = p.multiProcessorCount;
    const size_t bytes = (size_t)ix->n_pad * ix->ld * ix->elem();
    hipError_t e = hipMalloc(&ix->rows, bytes);
    if (e != hipSuccess) {
        delete ix;
        return fail(TS_ERR_NOMEM, "hipMalloc of %zu bytes for the index failed: %s", bytes, hipGetErrorString(e));
    }
    e = hipStreamCreateWithFlags(&ix->stream, hipStreamDefault);
    if (e == hipSuccess) e = hipMemsetAsync(ix->rows, 0, bytes, ix->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ix->stream);
    if (e != hipSuccess) {
        hipFree(ix->rows);
        delete ix;
        return fail(TS_ERR_HIP, "index initialisation failed: %s", hipGetErrorString(e));
    }
    *out = ix;
    return TS_OK;
}

extern "C" int ts_index_view(ts_index* src, ts_index** out) {
    if (!out) return fail(TS_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!src) return fail(TS_ERR_INVALID, "index is NULL");
    std::lock_guard<std::mutex> lock(src->mu);      // not while the rows are being moved by an append
    HIP_TRY(hipSetDevice(src->device));
    ts_index* ix = new (std::nothrow) ts_index();
    if (!ix) return fail(TS_ERR_NOMEM, "host allocation failed");
    ix->device = src->device;
    ix->n = src->n;
    ix->n_pad = src->n_pad;
    ix->d = src->d;
    ix->ld = src->ld;
    ix->dtype = src->dtype;
    ix->metric = src->metric;
    ix->row_offset = src->row_offset;
    ix->cu_count = src->cu_count;
    ix->rows = src->rows;
    ix->id_map = src->id_map;
    ix->borrowed = true;
    if (hipStreamCreateWithFlags(&ix->stream, hipStreamDefault) != hipSuccess) {
        delete ix;
        return fail(TS_ERR_HIP, "stream creation failed");
    }
    ix->parent = src;
    src->nviews.fetch_add(1);
    *out = ix;
    return TS_OK;
}

extern "C" int ts_index_destroy(ts_index* ix) {
    if (!ix) return TS_OK;
    if (ix->nviews.load() > 0)
        return fail(TS_ERR_UNSUPPORTED, "the index has %d live views: destroy them first (they read its rows)", ix->nviews.load());
    hipSetDevice(ix->device);
    if (ix->ordered && ix->order_ev) hipEventSynchronize(ix->order_ev);   // scratch still in use by a call on a caller's stream
    if (ix->stream) hipStreamSynchronize(ix->stream);
    if (ix->borrowed) {
        ix->rows = nullptr;
        ix->id_map = nullptr;
        if (ix->parent) ix->parent->nviews.fetch_sub(1);
    }
    if (ix->attached) ix->rows = nullptr;
    void* ptrs[] = {ix->rows,  ix->stage,   ix->qstore,   ix->qf32,     ix->cand,       ix->count,  ix->thr, ix->priv, ix->pcount, ix->sample, ix->mask_dev, ix->bias_dev, ix->rank_buf, ix->id_map,
                    ix->fb_list, ix->fb_count, ix->stat, ix->partial, ix->partial2, ix->res_scores, ix->res_idx, ix->dbg, ix->part, ix->wg_ticks};
    for (void* p : ptrs)
        if (p) hipFree(p);
    for (hipEvent_t e : ix->ev_pool) hipEventDestroy(e);
    if (ix->order_ev) hipEventDestroy(ix->order_ev);
    if (ix->stream) hipStreamDestroy(ix->stream);
    delete ix;
    return TS_OK;
}

extern "C" int ts_index_set_row_offset(ts_index* ix, int64_t off) {
    if (!ix || off < 0) return fail(TS_ERR_INVALID, "bad argument");
    ix->row_offset = off;
    return TS_OK;
}

extern "C" int ts_index_set_option(ts_index* ix, const char* name, int32_t value) {
    if (!ix || !name) return fail(TS_ERR_INVALID, "NULL argument");
    for (int i = 0; i < K_COUNT; ++i)
        if (!strcmp(name, kKnobNames[i])) {
#ifndef TS_DIAG
            if (i == K_MFMA_VARIANT && value != 0)
                return fail(TS_ERR_UNSUPPORTED, "TS_MFMA_VARIANT = %d: the timing-only kernel variants are compiled into the "
                            "diagnostic build only (make -C theoremsearch_amd/csrc diag; TS_LIB selects it)", value);
#endif
            std::lock_guard<std::mutex> lock(ix->mu);
            ix->knobs.v[i] = value;
            ix->knobs.set[i] = true;
            return TS_OK;
        }
    return fail(TS_ERR_INVALID, "unknown option '%s'", name);
}

extern "C" int ts_index_reset_option(ts_index* ix, const char* name) {
    if (!ix || !name) return fail(TS_ERR_INVALID, "NULL argument");
    for (int i = 0; i < K_COUNT; ++i)
        if (!strcmp(name, kKnobNames[i])) {
            std::lock_guard<std::mutex> lock(ix->mu);
            ix->knobs.set[i] = false;
            return TS_OK;
        }
    return fail(TS_ERR_INVALID, "unknown option '%s'", name);
}

extern "C" int ts_index_subset(ts_index* src, const int64_t* rows, int64_t nrows, ts_index** out) {
    if (!out) return fail(TS_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!src || (!rows && nrows > 0) || nrows < 0) return fail(TS_ERR_INVALID, "bad argument");
    if (src->id_map) return fail(TS_ERR_UNSUPPORTED, "subset of a subset index: take the subset of the original index");
    for (int64_t i = 0; i < nrows; ++i) {
        const int64_t r = rows[i] - src->row_offset;
        if (r < 0 || r >= src->n) return fail(TS_ERR_INVALID, "rows[%lld] = %lld is not in the index", (long long)i, (long long)rows[i]);
        if (i && rows[i] <= rows[i - 1]) return fail(TS_ERR_INVALID, "rows must be strictly ascending (at %lld)", (long long)i);
    }
    ts_index* ix = nullptr;
    TS_TRY(ts_index_create(src->device, nrows, src->d, src->dtype, src->metric, &ix));
    if (nrows == 0) {
        *out = ix;
        return TS_OK;
    }
    std::lock_guard<std::mutex> lock(src->mu);
    hipStream_t src_own;
    StreamScope src_scope;
    if (enter_stream(src, nullptr, &src_own, &src_scope) != TS_OK) {   // uploads enqueued on the source's own or a caller's stream
        ts_index_destroy(ix);
        return TS_ERR_HIP;
    }
    hipStreamSynchronize(src_own);
    hipError_t e = hipMalloc((void**)&ix->id_map, (size_t)nrows * 8);
    if (e == hipSuccess) e = hipMemcpyAsync(ix->id_map, rows, (size_t)nrows * 8, hipMemcpyHostToDevice, ix->stream);
    if (e == hipSuccess) {
        const int grid = (int)std::min<int64_t>((nrows + 3) / 4, 8192);
        gather_rows_kernel<<<grid, 256, 0, ix->stream>>>((const unsigned char*)src->rows, (unsigned char*)ix->rows, ix->id_map,
                                                         src->row_offset, nrows, (int64_t)src->ld * src->elem());
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ix->stream);
    if (e != hipSuccess) {
        ts_index_destroy(ix);
        return fail(TS_ERR_HIP, "subset copy failed: %s", hipGetErrorString(e));
    }
    *out = ix;
    return TS_OK;
}

extern "C" int ts_index_synchronize(ts_index* ix) {
    if (!ix) return fail(TS_ERR_INVALID, "index is NULL");
    std::lock_guard<std::mutex> lock(ix->mu);
    HIP_TRY(hipSetDevice(ix->device));
    if (ix->ordered && ix->order_ev) HIP_TRY(hipEventSynchronize(ix->order_ev));   // the end of the last call, whatever stream it ran on
    HIP_TRY(hipStreamSynchronize(ix->stream));
    ix->ordered = false;            // nothing in flight: the next call orders behind nothing
    ix->last_stream = nullptr;
    return TS_OK;
}

// `stream` waits for the end of the last call on this handle (its order event): how a caller's side stream picks up the
// results of a search without recording an event of its own on the search's stream (one marker packet less per step).
extern "C" int ts_index_wait_order(ts_index* ix, void* stream) {
    if (!ix) return fail(TS_ERR_INVALID, "index is NULL");
    std::lock_guard<std::mutex> lock(ix->mu);
    HIP_TRY(hipSetDevice(ix->device));
    if (ix->ordered && ix->order_ev && ix->last_stream != (hipStream_t)stream) HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, ix->order_ev, 0));
    return TS_OK;
}

extern "C" int ts_index_stream(const ts_index* ix, void** stream) {
    if (!ix || !stream) return fail(TS_ERR_INVALID, "NULL argument");
    *stream = (void*)ix->stream;
    return TS_OK;
}

extern "C" int ts_index_info(const ts_index* ix, int64_t* n, int32_t* d, int32_t* dtype, int32_t* metric, int64_t* ld,
                             int64_t* row_offset, void** rows) {
    if (!ix) return fail(TS_ERR_INVALID, "index is NULL");
    if (n) *n = ix->n;
    if (d) *d = ix->d;
    if (dtype) *dtype = ix->dtype;
    if (metric) *metric = ix->metric;
    if (ld) *ld = ix->ld;
    if (row_offset) *row_offset = ix->row_offset;
    if (rows) *rows = ix->rows;
    return TS_OK;
}

template <int SRC, int DST>
static void launch_prep(bool normalize, const void* src, int64_t src_ld, void* dst, float* f32copy, int64_t ld, int d,
                        int64_t nrows, int64_t rows_total, hipStream_t st) {
    const int64_t waves = std::max<int64_t>(1, rows_total);
    const int grid = (int)std::min<int64_t>((waves + 3) / 4, 4096);
    if (normalize)
        prep_rows_kernel<SRC, DST, true><<<grid, 256, 0, st>>>(src, src_ld, dst, f32copy, ld, d, nrows, rows_total);
    else
        prep_rows_kernel<SRC, DST, false><<<grid, 256, 0, st>>>(src, src_ld, dst, f32copy, ld, d, nrows, rows_total);
}

static int prep_dispatch(int src_dtype, int dst_dtype, bool normalize, const void* src, int64_t src_ld, void* dst,
                         float* f32copy, int64_t ld, int d, int64_t nrows, int64_t rows_total, hipStream_t st) {
    if (src_dtype == TS_F32 && dst_dtype == TS_F32)
        launch_prep<0, 0>(normalize, src, src_ld, dst, f32copy, ld, d, nrows, rows_total, st);
    else if (src_dtype == TS_F32 && dst_dtype == TS_BF16)
        launch_prep<0, 1>(normalize, src, src_ld, dst, f32copy, ld, d, nrows, rows_total, st);
    else if (src_dtype == TS_BF16 && dst_dtype == TS_F32)
        launch_prep<1, 0>(normalize, src, src_ld, dst, f32copy, ld, d, nrows, rows_total, st);
    else
        launch_prep<1, 1>(normalize, src, src_ld, dst, f32copy, ld, d, nrows, rows_total, st);
    HIP_TRY(hipGetLastError());
    return TS_OK;
}

static int check_rows(const ts_index* ix, const void* p, int src_dtype, int64_t row0, int64_t nrows, bool write) {
    if (!ix || !p) return fail(TS_ERR_INVALID, "NULL argument");
    if (src_dtype != TS_F32 && src_dtype != TS_BF16) return fail(TS_ERR_INVALID, "src_dtype %d", src_dtype);
    if (write && ix->id_map) return fail(TS_ERR_UNSUPPORTED, "a subset index is read-only");
    if (write && ix->borrowed) return fail(TS_ERR_UNSUPPORTED, "a view is read-only: upload through the handle that owns the rows");
    if (row0 < 0 || nrows < 0 || row0 + nrows > ix->n)
        return fail(TS_ERR_INVALID, "rows [%lld, %lld) outside the index of %lld rows", (long long)row0,
                    (long long)(row0 + nrows), (long long)ix->n);
    return TS_OK;
}

// Uploads: the *_locked forms run under ix->mu (the public entry points and the append calls take it).
static int upload_device_locked(ts_index* ix, const void* dev_rows, int src_dtype, int64_t src_ld, int64_t row0, int64_t nrows,
                                void* stream) {
    TS_TRY(check_rows(ix, dev_rows, src_dtype, row0, nrows, true));
    if (src_ld < ix->d) return fail(TS_ERR_INVALID, "src_ld %lld < d %d", (long long)src_ld, ix->d);
    if (nrows == 0) return TS_OK;
    HIP_TRY(hipSetDevice(ix->device));
    hipStream_t st;
    StreamScope scope;
    TS_TRY(enter_stream(ix, stream, &st, &scope));
    char* dst = (char*)ix->rows + (size_t)row0 * ix->ld * ix->elem();
    return prep_dispatch(src_dtype, ix->dtype, ix->metric == TS_METRIC_COS, dev_rows, src_ld, dst, nullptr, ix->ld, ix->d,
                         nrows, nrows, st);
}

static int upload_host_locked(ts_index* ix, const void* host_rows, int src_dtype, int64_t row0, int64_t nrows) {
    TS_TRY(check_rows(ix, host_rows, src_dtype, row0, nrows, true));
    if (nrows == 0) return TS_OK;
    HIP_TRY(hipSetDevice(ix->device));
    const size_t src_elem = src_dtype == TS_BF16 ? 2 : 4;
    const size_t src_row = (size_t)ix->d * src_elem;
    hipStream_t own;
    StreamScope scope;
    TS_TRY(enter_stream(ix, nullptr, &own, &scope));   // rows / the stage buffer may still feed a call enqueued on a caller's stream
    if (src_dtype == ix->dtype && ix->metric == TS_METRIC_IP && ix->ld == ix->d) {
        // stored as given: straight copy into place
        HIP_TRY(hipMemcpyAsync((char*)ix->rows + (size_t)row0 * src_row, host_rows, (size_t)nrows * src_row, hipMemcpyHostToDevice, own));
        HIP_TRY(hipStreamSynchronize(own));
        return TS_OK;
    }
    TS_TRY(ensure(&ix->stage, &ix->stage_bytes, kStageBytes));
    const int64_t rows_per = std::max<int64_t>(1, (int64_t)(kStageBytes / src_row));
    for (int64_t r = 0; r < nrows; r += rows_per) {
        const int64_t cnt = std::min(rows_per, nrows - r);
        HIP_TRY(hipMemcpyAsync(ix->stage, (const char*)host_rows + (size_t)r * src_row, (size_t)cnt * src_row,
                               hipMemcpyHostToDevice, own));
        char* dst = (char*)ix->rows + (size_t)(row0 + r) * ix->ld * ix->elem();
        TS_TRY(prep_dispatch(src_dtype, ix->dtype, ix->metric == TS_METRIC_COS, ix->stage, ix->d, dst, nullptr, ix->ld,
                             ix->d, cnt, cnt, own));
        HIP_TRY(hipStreamSynchronize(own));  // the stage buffer is reused by the next chunk
    }
    return TS_OK;
}

extern "C" int ts_index_upload_device(ts_index* ix, const void* dev_rows, int src_dtype, int64_t src_ld, int64_t row0,
                                      int64_t nrows, void* stream) {
    if (!ix) return fail(TS_ERR_INVALID, "index is NULL");
    std::lock_guard<std::mutex> lock(ix->mu);
    return upload_device_locked(ix, dev_rows, src_dtype, src_ld, row0, nrows, stream);
}

extern "C" int ts_index_upload(ts_index* ix, const void* host_rows, int src_dtype, int64_t row0, int64_t nrows) {
    if (!ix) return fail(TS_ERR_INVALID, "index is NULL");
    std::lock_guard<std::mutex> lock(ix->mu);
    return upload_host_locked(ix, host_rows, src_dtype, row0, nrows);
}

// ---- growth: reserve / append ------------------------------------------------------
// The allocation holds n_pad rows (a multiple of 256, zero beyond n); appending past it moves the rows to a larger
// allocation (1.5x, at least what is asked).  Refused while views of the index exist (they hold the old pointer).
static int grow_locked(ts_index* ix, int64_t want_rows) {
    if (want_rows > 0xFFFFFFF0ll) return fail(TS_ERR_INVALID, "capacity %lld out of range", (long long)want_rows);
    const int64_t new_pad = std::max<int64_t>(kRowPad, (want_rows + kRowPad - 1) / kRowPad * kRowPad);
    if (new_pad <= ix->n_pad) return TS_OK;
    if (ix->borrowed || ix->id_map) return fail(TS_ERR_UNSUPPORTED, "a view / subset index cannot grow");
    if (ix->attached) return fail(TS_ERR_UNSUPPORTED, "an index over attached rows cannot grow: the rows belong to the caller");
    if (ix->nviews.load() > 0) return fail(TS_ERR_UNSUPPORTED, "the index has %d live views: destroy them before growing it", ix->nviews.load());
    HIP_TRY(hipSetDevice(ix->device));
    const size_t row_bytes = (size_t)ix->ld * ix->elem();
    const size_t old_bytes = (size_t)ix->n_pad * row_bytes, new_bytes = (size_t)new_pad * row_bytes;
    void* fresh = nullptr;
    hipError_t e = hipMalloc(&fresh, new_bytes);
    if (e != hipSuccess) return fail(TS_ERR_NOMEM, "hipMalloc of %zu bytes for the grown index failed: %s", new_bytes, hipGetErrorString(e));
    hipStream_t own;
    StreamScope scope;
    int rc = enter_stream(ix, nullptr, &own, &scope);
    if (rc == TS_OK) {
        e = hipMemcpyAsync(fresh, ix->rows, old_bytes, hipMemcpyDeviceToDevice, own);
        if (e == hipSuccess) e = hipMemsetAsync((char*)fresh + old_bytes, 0, new_bytes - old_bytes, own);
        if (e == hipSuccess) e = hipStreamSynchronize(own);
        if (e != hipSuccess) rc = fail(TS_ERR_HIP, "moving the rows failed: %s", hipGetErrorString(e));
    }
    if (rc != TS_OK) {
        hipFree(fresh);
        return rc;
    }
    hipFree(ix->rows);
    ix->rows = fresh;
    ix->n_pad = new_pad;
    return TS_OK;
}

// Zero-copy: the index adopts rows that already sit in device memory (SURVEY.md section 8b "ts_index_attach_device": the
// encoder's output tensor as the corpus).  The rows must be what the kernels multiply: the index's storage dtype, row
// stride = the index's ld (d padded to 64 elements, zeros in the padding), already normalised when the metric is cosine,
// and the allocation must hold capacity_rows >= n rounded up to 256 rows (the matrix kernels read whole 32-row tiles;
// rows past n are never returned).  The caller keeps ownership and must keep the memory alive and unchanged while
// searches run; uploads into an attached index write into the caller's memory.
extern "C" int ts_index_attach_device(ts_index* ix, void* dev_rows, int64_t capacity_rows) {
    if (!ix || !dev_rows) return fail(TS_ERR_INVALID, "NULL argument");
    std::lock_guard<std::mutex> lock(ix->mu);
    if (ix->id_map || ix->parent) return fail(TS_ERR_UNSUPPORTED, "a view / subset index cannot adopt rows");
    if (ix->nviews.load() > 0) return fail(TS_ERR_UNSUPPORTED, "the index has live views");
    const int64_t need = std::max<int64_t>(kRowPad, (ix->n + kRowPad - 1) / kRowPad * kRowPad);
    if (capacity_rows < need)
        return fail(TS_ERR_INVALID, "the attached allocation holds %lld rows, %lld are needed (n = %lld rounded up to %d)",
                    (long long)capacity_rows, (long long)need, (long long)ix->n, kRowPad);
    if (((uintptr_t)dev_rows & 15) != 0) return fail(TS_ERR_INVALID, "attached rows must be 16-byte aligned");
    HIP_TRY(hipSetDevice(ix->device));
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, dev_rows) != hipSuccess || attr.type != hipMemoryTypeDevice || attr.device != ix->device) {
        (void)hipGetLastError();
        return fail(TS_ERR_INVALID, "attached rows are not device memory of device %d", ix->device);
    }
    hipStream_t own;
    StreamScope scope;
    TS_TRY(enter_stream(ix, nullptr, &own, &scope));
    HIP_TRY(hipStreamSynchronize(own));
    if (!ix->attached && ix->rows) HIP_TRY(hipFree(ix->rows));
    ix->rows = dev_rows;
    ix->n_pad = capacity_rows / kRowPad * kRowPad;
    ix->attached = true;       // destroy / grow must not free it
    return TS_OK;
}

extern "C" int ts_index_reserve(ts_index* ix, int64_t capacity_rows) {
    if (!ix || capacity_rows < 0) return fail(TS_ERR_INVALID, "bad argument");
    std::lock_guard<std::mutex> lock(ix->mu);
    return grow_locked(ix, capacity_rows);
}

static int append_locked(ts_index* ix, int64_t nrows, int64_t* first_row) {
    if (nrows < 0) return fail(TS_ERR_INVALID, "nrows = %lld", (long long)nrows);
    if (ix->borrowed || ix->id_map) return fail(TS_ERR_UNSUPPORTED, "a view / subset index is read-only");
    if (ix->n + nrows > ix->n_pad) TS_TRY(grow_locked(ix, std::max(ix->n + nrows, ix->n_pad + ix->n_pad / 2)));
    if (first_row) *first_row = ix->n + ix->row_offset;
    return TS_OK;
}

extern "C" int ts_index_append(ts_index* ix, const void* host_rows, int src_dtype, int64_t nrows, int64_t* first_row) {
    if (!ix || (!host_rows && nrows > 0)) return fail(TS_ERR_INVALID, "NULL argument");
    std::lock_guard<std::mutex> lock(ix->mu);
    TS_TRY(append_locked(ix, nrows, first_row));
    const int64_t old_n = ix->n;
    ix->n += nrows;
    const int rc = nrows ? upload_host_locked(ix, host_rows, src_dtype, old_n, nrows) : TS_OK;
    if (rc != TS_OK) ix->n = old_n;
    return rc;
}

extern "C" int ts_index_append_device(ts_index* ix, const void* dev_rows, int src_dtype, int64_t src_ld, int64_t nrows,
                                      void* stream, int64_t* first_row) {
    if (!ix || (!dev_rows && nrows > 0)) return fail(TS_ERR_INVALID, "NULL argument");
    std::lock_guard<std::mutex> lock(ix->mu);
    TS_TRY(append_locked(ix, nrows, first_row));
    const int64_t old_n = ix->n;
    ix->n += nrows;
    const int rc = nrows ? upload_device_locked(ix, dev_rows, src_dtype, src_ld, old_n, nrows, stream) : TS_OK;
    if (rc != TS_OK) ix->n = old_n;
    return rc;
}

extern "C" int ts_index_download(ts_index* ix, void* host_rows, int64_t row0, int64_t nrows) {
    if (!ix) return fail(TS_ERR_INVALID, "index is NULL");
    std::lock_guard<std::mutex> lock(ix->mu);
    TS_TRY(check_rows(ix, host_rows, ix->dtype, row0, nrows, false));
    if (nrows == 0) return TS_OK;
    HIP_TRY(hipSetDevice(ix->device));
    const size_t row_bytes = (size_t)ix->d * ix->elem();
    TS_TRY(ensure(&ix->stage, &ix->stage_bytes, kStageBytes));
    hipStream_t own;
    StreamScope scope;
    TS_TRY(enter_stream(ix, nullptr, &own, &scope));
    const int64_t rows_per = std::max<int64_t>(1, (int64_t)(kStageBytes / row_bytes));
    for (int64_t r = 0; r < nrows; r += rows_per) {
        const int64_t cnt = std::min(rows_per, nrows - r);
        const char* src = (const char*)ix->rows + (size_t)(row0 + r) * ix->ld * ix->elem();
        if (ix->dtype == TS_F32)
            unpad_rows_kernel<0><<<1024, 256, 0, ix->stream>>>(src, ix->ld, ix->stage, ix->d, cnt);
        else
            unpad_rows_kernel<1><<<1024, 256, 0, ix->stream>>>(src, ix->ld, ix->stage, ix->d, cnt);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync((char*)host_rows + (size_t)r * row_bytes, ix->stage, (size_t)cnt * row_bytes,
                               hipMemcpyDeviceToHost, ix->stream));
        HIP_TRY(hipStreamSynchronize(ix->stream));
    }
    return TS_OK;
}

// ---------------------------------------------------------------------------------------------
// search
// ---------------------------------------------------------------------------------------------
static inline bool mfma_dim(int d) { return d == 384 || d == 512 || d == 768 || d == 1024; }
// indexes the batched MFMA path serves: bf16 at the four widths, fp32 at d = 768 (kernels_mfma_f32.h) and d = 1024
// (kernels_mfma16.h, F32) on the exact-fp32 matrix instructions
// d = 384 / 512 on the 16x16 kernel exist as the full pass only: they need the usual two-level search (dense threshold
// sample + full pass), not the guaranteed chain (TS_MFMA_STAT=0) or the list-form sample (TS_MFMA_SAMPLE=0)
static inline bool two_level_search(const ts_index* ix) {
    return ix->knobs.get(K_MFMA_STAT, 1) != 0 && ix->knobs.get(K_MFMA_SAMPLE, 1) != 0;
}
static inline bool mfma_index(const ts_index* ix) {
    if (ix->dtype == TS_BF16) return mfma_dim(ix->d);
    if (ix->knobs.get(K_MFMA_F32, 16) == 0) return false;
    return ix->d == 768 || ix->d == 1024 || ((ix->d == 384 || ix->d == 512) && two_level_search(ix));
}

static int ensure_search_scratch(ts_index* ix, int k) {
    // each buffer on its own: a failed allocation leaves the others as they are and is retried by the next call
    auto need = [](auto** slot, size_t bytes, bool zero) -> int {
        if (*slot) return TS_OK;
        void* p = nullptr;
        HIP_TRY(hipMalloc(&p, bytes));
        if (zero) {
            const hipError_t e = hipMemset(p, 0, bytes);
            if (e != hipSuccess) {
                hipFree(p);
                return fail(TS_ERR_HIP, "hipMemset of search scratch failed: %s", hipGetErrorString(e));
            }
        }
        *slot = (std::remove_pointer_t<decltype(slot)>)p;
        return TS_OK;
    };
    TS_TRY(need(&ix->qstore, (size_t)kQBlock * ix->ld * ix->elem(), false));
    TS_TRY(need(&ix->qf32, (size_t)kQBlock * ix->ld * 4, false));
    TS_TRY(need(&ix->count, (size_t)kQBlock * 4, true));
    TS_TRY(need(&ix->thr, (size_t)kQBlock * 4, false));
    TS_TRY(need(&ix->fb_list, (size_t)kQBlock * 4, false));
    TS_TRY(need(&ix->fb_count, 16, true));
    TS_TRY(need(&ix->stat, 16, true));
    if (mfma_index(ix)) TS_TRY(need(&ix->cand, (size_t)kQBlock * kCandCap * 8, false));
    // scan partials: [256 slots][grid][k] keys, twice (ping-pong for the select rounds)
    const size_t grid = (size_t)ix->cu_count * kScanGridPerCU;
    const size_t want = (size_t)kQBlock * grid * (size_t)k * 8;
    if (ix->partial_bytes < want) {
        if (ix->partial) HIP_TRY(hipFree(ix->partial));
        if (ix->partial2) HIP_TRY(hipFree(ix->partial2));
        ix->partial = ix->partial2 = nullptr;
        ix->partial_bytes = 0;
        HIP_TRY(hipMalloc((void**)&ix->partial, want));
        HIP_TRY(hipMalloc((void**)&ix->partial2, want / 8 + 4096 * 8));
        ix->partial_bytes = want;
    }
    return TS_OK;
}

template <int DT, int CH, int G, bool EMIT>
static void launch_scan_spec(int qb, int kr, int grid, hipStream_t st, const ScanArgs& a) {
    if (EMIT) {
        if (qb == 4) scan_kernel<DT, CH, G, 4, 1, true><<<grid, 256, 0, st>>>(a);
        else scan_kernel<DT, CH, G, 1, 1, true><<<grid, 256, 0, st>>>(a);
        return;
    }
    if (qb == 4) {
        if (kr == 1) scan_kernel<DT, CH, G, 4, 1, false><<<grid, 256, 0, st>>>(a);
        else scan_kernel<DT, CH, G, 4, 4, false><<<grid, 256, 0, st>>>(a);
    } else {
        if (kr == 1) scan_kernel<DT, CH, G, 1, 1, false><<<grid, 256, 0, st>>>(a);
        else scan_kernel<DT, CH, G, 1, 4, false><<<grid, 256, 0, st>>>(a);
    }
}

template <int DT, bool EMIT>
static void launch_scan_generic(int qb, int kr, int grid, hipStream_t st, const ScanArgs& a) {
    const size_t lds = 8192 + (size_t)a.ld * 4 * (qb == 4 ? 4 : 1);
    auto go = [&](auto kern) {
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        kern<<<grid, 256, lds, st>>>(a);
    };
    if (qb == 4) {
        if (EMIT || kr == 1) go(scan_generic_kernel<DT, 1, EMIT, 4>);
        else go(scan_generic_kernel<DT, 4, EMIT, 1>);   // k > 64: four lists of 4 keys per lane do not fit; one query per pass
    } else {
        if (EMIT || kr == 1) go(scan_generic_kernel<DT, 1, EMIT, 1>);
        else go(scan_generic_kernel<DT, 4, EMIT, 1>);
    }
}

// One scan pass configuration for (dtype, ld); returns the query-batch width used.
template <bool EMIT>
static int launch_scan(const ts_index* ix, ScanArgs a, int qb_pref, hipStream_t st, int grid) {
    const int kr = (a.k <= 64) ? 1 : 4;
    const bool force_generic = ix->knobs.get(K_SCAN_GENERIC, 0) != 0;
    if (!force_generic && ix->dtype == TS_F32 && ix->ld == 768) { launch_scan_spec<0, 3, 64, EMIT>(qb_pref, kr, grid, st, a); return qb_pref; }
    if (!force_generic && ix->dtype == TS_F32 && ix->ld == 1024) { launch_scan_spec<0, 4, 64, EMIT>(qb_pref, kr, grid, st, a); return qb_pref; }
    if (!force_generic && ix->dtype == TS_BF16 && ix->ld == 768) { launch_scan_spec<1, 3, 32, EMIT>(qb_pref, kr, grid, st, a); return qb_pref; }
    if (!force_generic && ix->dtype == TS_BF16 && ix->ld == 1024) { launch_scan_spec<1, 2, 64, EMIT>(qb_pref, kr, grid, st, a); return qb_pref; }
    // the other common embedding widths (MiniLM-class 384, 512): same kernel, narrower lane groups
    if (!force_generic && ix->dtype == TS_F32 && ix->ld == 384) { launch_scan_spec<0, 3, 32, EMIT>(qb_pref, kr, grid, st, a); return qb_pref; }
    if (!force_generic && ix->dtype == TS_BF16 && ix->ld == 384) { launch_scan_spec<1, 3, 16, EMIT>(qb_pref, kr, grid, st, a); return qb_pref; }
    if (!force_generic && ix->dtype == TS_F32 && ix->ld == 512) { launch_scan_spec<0, 2, 64, EMIT>(qb_pref, kr, grid, st, a); return qb_pref; }
    if (!force_generic && ix->dtype == TS_BF16 && ix->ld == 512) { launch_scan_spec<1, 2, 32, EMIT>(qb_pref, kr, grid, st, a); return qb_pref; }
    // any other width: queries staged in LDS, 4 per pass while they fit (ld <= 8192) and k <= 64
    const int qb = (qb_pref == 4 && a.ld <= 8192 && (EMIT || kr == 1)) ? 4 : 1;
    if (ix->dtype == TS_F32) launch_scan_generic<0, EMIT>(qb, kr, grid, st, a);
    else launch_scan_generic<1, EMIT>(qb, kr, grid, st, a);
    return qb;
}

// Reduce [slots][m] partial keys to the final k per query: select rounds of 4096-key segments.
static int run_select_rounds(ts_index* ix, int slots, int m, int k, float* out_scores, int64_t* out_idx, const int* qlist,
                             const int* qcount, hipStream_t st) {
    const u64* in = ix->partial;
    u64* scratch[2] = {ix->partial2, ix->partial};
    int which = 0;
    int64_t in_stride = m;
    for (;;) {
        if (m > 1024 && m <= kHistSelectMax) {
            // the usual case (k <= 12 over 1024 workgroups, or k up to 256 over the fewer workgroups scan_search uses on a
            // small corpus): one launch, histogram cut instead of rounds of bitonic sorts
            if (!ix->attr_done_hist) {
                HIP_TRY(hipFuncSetAttribute((const void*)select_hist_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, kHistSelectLds));
                HIP_TRY(hipFuncSetAttribute((const void*)select_hist_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, kHistSelectLds));
                ix->attr_done_hist = true;
            }
            SelectArgs a;
            memset(&a, 0, sizeof(a));
            a.in = in;
            a.in_stride = in_stride;
            a.m = m;
            a.kout = k;
            a.k_user = k;
            a.row_offset = ix->row_offset;
            a.id_map = ix->id_map;
            a.qlist = qlist;
            a.qcount = qcount;
            a.out_scores = out_scores;
            a.out_idx = out_idx;
            if (k <= 64) select_hist_kernel<1><<<slots, kLevelThreads, kHistSelectLds, st>>>(a);
            else select_hist_kernel<4><<<slots, kLevelThreads, kHistSelectLds, st>>>(a);
            HIP_TRY(hipGetLastError());
            return TS_OK;
        }
        SelectArgs a;
        memset(&a, 0, sizeof(a));
        a.in = in;
        a.in_stride = in_stride;
        a.m = m;
        a.kout = k;
        a.k_user = k;
        a.row_offset = ix->row_offset;
        a.id_map = ix->id_map;
        a.qlist = qlist;
        a.qcount = qcount;
        if (m <= 1024 || (k > 64 && m <= 4096)) {
            a.out_scores = out_scores;
            a.out_idx = out_idx;
            if (m <= 1024) select_kernel<1024><<<dim3(1, slots), 256, 0, st>>>(a);
            else select_kernel<4096><<<dim3(1, slots), 256, 0, st>>>(a);
            HIP_TRY(hipGetLastError());
            return TS_OK;
        }
        // intermediate round: many small sorts in parallel beat a few big ones (a 4096-key bitonic
        // sort by one workgroup costs ~80 us, a 1024-key one ~15 us)
        const int seg = (m > 65536) ? 4096 : 1024;
        const int nseg = (m + seg - 1) / seg;
        a.out = scratch[which];
        a.out_stride = (int64_t)nseg * k;
        if (seg == 4096) select_kernel<4096><<<dim3(nseg, slots), 256, 0, st>>>(a);
        else select_kernel<1024><<<dim3(nseg, slots), 256, 0, st>>>(a);
        HIP_TRY(hipGetLastError());
        in = scratch[which];
        in_stride = a.out_stride;
        m = nseg * k;
        which ^= 1;
    }
}

// `qbuf`: fp32 queries to read instead of the prepared copy; `qb16`: bf16 queries to read in place (the caller's matrix).
static int scan_search(ts_index* ix, int nq, int k, float* out_scores, int64_t* out_idx, const int* qlist, const int* qcount,
                       hipStream_t st, const float* qbuf = nullptr, const unsigned short* qb16 = nullptr) {
    int grid = ix->cu_count * kScanGridPerCU;
    // Large k over a small corpus (app_showcase_model.py:96: topk(200) over a few thousand theorems): every workgroup
    // hands k keys to the select, and 1,024 x 200 of them cost three rounds of sorts (150 us) for a scan of 10 us.  Few
    // enough workgroups that ONE histogram select takes all their keys.
    if (k > 64 && ix->n <= 16384) grid = std::min(grid, std::max(8, kHistSelectMax / k));
    ScanArgs a;
    memset(&a, 0, sizeof(a));
    a.corpus = ix->rows;
    a.ld = ix->ld;
    a.n = ix->n;
    a.qbuf = qb16 ? nullptr : (qbuf ? qbuf : ix->qf32);
    a.qb16 = qb16;
    a.qlist = qlist;
    a.qcount = qcount;
    a.nq = nq;
    a.k = k;
    a.partial = ix->partial;
    a.row_mask = ix->active_mask;
    a.bias = ix->active_bias;
    a.bias_w = ix->active_bias_w;
    // The exact re-run of the MFMA path (device-side query count, almost always zero) is ONE launch: the workgroup that
    // finishes last reduces the partial lists itself (scan_finish), so the common case pays one empty launch, not one per
    // select round as well.
    // (Tried for the app's own shape too - one to four queries, small k - in place of the separate histogram select:
    // 0.471 -> 0.505 ms per search on 1M x 768 fp32, 53 instead of 33 us on 1,000 rows: every workgroup's release fence and
    // the last workgroup's serial sweep of 10,240 keys cost more than the second launch.  The re-run path only.)
    const bool one_launch = qcount != nullptr;
    if (one_launch) {
        a.done_ctr = (unsigned*)ix->fb_count + 2;     // zeroed with the block, left zeroed by the kernel
        a.out_scores = out_scores;
        a.out_idx = out_idx;
        a.row_offset = ix->row_offset;
        a.id_map = ix->id_map;
        // an almost always empty launch: one workgroup per CU dispatches (and drains) faster than four; when it does
        // run, a pass at a lower share of the HBM rate is the price of the rare query the estimate failed for
        grid = std::min(grid, ix->cu_count);
        if (ix->rebalance_pending && ix->rebalance_grid <= 256) {
            a.part = ix->part;
            a.wg_ticks = ix->wg_ticks;
            a.part_g = ix->rebalance_grid;
            const int b = ix->knobs.get(K_MFMA_BALANCE, 1);      // TS_MFMA_BALANCE = n > 1: gain n / 10 (default 0.7)
            a.part_gain = (b >= 2 && b <= 10) ? 0.1f * (float)b : 0.7f;
        }
        ix->rebalance_pending = false;
    }
    // k > 64 keeps 4 keys per lane and query: on bf16 x 768 four queries at once need all 256 VGPRs, one wave per SIMD
    // (measured 0.18 of the HBM rate against 0.8 for one query per pass); the other shapes keep two waves
    const bool wide_k_one_wave = k > 64 && ix->dtype == TS_BF16 && (ix->ld == 768 || ix->ld == 384);
    const int qb = ((nq >= 2 || qcount) && !wide_k_one_wave) ? 4 : 1;
    hipEvent_t stop = qcount ? nullptr : prof_begin(ix, st, ix->n);  // the MFMA path's fall-back pass is not bracketed
    launch_scan<false>(ix, a, qb, st, grid);
    prof_end(stop, st);
    HIP_TRY(hipGetLastError());
    if (one_launch) return TS_OK;
    return run_select_rounds(ix, nq, grid * k, k, out_scores, out_idx, qlist, qcount, st);
}

struct Level { int64_t stride, ntiles; int run; };

// Threshold levels of the MFMA path, sparsest first.  Level i visits runs of `run` consecutive tiles
// every run * stride tiles (stride 1 = every tile = the full pass) and passes on to level i+1 the
// kk-th best score it saw as that level's pass threshold: a lower bound of the final kk-th best, so
// nothing that belongs to the answer is ever dropped.  Expected candidates per query in level i+1 =
// kk * rows(i+1) / rows(i): the full pass is planned for `target` candidates (few trips through
// the append path), the sparser levels for up to kCandCap / 4 (they are short anyway); the first
// level is small enough to run unthresholded.
// Inverse of the standard normal CDF (Acklam's rational approximation, |error| < 1.2e-9): the z with P(X > z) = p.
static double normal_tail_z(double p) {
    if (p <= 0.0) return 8.0;
    if (p >= 0.5) return 0.0;
    const double q = std::sqrt(-2.0 * std::log(p));
    static const double c[] = {-7.784894002430293e-03, -3.223964580411365e-01, -2.400758277161838e+00,
                               -2.549732539343734e+00, 4.374664141464968e+00, 2.938163982698783e+00};
    static const double d[] = {7.784695709041462e-03, 3.224671290700398e-01, 2.445134137142996e+00, 3.754408661907416e+00};
    if (p < 0.02425)
        return -(((((c[0] * q + c[1]) * q + c[2]) * q + c[3]) * q + c[4]) * q + c[5]) /
               ((((d[0] * q + d[1]) * q + d[2]) * q + d[3]) * q + 1.0);
    // central region
    static const double a[] = {-3.969683028665376e+01, 2.209460984245205e+02, -2.759285104469687e+02,
                               1.383577518672690e+02, -3.066479806614716e+01, 2.506628277459239e+00};
    static const double b[] = {-5.447609879822406e+01, 1.615858368580409e+02, -1.556989798598866e+02,
                               6.680131188771972e+01, -1.328068155288572e+01};
    const double x = (1.0 - p) - 0.5, r = x * x;
    return (((((a[0] * r + a[1]) * r + a[2]) * r + a[3]) * r + a[4]) * r + a[5]) * x /
           (((((b[0] * r + b[1]) * r + b[2]) * r + b[3]) * r + b[4]) * r + 1.0);
}

static int mfma_target_cands(const Knobs& kn, int64_t n, int kk) {
    // Cost model fitted on 10M / 1.25M x 768, batch 256: a sample row costs ~0.4 ns, a candidate of the next
    // level ~0.27 us per query (the append path is ~1 us of wave time).  Minimising kk * N * c_row / F + c_cand * F
    // gives F ~ 512 * sqrt(N / 1e7) candidates per query for the full pass.
    int target = (int)(512.0 * std::sqrt(std::max<double>((double)n, 1.0) / 1e7));
    target = std::max(target, 8 * kk);  // large k: keep the level ratio >= 8, or the sparse levels cost as much as the pass
    return std::min(2048, std::max(64, kn.get(K_MFMA_TARGET_CANDS, target)));
}

static std::vector<Level> plan_levels(const Knobs& kn, int64_t n, int kk, bool statistical) {
    const int64_t T = (n + kTileRows - 1) / kTileRows;
    const int target = mfma_target_cands(kn, n, kk);
    auto pow2_ratio = [&](int cands) { int64_t r = 2; while (r * 2 * kk <= cands) r *= 2; return r; };
    const int64_t r_last = pow2_ratio(target);
    // The sparsest level runs unthresholded: every score becomes a candidate, so it may hold at most
    // kLevelSortMax rows (what one select sorts) and one tile per workgroup (16 entries per private list).
    // sample size: 8192 rows for large corpora; below 4M rows half of that estimates the threshold as well (the
    // guaranteed bound k * N / sample stays small) and its pass + select are 13 us shorter - 2 % of a 1.25M-row shard
    const int first_default = (statistical && n < 4000000) ? kLevelSortMax / 2 : kLevelSortMax;
    const int64_t first_rows = std::min<int64_t>(kLevelSortMax, (int64_t)kn.get(K_MFMA_FIRST_ROWS, first_default));
    const int64_t r_cap = std::max<int64_t>(2, pow2_ratio(kn.get(K_MFMA_TARGET_SPARSE, 1280)));
    std::vector<Level> lv;
    int64_t stride = 1;
    for (;;) {
        const int64_t nt = (T + stride - 1) / stride;
        // sampling in runs of consecutive tiles (shared DRAM pages / TLB entries) measured no different from
        // single tiles; kept as a knob
        const int run = (stride > 1 && nt >= 8 * 256) ? kn.get(K_MFMA_RUN, 1) : 1;
        lv.push_back({stride, nt, run});
        if (nt * kTileRows <= first_rows) break;  // every score of this level fits: it can run unthresholded
        if (statistical) {
            // one unthresholded sample of up to first_rows rows; its select extrapolates the threshold of the full pass
            int64_t need = 2;
            while (((T + need - 1) / need) * kTileRows > first_rows) need *= 2;
            stride = need;
        } else if (lv.size() == 1) {
            stride *= r_last;
        } else {
            // smallest ratio that reaches the unthresholded size in one step, if the cap allows it
            int64_t need = 2;
            while (need < r_cap && ((T + stride * need - 1) / (stride * need)) * kTileRows > first_rows) need *= 2;
            stride *= need;
        }
    }
    std::reverse(lv.begin(), lv.end());
    return lv;
}

template <int D, int GROUPS>
static int launch_mfma(bool full_pass, int variant, int grid, hipStream_t st, const MfmaArgs& a) {
    constexpr int lds = MfmaDims<D>::kLds;
    // the dynamic-LDS limit is set per function AND per device: one bit per device, per instantiation
    static std::atomic<unsigned long long> attr_done{0};
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(attr_done.load(std::memory_order_acquire) & bit)) {
        HIP_TRY(hipFuncSetAttribute((const void*)mfma_topk_kernel<D, GROUPS, 0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        HIP_TRY(hipFuncSetAttribute((const void*)mfma_topk_kernel<D, GROUPS, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
#ifdef TS_DIAG
        HIP_TRY(hipFuncSetAttribute((const void*)mfma_topk_kernel<D, GROUPS, 1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        HIP_TRY(hipFuncSetAttribute((const void*)mfma_topk_kernel<D, GROUPS, 2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        HIP_TRY(hipFuncSetAttribute((const void*)mfma_topk_kernel<D, GROUPS, 3, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        HIP_TRY(hipFuncSetAttribute((const void*)mfma_topk_kernel<D, GROUPS, 4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        HIP_TRY(hipFuncSetAttribute((const void*)mfma_topk_kernel<D, GROUPS, 5, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        HIP_TRY(hipFuncSetAttribute((const void*)mfma_topk_kernel<D, GROUPS, 6, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        HIP_TRY(hipFuncSetAttribute((const void*)mfma_topk_kernel<D, GROUPS, 7, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
#endif
        attr_done.fetch_or(bit, std::memory_order_release);
    }
    if (!full_pass) mfma_topk_kernel<D, GROUPS, 0, true><<<grid, kMfmaThreads, lds, st>>>(a);
#ifdef TS_DIAG
    else if (variant == 1) mfma_topk_kernel<D, GROUPS, 1, false><<<grid, kMfmaThreads, lds, st>>>(a);
    else if (variant == 2) mfma_topk_kernel<D, GROUPS, 2, false><<<grid, kMfmaThreads, lds, st>>>(a);
    else if (variant == 3) mfma_topk_kernel<D, GROUPS, 3, false><<<grid, kMfmaThreads, lds, st>>>(a);
    else if (variant == 4) mfma_topk_kernel<D, GROUPS, 4, false><<<grid, kMfmaThreads, lds, st>>>(a);
    else if (variant == 5) mfma_topk_kernel<D, GROUPS, 5, false><<<grid, kMfmaThreads, lds, st>>>(a);
    else if (variant == 6) mfma_topk_kernel<D, GROUPS, 6, false><<<grid, kMfmaThreads, lds, st>>>(a);
    else if (variant == 7) mfma_topk_kernel<D, GROUPS, 7, false><<<grid, kMfmaThreads, lds, st>>>(a);
#endif
    else mfma_topk_kernel<D, GROUPS, 0, false><<<grid, kMfmaThreads, lds, st>>>(a);
    (void)variant;
    HIP_TRY(hipGetLastError());
    return TS_OK;
}

template <int D, int NB>
static int launch_mfma16(bool full_pass, int variant, int grid, hipStream_t st, const MfmaArgs& a) {
    constexpr int lds = Mfma16Dims<D>::kLds + kMfma16StageBytes;
    static_assert(lds <= 160 * 1024, "DMA ring + staged candidates must fit the CU's LDS");
    // d = 384 / 512: the full pass only (the threshold sample has its own kernel; the thresholded sparse levels of the
    // guaranteed chain run the 32x32 kernel for these widths: use_shape16)
    constexpr bool kSparseToo = (D == 768 || D == 1024);
#ifdef TS_DIAG
    constexpr bool kDiag = (D == 768 && NB == 4);     // the timing-only variants exist for the headline shape only
#else
    constexpr bool kDiag = false;                     // ... and in the diagnostic build only (make diag)
#endif
    static std::atomic<unsigned long long> attr_done{0};
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(attr_done.load(std::memory_order_acquire) & bit)) {
        HIP_TRY(hipFuncSetAttribute((const void*)mfma16_topk_kernel<D, NB, 0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        if constexpr (kSparseToo)
            HIP_TRY(hipFuncSetAttribute((const void*)mfma16_topk_kernel<D, NB, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        if constexpr (kDiag) {
            HIP_TRY(hipFuncSetAttribute((const void*)mfma16_topk_kernel<D, NB, 1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            HIP_TRY(hipFuncSetAttribute((const void*)mfma16_topk_kernel<D, NB, 2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            HIP_TRY(hipFuncSetAttribute((const void*)mfma16_topk_kernel<D, NB, 3, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            HIP_TRY(hipFuncSetAttribute((const void*)mfma16_topk_kernel<D, NB, 4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            HIP_TRY(hipFuncSetAttribute((const void*)mfma16_topk_kernel<D, NB, 7, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            HIP_TRY(hipFuncSetAttribute((const void*)mfma16_topk_kernel<D, NB, 5, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            HIP_TRY(hipFuncSetAttribute((const void*)mfma16_topk_kernel<D, NB, 6, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        }
        attr_done.fetch_or(bit, std::memory_order_release);
    }
    if (!full_pass) {
        if constexpr (kSparseToo) mfma16_topk_kernel<D, NB, 0, true><<<grid, kMfmaThreads, lds, st>>>(a);
        else return fail(TS_ERR_INTERNAL, "no sparse level of the 16x16 kernel at d = %d", D);
    } else if (kDiag && variant != 0) {
        if constexpr (kDiag) {
            if (variant == 1) mfma16_topk_kernel<D, NB, 1, false><<<grid, kMfmaThreads, lds, st>>>(a);
            else if (variant == 2) mfma16_topk_kernel<D, NB, 2, false><<<grid, kMfmaThreads, lds, st>>>(a);
            else if (variant == 3) mfma16_topk_kernel<D, NB, 3, false><<<grid, kMfmaThreads, lds, st>>>(a);
            else if (variant == 4) mfma16_topk_kernel<D, NB, 4, false><<<grid, kMfmaThreads, lds, st>>>(a);
            else if (variant == 7) mfma16_topk_kernel<D, NB, 7, false><<<grid, kMfmaThreads, lds, st>>>(a);
            else if (variant == 5) mfma16_topk_kernel<D, NB, 5, false><<<grid, kMfmaThreads, lds, st>>>(a);
            else if (variant == 6) mfma16_topk_kernel<D, NB, 6, false><<<grid, kMfmaThreads, lds, st>>>(a);
            else mfma16_topk_kernel<D, NB, 0, false><<<grid, kMfmaThreads, lds, st>>>(a);
        }
    } else {
        mfma16_topk_kernel<D, NB, 0, false><<<grid, kMfmaThreads, lds, st>>>(a);
    }
    HIP_TRY(hipGetLastError());
    return TS_OK;
}

static int launch_mfma_f32(bool full_pass, int variant, int grid, hipStream_t st, const MfmaArgs& a) {
    constexpr int lds = MfmaF32Dims::kLds;
    static std::atomic<unsigned long long> attr_done{0};
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(attr_done.load(std::memory_order_acquire) & bit)) {
        HIP_TRY(hipFuncSetAttribute((const void*)mfma_f32_topk_kernel<0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        HIP_TRY(hipFuncSetAttribute((const void*)mfma_f32_topk_kernel<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
#ifdef TS_DIAG
        HIP_TRY(hipFuncSetAttribute((const void*)mfma_f32_topk_kernel<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
#endif
        attr_done.fetch_or(bit, std::memory_order_release);
    }
    if (!full_pass) mfma_f32_topk_kernel<0, true><<<grid, kMfmaThreads, lds, st>>>(a);
#ifdef TS_DIAG
    else if (variant == 1) mfma_f32_topk_kernel<1, false><<<grid, kMfmaThreads, lds, st>>>(a);
#endif
    else mfma_f32_topk_kernel<0, false><<<grid, kMfmaThreads, lds, st>>>(a);
    (void)variant;
    HIP_TRY(hipGetLastError());
    return TS_OK;
}

// Which MFMA shape serves this index: d = 768 runs the 16x16x32 kernel (kernels_mfma16.h) unless TS_MFMA_SHAPE=32 asks for
// the 32x32x16 one (kernels_mfma.h), which also serves the other widths.
static bool use_shape16(const ts_index* ix) {
    // d = 384 / 512 (round 3): the 16x16 kernel has the full pass only for these widths, so it serves them when the search
    // is the usual two-level one (dense threshold sample + full pass); the guaranteed chain (TS_MFMA_STAT=0) and the
    // list-form sample (TS_MFMA_SAMPLE=0) run the 32x32 kernel (bf16) - fp32 at these widths has no other matrix kernel
    const bool narrow = ix->d == 384 || ix->d == 512;
    const bool two_level = two_level_search(ix);
    // fp32: the 16x16x4 form of the same kernel (10M x 768, 256 queries: 14.1 ms a pass against 14.8 ms of the 32x32x2
    // kernel, which TS_MFMA_F32=32 still selects for d = 768)
    if (ix->dtype == TS_F32) return ix->d != 768 || ix->knobs.get(K_MFMA_F32, 16) != 32;
    if (narrow && !two_level) return false;
    return ix->dtype == TS_BF16 && mfma_dim(ix->d) && ix->knobs.get(K_MFMA_SHAPE, 16) != 32;
}

// fp32 index: mfma16_topk_kernel<D, NB, ., ., F32 = true>.  d = 1024: one block of 16 queries per wave, 64 per launch;
// d = 768: two blocks, 128 per launch (one block when the batch has at most 64 queries)
template <int D, int NB>
static int launch_mfma16_f32(bool full_pass, int grid, hipStream_t st, const MfmaArgs& a) {
    constexpr int lds = Mfma16Dims<2 * D>::kLds + kMfma16StageBytes;
    static_assert(lds <= 160 * 1024, "DMA ring + staged candidates must fit the CU's LDS");
    constexpr bool kSparseToo = (D == 768 || D == 1024);        // d = 384 / 512: the full pass only (as launch_mfma16)
    static std::atomic<unsigned long long> attr_done{0};
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(attr_done.load(std::memory_order_acquire) & bit)) {
        HIP_TRY(hipFuncSetAttribute((const void*)mfma16_topk_kernel<D, NB, 0, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        if constexpr (kSparseToo)
            HIP_TRY(hipFuncSetAttribute((const void*)mfma16_topk_kernel<D, NB, 0, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_done.fetch_or(bit, std::memory_order_release);
    }
    if (!full_pass) {
        if constexpr (kSparseToo) mfma16_topk_kernel<D, NB, 0, true, true><<<grid, kMfmaThreads, lds, st>>>(a);
        else return fail(TS_ERR_INTERNAL, "no sparse level of the 16x16 kernel at d = %d", D);
    } else {
        mfma16_topk_kernel<D, NB, 0, false, true><<<grid, kMfmaThreads, lds, st>>>(a);
    }
    HIP_TRY(hipGetLastError());
    return TS_OK;
}

// Queries one launch of the MFMA kernel serves for this index / batch: d = 768 holds two query groups per wave
// (256 queries; one group = half the matrix work when the batch is <= 128), d = 1024 one (128 queries).
static int mfma_block_queries(const ts_index* ix, int nq) {
    if (ix->dtype == TS_F32) {
        if (ix->d == 1024) return 64;                          // one block of 16 queries x 4 waves
        if (use_shape16(ix)) return nq <= 64 ? 64 : 128;       // one or two blocks per wave (d = 384, 512, 768)
        return kMfmaF32Queries;                                // 32x32x2 kernel: 32 queries x 4 waves
    }
    if (use_shape16(ix)) {
        // 16 queries x NB blocks x 4 waves; d = 1024 has registers for 3 blocks per wave, and a batch of more than 192
        // queries is cut into equal launches (two of 128 for 256: both then stream at the HBM rate)
        const int max_nb = ix->d == 1024 ? 3 : 4;
        const int blocks = (std::min(nq, 256) + 63) / 64;
        if (blocks <= max_nb) return 64 * std::max(1, blocks);
        return 64 * ((blocks + 1) / 2);
    }
    if (ix->d == 1024) return 128;
    return nq <= 128 && ix->knobs.get(K_MFMA_GROUPS, 0) != 2 ? 128 : 256;
}

// `qmat`: the queries as the kernels multiply them (storage dtype, row stride d = ld, a whole launch's worth of rows):
// the prepared copy, or the caller's own device matrix when it already has that form (`in_place`).
static int mfma_search(ts_index* ix, int nq, int k, float* out_scores, int64_t* out_idx, hipStream_t st, ts_search_stats* stats,
                       const void* qmat, bool in_place) {
    // threshold rank: the k-th best of a sample is already a valid lower bound of the final k-th best; private
    // lists + spill absorb the run-to-run spread of the candidate count, so no safety margin in the rank
    const int kk = std::max(k, ix->knobs.get(K_MFMA_MIN_RANK, 1));
    const int variant = ix->knobs.get(K_MFMA_VARIANT, 0);
    const bool shape16 = use_shape16(ix);
    const int groups = shape16 ? 0 : mfma_block_queries(ix, nq) / 128;
    const int nb16 = shape16 ? mfma_block_queries(ix, nq) / 64 : 0;
    if (!ix->attr_done) {
        HIP_TRY(hipFuncSetAttribute((const void*)level_select_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, kLevelLds));
        HIP_TRY(hipFuncSetAttribute((const void*)level_select_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, kLevelLds));
        HIP_TRY(hipFuncSetAttribute((const void*)sample_select_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, kLevelLds));
        HIP_TRY(hipFuncSetAttribute((const void*)sample_select_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, kLevelLds));
        ix->attr_done = true;
    }
    // The sparsest level (every score a candidate, at most kLevelSortMax rows) runs as a dense score matrix + one select per
    // query (kernels_sample.h) instead of the full-pass kernel over the sample + a gather from lane-private lists; the
    // latter stays selectable (TS_MFMA_SAMPLE=0) as the A/B partner and serves the thresholded sparse levels of the
    // guaranteed chain (TS_MFMA_STAT=0).
    const bool dense_sample = ix->knobs.get(K_MFMA_SAMPLE, 1) != 0;
    if (dense_sample && !ix->sample) HIP_TRY(hipMalloc((void**)&ix->sample, (size_t)kMfmaQ * kLevelSortMax * 4));
    const int grid = std::max(1, std::min(ix->knobs.get(K_MFMA_GRID, ix->cu_count), 2048));
    // lane-private candidate lists: 2 writers x 32 entries per workgroup and query (32x32 shape) or 4 x 16 (16x16 shape)
    const int nwriters = (shape16 ? 4 : 2) * grid;
    const int priv_cap = shape16 ? kMfma16PrivCap : kMfmaPrivCap;
    if (ix->priv_writers < 4 * grid) {
        if (ix->priv) HIP_TRY(hipFree(ix->priv));
        if (ix->pcount) HIP_TRY(hipFree(ix->pcount));
        ix->priv = nullptr; ix->pcount = nullptr; ix->priv_writers = 0;
        static_assert(4 * kMfma16PrivCap == 2 * kMfmaPrivCap, "both shapes use the same list bytes per workgroup");
        HIP_TRY(hipMalloc((void**)&ix->priv, (size_t)kMfmaQ * 4 * grid * kMfma16PrivCap * 8));
        HIP_TRY(hipMalloc((void**)&ix->pcount, (size_t)kMfmaQ * 4 * grid * 4));
        ix->priv_writers = 4 * grid;
    }
    // Threshold of the full pass: by default extrapolated from ONE unthresholded sample (Gaussian tail of the
    // sample's scores, verified afterwards by the candidate count); TS_MFMA_STAT=0 selects the chain of
    // guaranteed lower bounds (more sample rows to scan, no re-runs ever).
    const bool statistical = ix->knobs.get(K_MFMA_STAT, 1) != 0;
    const std::vector<Level> lv = plan_levels(ix->knobs, ix->n, kk, statistical);
    // Expected candidates per query of the full pass under the estimate.  Every candidate costs the pass ~0.3 us of one
    // CU's time (the appending wave holds the other three at the next barrier), whatever N: 160 per query were 10 % of
    // a 1.25M-row shard's pass and 1 % of the 10M pass; an under-filled query (fewer than k back) costs an exact scan
    // pass.  6 k (at least 64) keeps the under-fill probability negligible for Gaussian-like scores (Poisson mean 64
    // against k = 10, estimate error e^+-0.15) - measured on 10M / 1.25M x 768: 160 / 96 / 64 / 40 expected candidates
    // -> 0 re-runs, 24 -> 5-7 re-runs per 256 queries; full pass 0.459 / 0.447 / 0.438 / 0.424 ms on the shard.
    const int stat_cands = std::min(2048, std::max(2 * kk, ix->knobs.get(K_MFMA_STAT_CANDS, std::max(64, 6 * kk))));
    // rows the candidates are drawn from: all of them, or the rows a filter allows (the sample sees only those too)
    const int64_t pop = ix->active_mask ? ix->active_allowed : ix->n;
    const float z_tail = (statistical && lv.size() == 2)
                             ? (float)normal_tail_z(std::min(0.25, (double)stat_cands / (double)std::max<int64_t>(pop, 1)))
                             : 0.0f;
    // Second estimate (exponential tail fit of the sample's order statistics, kernels_select.h), for score distributions
    // with heavier tails than a Gaussian.  Only where it is needed: when the guaranteed bound alone (the kk-th best of
    // the sample admits ~kk * N / sample rows) would swamp the candidate buffer - large corpora; it aims at
    // max(2048, 8 kk) expected candidates, a quarter of the buffer.
    const double sample_rows = (double)std::max<int64_t>(1, lv[0].ntiles * kTileRows);
    const bool bound_swamps = (double)kk * (double)ix->n / sample_rows > 0.5 * kCandCap;
    const float tail_p = (z_tail > 0.0f && bound_swamps && ix->knobs.get(K_MFMA_TAIL_FIT, 1))
                             ? (float)std::min(0.25, (double)std::max(2048, 8 * kk) / (double)std::max<int64_t>(pop, 1))
                             : 0.0f;
    // Feedback partition of the full pass (16x16 kernel): the final select moves the workgroups' tile boundaries towards
    // equal finishing times for the next search (kernels_select.h, rebalance_tiles).  The table starts as equal shares and
    // is re-made whenever the grid or the number of tiles changes.
    const int64_t full_tiles = lv.back().ntiles;
    const bool balance = shape16 && ix->knobs.get(K_MFMA_BALANCE, 1) != 0 && grid >= 8 && grid <= 256 && lv.back().stride == 1 &&
                         lv.back().run == 1 && full_tiles >= 32 * (int64_t)grid && (variant == 0 || variant == 3);
    if (balance && (ix->part_g != grid || ix->part_ntiles != full_tiles)) {
        if (ix->part_g != grid) {
            if (ix->part) HIP_TRY(hipFree(ix->part));
            if (ix->wg_ticks) HIP_TRY(hipFree(ix->wg_ticks));
            ix->part = nullptr; ix->wg_ticks = nullptr; ix->part_g = 0; ix->part_ntiles = -1;
            HIP_TRY(hipMalloc((void**)&ix->part, (size_t)(grid + 1) * 8));
            HIP_TRY(hipMalloc((void**)&ix->wg_ticks, (size_t)grid * 4));
            ix->part_g = grid;
        }
        std::vector<int64_t> equal((size_t)grid + 1);
        for (int w = 0; w <= grid; ++w) equal[w] = full_tiles * (int64_t)w / grid;
        HIP_TRY(hipMemcpyAsync(ix->part, equal.data(), equal.size() * 8, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemsetAsync(ix->wg_ticks, 0, (size_t)grid * 4, st));
        HIP_TRY(hipStreamSynchronize(st));          // `equal` is a local; this happens once per (grid, size)
        ix->part_ntiles = full_tiles;
    }
    for (size_t i = 0; i < lv.size(); ++i) {
        const bool full_pass = (i + 1 == lv.size());
        if (i == 0 && !full_pass && dense_sample && lv[0].ntiles * kTileRows <= kLevelSortMax) {
            SampleArgs sa;
            memset(&sa, 0, sizeof(sa));
            sa.corpus = ix->rows;
            sa.n = ix->n;
            sa.ld = (int)ix->ld;
            sa.ntiles = lv[0].ntiles;
            sa.tile_stride = lv[0].stride;
            sa.run = lv[0].run;
            sa.q = qmat;
            sa.nq = nq;
            sa.row_mask = ix->active_mask;
            sa.scores = ix->sample;
            sa.row_stride = (int)((lv[0].ntiles * kTileRows + 63) / 64 * 64);
            sa.fb_count = ix->fb_count;
            sa.stat = ix->stat;
            // 32 rows per workgroup and one 64-query chunk: 512 workgroups of 50 KB LDS at 4,096 rows x 256 queries, two to
            // a CU (64-row workgroups serving two chunks each measured the same: 15.2 / 25.0 us against 14.9 / 24.3 us at
            // 4,096 / 8,192 rows - the launch is latency, not work)
            const bool f32 = ix->dtype == TS_F32;
            const int nchunks = (nq + 63) / 64;
            const int wg_rows = 32;
            sa.chunks_per_wg = 1;
            const dim3 sgrid((unsigned)(sa.row_stride / wg_rows), (unsigned)((nchunks + sa.chunks_per_wg - 1) / sa.chunks_per_wg));
            const int slds = sample_lds_bytes(wg_rows, (int)(ix->ld * ix->elem()));
            if (slds > 160 * 1024) return fail(TS_ERR_INTERNAL, "threshold sample: rows of %lld bytes do not fit the LDS", (long long)(ix->ld * ix->elem()));
            {
                static std::atomic<unsigned long long> sample_attr{0};
                int dev = 0;
                HIP_TRY(hipGetDevice(&dev));
                const unsigned long long bit = 1ull << (dev & 63);
                if (!(sample_attr.load(std::memory_order_acquire) & bit)) {
                    HIP_TRY(hipFuncSetAttribute((const void*)sample_scores_kernel<true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                    HIP_TRY(hipFuncSetAttribute((const void*)sample_scores_kernel<false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                    sample_attr.fetch_or(bit, std::memory_order_release);
                }
            }
            if (f32) sample_scores_kernel<true, 2><<<sgrid, 256, slds, st>>>(sa);
            else sample_scores_kernel<false, 2><<<sgrid, 256, slds, st>>>(sa);
            HIP_TRY(hipGetLastError());
            LevelArgs l;
            memset(&l, 0, sizeof(l));
            l.count = ix->count;
            l.kk = kk;
            l.thr = ix->thr;
            l.z_tail = z_tail;
            l.tail_p = tail_p;
            l.tail_z = (float)normal_tail_z(std::min(0.25, 32.0 / sample_rows));
            l.nq = nq;
            if (kk <= 64) sample_select_kernel<1><<<nq, kLevelThreads, kLevelLds, st>>>(l, ix->sample, sa.row_stride, (int)(lv[0].ntiles * kTileRows));
            else sample_select_kernel<4><<<nq, kLevelThreads, kLevelLds, st>>>(l, ix->sample, sa.row_stride, (int)(lv[0].ntiles * kTileRows));
            HIP_TRY(hipGetLastError());
            continue;
        }
        MfmaArgs a;
        a.corpus = (const unsigned short*)ix->rows;
        a.n = ix->n;
        a.ntiles = lv[i].ntiles;
        a.tile_stride = lv[i].stride;
        a.run = lv[i].run;
        a.q = (const unsigned short*)qmat;
        a.thr = ix->thr;
        a.nq = ix->knobs.get(K_MFMA_NO_IDLE, 0) ? 256 : nq;
        a.row_mask = ix->active_mask;
        a.ahead = ix->knobs.get(K_MFMA_AHEAD, 0);
        a.priv = ix->priv;
        a.pcount = ix->pcount;
        a.cand = ix->cand;
        a.count = ix->count;
        a.cap = kCandCap;
        a.first_level = (i == 0) ? 1 : 0;      // thresholds and per-search counters are initialised inside the first launch of a search
        a.nq_real = nq;
        a.fb_count = ix->fb_count;
        a.stat = ix->stat;
        a.part = (balance && full_pass) ? ix->part : nullptr;
        a.wg_ticks = (balance && full_pass) ? ix->wg_ticks : nullptr;
        a.dbg = nullptr;
#ifdef TS_DIAG
        if (variant >= 3) {
            if (!ix->dbg) HIP_TRY(hipMalloc((void**)&ix->dbg, 2048 * 4 * 4 * 8));
            a.dbg = ix->dbg;
        }
#endif
        hipEvent_t stop = full_pass ? prof_begin(ix, st, ix->n) : nullptr;  // only the full pass is bracketed
        int rc;
        if (ix->dtype == TS_F32 && ix->d == 1024) rc = launch_mfma16_f32<1024, 1>(full_pass, grid, st, a);
        else if (ix->dtype == TS_F32 && ix->d == 512) rc = (nb16 == 1) ? launch_mfma16_f32<512, 1>(full_pass, grid, st, a) : launch_mfma16_f32<512, 2>(full_pass, grid, st, a);
        else if (ix->dtype == TS_F32 && ix->d == 384) rc = (nb16 == 1) ? launch_mfma16_f32<384, 1>(full_pass, grid, st, a) : launch_mfma16_f32<384, 2>(full_pass, grid, st, a);
        else if (ix->dtype == TS_F32 && shape16 && nb16 == 1) rc = launch_mfma16_f32<768, 1>(full_pass, grid, st, a);
        else if (ix->dtype == TS_F32 && shape16) rc = launch_mfma16_f32<768, 2>(full_pass, grid, st, a);
        else if (ix->dtype == TS_F32) rc = launch_mfma_f32(full_pass, variant, grid, st, a);
        else if (shape16 && ix->d == 512) rc = (nb16 == 4) ? launch_mfma16<512, 4>(full_pass, variant, grid, st, a) : (nb16 == 3) ? launch_mfma16<512, 3>(full_pass, variant, grid, st, a) : (nb16 == 2) ? launch_mfma16<512, 2>(full_pass, variant, grid, st, a) : launch_mfma16<512, 1>(full_pass, variant, grid, st, a);
        else if (shape16 && ix->d == 384) rc = (nb16 == 4) ? launch_mfma16<384, 4>(full_pass, variant, grid, st, a) : (nb16 == 3) ? launch_mfma16<384, 3>(full_pass, variant, grid, st, a) : (nb16 == 2) ? launch_mfma16<384, 2>(full_pass, variant, grid, st, a) : launch_mfma16<384, 1>(full_pass, variant, grid, st, a);
        else if (shape16 && ix->d == 1024 && nb16 == 3) rc = launch_mfma16<1024, 3>(full_pass, variant, grid, st, a);
        else if (shape16 && ix->d == 1024 && nb16 == 2) rc = launch_mfma16<1024, 2>(full_pass, variant, grid, st, a);
        else if (shape16 && ix->d == 1024) rc = launch_mfma16<1024, 1>(full_pass, variant, grid, st, a);
        else if (shape16 && nb16 == 4) rc = launch_mfma16<768, 4>(full_pass, variant, grid, st, a);
        else if (shape16 && nb16 == 3) rc = launch_mfma16<768, 3>(full_pass, variant, grid, st, a);
        else if (shape16 && nb16 == 2) rc = launch_mfma16<768, 2>(full_pass, variant, grid, st, a);
        else if (shape16) rc = launch_mfma16<768, 1>(full_pass, variant, grid, st, a);
        else if (ix->d == 1024) rc = launch_mfma<1024, 1>(full_pass, variant, grid, st, a);
        else if (ix->d == 512) rc = (groups == 1) ? launch_mfma<512, 1>(full_pass, variant, grid, st, a) : launch_mfma<512, 2>(full_pass, variant, grid, st, a);
        else if (ix->d == 384) rc = (groups == 1) ? launch_mfma<384, 1>(full_pass, variant, grid, st, a) : launch_mfma<384, 2>(full_pass, variant, grid, st, a);
        else if (groups == 1) rc = launch_mfma<768, 1>(full_pass, variant, grid, st, a);
        else rc = launch_mfma<768, 2>(full_pass, variant, grid, st, a);
        prof_end(stop, st);
        TS_TRY(rc);
#ifdef TS_DIAG
        if (a.dbg && full_pass && shape16 && variant == 3) {
            // clock probe (MI355X_MICROARCH.md "DVFS give-back" item 6): shader cycles / 100 MHz ticks around the tile loop,
            // median over workgroups
            std::vector<unsigned long long> h((size_t)grid * 4);
            HIP_TRY(hipStreamSynchronize(st));
            HIP_TRY(hipMemcpy(h.data(), a.dbg, h.size() * 8, hipMemcpyDeviceToHost));
            std::vector<double> ghz, cpu_;
            for (int w = 0; w < grid; ++w)
                if (h[w * 4 + 1] > 0 && h[w * 4 + 2] > 0) {
                    ghz.push_back((double)h[w * 4] / (double)h[w * 4 + 1] * 0.1);
                    cpu_.push_back((double)h[w * 4] / (double)h[w * 4 + 2]);
                }
            if (!ghz.empty()) {
                std::sort(ghz.begin(), ghz.end());
                std::sort(cpu_.begin(), cpu_.end());
                ix->probe_ghz = ghz[ghz.size() / 2];
                ix->probe_cycles_per_unit = cpu_[cpu_.size() / 2];
                ix->probe_units = (double)h[2];
                if (ix->knobs.get(K_PROBE_SPREAD, 0)) {
                    // the launch ends with its slowest workgroup: time inside the tile loop per workgroup (100 MHz ticks),
                    // and its mean by workgroup id % 8 (the XCD under round-robin dispatch)
                    std::vector<double> us;
                    double xm[8] = {0}, xn[8] = {0};
                    for (int w = 0; w < grid; ++w)
                        if (h[w * 4 + 1] > 0) {
                            us.push_back((double)h[w * 4 + 1] * 0.01);
                            xm[w & 7] += us.back();
                            xn[w & 7] += 1;
                        }
                    std::sort(us.begin(), us.end());
                    fprintf(stderr, "[tsearch probe] tile loop per workgroup: min %.1f us, median %.1f, max %.1f; mean by id %% 8:", us.front(),
                            us[us.size() / 2], us.back());
                    for (int x = 0; x < 8; ++x) fprintf(stderr, " %.1f", xm[x] / std::max(1.0, xn[x]));
                    fprintf(stderr, "\n");
                }
            }
        } else if (a.dbg && full_pass && shape16 && variant == 5) {
            std::vector<unsigned long long> h((size_t)grid * 16);
            HIP_TRY(hipStreamSynchronize(st));
            HIP_TRY(hipMemcpy(h.data(), a.dbg, h.size() * 8, hipMemcpyDeviceToHost));
            const double units = (double)(lv.back().ntiles * MfmaDims<768>::kUnits) / grid;
            for (int wv = 0; wv < 4; ++wv) {
                double tot = 0, vm = 0, bar = 0, dma = 0;
                for (int w = wv; w < grid * 4; w += 4) { tot += h[w * 4]; vm += h[w * 4 + 1]; bar += h[w * 4 + 2]; dma += h[w * 4 + 3]; }
                fprintf(stderr, "[tsearch stamps16] wave %d per unit: total %.0f cycles, vmcnt wait %.0f, barrier wait %.0f, DMA issue %.0f (6 pieces; stamp cost ~40 each included)\n",
                        wv, tot / grid / units, vm / grid / units, bar / grid / units, dma / grid / units);
            }
        } else if (a.dbg && full_pass && !shape16) {
            std::vector<unsigned long long> h((size_t)grid * 16);
            HIP_TRY(hipStreamSynchronize(st));
            HIP_TRY(hipMemcpy(h.data(), a.dbg, h.size() * 8, hipMemcpyDeviceToHost));
            for (int wv = 0; wv < 4; ++wv) {  // by wave of the workgroup: with small batches the waves differ
                double tot = 0, vm = 0, bar = 0, units = 0;
                for (int w = wv; w < grid * 4; w += 4) { tot += h[w * 4]; vm += h[w * 4 + 1]; bar += h[w * 4 + 2]; units += h[w * 4 + 3]; }
                fprintf(stderr, "[tsearch stamps] wave %d per unit: total %.0f cycles, vmcnt wait %.0f, barrier wait %.0f (units/wave %.0f)\n",
                        wv, tot / units, vm / units, bar / units, units / grid);
            }
        }
#endif
        LevelArgs l;
        memset(&l, 0, sizeof(l));
        l.priv = ix->priv;
        l.pcount = ix->pcount;
        l.nwriters = (shape16 && full_pass) ? 0 : nwriters;   // the 16x16 full pass stages its candidates in LDS: shared lists only
        l.priv_cap = priv_cap;
        l.cand = ix->cand;
        l.count = ix->count;
        l.cap = kCandCap;
        l.kk = kk;
        l.thr = ix->thr;
        l.final_level = full_pass;
        l.z_tail = full_pass ? 0.0f : z_tail;
        l.tail_p = full_pass ? 0.0f : tail_p;
        l.tail_z = (float)normal_tail_z(std::min(0.25, 32.0 / sample_rows));
        // fewer candidates back than there are answers = the threshold was too high (an estimate that overshot, or a sample
        // score that differs from the pass's in the last bit): exact re-run
        l.min_fill = (int)std::min<int64_t>(k, pop);
        l.out_scores = out_scores;
        l.out_idx = out_idx;
        l.k_user = k;
        l.row_offset = ix->row_offset;
        l.id_map = ix->id_map;
        l.fb_list = ix->fb_list;
        l.fb_count = ix->fb_count;
        l.stat_candidates = ix->stat;
        l.nq = nq;
        if (kk <= 64) level_select_kernel<1><<<nq, kLevelThreads, kLevelLds, st>>>(l);
        else level_select_kernel<4><<<nq, kLevelThreads, kLevelLds, st>>>(l);
        HIP_TRY(hipGetLastError());
    }
    // block 0 of the launch below moves the tile boundaries of the pass just finished for the next search
    ix->rebalance_pending = balance;
    ix->rebalance_grid = grid;
    // exact fall-back for queries that lost candidates (device-side count; one empty launch when 0)
    if (!in_place) TS_TRY(scan_search(ix, nq, k, out_scores, out_idx, ix->fb_list, ix->fb_count, st));
    else if (ix->dtype == TS_F32) TS_TRY(scan_search(ix, nq, k, out_scores, out_idx, ix->fb_list, ix->fb_count, st, (const float*)qmat));
    else TS_TRY(scan_search(ix, nq, k, out_scores, out_idx, ix->fb_list, ix->fb_count, st, nullptr, (const unsigned short*)qmat));
    if (stats) stats->levels = (int)lv.size();
    return TS_OK;
}

// Largest batch the streaming scan still serves faster than the MFMA path: one scan pass serves 4 queries at the HBM
// rate, and one launch of the matrix kernels (64 queries or more) costs less than two scan passes on both storage types
// (1M x 768 fp32, 5-8 queries: 0.99 ms through the scan, 0.74 ms through the 16x16x4 kernel; 10M x 768 bf16: 4.44 against
// 2.21 ms).  Large k (4 keys per lane in the scan) moves it down to 1.
static int scan_max_queries(const ts_index* ix, int k) {
    if (k > 64) return 1;
    return ix->knobs.get(K_SCAN_MAX_QUERIES, 4);
}

struct BiasSpec {       // ts_search_biased: rank by score + weight * bias[row]
    const float* bias = nullptr;
    int on_device = 0;
    float weight = 0.f;
    float* out_sims = nullptr;   // optional: raw similarities of the results, where the scores go
};

static int search_impl(ts_index* ix, const void* queries, int q_dtype, int q_on_device, int32_t nq, int32_t k,
                       float* out_scores, int64_t* out_idx, int out_on_device, void* stream, int algo,
                       ts_search_stats* stats, const uint32_t* row_mask = nullptr, int mask_on_device = 0,
                       const BiasSpec* bias = nullptr) {
    if (stats) memset(stats, 0, sizeof(*stats));
    if (!ix || !queries || !out_scores || !out_idx) return fail(TS_ERR_INVALID, "NULL argument");
    if (q_dtype != TS_F32 && q_dtype != TS_BF16) return fail(TS_ERR_INVALID, "q_dtype %d", q_dtype);
    if (nq < 0) return fail(TS_ERR_INVALID, "nq = %d", nq);
    if (k < 1 || k > TS_MAX_K) return fail(TS_ERR_INVALID, "k = %d outside [1, %d]", k, TS_MAX_K);
    if (algo < TS_ALGO_AUTO || algo > TS_ALGO_MFMA) return fail(TS_ERR_INVALID, "algo %d", algo);
    const bool mfma_ok = mfma_index(ix) && ix->n >= 1;
    if (algo == TS_ALGO_MFMA && !mfma_ok)
        return fail(TS_ERR_UNSUPPORTED, "the MFMA path needs a bf16 or fp32 index with d = 384, 512, 768 or 1024");
    if (nq == 0) return TS_OK;
    std::lock_guard<std::mutex> lock(ix->mu);
    HIP_TRY(hipSetDevice(ix->device));
    hipStream_t st;
    StreamScope scope;
    TS_TRY(enter_stream(ix, stream, &st, &scope));
    TS_TRY(ensure_search_scratch(ix, k));
    struct MaskScope {  // the bitmask and the bias are properties of this call only
        ts_index* ix;
        ~MaskScope() { ix->active_mask = nullptr; ix->active_bias = nullptr; }
    } mask_scope{ix};
    if (bias) {
        // the additive term is applied where the row is known and the key is made: the scan kernel (four queries per pass at
        // the HBM rate).  The matrix kernels test a whole accumulator tile against one threshold per query; a per-row term
        // of the size of w * ln(citations) (several standard deviations of the scores) leaves no threshold that prunes.
        if (ix->id_map) return fail(TS_ERR_UNSUPPORTED, "biased search on a subset index");
        if (algo == TS_ALGO_MFMA) return fail(TS_ERR_UNSUPPORTED, "the biased search runs on the scan kernel");
        algo = TS_ALGO_SCAN;
        if (bias->on_device) {
            ix->active_bias = bias->bias;
        } else {
            TS_TRY(ensure((void**)&ix->bias_dev, &ix->bias_bytes, std::max<size_t>((size_t)ix->n * 4, 4)));
            HIP_TRY(hipMemcpyAsync(ix->bias_dev, bias->bias, (size_t)ix->n * 4, hipMemcpyHostToDevice, st));
            ix->active_bias = ix->bias_dev;
        }
        ix->active_bias_w = bias->weight;
    }
    if (row_mask) {
        const size_t words = (size_t)((ix->n + 31) / 32);
        if (mask_on_device) {
            ix->active_mask = row_mask;
        } else {
            TS_TRY(ensure((void**)&ix->mask_dev, &ix->mask_bytes, std::max<size_t>(words * 4, 4)));
            HIP_TRY(hipMemcpyAsync(ix->mask_dev, row_mask, words * 4, hipMemcpyHostToDevice, st));
            ix->active_mask = ix->mask_dev;
        }
        // Batches behind a host mask that keeps at least a tenth of the rows run the MFMA path: the bit is tested in its
        // append path and the threshold estimates are made for the allowed rows (the sample sees only those).  Sparser
        // masks leave the sample too few allowed rows to estimate from; device masks would need a count + sync first:
        // both go through the scan kernel, 4 queries per pass (or through a subset index).
        bool dense_host_mask = false;
        if (!mask_on_device && mfma_index(ix) && nq > scan_max_queries(ix, k) &&
            ix->n >= ix->knobs.get(K_MFMA_MIN_ROWS, 16384) && algo != TS_ALGO_SCAN) {
            int64_t allowed = 0;
            for (size_t w = 0; w < words; ++w) allowed += __builtin_popcount(row_mask[w]);
            const int64_t tail_bits = (int64_t)words * 32 - ix->n;   // bits past the last row do not count
            if (tail_bits > 0 && words > 0) allowed -= __builtin_popcount(row_mask[words - 1] >> (32 - tail_bits));
            ix->active_allowed = allowed;
            dense_host_mask = allowed * 10 >= ix->n;
        }
        if (algo == TS_ALGO_MFMA && !dense_host_mask)
            return fail(TS_ERR_UNSUPPORTED, "the MFMA path serves host masks that keep at least a tenth of the rows, for more than 4 queries");
        algo = dense_host_mask ? TS_ALGO_MFMA : TS_ALGO_SCAN;
    }
    int use = algo;
    // The scan serves 4 queries per pass at the HBM rate; the MFMA path serves up to 256 per pass but its pass is
    // ~1.7x longer (matrix + HBM load drops the clock): a handful of queries is faster through the scan.
    if (use == TS_ALGO_AUTO)
        use = (mfma_ok && ix->n >= ix->knobs.get(K_MFMA_MIN_ROWS, 16384) && nq > scan_max_queries(ix, k)) ? TS_ALGO_MFMA : TS_ALGO_SCAN;
    if (stats) stats->algo = use;

    float* dscores = out_scores;
    int64_t* didx = out_idx;
    if (!out_on_device) {
        const size_t want = (size_t)nq * k;
        if (ix->res_cap < want) {
            if (ix->res_scores) HIP_TRY(hipFree(ix->res_scores));
            if (ix->res_idx) HIP_TRY(hipFree(ix->res_idx));
            ix->res_scores = nullptr; ix->res_idx = nullptr; ix->res_cap = 0;
            HIP_TRY(hipMalloc((void**)&ix->res_scores, want * 4));
            HIP_TRY(hipMalloc((void**)&ix->res_idx, want * 8));
            ix->res_cap = want;
        }
        dscores = ix->res_scores;
        didx = ix->res_idx;
    }
    const size_t q_elem = q_dtype == TS_BF16 ? 2 : 4;
    if (!q_on_device) TS_TRY(ensure(&ix->stage, &ix->stage_bytes, kStageBytes));

    // queries are served in blocks: 256 per pass, or what one launch of the MFMA kernel holds
    const int block = (use == TS_ALGO_MFMA) ? mfma_block_queries(ix, nq) : kQBlock;
    for (int q0 = 0; q0 < nq; q0 += block) {
        const int nb = std::min(block, nq - q0);
        const void* qsrc = (const char*)queries + (size_t)q0 * ix->d * q_elem;
        float* os = dscores + (size_t)q0 * k;
        int64_t* oi = didx + (size_t)q0 * k;
        // One fp32 query against an fp32 inner-product index whose rows are not padded (the single query of the apps,
        // streamlit_app.py:173, app_showcase_model.py:92; configs[1]): nothing to normalise, round or pad - the scan
        // reads the query where it is (device) or where the copy puts it (host).  No preparation launch.
        if (use == TS_ALGO_SCAN && nb == 1 && nq == 1 && q_dtype == TS_F32 && ix->dtype == TS_F32 &&
            ix->metric == TS_METRIC_IP && ix->ld == ix->d && ((uintptr_t)qsrc & 3) == 0) {
            const float* qb = (const float*)qsrc;
            if (!q_on_device) {
                HIP_TRY(hipMemcpyAsync(ix->qf32, qsrc, (size_t)ix->d * 4, hipMemcpyHostToDevice, st));
                qb = ix->qf32;
            }
            TS_TRY(scan_search(ix, 1, k, os, oi, nullptr, nullptr, st, qb));
            continue;
        }
        // Device queries that already are what the matrix kernels multiply - the index's storage type, an inner-product
        // index (nothing to normalise), rows not padded, a whole launch's worth of them, 16-byte aligned - are read where
        // they lie: no preparation launch (the encoder's fused pooling writes this form, ts_pool_normalize with
        // out_dtype = the index's; bench.py's resident query batch).  They must stay unchanged until the search has run.
        if (use == TS_ALGO_MFMA && q_on_device && q_dtype == ix->dtype && ix->metric == TS_METRIC_IP && ix->ld == ix->d &&
            nb == block && ((uintptr_t)qsrc & 15) == 0) {
            TS_TRY(mfma_search(ix, nb, k, os, oi, st, stats, qsrc, true));
            continue;
        }
        if (!q_on_device) {
            HIP_TRY(hipMemcpyAsync(ix->stage, qsrc, (size_t)nb * ix->d * q_elem, hipMemcpyHostToDevice, st));
            qsrc = ix->stage;
        }
        // normalise (COS), round to the storage type, zero-pad to 256 rows x ld; fp32 copy for the scan
        TS_TRY(prep_dispatch(q_dtype, ix->dtype, ix->metric == TS_METRIC_COS, qsrc, ix->d, ix->qstore, ix->qf32, ix->ld, ix->d,
                             nb, kQBlock, st));
        if (use == TS_ALGO_MFMA) TS_TRY(mfma_search(ix, nb, k, os, oi, st, stats, ix->qstore, false));
        else TS_TRY(scan_search(ix, nb, k, os, oi, nullptr, nullptr, st));
    }
    DevBuf sims_tmp;
    if (bias && bias->out_sims) {
        float* dsims = bias->out_sims;
        const int64_t cnt = (int64_t)nq * k;
        if (!out_on_device) {
            HIP_TRY(sims_tmp.alloc((size_t)cnt * 4));
            dsims = sims_tmp.as<float>();
        }
        unbias_kernel<<<(unsigned)((cnt + 255) / 256), 256, 0, st>>>(dscores, didx, ix->active_bias, ix->active_bias_w, ix->row_offset, dsims, cnt);
        HIP_TRY(hipGetLastError());
        if (!out_on_device) HIP_TRY(hipMemcpyAsync(bias->out_sims, dsims, (size_t)cnt * 4, hipMemcpyDeviceToHost, st));
    }
    if (!out_on_device) {
        HIP_TRY(hipMemcpyAsync(out_scores, dscores, (size_t)nq * k * 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(out_idx, didx, (size_t)nq * k * 8, hipMemcpyDeviceToHost, st));
    }
    if (!out_on_device || stats) {
        int fb = 0;
        unsigned long long cands = 0;
        if (stats && use == TS_ALGO_MFMA) {
            HIP_TRY(hipMemcpyAsync(&fb, ix->fb_count, 4, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(&cands, ix->stat, 8, hipMemcpyDeviceToHost, st));
        }
        HIP_TRY(hipStreamSynchronize(st));
        if (stats && use == TS_ALGO_MFMA) {
            stats->fallback_queries = fb;
            stats->candidates = (int64_t)cands;
        }
    }
    return TS_OK;
}

extern "C" int ts_search_ex(ts_index* ix, const void* queries, int q_dtype, int q_on_device, int32_t nq, int32_t k,
                            float* out_scores, int64_t* out_idx, int out_on_device, void* stream, int algo,
                            ts_search_stats* stats) {
    return search_impl(ix, queries, q_dtype, q_on_device, nq, k, out_scores, out_idx, out_on_device, stream, algo, stats);
}

extern "C" int ts_search(ts_index* ix, const void* queries, int q_dtype, int q_on_device, int32_t nq, int32_t k,
                         float* out_scores, int64_t* out_idx, int out_on_device, void* stream) {
    return search_impl(ix, queries, q_dtype, q_on_device, nq, k, out_scores, out_idx, out_on_device, stream, TS_ALGO_AUTO,
                       nullptr);
}

extern "C" int ts_search_filtered(ts_index* ix, const void* queries, int q_dtype, int q_on_device, int32_t nq, int32_t k,
                                  const uint32_t* row_mask, int mask_on_device, float* out_scores, int64_t* out_idx,
                                  int out_on_device, void* stream) {
    if (!row_mask) return fail(TS_ERR_INVALID, "row_mask is NULL");
    return search_impl(ix, queries, q_dtype, q_on_device, nq, k, out_scores, out_idx, out_on_device, stream, TS_ALGO_AUTO,
                       nullptr, row_mask, mask_on_device);
}

extern "C" int ts_search_filtered_ex(ts_index* ix, const void* queries, int q_dtype, int q_on_device, int32_t nq, int32_t k,
                                     const uint32_t* row_mask, int mask_on_device, float* out_scores, int64_t* out_idx,
                                     int out_on_device, void* stream, int algo, ts_search_stats* stats) {
    if (!row_mask) return fail(TS_ERR_INVALID, "row_mask is NULL");
    return search_impl(ix, queries, q_dtype, q_on_device, nq, k, out_scores, out_idx, out_on_device, stream, algo, stats,
                       row_mask, mask_on_device);
}

extern "C" int ts_search_biased(ts_index* ix, const void* queries, int q_dtype, int q_on_device, int32_t nq, int32_t k,
                                const float* bias, int bias_on_device, float weight, const uint32_t* row_mask, int mask_on_device,
                                float* out_scores, float* out_sims, int64_t* out_idx, int out_on_device, void* stream) {
    if (!bias) return fail(TS_ERR_INVALID, "bias is NULL");
    if (!(weight == weight) || std::isinf(weight)) return fail(TS_ERR_INVALID, "weight must be finite");
    BiasSpec b;
    b.bias = bias;
    b.on_device = bias_on_device;
    b.weight = weight;
    b.out_sims = out_sims;
    return search_impl(ix, queries, q_dtype, q_on_device, nq, k, out_scores, out_idx, out_on_device, stream, TS_ALGO_AUTO, nullptr,
                       row_mask, mask_on_device, &b);
}

template <int DT, int CH, int G>
static void launch_rank_spec(int qb, int grid, hipStream_t st, const RankArgs& a) {
    if (qb == 4) rank_kernel<DT, CH, G, 4><<<grid, 256, 0, st>>>(a);
    else rank_kernel<DT, CH, G, 1><<<grid, 256, 0, st>>>(a);
}

static void launch_rank(const ts_index* ix, const RankArgs& a, hipStream_t st, int grid) {
    const int qb = a.nq >= 2 ? 4 : 1;
    const bool force_generic = ix->knobs.get(K_SCAN_GENERIC, 0) != 0;
    if (!force_generic && ix->dtype == TS_F32 && ix->ld == 768) return launch_rank_spec<0, 3, 64>(qb, grid, st, a);
    if (!force_generic && ix->dtype == TS_F32 && ix->ld == 1024) return launch_rank_spec<0, 4, 64>(qb, grid, st, a);
    if (!force_generic && ix->dtype == TS_BF16 && ix->ld == 768) return launch_rank_spec<1, 3, 32>(qb, grid, st, a);
    if (!force_generic && ix->dtype == TS_BF16 && ix->ld == 1024) return launch_rank_spec<1, 2, 64>(qb, grid, st, a);
    if (!force_generic && ix->dtype == TS_F32 && ix->ld == 384) return launch_rank_spec<0, 3, 32>(qb, grid, st, a);
    if (!force_generic && ix->dtype == TS_BF16 && ix->ld == 384) return launch_rank_spec<1, 3, 16>(qb, grid, st, a);
    if (!force_generic && ix->dtype == TS_F32 && ix->ld == 512) return launch_rank_spec<0, 2, 64>(qb, grid, st, a);
    if (!force_generic && ix->dtype == TS_BF16 && ix->ld == 512) return launch_rank_spec<1, 2, 32>(qb, grid, st, a);
    const size_t lds = (size_t)a.ld * 4;
    if (ix->dtype == TS_F32) {
        hipFuncSetAttribute((const void*)rank_generic_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        rank_generic_kernel<0><<<grid, 256, lds, st>>>(a);
    } else {
        hipFuncSetAttribute((const void*)rank_generic_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        rank_generic_kernel<1><<<grid, 256, lds, st>>>(a);
    }
}

// ---------------------------------------------------------------------------------------------
// ingestion: pgvector text rows (host code, no device)
// ---------------------------------------------------------------------------------------------
extern "C" int ts_parse_pgvector_text(const char* text, int64_t len, int32_t d, float* out, int64_t max_rows, int64_t* rows_parsed,
                                      int64_t* consumed) {
    if (!text || !out || !rows_parsed || len < 0 || d <= 0 || max_rows < 0) return fail(TS_ERR_INVALID, "bad argument");
    int64_t pos = 0, rows = 0, done = 0;
    char tok[64];
    while (rows < max_rows) {
        while (pos < len && text[pos] != '[') ++pos;  // anything before the bracket (ids, tabs, quotes, newlines) is skipped
        if (pos >= len) break;
        int64_t p = pos + 1;
        int col = 0;
        bool closed = false;
        while (p < len) {
            while (p < len && (isspace((unsigned char)text[p]) || text[p] == ',')) ++p;
            if (p < len && text[p] == ']') { closed = true; ++p; break; }
            int t = 0;
            while (p < len && text[p] != ',' && text[p] != ']' && !isspace((unsigned char)text[p]) && t < 63) tok[t++] = text[p++];
            if (p >= len) break;  // value cut off by the end of the buffer: the caller resumes at `consumed`
            tok[t] = 0;
            if (t == 63 && text[p] != ',' && text[p] != ']' && !isspace((unsigned char)text[p]))
                return fail(TS_ERR_INVALID, "row %lld: numeric literal longer than 63 characters", (long long)rows);
            // vector_in takes decimal literals only: strtof would also accept nan / inf / hex floats
            for (int c = 0; c < t; ++c)
                if (!(isdigit((unsigned char)tok[c]) || tok[c] == '+' || tok[c] == '-' || tok[c] == '.' || tok[c] == 'e' || tok[c] == 'E'))
                    return fail(TS_ERR_INVALID, "row %lld: cannot parse '%s'", (long long)rows, tok);
            char* end = nullptr;
            const float v = strtof(tok, &end);  // what pgvector's vector_in does: one correctly rounded fp32 conversion
            if (end == tok || *end != 0) return fail(TS_ERR_INVALID, "row %lld: cannot parse '%s'", (long long)rows, tok);
            if (!std::isfinite(v)) return fail(TS_ERR_INVALID, "row %lld: '%s' is not a finite float", (long long)rows, tok);
            if (col >= d) return fail(TS_ERR_INVALID, "row %lld has more than %d values", (long long)rows, d);
            out[rows * d + col++] = v;
        }
        if (!closed) break;  // incomplete row at the end of the buffer
        if (col != d) return fail(TS_ERR_INVALID, "row %lld has %d values, expected %d", (long long)rows, col, d);
        ++rows;
        pos = done = p;
    }
    *rows_parsed = rows;
    if (consumed) *consumed = done;
    return TS_OK;
}

// host twin of ord_f32 (common.h): the score half of a key
static u32 host_ord_f32(float s) {
    s = s + 0.0f;
    u32 u;
    memcpy(&u, &s, 4);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// target_rows != NULL: rank of that row (its score is computed by the kernel);  otherwise target_scores / target_ids:
// number of rows of THIS index that rank before a document with that score and global id (it may live on another shard).
static int rank_impl(ts_index* ix, const void* queries, int q_dtype, int q_on_device, int32_t nq, const int64_t* target_rows,
                     const float* target_scores, const int64_t* target_ids, int64_t* out_rank, float* out_score, void* stream) {
    if (!ix || !queries || !out_rank) return fail(TS_ERR_INVALID, "NULL argument");
    if (q_dtype != TS_F32 && q_dtype != TS_BF16) return fail(TS_ERR_INVALID, "q_dtype %d", q_dtype);
    if (nq < 0) return fail(TS_ERR_INVALID, "nq = %d", nq);
    if (nq == 0) return TS_OK;
    if (ix->id_map) return fail(TS_ERR_UNSUPPORTED, "rank / count on a subset index");
    std::lock_guard<std::mutex> lock(ix->mu);
    HIP_TRY(hipSetDevice(ix->device));
    hipStream_t st;
    StreamScope scope;
    TS_TRY(enter_stream(ix, stream, &st, &scope));
    TS_TRY(ensure_search_scratch(ix, 1));
    constexpr size_t kPer = 8 + 8 + 4;
    TS_TRY(ensure(&ix->rank_buf, &ix->rank_bytes, (size_t)kQBlock * kPer));
    int64_t* d_target = (int64_t*)ix->rank_buf;  // rows, or ready-made keys
    unsigned long long* d_counts = (unsigned long long*)(d_target + kQBlock);
    float* d_tscore = (float*)(d_counts + kQBlock);
    const size_t q_elem = q_dtype == TS_BF16 ? 2 : 4;
    if (!q_on_device) TS_TRY(ensure(&ix->stage, &ix->stage_bytes, kStageBytes));
    std::vector<int64_t> local(kQBlock);
    std::vector<unsigned long long> counts(kQBlock);
    std::vector<float> tscore(kQBlock);
    for (int q0 = 0; q0 < nq; q0 += kQBlock) {
        const int nb = std::min(kQBlock, nq - q0);
        const void* qsrc = (const char*)queries + (size_t)q0 * ix->d * q_elem;
        if (!q_on_device) {
            HIP_TRY(hipMemcpyAsync(ix->stage, qsrc, (size_t)nb * ix->d * q_elem, hipMemcpyHostToDevice, st));
            qsrc = ix->stage;
        }
        TS_TRY(prep_dispatch(q_dtype, ix->dtype, ix->metric == TS_METRIC_COS, qsrc, ix->d, ix->qstore, ix->qf32, ix->ld, ix->d, nb,
                             kQBlock, st));
        for (int i = 0; i < nb; ++i) {
            if (target_rows) {
                const int64_t r = target_rows[q0 + i] - ix->row_offset;
                local[i] = (r >= 0 && r < ix->n) ? r : -1;
            } else {
                // key of (score, global id) in this shard's key space: a document before the shard loses every tie
                // (low word all ones), one behind it wins every tie (low word zero)
                const float sc = target_scores[q0 + i];
                const int64_t r = target_ids[q0 + i] - ix->row_offset;
                const u64 low = r < 0 ? 0xFFFFFFFFull : (r >= ix->n ? 0ull : (u64)(0xFFFFFFFFu - (u32)r));
                local[i] = (sc == sc) ? (int64_t)(((u64)host_ord_f32(sc) << 32) | low) : -1;  // NaN: all ones, nothing counts
            }
        }
        HIP_TRY(hipMemcpyAsync(d_target, local.data(), (size_t)nb * 8, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemsetAsync(d_counts, 0, (size_t)nb * 8, st));
        RankArgs a;
        memset(&a, 0, sizeof(a));
        a.corpus = ix->rows;
        a.ld = ix->ld;
        a.n = ix->n;
        a.qbuf = ix->qf32;
        a.nq = nb;
        a.target = d_target;
        a.tkey = target_rows ? nullptr : (const u64*)d_target;
        a.counts = d_counts;
        a.tscore = d_tscore;
        hipEvent_t stop = prof_begin(ix, st, ix->n);
        launch_rank(ix, a, st, ix->cu_count * kScanGridPerCU);
        prof_end(stop, st);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(counts.data(), d_counts, (size_t)nb * 8, hipMemcpyDeviceToHost, st));
        if (target_rows) HIP_TRY(hipMemcpyAsync(tscore.data(), d_tscore, (size_t)nb * 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));  // `local` is reused by the next block
        for (int i = 0; i < nb; ++i) {
            if (target_rows) {
                const bool ok = local[i] >= 0 && tscore[i] == tscore[i];
                out_rank[q0 + i] = ok ? (int64_t)counts[i] : -1;
                if (out_score) out_score[q0 + i] = tscore[i];
            } else {
                out_rank[q0 + i] = (target_scores[q0 + i] == target_scores[q0 + i]) ? (int64_t)counts[i] : -1;
            }
        }
    }
    return TS_OK;
}

extern "C" int ts_rank_of(ts_index* ix, const void* queries, int q_dtype, int q_on_device, int32_t nq, const int64_t* target_rows,
                          int64_t* out_rank, float* out_score, void* stream) {
    if (!target_rows) return fail(TS_ERR_INVALID, "target_rows is NULL");
    return rank_impl(ix, queries, q_dtype, q_on_device, nq, target_rows, nullptr, nullptr, out_rank, out_score, stream);
}

extern "C" int ts_count_above(ts_index* ix, const void* queries, int q_dtype, int q_on_device, int32_t nq, const float* target_scores,
                              const int64_t* target_ids, int64_t* out_counts, void* stream) {
    if (!target_scores || !target_ids) return fail(TS_ERR_INVALID, "NULL argument");
    return rank_impl(ix, queries, q_dtype, q_on_device, nq, nullptr, target_scores, target_ids, out_counts, nullptr, stream);
}

extern "C" int ts_scores(ts_index* ix, const void* queries, int q_dtype, int q_on_device, int32_t nq, float* out,
                         int out_on_device, void* stream) {
    if (!ix || !queries || !out) return fail(TS_ERR_INVALID, "NULL argument");
    if (q_dtype != TS_F32 && q_dtype != TS_BF16) return fail(TS_ERR_INVALID, "q_dtype %d", q_dtype);
    if (nq < 0) return fail(TS_ERR_INVALID, "nq = %d", nq);
    if (nq == 0 || ix->n == 0) return TS_OK;
    std::lock_guard<std::mutex> lock(ix->mu);
    HIP_TRY(hipSetDevice(ix->device));
    hipStream_t st;
    StreamScope scope;
    TS_TRY(enter_stream(ix, stream, &st, &scope));
    TS_TRY(ensure_search_scratch(ix, 1));
    float* dout = out;
    DevBuf tmp;
    if (!out_on_device) {
        // the score matrix is for the evaluation script's small shapes: refuse what cannot fit beside the index
        const size_t want = (size_t)nq * (size_t)ix->n * 4;
        size_t free_b = 0, total_b = 0;
        HIP_TRY(hipMemGetInfo(&free_b, &total_b));
        if (want > free_b / 2)
            return fail(TS_ERR_NOMEM, "score matrix of %d x %lld floats (%zu bytes) does not fit: use ts_search / ts_rank_of",
                        nq, (long long)ix->n, want);
        HIP_TRY(tmp.alloc(want));
        dout = tmp.as<float>();
    }
    const size_t q_elem = q_dtype == TS_BF16 ? 2 : 4;
    if (!q_on_device) TS_TRY(ensure(&ix->stage, &ix->stage_bytes, kStageBytes));
    int rc = TS_OK;
    for (int q0 = 0; q0 < nq && rc == TS_OK; q0 += kQBlock) {
        const int nb = std::min(kQBlock, nq - q0);
        const void* qsrc = (const char*)queries + (size_t)q0 * ix->d * q_elem;
        if (!q_on_device) {
            if (hipMemcpyAsync(ix->stage, qsrc, (size_t)nb * ix->d * q_elem, hipMemcpyHostToDevice, st) != hipSuccess) {
                rc = fail(TS_ERR_HIP, "copy of the queries failed");
                break;
            }
            qsrc = ix->stage;
        }
        rc = prep_dispatch(q_dtype, ix->dtype, ix->metric == TS_METRIC_COS, qsrc, ix->d, ix->qstore, ix->qf32, ix->ld, ix->d, nb,
                           kQBlock, st);
        if (rc != TS_OK) break;
        ScanArgs a;
        memset(&a, 0, sizeof(a));
        a.corpus = ix->rows;
        a.ld = ix->ld;
        a.n = ix->n;
        a.qbuf = ix->qf32;
        a.nq = nb;
        a.k = 1;
        a.scores = dout + (size_t)q0 * ix->n;
        launch_scan<true>(ix, a, nb >= 2 ? 4 : 1, st, ix->cu_count * kScanGridPerCU);
        if (hipGetLastError() != hipSuccess) rc = fail(TS_ERR_HIP, "score kernel launch failed");
    }
    if (!out_on_device) {
        if (rc == TS_OK && hipMemcpyAsync(out, dout, (size_t)nq * ix->n * 4, hipMemcpyDeviceToHost, st) != hipSuccess)
            rc = fail(TS_ERR_HIP, "copy of the score matrix failed");
        if (hipStreamSynchronize(st) != hipSuccess && rc == TS_OK) rc = fail(TS_ERR_HIP, "stream synchronize failed");
    }
    return rc;
}

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

extern "C" int ts_pool_normalize(int device, const void* hidden, int h_dtype, const int64_t* attention_mask, int64_t n,
                                 int32_t seq, int32_t d, int pooling, int normalize, void* out, int out_dtype, int64_t out_ld,
                                 void* stream) {
    if (!hidden || !attention_mask || !out) return fail(TS_ERR_INVALID, "NULL argument");
    if ((h_dtype != TS_F32 && h_dtype != TS_BF16) || (out_dtype != TS_F32 && out_dtype != TS_BF16))
        return fail(TS_ERR_INVALID, "dtype");
    if (n < 0 || seq < 1 || d < 1 || d > 4096 || out_ld < d) return fail(TS_ERR_INVALID, "bad shape (d must be <= 4096)");
    if (pooling < TS_POOL_MEAN || pooling > TS_POOL_CLS) return fail(TS_ERR_INVALID, "pooling %d", pooling);
    if (n == 0) return TS_OK;
    TS_TRY(check_device(device));
    HIP_TRY(hipSetDevice(device));
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)n);
    // the encoders' shapes take the vector form (16-byte loads, tokens dealt over thread groups); anything else the general one
    const int vec = h_dtype == TS_BF16 ? 8 : 4;
    const bool vform = d % vec == 0 && d / vec <= 256 && seq <= kPoolVecSeq && ((uintptr_t)hidden & 15) == 0;
#define TS_POOL_LAUNCH(H, O)                                                                                                  \
    do {                                                                                                                      \
        if (vform) pool_normalize_vec_kernel<H, O><<<grid, 256, 0, st>>>(hidden, attention_mask, seq, d, pooling, normalize, out, out_ld); \
        else pool_normalize_kernel<H, O><<<grid, 256, 0, st>>>(hidden, attention_mask, seq, d, pooling, normalize, out, out_ld); \
    } while (0)
    if (h_dtype == TS_F32 && out_dtype == TS_F32) TS_POOL_LAUNCH(0, 0);
    else if (h_dtype == TS_F32) TS_POOL_LAUNCH(0, 1);
    else if (out_dtype == TS_F32) TS_POOL_LAUNCH(1, 0);
    else TS_POOL_LAUNCH(1, 1);
#undef TS_POOL_LAUNCH
    HIP_TRY(hipGetLastError());
    return TS_OK;
}

extern "C" int ts_add_layernorm(int device, const void* a, const void* b, const void* gamma, const void* beta, float eps, int64_t rows,
                                int32_t d, int dtype, void* out, void* stream) {
    if (!a || !b || !gamma || !beta || !out) return fail(TS_ERR_INVALID, "NULL argument");
    if (dtype != TS_F32 && dtype != TS_BF16) return fail(TS_ERR_INVALID, "dtype %d", dtype);
    const int vec = dtype == TS_BF16 ? 8 : 4;
    if (rows < 0 || d < vec || d % vec || d > 64 * kLnMax * vec)
        return fail(TS_ERR_INVALID, "d = %d must be a multiple of %d and at most %d", d, vec, 64 * kLnMax * vec);
    if ((((uintptr_t)a | (uintptr_t)b | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)out) & 15) != 0)
        return fail(TS_ERR_INVALID, "buffers must be 16-byte aligned");
    if (rows == 0) return TS_OK;
    TS_TRY(check_device(device));
    HIP_TRY(hipSetDevice(device));
    const unsigned grid = (unsigned)((rows + 3) / 4);
    const int per_lane = (d / vec + 63) / 64;              // 16-byte accesses per lane
    hipStream_t st = (hipStream_t)stream;
#define TS_LN_LAUNCH(DT_)                                                                                        \
    do {                                                                                                         \
        if (per_lane <= 1) add_layernorm_kernel<DT_, 1><<<grid, 256, 0, st>>>(a, b, gamma, beta, eps, rows, d, out); \
        else if (per_lane <= 2) add_layernorm_kernel<DT_, 2><<<grid, 256, 0, st>>>(a, b, gamma, beta, eps, rows, d, out); \
        else add_layernorm_kernel<DT_, 4><<<grid, 256, 0, st>>>(a, b, gamma, beta, eps, rows, d, out);           \
    } while (0)
    if (dtype == TS_F32) TS_LN_LAUNCH(0);
    else TS_LN_LAUNCH(1);
#undef TS_LN_LAUNCH
    HIP_TRY(hipGetLastError());
    return TS_OK;
}

extern "C" int ts_embed_layernorm(int device, const int64_t* ids, const int64_t* type_ids, const void* word, const void* pos,
                                  const void* type, int64_t n_word, int64_t n_pos, int64_t n_type, const void* gamma, const void* beta,
                                  float eps, int64_t tokens, int32_t seq, int32_t d, int dtype, void* out, void* stream) {
    if (!ids || !word || !pos || !type || !gamma || !beta || !out) return fail(TS_ERR_INVALID, "NULL argument");
    if (dtype != TS_F32 && dtype != TS_BF16) return fail(TS_ERR_INVALID, "dtype %d", dtype);
    const int vec = dtype == TS_BF16 ? 8 : 4;
    if (tokens < 0 || seq < 1 || d < vec || d % vec || d > 64 * kLnMax * vec)
        return fail(TS_ERR_INVALID, "d = %d must be a multiple of %d and at most %d; seq >= 1", d, vec, 64 * kLnMax * vec);
    if (n_word < 1 || n_pos < 1 || n_type < 1) return fail(TS_ERR_INVALID, "empty embedding table");
    if ((((uintptr_t)word | (uintptr_t)pos | (uintptr_t)type | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)out) & 15) != 0)
        return fail(TS_ERR_INVALID, "tables and output must be 16-byte aligned");
    if (tokens == 0) return TS_OK;
    TS_TRY(check_device(device));
    HIP_TRY(hipSetDevice(device));
    const unsigned grid = (unsigned)((tokens + 3) / 4);
    const int per_lane = (d / vec + 63) / 64;
    hipStream_t st = (hipStream_t)stream;
#define TS_EMB_LAUNCH(DT_, LN_)                                                                                              \
    embed_layernorm_kernel<DT_, LN_><<<grid, 256, 0, st>>>(ids, type_ids, word, pos, type, n_word, n_pos, n_type, gamma, beta, eps, \
                                                          tokens, seq, d, out)
    if (dtype == TS_F32) {
        if (per_lane <= 1) TS_EMB_LAUNCH(0, 1);
        else if (per_lane <= 2) TS_EMB_LAUNCH(0, 2);
        else TS_EMB_LAUNCH(0, 4);
    } else {
        if (per_lane <= 1) TS_EMB_LAUNCH(1, 1);
        else if (per_lane <= 2) TS_EMB_LAUNCH(1, 2);
        else TS_EMB_LAUNCH(1, 4);
    }
#undef TS_EMB_LAUNCH
    HIP_TRY(hipGetLastError());
    return TS_OK;
}

extern "C" int ts_attention_short(int device, const void* qkv, const int64_t* attention_mask, int32_t batch, int32_t seq, int32_t heads,
                                 int32_t head_dim, void* out, void* stream) {
    if (!qkv || !out) return fail(TS_ERR_INVALID, "NULL argument");
    if (batch < 0 || seq < 1 || heads < 1) return fail(TS_ERR_INVALID, "batch = %d, seq = %d, heads = %d", batch, seq, heads);
    if (head_dim != 64 || seq > kAttnMaxSeq)
        return fail(TS_ERR_UNSUPPORTED, "head size %d / %d tokens: this kernel serves head size 64 and at most %d tokens", head_dim, seq,
                    kAttnMaxSeq);
    if ((((uintptr_t)qkv | (uintptr_t)out) & 15) != 0) return fail(TS_ERR_INVALID, "qkv and out must be 16-byte aligned");
    if (batch == 0) return TS_OK;
    TS_TRY(check_device(device));
    HIP_TRY(hipSetDevice(device));
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = (unsigned)(((int64_t)batch * heads + 3) / 4);
    const unsigned short* in = (const unsigned short*)qkv;
    unsigned short* o = (unsigned short*)out;
    switch ((seq + 15) / 16) {
        case 1: attention_short_kernel<1><<<grid, 256, 0, st>>>(in, attention_mask, batch, seq, heads, o); break;
        case 2: attention_short_kernel<2><<<grid, 256, 0, st>>>(in, attention_mask, batch, seq, heads, o); break;
        case 3: attention_short_kernel<3><<<grid, 256, 0, st>>>(in, attention_mask, batch, seq, heads, o); break;
        default: attention_short_kernel<4><<<grid, 256, 0, st>>>(in, attention_mask, batch, seq, heads, o); break;
    }
    HIP_TRY(hipGetLastError());
    return TS_OK;
}

extern "C" int ts_index_profile_enable(ts_index* ix, int enable) {
    if (!ix) return fail(TS_ERR_INVALID, "index is NULL");
    std::lock_guard<std::mutex> lock(ix->mu);
    ix->profiling = enable != 0;
    ix->ev_used = 0;
    if (enable) {
        // events are created here, not inside the loop that is being measured
        HIP_TRY(hipSetDevice(ix->device));
        while (ix->ev_pool.size() < 512) {
            hipEvent_t e = nullptr;
            if (hipEventCreate(&e) != hipSuccess) break;
            ix->ev_pool.push_back(e);
        }
    }
    return TS_OK;
}

extern "C" int ts_index_probe_read(ts_index* ix, double* ghz, double* cycles_per_unit, double* units_per_workgroup) {
    if (!ix) return fail(TS_ERR_INVALID, "index is NULL");
    std::lock_guard<std::mutex> lock(ix->mu);
    if (ghz) *ghz = ix->probe_ghz;
    if (cycles_per_unit) *cycles_per_unit = ix->probe_cycles_per_unit;
    if (units_per_workgroup) *units_per_workgroup = ix->probe_units;
    return TS_OK;
}

extern "C" int ts_index_profile_read(ts_index* ix, int64_t* launches, double* total_ms, int64_t* rows_per_launch) {
    if (!ix) return fail(TS_ERR_INVALID, "index is NULL");
    std::lock_guard<std::mutex> lock(ix->mu);
    HIP_TRY(hipSetDevice(ix->device));
    double sum = 0.0;
    int64_t cnt = 0;
    for (size_t i = 0; i + 1 < ix->ev_used; i += 2) {
        HIP_TRY(hipEventSynchronize(ix->ev_pool[i + 1]));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, ix->ev_pool[i], ix->ev_pool[i + 1]));
        sum += ms;
        ++cnt;
    }
    ix->ev_used = 0;
    if (launches) *launches = cnt;
    if (total_ms) *total_ms = sum;
    if (rows_per_launch) *rows_per_launch = ix->prof_rows;
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

// ---------------------------------------------------------------------------------------------
// timers
// ---------------------------------------------------------------------------------------------
extern "C" int ts_timer_create(int device, ts_timer** out) {
    if (!out) return fail(TS_ERR_INVALID, "out is NULL");
    TS_TRY(check_device(device));
    HIP_TRY(hipSetDevice(device));
    ts_timer* t = new (std::nothrow) ts_timer();
    if (!t) return fail(TS_ERR_NOMEM, "host allocation failed");
    t->device = device;
    HIP_TRY(hipEventCreate(&t->a));
    HIP_TRY(hipEventCreate(&t->b));
    *out = t;
    return TS_OK;
}
extern "C" int ts_timer_start(ts_timer* t, void* stream) {
    if (!t) return fail(TS_ERR_INVALID, "timer is NULL");
    HIP_TRY(hipSetDevice(t->device));
    HIP_TRY(hipEventRecord(t->a, (hipStream_t)stream));
    return TS_OK;
}
extern "C" int ts_timer_stop(ts_timer* t, void* stream) {
    if (!t) return fail(TS_ERR_INVALID, "timer is NULL");
    HIP_TRY(hipSetDevice(t->device));
    HIP_TRY(hipEventRecord(t->b, (hipStream_t)stream));
    return TS_OK;
}
extern "C" int ts_timer_elapsed_ms(ts_timer* t, float* ms) {
    if (!t || !ms) return fail(TS_ERR_INVALID, "NULL argument");
    HIP_TRY(hipSetDevice(t->device));
    HIP_TRY(hipEventSynchronize(t->b));
    HIP_TRY(hipEventElapsedTime(ms, t->a, t->b));
    return TS_OK;
}
extern "C" int ts_timer_destroy(ts_timer* t) {
    if (!t) return TS_OK;
    hipSetDevice(t->device);
    if (t->a) hipEventDestroy(t->a);
    if (t->b) hipEventDestroy(t->b);
    delete t;
    return TS_OK;
}

#include "shards.inc"
