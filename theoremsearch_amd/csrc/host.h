// Host side shared by the translation units of libtsearch.so (index.hip, search.hip, search_mfma.hip, the launch_*.hip
// files, encoder_ops.hip, shards.hip): error reporting, the per-handle knobs, the handles themselves, stream ordering.
// Internal: nothing here is part of the C ABI (include/tsearch.h); the library is built with hidden visibility.
#pragma once
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

using namespace ts;

// ---------------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------------
inline thread_local char g_err[512] = "";

inline int fail(int code, const char* fmt, ...) {
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
    K_MFMA_PAIR, K_MFMA_PAIR_LAG, K_COUNT
};
inline const char* const kKnobNames[K_COUNT] = {
    "TS_MFMA_MIN_RANK", "TS_MFMA_VARIANT", "TS_MFMA_GROUPS", "TS_MFMA_GRID", "TS_MFMA_STAT", "TS_MFMA_STAT_CANDS",
    "TS_MFMA_TAIL_FIT", "TS_MFMA_NO_IDLE", "TS_MFMA_AHEAD", "TS_MFMA_TARGET_CANDS", "TS_MFMA_FIRST_ROWS",
    "TS_MFMA_TARGET_SPARSE", "TS_MFMA_RUN", "TS_MFMA_MIN_ROWS", "TS_MFMA_SHAPE", "TS_MFMA_F32", "TS_SCAN_GENERIC",
    "TS_SCAN_MAX_QUERIES", "TS_MFMA_BALANCE", "TS_PROBE_SPREAD", "TS_MFMA_SAMPLE", "TS_MFMA_PAIR", "TS_MFMA_PAIR_LAG"};
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
        // the timing-only kernel variants and the knobs only the A/B tools ever turned exist in the diagnostic build only
        // (make diag): the product library runs their defaults
        for (int i = 0; i < K_COUNT; ++i)
            if (diag_only((Knob)i)) { set[i] = false; v[i] = 0; }
#endif
    }
    static bool diag_only(Knob k) {
        return k == K_MFMA_VARIANT || k == K_MFMA_MIN_RANK || k == K_MFMA_GROUPS || k == K_MFMA_STAT_CANDS || k == K_MFMA_NO_IDLE ||
               k == K_MFMA_TARGET_CANDS || k == K_MFMA_TARGET_SPARSE || k == K_MFMA_MIN_ROWS || k == K_SCAN_GENERIC ||
               k == K_SCAN_MAX_QUERIES || k == K_PROBE_SPREAD;
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
    bool rebalance_pending = false, rebalance_in_rerun = false; int rebalance_grid = 0;  // the exact re-run's launch also moves the full pass's tile boundaries
    int* fb_list = nullptr;     int* fb_count = nullptr;     u32* stat = nullptr;   // [kQBlock] candidates per query of the last final select
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
    unsigned* pair_pos = nullptr;                            // paired full pass: the tile each workgroup has reached (one word per workgroup of the largest grid)
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

inline int ensure(void** p, size_t* have, size_t want) {
    if (*have >= want && *p) return TS_OK;
    if (*p) HIP_TRY(hipFree(*p));
    *p = nullptr;
    *have = 0;
    HIP_TRY(hipMalloc(p, want));
    *have = want;
    return TS_OK;
}

// The staging buffer between host and device holds what ONE call moves, up to kStageBytes (larger transfers pass through it in
// pieces), in steps of 1 MiB; it only grows.  (A fixed 256 MiB per index cost the throw-away index of util.cos_sim -
// compare_embeddings.py:61, a thousand rows - 5.6 ms of hipMalloc in its upload and 0.9 ms of hipFree in its close around a
// 0.4 ms score kernel: profiles/r05m_cos_sim_phases.json.)  `floor_bytes`: the largest single piece the caller will put there.
inline int ensure_stage(ts_index* ix, size_t need, size_t floor_bytes) {
    const size_t gran = (size_t)1 << 20;
    const size_t want = std::max(std::min(kStageBytes, need), std::max(floor_bytes, (size_t)1));
    return ensure(&ix->stage, &ix->stage_bytes, (want + gran - 1) / gran * gran);
}

// Event bracket around one launch: prof_begin records the start event and returns the stop event
// (NULL when profiling is off); the caller records it with prof_end after the launch.
inline hipEvent_t prof_begin(ts_index* ix, hipStream_t st, int64_t rows) {
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
inline void prof_end(hipEvent_t stop, hipStream_t st) {
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
        // a call that is being captured into a HIP graph records nothing: an event recorded during capture belongs to the
        // graph and cannot order a later call on another stream (include/tsearch.h: captured calls are ordered by the caller)
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cap) != hipSuccess) (void)hipGetLastError();
        if (cap != hipStreamCaptureStatusNone) return;
        if (hipEventRecord(ix->order_ev, st) == hipSuccess) {
            ix->last_stream = st;
            ix->ordered = true;
        } else {
            (void)hipGetLastError();
        }
    }
};
inline int enter_stream(ts_index* ix, void* stream, hipStream_t* out, StreamScope* scope) {
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

inline int check_device(int device) {
    int c = 0;
    ts_device_count(&c);
    if (c <= 0) return fail(TS_ERR_NODEVICE, "no HIP device visible (libtsearch has no CPU path)");
    if (device < 0 || device >= c) return fail(TS_ERR_INVALID, "device %d out of range [0, %d)", device, c);
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

// Which MFMA shape serves this index: d = 768 runs the 16x16x32 kernel (kernels_mfma16.h) unless TS_MFMA_SHAPE=32 asks for
// the 32x32x16 one (kernels_mfma.h), which also serves the other widths.
static inline bool use_shape16(const ts_index* ix) {
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

// ---------------------------------------------------------------------------------------------
// functions one translation unit defines for the others
// ---------------------------------------------------------------------------------------------
namespace ts { struct MfmaArgs; }
// index.hip: normalise / convert / pad rows (uploads, query preparation)
int prep_dispatch(int src_dtype, int dst_dtype, bool normalize, const void* src, int64_t src_ld, void* dst, float* f32copy,
                  int64_t ld, int d, int64_t nrows, int64_t rows_total, hipStream_t st);
// search.hip: the streaming scan + its select (also the exact re-run of the matrix path: qlist / qcount on the device)
int scan_search(ts_index* ix, int nq, int k, float* out_scores, int64_t* out_idx, const int* qlist, const int* qcount,
                hipStream_t st, const float* qbuf = nullptr, const unsigned short* qb16 = nullptr);
// search_mfma.hip: the matrix path (threshold sample, full pass, final select, re-run)
int mfma_block_queries(const ts_index* ix, int nq);
int mfma_search(ts_index* ix, int nq, int k, float* out_scores, int64_t* out_idx, hipStream_t st, ts_search_stats* stats,
                const void* qmat, bool in_place);
// launch_mfma16.hip / launch_mfma16_f32.hip / launch_mfma32.hip: one launch of a matrix kernel (a full pass or a sparse level)
int launch_pass_mfma16(int d, int nb, bool full_pass, int variant, int grid, hipStream_t st, const ts::MfmaArgs& a);
int launch_pass_mfma16_f32(int d, int nb, bool full_pass, int grid, hipStream_t st, const ts::MfmaArgs& a);
int launch_pass_mfma32(int d, int groups, bool full_pass, int variant, int grid, hipStream_t st, const ts::MfmaArgs& a);
int launch_pass_mfma32_f32(bool full_pass, int variant, int grid, hipStream_t st, const ts::MfmaArgs& a);
