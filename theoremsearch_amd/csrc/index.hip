// libtsearch.so - C ABI (include/tsearch.h), part 1: library / device, the index handle (create, view, subset, options,
// uploads, growth, attach, download), ingestion of pgvector text, kernel timing, timers.  Host side only.  No CPU compute
// path exists: without a HIP device every compute entry point fails with TS_ERR_NODEVICE.
#include "host.h"
#include "kernels_prep.h"

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
                    ix->fb_list, ix->fb_count, ix->stat, ix->partial, ix->partial2, ix->res_scores, ix->res_idx, ix->dbg, ix->part, ix->wg_ticks, ix->pair_pos};
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
            if (i != K_MFMA_VARIANT && Knobs::diag_only((Knob)i))
                return fail(TS_ERR_UNSUPPORTED, "%s is an option of the diagnostic build only (make -C theoremsearch_amd/csrc diag)", name);
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

extern "C" int ts_copy_device(int device, void* dst, const void* src, int64_t bytes, void* stream) {
    if (bytes < 0 || ((!dst || !src) && bytes > 0)) return fail(TS_ERR_INVALID, "bad argument");
    if (bytes == 0) return TS_OK;
    TS_TRY(check_device(device));
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
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

int prep_dispatch(int src_dtype, int dst_dtype, bool normalize, const void* src, int64_t src_ld, void* dst,
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
    TS_TRY(ensure_stage(ix, (size_t)nrows * src_row, src_row));
    const int64_t rows_per = std::max<int64_t>(1, (int64_t)(ix->stage_bytes / src_row));
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
    hipStream_t own;
    StreamScope scope;
    TS_TRY(enter_stream(ix, nullptr, &own, &scope));
    TS_TRY(ensure_stage(ix, (size_t)nrows * row_bytes, row_bytes));
    const int64_t rows_per = std::max<int64_t>(1, (int64_t)(ix->stage_bytes / row_bytes));
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
