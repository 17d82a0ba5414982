// libtsearch.so - C ABI (include/tsearch.h), part 2: the search entry points (ts_search*, ts_rank_of, ts_count_above,
// ts_scores), the streaming scan path and its selects.  The matrix path lives in search_mfma.hip.
#include "host.h"
#include "kernels_scan.h"
#include "kernels_select.h"

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
    TS_TRY(need(&ix->stat, (size_t)kQBlock * 4, true));
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
int scan_search(ts_index* ix, int nq, int k, float* out_scores, int64_t* out_idx, const int* qlist, const int* qcount,
                hipStream_t st, const float* qbuf, const unsigned short* qb16) {
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
        if (ix->rebalance_pending && ix->rebalance_in_rerun && ix->rebalance_grid <= 256) {
            a.part = ix->part;
            a.wg_ticks = ix->wg_ticks;
            a.part_g = ix->rebalance_grid;
            const int b = ix->knobs.get(K_MFMA_BALANCE, 1);      // TS_MFMA_BALANCE = n > 1: gain n / 10 (default 0.7)
            a.part_gain = (b >= 2 && b <= 10) ? 0.1f * (float)b : 0.7f;
            ix->rebalance_pending = false;
        }
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
    if (!q_on_device) TS_TRY(ensure_stage(ix, (size_t)std::min(nq, kQBlock) * ix->d * q_elem, 0));   // one block of queries at a time

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
        std::vector<u32> cands_q;
        const int last_nb = nq - (nq - 1) / block * block;     // queries of the last block: what the counters describe
        if (stats && use == TS_ALGO_MFMA) {
            cands_q.resize((size_t)last_nb);
            HIP_TRY(hipMemcpyAsync(&fb, ix->fb_count, 4, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(cands_q.data(), ix->stat, (size_t)last_nb * 4, hipMemcpyDeviceToHost, st));
        }
        HIP_TRY(hipStreamSynchronize(st));
        if (stats && use == TS_ALGO_MFMA) {
            stats->fallback_queries = fb;
            int64_t cands = 0;
            for (u32 c_ : cands_q) cands += c_;
            stats->candidates = cands;
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
    if (!q_on_device) TS_TRY(ensure_stage(ix, (size_t)std::min(nq, kQBlock) * ix->d * q_elem, 0));   // one block of queries at a time
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
    if (!q_on_device) TS_TRY(ensure_stage(ix, (size_t)std::min(nq, kQBlock) * ix->d * q_elem, 0));   // one block of queries at a time
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

