// The matrix path of a search: planning of the threshold levels, the dense threshold sample and its
// select, the full pass (launch_mfma*.hip hold the kernel instantiations), the final select, the exact re-run.
#include "host.h"
#include "kernels_mfma.h"
#include "kernels_mfma16.h"
#include "kernels_mfma_f32.h"
#include "kernels_sample.h"
#include "kernels_select.h"

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

// Queries one launch of the MFMA kernel serves for this index / batch: d = 768 holds two query groups per wave
// (256 queries; one group = half the matrix work when the batch is <= 128), d = 1024 one (128 queries).
// bf16 x 1024 (the production table, rds_schema.sql:50-56: vector(1024)) holds 3 blocks of 16 queries per wave: 192 per
// workgroup.  A batch of 193 .. 256 queries runs as ONE launch of workgroup PAIRS (MfmaArgs::pair): both workgroups of a pair
// walk the same tiles with 128 queries each, two blocks per wave, so the corpus crosses HBM once for the whole batch (the
// pair's second read of a tile is served by the XCD's L2 / the memory-side cache) instead of once per 128 queries.
constexpr int kMfmaMaxGrid = 2048;   // workgroups of one pass (TS_MFMA_GRID is clamped to it; the pairs' position words are sized by it)
static int mfma_grid(const ts_index* ix) { return std::max(1, std::min(ix->knobs.get(K_MFMA_GRID, ix->cu_count), kMfmaMaxGrid)); }
static bool mfma_pairs(const ts_index* ix, int nq) {
    return ix->dtype == TS_BF16 && ix->d == 1024 && nq > 192 && use_shape16(ix) && two_level_search(ix) &&
           ix->knobs.get(K_MFMA_PAIR, 1) != 0 && mfma_grid(ix) % 16 == 0;
}

int mfma_block_queries(const ts_index* ix, int nq) {
    if (mfma_pairs(ix, nq)) return 256;
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
int mfma_search(ts_index* ix, int nq, int k, float* out_scores, int64_t* out_idx, hipStream_t st, ts_search_stats* stats,
                       const void* qmat, bool in_place) {
    // threshold rank: the k-th best of a sample is already a valid lower bound of the final k-th best; private
    // lists + spill absorb the run-to-run spread of the candidate count, so no safety margin in the rank
    const int kk = std::max(k, ix->knobs.get(K_MFMA_MIN_RANK, 1));
    const int variant = ix->knobs.get(K_MFMA_VARIANT, 0);
    const bool shape16 = use_shape16(ix);
    const int groups = shape16 ? 0 : mfma_block_queries(ix, nq) / 128;
    const bool pair = mfma_pairs(ix, nq);
    const int nb16 = pair ? 2 : (shape16 ? mfma_block_queries(ix, nq) / 64 : 0);
    if (!ix->attr_done) {
        HIP_TRY(hipFuncSetAttribute((const void*)level_select_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, kLevelLds));
        HIP_TRY(hipFuncSetAttribute((const void*)level_select_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, kLevelLds));
        ix->attr_done = true;
    }
    // The sparsest level (every score a candidate, at most kLevelSortMax rows) runs as a dense score matrix + one select per
    // query (kernels_sample.h) instead of the full-pass kernel over the sample + a gather from lane-private lists; the
    // latter stays selectable (TS_MFMA_SAMPLE=0) as the A/B partner and serves the thresholded sparse levels of the
    // guaranteed chain (TS_MFMA_STAT=0).
    const bool dense_sample = ix->knobs.get(K_MFMA_SAMPLE, 1) != 0;
    if (dense_sample && !ix->sample) HIP_TRY(hipMalloc((void**)&ix->sample, (size_t)kMfmaQ * kLevelSortMax * 4));
    const int grid = mfma_grid(ix);
    const int wgs = pair ? grid / 2 : grid;                 // tile ranges of the full pass: one per workgroup, or one per pair
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
    const bool balance = shape16 && ix->knobs.get(K_MFMA_BALANCE, 1) != 0 && wgs >= 8 && wgs <= 256 && lv.back().stride == 1 &&
                         lv.back().run == 1 && full_tiles >= 32 * (int64_t)wgs && (variant == 0 || variant == 3);
    if (balance && (ix->part_g != wgs || ix->part_ntiles != full_tiles)) {
        if (ix->part_g != wgs) {
            if (ix->part) HIP_TRY(hipFree(ix->part));
            if (ix->wg_ticks) HIP_TRY(hipFree(ix->wg_ticks));
            ix->part = nullptr; ix->wg_ticks = nullptr; ix->part_g = 0; ix->part_ntiles = -1;
            HIP_TRY(hipMalloc((void**)&ix->part, (size_t)(wgs + 1) * 8));
            HIP_TRY(hipMalloc((void**)&ix->wg_ticks, (size_t)wgs * 4));
            ix->part_g = wgs;
        }
        std::vector<int64_t> equal((size_t)wgs + 1);
        for (int w = 0; w <= wgs; ++w) equal[w] = full_tiles * (int64_t)w / wgs;
        HIP_TRY(hipMemcpyAsync(ix->part, equal.data(), equal.size() * 8, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemsetAsync(ix->wg_ticks, 0, (size_t)wgs * 4, st));
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
            // 32 rows per workgroup and one 64-query chunk: 512 workgroups of 50 KB LDS at 4,096 rows x 256 queries, two to
            // a CU (64-row workgroups serving two chunks each measured the same: 15.2 / 25.0 us against 14.9 / 24.3 us at
            // 4,096 / 8,192 rows - the launch is latency, not work)
            const bool f32 = ix->dtype == TS_F32;
            const int nchunks = (nq + 63) / 64;
            const int wg_rows = 32;
            // the previous search's full pass left its workgroups' times: one extra workgroup of this launch moves the
            // tile boundaries before this search's pass reads them
            if (ix->rebalance_pending && balance && ix->rebalance_grid == wgs && ix->rebalance_grid <= 256) {
                sa.part = ix->part;
                sa.wg_ticks = ix->wg_ticks;
                sa.part_g = ix->rebalance_grid;
                const int b = ix->knobs.get(K_MFMA_BALANCE, 1);      // TS_MFMA_BALANCE = n > 1: gain n / 10 (default 0.7)
                sa.part_gain = (b >= 2 && b <= 10) ? 0.1f * (float)b : 0.7f;
            }
            ix->rebalance_pending = false;
            const dim3 sgrid((unsigned)(sa.row_stride / wg_rows), (unsigned)(nchunks + (sa.part ? 1 : 0)));
            const int slds = sample_lds_bytes(wg_rows, (int)(ix->ld * ix->elem()));
            constexpr int kSampleLdsMax = 144 * 1024;   // dynamic part; the kernel also has a few hundred static bytes (rebalance_tiles)
            if (slds > kSampleLdsMax) return fail(TS_ERR_INTERNAL, "threshold sample: rows of %lld bytes do not fit the LDS", (long long)(ix->ld * ix->elem()));
            {
                static std::atomic<unsigned long long> sample_attr{0};
                int dev = 0;
                HIP_TRY(hipGetDevice(&dev));
                const unsigned long long bit = 1ull << (dev & 63);
                if (!(sample_attr.load(std::memory_order_acquire) & bit)) {
                    HIP_TRY(hipFuncSetAttribute((const void*)sample_scores_kernel<true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, kSampleLdsMax));
                    HIP_TRY(hipFuncSetAttribute((const void*)sample_scores_kernel<false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, kSampleLdsMax));
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
            // where the select cuts first: ~2 kl of the sample's live rows above it on Gaussian-like scores
            const int kl = tail_p > 0.0f ? std::max(kk, 32) : kk;
            const double live = sample_rows * (double)pop / (double)std::max<int64_t>(ix->n, 1);
            const float z_sel = (float)normal_tail_z(std::min(0.25, 2.0 * kl / std::max(live, 1.0)));
            sample_select_fast_kernel<kSelThreads><<<nq, kSelThreads, 0, st>>>(l, ix->sample, sa.row_stride, z_sel);
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
        a.part = (balance && full_pass) ? ix->part : nullptr;
        a.wg_ticks = (balance && full_pass) ? ix->wg_ticks : nullptr;
        a.pair = (pair && full_pass) ? (ix->knobs.get(K_MFMA_PAIR, 2) == 1 ? 1 : 2) : 0;      // 2: the k-split form (TS_MFMA_PAIR=1: two blocks per wave over the whole row)
        a.pair_pos = nullptr;
        a.pair_lag = 0;
        if (a.pair && ix->knobs.get(K_MFMA_PAIR_LAG, 1) > 0) {
            if (!ix->pair_pos) {
                // one word per workgroup of the pass (index 2 * pair + half < grid): sized for the largest grid the option allows
                HIP_TRY(hipMalloc((void**)&ix->pair_pos, kMfmaMaxGrid * sizeof(unsigned)));
                HIP_TRY(hipMemsetAsync(ix->pair_pos, 0, kMfmaMaxGrid * sizeof(unsigned), st));
            }
            a.pair_pos = ix->pair_pos;
            a.pair_lag = ix->knobs.get(K_MFMA_PAIR_LAG, 1);
        }
        a.dbg = nullptr;
#ifdef TS_DIAG
        if (variant >= 3) {
            if (!ix->dbg) HIP_TRY(hipMalloc((void**)&ix->dbg, 2048 * 4 * 4 * 8));
            a.dbg = ix->dbg;
        }
#endif
        hipEvent_t stop = full_pass ? prof_begin(ix, st, ix->n) : nullptr;  // only the full pass is bracketed
        int rc;
        if (ix->dtype == TS_F32 && shape16) rc = launch_pass_mfma16_f32(ix->d, nb16, full_pass, grid, st, a);
        else if (ix->dtype == TS_F32) rc = launch_pass_mfma32_f32(full_pass, variant, grid, st, a);
        else if (shape16) rc = launch_pass_mfma16(ix->d, nb16, full_pass, variant, grid, st, a);
        else rc = launch_pass_mfma32(ix->d, groups, full_pass, variant, grid, st, a);
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
        l.stat_q = ix->stat;
        l.nq = nq;
        if (kk <= 64) level_select_kernel<1><<<nq, kLevelThreads, kLevelLds, st>>>(l);
        else level_select_kernel<4><<<nq, kLevelThreads, kLevelLds, st>>>(l);
        HIP_TRY(hipGetLastError());
    }
    // the pass just finished left its workgroups' times: the next search's sample launch moves the tile boundaries (without
    // a dense sample: block 0 of the re-run launch below)
    ix->rebalance_pending = balance;
    ix->rebalance_grid = wgs;
    ix->rebalance_in_rerun = !(dense_sample && lv.size() == 2);
    // exact fall-back for queries that lost candidates (device-side count; one empty launch when 0)
    if (!in_place) TS_TRY(scan_search(ix, nq, k, out_scores, out_idx, ix->fb_list, ix->fb_count, st));
    else if (ix->dtype == TS_F32) TS_TRY(scan_search(ix, nq, k, out_scores, out_idx, ix->fb_list, ix->fb_count, st, (const float*)qmat));
    else TS_TRY(scan_search(ix, nq, k, out_scores, out_idx, ix->fb_list, ix->fb_count, st, nullptr, (const unsigned short*)qmat));
    if (stats) stats->levels = (int)lv.size();
    return TS_OK;
}
