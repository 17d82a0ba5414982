// Selection kernels: exact top-k of candidate key lists (bitonic sort in LDS, one workgroup per
// list segment), the per-level threshold update of the MFMA path, and the cross-shard merge.
//
// Replaces the reference's full sorts: np.argsort(-scores)[:k] (app_scratchpad.py:130,
// compare_embeddings.py:52), torch.topk(scores, k, sorted=True) (app_showcase_model.py:96) and
// Postgres' top-N heapsort behind ORDER BY ... LIMIT k (streamlit_app.py:282-283).  Order rule:
// score descending, then row ascending (unsigned order of the 64-bit keys, common.h).
#pragma once
#include <math.h>

#include "common.h"

namespace ts {

#define TS_MAX_K_INTERNAL 256

__device__ __forceinline__ double wave_sum_f64_sel(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

struct SelectArgs {
    const u64* in;       // [slot][in_stride] keys (any order; 0 = empty)
    int64_t in_stride;
    int m;               // keys per slot
    int kout;            // keys written per segment
    u64* out;            // intermediate: [slot][out_stride], segment s writes at s * kout
    int64_t out_stride;
    float* out_scores;   // final round (one segment): [query][k_user]
    int64_t* out_idx;
    int k_user;
    int64_t row_offset;
    const int64_t* id_map;  // optional: local row -> global id (subset indexes); replaces row + row_offset
    const int* qlist;    // optional slot -> query id
    const int* qcount;   // optional device-side slot count
};

// grid = (segments, slots); 256 threads; SEG keys of LDS.
template <int SEG>
__global__ void __launch_bounds__(256) select_kernel(SelectArgs a) {
    __shared__ u64 keys[SEG];
    const int slot = blockIdx.y;
    if (a.qcount && slot >= *a.qcount) return;
    const int seg = blockIdx.x;
    const int begin = seg * SEG;
    const int cnt = min(SEG, a.m - begin);
    int P = 1;
    while (P < cnt) P <<= 1;
    if (P < 2) P = 2;
    const u64* src = a.in + (int64_t)slot * a.in_stride + begin;
    for (int i = threadIdx.x; i < P; i += blockDim.x) keys[i] = (i < cnt) ? src[i] : 0ull;
    bitonic_sort_desc(keys, P, threadIdx.x, blockDim.x);
    if (a.out_scores) {
        const int qid = a.qlist ? a.qlist[slot] : slot;
        for (int i = threadIdx.x; i < a.k_user; i += blockDim.x) {
            const u64 key = (i < P) ? keys[i] : 0ull;
            a.out_scores[(int64_t)qid * a.k_user + i] = key ? key_score(key) : -INFINITY;
            a.out_idx[(int64_t)qid * a.k_user + i] = !key ? -1 : a.id_map ? a.id_map[key_row(key)] : (int64_t)key_row(key) + a.row_offset;
        }
    } else {
        u64* dst = a.out + (int64_t)slot * a.out_stride + (int64_t)seg * a.kout;
        for (int i = threadIdx.x; i < a.kout; i += blockDim.x) dst[i] = (i < P) ? keys[i] : 0ull;
    }
}

// MFMA path, after each threshold level: one workgroup per query gathers that query's candidates
// (the writers' private lists plus the shared spill list), sorts them, and
//   sample level: sets thr[q] = score of the kk-th best candidate (-inf if fewer);
//   final level : writes the k results, or appends q to the fall-back list when candidates were
//                 lost (shared list overflow, or more candidates than the sort buffer holds).
struct LevelArgs {
    const u64* priv;     // [nq][nwriters][priv_cap]
    const u32* pcount;   // [nq][nwriters] entries produced per writer (> priv_cap: the rest spilled)
    int nwriters;
    int priv_cap;
    const u64* cand;     // [nq][cap] shared spill lists
    u32* count;          // [nq] appended to the shared list (may exceed cap); reset here
    int cap;
    int kk;              // threshold rank, >= k_user
    float* thr;          // [256]
    int final_level;
    float z_tail;        // sample level: > 0 = also apply the Gaussian-tail estimate mean + z * std of the sample
    float tail_p;        // sample level: > 0 = also apply the exponential-tail fit for this exceedance probability
    float tail_z;        // standard-normal quantile of the sample's 32nd best (z with P(X > z) = 32 / sample rows)
    int min_fill;        // final level: fewer candidates than this = the estimate was too high: exact re-run
    float* out_scores;
    int64_t* out_idx;
    int k_user;
    int64_t row_offset;
    const int64_t* id_map;
    int* fb_list;
    int* fb_count;
    unsigned long long* stat_candidates;  // sum of candidates seen at the final level
};

constexpr int kLevelSortMax = 8192;
constexpr int kLevelThreads = 512;
constexpr int kLevelLds = kLevelSortMax * 8 + (kLevelThreads / 64) * TS_MAX_K_INTERNAL * 8 + 16;

// KR = key registers per lane of the per-wave running top-k (1: kk <= 64, 4: kk <= 256).
template <int KR>
__global__ void __launch_bounds__(kLevelThreads) level_select_kernel(LevelArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u64* keys = (u64*)smem;                                   // kLevelSortMax gathered candidates
    u64* best = (u64*)(smem + kLevelSortMax * 8);             // 8 waves x kk, then sorted
    u32* ctr = (u32*)(smem + kLevelSortMax * 8 + (kLevelThreads / 64) * TS_MAX_K_INTERNAL * 8);
    u32& fill = ctr[0];
    u32& produced = ctr[1];
    u32& base_shared = ctr[2];
    const int q = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if (threadIdx.x == 0) {
        fill = 0;
        produced = 0;
    }
    __syncthreads();
    // gather: private lists ...
    for (int w = threadIdx.x; w < a.nwriters; w += blockDim.x) {
        const u32 made = a.pcount[(int64_t)q * a.nwriters + w];
        if (made == 0) continue;
        const u32 n = min(made, (u32)a.priv_cap);
        const u64* src = a.priv + ((int64_t)q * a.nwriters + w) * a.priv_cap;
        const u32 at = atomicAdd(&fill, n);
        atomicAdd(&produced, n);
        for (u32 e = 0; e < n; ++e)
            if (at + e < (u32)kLevelSortMax) keys[at + e] = src[e];
    }
    // ... and the shared spill list
    const u32 raw = a.count[q];
    const u32 ns = min(raw, (u32)a.cap);
    __syncthreads();
    if (threadIdx.x == 0) {
        base_shared = fill;
        fill += ns;
        produced += raw;
    }
    __syncthreads();
    for (u32 i = threadIdx.x; i < ns; i += blockDim.x)
        if (base_shared + i < (u32)kLevelSortMax) keys[base_shared + i] = a.cand[(int64_t)q * a.cap + i];
    __syncthreads();
    const u32 total = fill;
    const bool lost = raw > (u32)a.cap || total > (u32)kLevelSortMax;
    const int cnt = (int)min(total, (u32)kLevelSortMax);
    const int kk = a.kk;
    // the sample level of the estimated threshold also needs the sample's 32 best for the tail fit
    constexpr int kTailM = 32;
    const bool tail_fit = !a.final_level && a.tail_p > 0.0f;
    const int kl = tail_fit ? max(kk, kTailM) : kk;  // entries kept per wave

    // every wave streams its slice through a running top-kk (a key enters only if it beats the wave's
    // kk-th: about kk ln(n / kk) insertions for n keys), then the waves' lists are merged by one sort
    WaveTopK<KR> tk;
    tk.init();
    for (int i0 = wave * 64; i0 < cnt; i0 += nw * 64) {
        const int i = i0 + lane;
        const u64 key = (i < cnt) ? keys[i] : 0ull;
        u64 m = __ballot(key > tk.thr);
        while (m) {
            const int src = __ffsll((long long)m) - 1;
            m &= m - 1;
            const u64 K = shfl_u64(key, src);
            if (K > tk.thr) tk.insert(K, kl, lane);
        }
    }
    int P = 2;
    while (P < nw * kl) P <<= 1;
    for (int i = threadIdx.x; i < P; i += blockDim.x) best[i] = 0ull;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < KR; ++r) {
        const int s = r * 64 + lane;
        if (s < kl) best[wave * kl + s] = tk.key[r];
    }
    bitonic_sort_desc(best, P, threadIdx.x, blockDim.x);

    if (threadIdx.x == 0) a.count[q] = 0;
    if (!a.final_level) {
        // the kk-th best of a sample is a lower bound of the final kk-th best: a guaranteed threshold (any subset
        // of the candidates still gives a valid bound, so `lost` is harmless here)
        const u64 kth = best[kk - 1];
        float thr = kth ? key_score(kth) : -INFINITY;
        // Heavier-than-Gaussian tails (clusters of near-duplicates: real corpora) make the Gaussian estimate far too
        // low and the full pass would drown in candidates.  Second estimate, from the sample's order statistics 8..32
        // (the top 7 are left out: outliers must not set the slope): an exponential tail fitted to their spacings
        // (E[x_j - x_32] = e * sum_{i=j}^{31} 1/i) and extrapolated to the exceedance probability tail_p, which the
        // host sets for ~2048 expected candidates - a quarter of the buffer, far above k - and only for corpora so large
        // that the guaranteed bound alone would swamp the buffer.  It is used only where the sample SHOWS a heavy tail:
        // its 32nd best lies more than half a standard deviation above where a Gaussian with the sample's mean and
        // variance puts it.  On Gaussian-like scores the fit is therefore never consulted - extrapolated over
        // ln(N / sample) it is noisier than the Gaussian estimate, and at 50M rows its overshoots sent one query per
        // batch to the exact re-run (an 11 ms scan pass per step, measured).  Like the Gaussian estimate it is only an
        // estimate that the final level verifies.
        float thr_tail = -INFINITY, x_m = -INFINITY;
        if (tail_fit && cnt >= 1024 && best[kTailM - 1] != 0ull) {
            x_m = key_score(best[kTailM - 1]);
            float spacing = 0.0f;
            for (int j = 8; j < kTailM; ++j) spacing += key_score(best[j - 1]) - x_m;
            const float e = spacing * (1.0f / 13.95928363f);
            const float ratio = ((float)kTailM / (float)cnt) / a.tail_p;  // how far beyond the sample's 32nd best
            if (ratio > 1.0f && e > 0.0f) thr_tail = x_m + e * __logf(ratio);
        }
        if (a.z_tail > 0.0f && cnt >= 256) {
            // ... and usually far too low for the next level when that level is much bigger.  The scores of one
            // query over the corpus are close to Gaussian (normalised, high-dimensional rows), so the sample's
            // mean + z * std estimates the score that only the wanted number of rows exceed.  NOT a bound: the
            // final level checks that at least min_fill candidates came back and re-runs the query exactly if not.
            double s1 = 0.0, s2 = 0.0;
            for (int i = threadIdx.x; i < cnt; i += blockDim.x) {
                const double v = (double)key_score(keys[i]);
                s1 += v;
                s2 += v * v;
            }
            s1 = wave_sum_f64_sel(s1);
            s2 = wave_sum_f64_sel(s2);
            double* red = (double*)best;  // the sorted list has been read (kth) by every thread that needs it
            __syncthreads();
            if (lane == 0) {
                red[2 * wave] = s1;
                red[2 * wave + 1] = s2;
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                double t1 = 0.0, t2 = 0.0;
                for (int w = 0; w < nw; ++w) {
                    t1 += red[2 * w];
                    t2 += red[2 * w + 1];
                }
                const double mean = t1 / cnt, var = fmax(t2 / cnt - mean * mean, 0.0), sd = sqrt(var);
                thr = fmaxf(thr, (float)(mean + (double)a.z_tail * sd));
                const bool heavy_tail = (double)x_m > mean + ((double)a.tail_z + 0.5) * sd;
                if (heavy_tail) thr = fmaxf(thr, thr_tail);
            }
        }
        if (threadIdx.x == 0) a.thr[q] = thr;
        return;
    }
    if (threadIdx.x == 0) atomicAdd(a.stat_candidates, (unsigned long long)produced);
    if (lost || (int)total < a.min_fill) {
        if (threadIdx.x == 0) a.fb_list[atomicAdd(a.fb_count, 1)] = q;
        return;
    }
    for (int i = threadIdx.x; i < a.k_user; i += blockDim.x) {
        const u64 key = best[i];
        a.out_scores[(int64_t)q * a.k_user + i] = key ? key_score(key) : -INFINITY;
        a.out_idx[(int64_t)q * a.k_user + i] = !key ? -1 : a.id_map ? a.id_map[key_row(key)] : (int64_t)key_row(key) + a.row_offset;
    }
}

// Cross-shard merge: per query, nparts * k_in (score, global id) pairs ->
// best k_out.  Global ids are 64-bit here, so the sort runs on (ordered score, id) pairs.
struct MergeArgs {
    const float* scores;   // part p: scores + p * part_stride (in floats), [nq][k_in]
    const int64_t* idx;    // part p: idx + p * part_stride_idx (in int64), [nq][k_in]
    int64_t part_stride, part_stride_idx;
    int nparts, nq, k_in, k_out;
    float* out_scores;     // [nq][k_out]
    int64_t* out_idx;
};

constexpr int kMergeMax = 4096;

__global__ void __launch_bounds__(256) merge_kernel(MergeArgs a) {
    __shared__ u32 so[kMergeMax];
    __shared__ int64_t si[kMergeMax];
    const int q = blockIdx.x;
    const int m = a.nparts * a.k_in;
    int P = 2;
    while (P < m) P <<= 1;
    for (int i = threadIdx.x; i < P; i += blockDim.x) {
        u32 o = 0;
        int64_t id = INT64_MAX;
        if (i < m) {
            const int part = i / a.k_in, j = i - part * a.k_in;
            const int64_t off = (int64_t)q * a.k_in + j;
            const float s = a.scores[(int64_t)part * a.part_stride + off];
            const int64_t r = a.idx[(int64_t)part * a.part_stride_idx + off];
            if (r >= 0 && s == s) {
                o = ord_f32(s);
                id = r;
            }
        }
        so[i] = o;
        si[i] = id;
    }
    for (int size = 2; size <= P; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int i = threadIdx.x; i < (P >> 1); i += blockDim.x) {
                const int lo = 2 * i - (i & (stride - 1));
                const int hi = lo + stride;
                const bool desc = ((lo & size) == 0);
                const u32 ao = so[lo], bo = so[hi];
                const int64_t ai = si[lo], bi = si[hi];
                const bool a_worse = (ao < bo) || (ao == bo && ai > bi);
                if (a_worse == desc && !(ao == bo && ai == bi)) {
                    so[lo] = bo; si[lo] = bi;
                    so[hi] = ao; si[hi] = ai;
                }
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < a.k_out; i += blockDim.x) {
        const bool ok = i < P && si[i] != INT64_MAX;
        a.out_scores[(int64_t)q * a.k_out + i] = ok ? unord_f32(so[i]) : -INFINITY;
        a.out_idx[(int64_t)q * a.k_out + i] = ok ? si[i] : -1;
    }
}

}  // namespace ts
