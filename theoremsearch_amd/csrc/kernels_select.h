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
    u32* stat_q;         // [nq] candidates seen at the final level (the host adds them up when statistics are asked for:
                         // one atomic per query on ONE counter was 256 serialised L2 operations at the end of every search)
    int nq;
};

constexpr int kLevelSortMax = 8192;
constexpr int kLevelThreads = 512;
constexpr int kLevelSmall = 2048;      // keys the direct sort takes (after the histogram cut)
constexpr int kLevelBins = 1024;
// LDS: gathered keys | short list (sorted) | per-wave lists of the streaming path | histogram | counters
constexpr int kLevelLds = kLevelSortMax * 8 + kLevelSmall * 8 + (kLevelThreads / 64) * TS_MAX_K_INTERNAL * 8 + kLevelBins * 4 + 64;

// The kl best of `cnt` keys in LDS, sorted descending (shared by level_select_kernel and select_hist_kernel); returns the
// sorted list (at least kl entries, zero = empty).  LDS areas: small[kLevelSmall], wlists[8 x TS_MAX_K_INTERNAL],
// hist[kLevelBins], ctr[16] (ctr[3] = nsmall must be 0 on entry, hist zeroed).  `want_stats`: mean / sd of the scores are
// needed by the caller even when the short path does not need them.
template <int KR>
__device__ __forceinline__ u64* lds_select_top(u64* keys, int cnt, int kl, u64* small, u64* wlists, u32* hist, u32* ctr,
                                               bool want_stats_in, double& mean, double& sd) {
    u32& nsmall = ctr[3];
    int& cut_bin = *(int*)&ctr[4];
    double* red = (double*)&ctr[8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const bool want_stats = cnt > kLevelSmall || want_stats_in;
    mean = 0.0;
    sd = 0.0;
    if (want_stats) {
        double s1 = 0.0, s2 = 0.0;
        for (int i = threadIdx.x; i < cnt; i += blockDim.x) {
            const double v = (double)key_score(keys[i]);
            s1 += v;
            s2 += v * v;
        }
        s1 = wave_sum_f64_sel(s1);
        s2 = wave_sum_f64_sel(s2);
        double* part = (double*)wlists;   // free until the streaming path
        if (lane == 0) {
            part[2 * wave] = s1;
            part[2 * wave + 1] = s2;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            double t1 = 0.0, t2 = 0.0;
            for (int w = 0; w < nw; ++w) {
                t1 += part[2 * w];
                t2 += part[2 * w + 1];
            }
            const double m = t1 / cnt;
            red[0] = m;
            red[1] = sqrt(fmax(t2 / cnt - m * m, 0.0));
        }
        __syncthreads();
        mean = red[0];
        sd = red[1];
    }

    u64* best = small;
    bool streamed = false;
    if (cnt <= kLevelSmall) {
        for (int i = threadIdx.x; i < cnt; i += blockDim.x) small[i] = keys[i];
        if (threadIdx.x == 0) nsmall = (u32)cnt;
    } else {
        // histogram cut: bins of (12 / 1024) sd over [mean - 4 sd, mean + 8 sd], bin 1023 = everything above
        const float lo = (float)(mean - 4.0 * sd);
        const float inv = (sd > 0.0) ? (float)((double)kLevelBins / (12.0 * sd)) : 0.0f;
        auto bin_of = [&](u64 key) -> int {
            const float t = (key_score(key) - lo) * inv;
            return (t >= (float)(kLevelBins - 1)) ? kLevelBins - 1 : (t > 0.0f ? (int)t : 0);
        };
        for (int i = threadIdx.x; i < cnt; i += blockDim.x) atomicAdd(&hist[bin_of(keys[i])], 1u);
        __syncthreads();
        if (wave == 0) {
            // one wave scans the bins from the top: lane l owns bins [1023 - 16 l - 15, 1023 - 16 l]
            u32 mine = 0;
            for (int j = 0; j < 16; ++j) mine += hist[kLevelBins - 1 - 16 * lane - j];
            u32 incl = mine;
            for (int off = 1; off < 64; off <<= 1) {
                const u32 up = __shfl_up((int)incl, off, 64);
                if (lane >= off) incl += up;
            }
            const u64 reach = __ballot(incl >= (u32)kl);
            int cb = 0;
            if (reach) {
                const int L = __ffsll((long long)reach) - 1;          // first lane group that reaches kl
                u32 run = (u32)__shfl((int)(incl - mine), L, 64);
                if (lane == L) {
                    int j = 0;
                    for (; j < 16; ++j) {
                        run += hist[kLevelBins - 1 - 16 * L - j];
                        if (run >= (u32)kl) break;
                    }
                    cut_bin = kLevelBins - 1 - 16 * L - min(j, 15);
                }
            } else if (lane == 0) {
                cut_bin = 0;
            }
            (void)cb;
        }
        __syncthreads();
        const int cb = cut_bin;
        for (int i = threadIdx.x; i < cnt; i += blockDim.x) {
            const u64 key = keys[i];
            if (bin_of(key) >= cb) {
                const u32 at = atomicAdd(&nsmall, 1u);
                if (at < (u32)kLevelSmall) small[at] = key;
            }
        }
    }
    __syncthreads();
    const u32 nsm = nsmall;
    if (nsm <= (u32)kLevelSmall) {
        int P = 2;
        while (P < (int)nsm) P <<= 1;
        if (P < kl) {
            P = 2;
            while (P < kl) P <<= 1;
        }
        for (int i = (int)nsm + threadIdx.x; i < P; i += blockDim.x) small[i] = 0ull;
        bitonic_sort_desc(small, P, threadIdx.x, blockDim.x);
    } else {
        // streaming path over all gathered keys
        streamed = true;
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
        __syncthreads();
        for (int i = threadIdx.x; i < P; i += blockDim.x) wlists[i] = 0ull;
        __syncthreads();
#pragma unroll
        for (int r = 0; r < KR; ++r) {
            const int s = r * 64 + lane;
            if (s < kl) wlists[wave * kl + s] = tk.key[r];
        }
        bitonic_sort_desc(wlists, P, threadIdx.x, blockDim.x);
        best = wlists;
    }
    (void)streamed;
    return best;

}

// KR = key registers per lane of the per-wave running top-k of the streaming path (1: kk <= 64, 4: kk <= 256).
//
// Selection of the `kl` best of up to 8192 gathered keys, two ways:
//   * short path (the usual one): at most kLevelSmall keys -> one bitonic sort of them; more keys -> a 1024-bin histogram
//     of the scores over [mean - 4 sd, mean + 8 sd] (LDS atomics), a scan of the bins from the top down to the bin that
//     holds the kl-th best key, and the sort of just the keys in the bins above it (plus that bin);
//   * streaming path (many equal scores in the cut bin: duplicates, saturated scores): every wave runs its slice through
//     a running top-kl (WaveTopK) and the eight lists are merged by one sort - what this kernel always did before.
// Both give the same `best[0 .. kl)`: the kl largest keys in descending order.
// Pass threshold of the next level from the kl best keys of a sample (`best`, sorted descending; cnt keys were sampled;
// mean / sd of all sampled scores).  Shared by level_select_kernel (candidate lists of a sample level) and
// sample_select_kernel (the dense sample matrix).
__device__ __forceinline__ float level_threshold(const LevelArgs& a, const u64* best, int cnt, int kk, bool tail_fit, double mean,
                                                 double sd) {
    constexpr int kTailM = 32;
    // the kk-th best of a sample is a lower bound of the final kk-th best: a guaranteed threshold (any subset
    // of the candidates still gives a valid bound, so lost candidates are harmless here)
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
        thr = fmaxf(thr, (float)(mean + (double)a.z_tail * sd));
        const bool heavy_tail = (double)x_m > mean + ((double)a.tail_z + 0.5) * sd;
        if (heavy_tail) thr = fmaxf(thr, thr_tail);
    }
    return thr;
}

// Sum over the 64 lanes of a wave on the DPP path (row / quad permutes inside the vector pipe, a few cycles each) instead of
// six dependent ds_bpermute round trips through the LDS crossbar: quad xor 1, xor 2, half-row mirror, row mirror leave every
// lane with its row's sum; row_bcast15 / row_bcast31 fold the rows into lane 63; the result is read from there (uniform).
#define TS_DPP(v, ctrl, rmask) ((u32)__builtin_amdgcn_update_dpp(0, (int)(v), (ctrl), (rmask), 0xf, false))
#define TS_DPP_QUAD_XOR1 0xB1      /* quad_perm [1,0,3,2] */
#define TS_DPP_QUAD_XOR2 0x4E      /* quad_perm [2,3,0,1] */
#define TS_DPP_HALF_MIRROR 0x141
#define TS_DPP_ROW_MIRROR 0x140
#define TS_DPP_BCAST15 0x142
#define TS_DPP_BCAST31 0x143
__device__ __forceinline__ double wave_total_f64(double v) {
#define TS_DPP_F64_STEP(ctrl, rmask)                                                                          \
    {                                                                                                          \
        const u64 b = (u64)__double_as_longlong(v);                                                            \
        const u32 lo = TS_DPP((u32)b, ctrl, rmask), hi = TS_DPP((u32)(b >> 32), ctrl, rmask);                  \
        v += __longlong_as_double((long long)(((u64)hi << 32) | lo));                                          \
    }
    TS_DPP_F64_STEP(TS_DPP_QUAD_XOR1, 0xf)
    TS_DPP_F64_STEP(TS_DPP_QUAD_XOR2, 0xf)
    TS_DPP_F64_STEP(TS_DPP_HALF_MIRROR, 0xf)
    TS_DPP_F64_STEP(TS_DPP_ROW_MIRROR, 0xf)
    TS_DPP_F64_STEP(TS_DPP_BCAST15, 0xa)      // rows 1 and 3 += the row before them (disabled rows add 0.0: old = 0)
    TS_DPP_F64_STEP(TS_DPP_BCAST31, 0xc)      // rows 2 and 3 += rows 0 + 1
#undef TS_DPP_F64_STEP
    const u64 b = (u64)__double_as_longlong(v);
    const u32 lo = (u32)__builtin_amdgcn_readlane((int)(u32)b, 63), hi = (u32)__builtin_amdgcn_readlane((int)(u32)(b >> 32), 63);
    return __longlong_as_double((long long)(((u64)hi << 32) | lo));
}

// Descending bitonic sort of 64 R values held by ONE wave, element e = 64 r + lane in v[r]: exchanges between lanes are
// shuffles, exchanges at distances of 64 and more stay inside the lane - no LDS, no barrier.  What the two short selects
// below run on: the kernels that bracket the full pass are latency, and a workgroup-wide sort of a hundred keys is 28
// barriers of it.
template <typename T> __device__ __forceinline__ T wave_xor(T v, int mask);
template <> __device__ __forceinline__ u32 wave_xor<u32>(u32 v, int mask) { return (u32)__shfl_xor((int)v, mask, 64); }
template <> __device__ __forceinline__ u64 wave_xor<u64>(u64 v, int mask) {
    const u32 lo = (u32)__shfl_xor((int)(u32)v, mask, 64), hi = (u32)__shfl_xor((int)(u32)(v >> 32), mask, 64);
    return ((u64)hi << 32) | lo;
}
template <typename T, int R>
__device__ __forceinline__ void wave_sort_desc(T (&v)[R], int lane) {
#pragma unroll
    for (int size = 2; size <= 64 * R; size <<= 1) {
#pragma unroll
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            if (stride >= 64) {
                const int rs = stride >> 6;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    if (r & rs) continue;
                    const bool desc = (((r << 6) & size) == 0);      // bit `size` of the element index lies in r here
                    const T a = v[r], b = v[r | rs];
                    const bool swap = (a < b) == desc;
                    v[r] = swap ? b : a;
                    v[r | rs] = swap ? a : b;
                }
            } else {
                const bool low = (lane & stride) == 0;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const bool desc = ((((r << 6) | lane) & size) == 0);
                    const T mine = v[r], other = wave_xor<T>(mine, stride);
                    const T hi = mine > other ? mine : other, lo = mine > other ? other : mine;
                    v[r] = (low == desc) ? hi : lo;
                }
            }
        }
    }
}

constexpr int kFinalFast = 128;     // candidates the one-wave form of the final select takes (2 keys per lane)

template <int KR>
__global__ void __launch_bounds__(kLevelThreads) level_select_kernel(LevelArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u64* keys = (u64*)smem;                                   // kLevelSortMax gathered candidates
    u64* small = (u64*)(smem + kLevelSortMax * 8);            // short list, then sorted
    u64* wlists = small + kLevelSmall;                        // streaming path: 8 waves x kl
    u32* hist = (u32*)(wlists + (kLevelThreads / 64) * TS_MAX_K_INTERNAL);
    u32* ctr = hist + kLevelBins;
    u32& fill = ctr[0];
    u32& produced = ctr[1];
    u32& base_shared = ctr[2];
    u32& nsmall = ctr[3];                                     // (ctr[4], ctr[8..11]: lds_select_top's cut bin, mean, sd)
    const int q = blockIdx.x;
    // The usual final level behind the 16x16 full pass: a few dozen candidates in the query's shared list and nothing else
    // (the threshold estimate aims at 6 k, at least 64).  One wave takes them straight into registers, sorts them there and
    // writes the answer; the other waves leave.  Two dependent round trips (count, keys) and no barrier, instead of the
    // gather / histogram / workgroup sort below.
    if (a.final_level && a.nwriters == 0 && a.k_user <= kFinalFast) {
        const u32 raw0 = a.count[q];                           // uniform over the workgroup
        if (raw0 <= (u32)kFinalFast && raw0 <= (u32)a.cap) {
            if (threadIdx.x >= 64) return;
            const int lane = threadIdx.x;
            u64 v[2];
#pragma unroll
            for (int r = 0; r < 2; ++r) v[r] = ((u32)(64 * r + lane) < raw0) ? a.cand[(int64_t)q * a.cap + 64 * r + lane] : 0ull;
            if (lane == 0) {
                a.count[q] = 0;
                a.stat_q[q] = raw0;
            }
            if ((int)raw0 < a.min_fill) {
                if (lane == 0) a.fb_list[atomicAdd(a.fb_count, 1)] = q;
                return;
            }
            wave_sort_desc<u64, 2>(v, lane);
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int i = 64 * r + lane;
                if (i < a.k_user) {
                    const u64 key = v[r];
                    a.out_scores[(int64_t)q * a.k_user + i] = key ? key_score(key) : -INFINITY;
                    a.out_idx[(int64_t)q * a.k_user + i] = !key ? -1 : a.id_map ? a.id_map[key_row(key)] : (int64_t)key_row(key) + a.row_offset;
                }
            }
            return;
        }
    }
    if (threadIdx.x == 0) {
        fill = 0;
        produced = 0;
        nsmall = 0;
    }
    for (int i = threadIdx.x; i < kLevelBins; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    // gather: private lists (a writer's entries are loaded together: one round trip, not one per entry) ...
    for (int w = threadIdx.x; w < a.nwriters; w += blockDim.x) {
        const u32 made = a.pcount[(int64_t)q * a.nwriters + w];
        if (made == 0) continue;
        const u32 n = min(made, (u32)a.priv_cap);
        const u64* src = a.priv + ((int64_t)q * a.nwriters + w) * a.priv_cap;
        const u32 at = atomicAdd(&fill, n);
        atomicAdd(&produced, n);
        u64 v[32];
#pragma unroll
        for (int e = 0; e < 32; ++e) v[e] = ((u32)e < n) ? src[e] : 0ull;
#pragma unroll
        for (int e = 0; e < 32; ++e)
            if ((u32)e < n && at + e < (u32)kLevelSortMax) keys[at + e] = v[e];
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
    const int kl = tail_fit ? max(kk, kTailM) : kk;  // entries wanted, sorted

    // mean / sd of the gathered scores (the histogram's range, and the Gaussian-tail estimate of the sample level) and the
    // kl best keys, sorted
    double mean = 0.0, sd = 0.0;
    u64* best = lds_select_top<KR>(keys, cnt, kl, small, wlists, hist, ctr, !a.final_level && a.z_tail > 0.0f && cnt >= 256, mean, sd);

    if (threadIdx.x == 0) a.count[q] = 0;
    if (!a.final_level) {
        const float thr = level_threshold(a, best, cnt, kk, tail_fit, mean, sd);
        if (threadIdx.x == 0) a.thr[q] = thr;
        return;
    }
    if (threadIdx.x == 0) a.stat_q[q] = produced;
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

// Sample level of the batched search, dense form (kernels_sample.h): one workgroup per query reads that query's row of the
// sample score matrix (coalesced; -inf = no such row / filtered out, NaN never counted) and sets thr[q] with the estimates
// level_select_kernel applies to a sample level (level_threshold) - without the gather from a thousand lane-private lists.
// Shaped for latency (round 4; round 3's form keyed the scores by position and ran them through lds_select_top: 16 us): the threshold needs the SCORES of the kl best sample rows, their mean and their
// standard deviation - not which rows they were.  One workgroup of 256 threads per query holds the whole row of the sample
// matrix in registers (8 x 16-byte loads per thread, all in flight at once), reduces count / sum / sum of squares, cuts at
// mean + z sd (`z_sel`: the host's guess for ~2 kl survivors on Gaussian-like scores), and one wave sorts the survivors in
// registers (wave_sort_desc).  A cut that leaves fewer than kl or more than kSelCap survivors is moved (three tries), then
// found exactly by bisection on the ordered 32-bit scores - piles of equal scores included.
// test_dense_threshold_sample_and_the_list_form_give_the_same_answers holds the answers against the list form's.
constexpr int kSelCap = 1024;        // survivors the sort takes (kl <= 256)

// 256 threads per query (one wave per query measured 20 us against 15: the hundreds of compare / count instructions are
// issue time, and four SIMDs share it).
constexpr int kSelThreads = 256;
template <int THREADS>   // = kSelThreads (a template so that every translation unit may include this header)
__global__ void __launch_bounds__(THREADS) sample_select_fast_kernel(LevelArgs a, const float* scores, int row_stride, float z_sel) {
    constexpr int NW = THREADS / 64;
    constexpr int J4 = kLevelSortMax / (4 * THREADS);         // 8 float4 per thread cover 8192 positions
    __shared__ double red[2 * NW];
    __shared__ u32 wcount[2][NW];
    __shared__ __attribute__((aligned(16))) u32 surv[kSelCap];
    __shared__ __attribute__((aligned(16))) u64 best[kSelCap];
    __shared__ u32 nsurv;
    const int q = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) {
        a.count[q] = 0;                                       // the shared list of the full pass starts empty
        nsurv = 0;
    }
    const float4* src = (const float4*)(scores + (int64_t)q * row_stride);
    const int n4 = row_stride >> 2;                           // row_stride is a multiple of 64; positions past the sample hold -inf
    float v[4 * J4];
#pragma unroll
    for (int j = 0; j < J4; ++j) {
        const int i = j * THREADS + threadIdx.x;
        const float4 t = (i < n4) ? src[i] : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        v[4 * j] = t.x; v[4 * j + 1] = t.y; v[4 * j + 2] = t.z; v[4 * j + 3] = t.w;
    }
    // live scores: not NaN, not -inf (no such row / filtered out).  From here on a score is its ordered 32-bit image
    // (0 = dead; every live score's image is above 0).
    double s1 = 0.0, s2 = 0.0;
    u32 nl = 0;
    u32 o[4 * J4];
#pragma unroll
    for (int e = 0; e < 4 * J4; ++e) {
        const bool live = v[e] == v[e] && v[e] > -INFINITY;
        const double d = live ? (double)v[e] : 0.0;
        s1 += d;
        s2 += d * d;
        nl += (u32)__popcll(__ballot(live));                  // wave-uniform: this wave's live count so far
        o[e] = live ? ord_f32(v[e]) : 0u;
    }
    s1 = wave_total_f64(s1);
    s2 = wave_total_f64(s2);
    if (lane == 0) { red[2 * wave] = s1; red[2 * wave + 1] = s2; wcount[0][wave] = nl; }
    __syncthreads();
    double t1 = 0.0, t2 = 0.0;
    int cnt = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) { t1 += red[2 * w]; t2 += red[2 * w + 1]; cnt += (int)wcount[0][w]; }
    const double mean = cnt > 0 ? t1 / cnt : 0.0;
    const double sd = cnt > 0 ? sqrt(fmax(t2 / cnt - mean * mean, 0.0)) : 0.0;
    const int kk = a.kk;
    constexpr int kTailM = 32;
    const bool tail_fit = a.tail_p > 0.0f;
    const int kl = min(tail_fit ? max(kk, kTailM) : kk, TS_MAX_K_INTERNAL);

    // survivors of a cut (>= cut, cut >= 1), counted over the workgroup (uniform result); the two count buffers alternate,
    // so one barrier per call is enough
    int flip = 1;
    auto count_ge = [&](u32 cut) __attribute__((always_inline)) -> u32 {
        u32 c = 0;
#pragma unroll
        for (int e = 0; e < 4 * J4; ++e) c += (u32)__popcll(__ballot(o[e] >= cut));
        if (lane == 0) wcount[flip][wave] = c;
        __syncthreads();
        u32 tot = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) tot += wcount[flip][w];
        flip ^= 1;
        return tot;
    };
    const u32 want = (u32)min(kl, cnt);                       // fewer live rows than kl: all of them
    u32 cut = 1u, c = 0;
    bool found = want == 0;
    if (!found && sd > 0.0) {
        float z = z_sel;
        for (int t = 0; t < 3 && !found; ++t) {
            cut = max(ord_f32((float)(mean + (double)z * sd)), 1u);
            c = count_ge(cut);
            if (c < want) z -= 0.75f;
            else if (c > (u32)kSelCap) z += 0.75f;
            else found = true;
        }
    }
    if (!found) {
        // exact: the largest cut that still leaves `want` survivors = the want-th best score itself; everything above it
        // (fewer than want <= 256 values) plus copies of it are the want best scores
        u32 lo = 1u, hi = 0xFFFFFFFFu;                        // count_ge(lo) >= want always; hi may fail
        if (count_ge(hi) >= want) lo = hi;
        while (lo < hi) {                                     // invariant: count_ge(lo) >= want, count_ge(hi + 1) < want
            const u32 mid = lo + (hi - lo) / 2u + ((hi - lo) & 1u);
            if (count_ge(mid) >= want) lo = mid;
            else hi = mid - 1u;
        }
        cut = lo;
        c = count_ge(cut);
    }
    // gather the survivors (order does not matter; equal scores beyond the capacity are the cut itself and are padded back)
    const bool over = c > (u32)kSelCap;                       // only after the exact search: a pile of scores equal to the cut
    const u32 take_cut = (over && cut != 0xFFFFFFFFu) ? cut + 1u : cut;   // then take what lies strictly above and pad with the cut
    const bool take_any = want > 0 && !(over && cut == 0xFFFFFFFFu);
#pragma unroll
    for (int e = 0; e < 4 * J4; ++e) {
        const bool in = take_any && o[e] >= take_cut;
        const u64 m = __ballot(in);
        if (m) {
            u32 base = 0;
            if (lane == 0) base = atomicAdd(&nsurv, (u32)__popcll(m));
            base = (u32)__builtin_amdgcn_readfirstlane((int)base);
            const u32 at = base + (u32)__popcll(m & ((1ull << lane) - 1ull));
            if (in && at < (u32)kSelCap) surv[at] = o[e];
        }
    }
    __syncthreads();
    const int ns = (int)min(nsurv, (u32)kSelCap);
    const int m = over ? max(ns, (int)want) : ns;             // `over`: surv[ns .. want) are copies of the cut (not stored)
    // sorted scores as keys (score << 32), entries past the survivors 0 = empty, as level_threshold expects (it reads up
    // to best[kl - 1])
    if (m <= THREADS) {
        // rank sort: thread t owns survivor t and counts the survivors that come before it - m broadcast reads of LDS and
        // two instructions each, no exchange network
        const u32 mine = (int)threadIdx.x < ns ? surv[threadIdx.x] : cut;
        int rank = 0;
        const uint4* s4 = (const uint4*)surv;                 // four survivors per read (entries past ns are not counted)
#pragma unroll 2
        for (int i = 0; i < ns; i += 4) {
            const uint4 x = s4[i >> 2];
            rank += (x.x > mine || (x.x == mine && i < (int)threadIdx.x)) ? 1 : 0;
            rank += (i + 1 < ns && (x.y > mine || (x.y == mine && i + 1 < (int)threadIdx.x))) ? 1 : 0;
            rank += (i + 2 < ns && (x.z > mine || (x.z == mine && i + 2 < (int)threadIdx.x))) ? 1 : 0;
            rank += (i + 3 < ns && (x.w > mine || (x.w == mine && i + 3 < (int)threadIdx.x))) ? 1 : 0;
        }
        if ((int)threadIdx.x >= ns) rank = (int)threadIdx.x;  // the padding (copies of the cut, below every survivor) and the empty tail
        best[rank] = (int)threadIdx.x < m ? ((u64)mine << 32) : 0ull;
    } else {
        for (int i = threadIdx.x; i < kSelCap; i += THREADS) best[i] = (i < ns) ? ((u64)surv[i] << 32) : (i < m ? ((u64)cut << 32) : 0ull);
        int P = 2;
        while (P < m) P <<= 1;
        bitonic_sort_desc(best, P, threadIdx.x, THREADS);
    }
    __syncthreads();
    if (wave != 0) return;
    const float thr = level_threshold(a, best, cnt, kk, tail_fit, mean, sd);
    if (lane == 0) a.thr[q] = thr;
}

// One-launch reduction of the scan's partial lists (up to kHistSelectMax keys per query: 1024 workgroups x k <= 12) to the
// final k: the keys are loaded into LDS (empty slots squeezed out by ballot), then lds_select_top's histogram cut + short
// sort.  Replaces two select_kernel rounds (10 bitonic sorts of 1024 keys + a final one: 22 + 7 us and a kernel boundary)
// by ~10 us: what the single-query searches of the apps wait for (streamlit_app.py:282-283, app_showcase_model.py:93-96).
constexpr int kHistSelectMax = 12288;   // 96 KiB of keys: with the short list, the per-wave lists and the histogram 132 KiB of LDS
constexpr int kHistSelectLds = kHistSelectMax * 8 + kLevelSmall * 8 + (kLevelThreads / 64) * TS_MAX_K_INTERNAL * 8 + kLevelBins * 4 + 64;
static_assert(kHistSelectLds <= 160 * 1024 && kLevelLds <= 160 * 1024, "select kernels must fit the CU's LDS");

template <int KR>   // 1: k <= 64, 4: k <= 256 (as level_select_kernel)
__global__ void __launch_bounds__(kLevelThreads) select_hist_kernel(SelectArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u64* keys = (u64*)smem;
    u64* small = keys + kHistSelectMax;
    u64* wlists = small + kLevelSmall;
    u32* hist = (u32*)(wlists + (kLevelThreads / 64) * TS_MAX_K_INTERNAL);
    u32* ctr = hist + kLevelBins;
    const int slot = blockIdx.x;
    if (a.qcount && slot >= *a.qcount) return;
    const int lane = threadIdx.x & 63;
    if (threadIdx.x == 0) {
        ctr[0] = 0;
        ctr[3] = 0;
    }
    for (int i = threadIdx.x; i < kLevelBins; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    const u64* src = a.in + (int64_t)slot * a.in_stride;
    const int m = min(a.m, kHistSelectMax);
    for (int i0 = (threadIdx.x & ~63); i0 < m; i0 += blockDim.x) {
        const int i = i0 + lane;
        const u64 key = (i < m) ? src[i] : 0ull;
        const u64 live = __ballot(key != 0ull);
        u32 base = 0;
        if (lane == 0 && live) base = atomicAdd(&ctr[0], (u32)__popcll(live));
        base = (u32)__shfl((int)base, 0, 64);
        if (key != 0ull) keys[base + __popcll(live & ((1ull << lane) - 1ull))] = key;
    }
    __syncthreads();
    const int cnt = (int)ctr[0];
    double mean, sd;
    u64* best = lds_select_top<KR>(keys, cnt, a.k_user, small, wlists, hist, ctr, false, mean, sd);
    const int qid = a.qlist ? a.qlist[slot] : slot;
    for (int i = threadIdx.x; i < a.k_user; i += blockDim.x) {
        const u64 key = best[i];
        a.out_scores[(int64_t)qid * a.k_user + i] = key ? key_score(key) : -INFINITY;
        a.out_idx[(int64_t)qid * a.k_user + i] = !key ? -1 : a.id_map ? a.id_map[key_row(key)] : (int64_t)key_row(key) + a.row_offset;
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

// CAP = LDS entries per workgroup (>= the power of two that holds nparts * k_in).  The exchange of the sharded search
// merges 8 x 10 candidates per query: with CAP = 128 the workgroup is one wave and 1.5 KB of LDS, so the merge on the side
// stream does not take CU slots from the next batch's threshold sample (the 4,096-entry form holds 48 KB per workgroup).
template <int CAP>
__global__ void __launch_bounds__(256) merge_kernel(MergeArgs a) {
    __shared__ u32 so[CAP];
    __shared__ int64_t si[CAP];
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

inline void launch_merge(const MergeArgs& a, hipStream_t st) {
    const int m = a.nparts * a.k_in;
    if (m <= 128) merge_kernel<128><<<a.nq, 64, 0, st>>>(a);
    else if (m <= 1024) merge_kernel<1024><<<a.nq, 256, 0, st>>>(a);
    else merge_kernel<kMergeMax><<<a.nq, 256, 0, st>>>(a);
}

}  // namespace ts
