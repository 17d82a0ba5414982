// libtsearch.so - C ABI (include/tsearch.h), part 5: ts_search_cpu, the ONE host-only entry of the library.
//
// SURVEY.md section 8b lists it among the minimum exports ("C++ fallback, same semantics, for C1"): BASELINE.json configs[0] is
// the reference's own CPU-runnable case - `util.cos_sim` + `argsort` over ~1k theorems (compare_embeddings.py:24-31,55-92) -
// and a deployment without an MI355X must still be able to run that plumbing.  It is NOT a fallback of anything: no device
// entry ever calls it, nothing selects it by itself (ts_search and friends return TS_ERR_NODEVICE without a device), and the
// only way in from Python is TheoremIndex(..., device=-1).  Stateless: rows and queries are host arrays, prepared on every
// call exactly as the device path prepares them once at upload (prep_rows_kernel: norm^2 in fp64, x / max(||x||, 1e-12);
// bf16 storage rounds to nearest even) - which is also what util.cos_sim does on every call.  Scores are fp32 dot products
// of the prepared values (eight partial sums, then a fixed tree); selection is the library's 64-bit key order
// (score descending, row ascending; NaN never ranks; -0 folds into +0), padding (-inf, -1).
#include <thread>

#include "host.h"

namespace {

inline u32 ord_host(float s) {
    s = s + 0.0f;
    u32 u;
    memcpy(&u, &s, 4);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
inline float unord_host(u32 o) {
    const u32 u = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
inline float bf16_bits_to_f32_host(unsigned short b) {
    const u32 u = (u32)b << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
inline float round_bf16_host(float f) {          // RNE to bf16, widened again; NaN stays NaN (common.h f32_to_bf16)
    u32 u;
    memcpy(&u, &f, 4);
    const unsigned short b = (f != f) ? (unsigned short)((u >> 16) | 0x0040u) : (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
    return bf16_bits_to_f32_host(b);
}

// One row (fp32 or bf16 bits) -> the values the kernels would multiply, in `out[0..d)`.
inline void prepare_row(const void* src, int src_dtype, int64_t row, int d, bool normalize, bool store_bf16, float* out) {
    if (src_dtype == TS_F32) {
        const float* p = (const float*)src + row * (int64_t)d;
        for (int c = 0; c < d; ++c) out[c] = p[c];
    } else {
        const unsigned short* p = (const unsigned short*)src + row * (int64_t)d;
        for (int c = 0; c < d; ++c) out[c] = bf16_bits_to_f32_host(p[c]);
    }
    if (normalize) {
        double ss = 0.0;
        for (int c = 0; c < d; ++c) ss += (double)out[c] * (double)out[c];
        const float denom = std::max((float)std::sqrt(ss), 1e-12f);
        for (int c = 0; c < d; ++c) out[c] = (float)((double)out[c] / (double)denom);
    }
    if (store_bf16)
        for (int c = 0; c < d; ++c) out[c] = round_bf16_host(out[c]);
}

inline float dot_f32(const float* a, const float* b, int d) {
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int c = 0;
    for (; c + 8 <= d; c += 8)
        for (int j = 0; j < 8; ++j) acc[j] = std::fmaf(a[c + j], b[c + j], acc[j]);
    for (; c < d; ++c) acc[c & 7] = std::fmaf(a[c], b[c], acc[c & 7]);
    return ((acc[0] + acc[4]) + (acc[2] + acc[6])) + ((acc[1] + acc[5]) + (acc[3] + acc[7]));
}

// the k largest keys seen so far: a min-heap on the key (root = the worst kept)
struct KeyHeap {
    std::vector<u64> h;
    int k = 0;
    void push(u64 key) {
        if ((int)h.size() < k) {
            h.push_back(key);
            std::push_heap(h.begin(), h.end(), std::greater<u64>());
        } else if (key > h.front()) {
            std::pop_heap(h.begin(), h.end(), std::greater<u64>());
            h.back() = key;
            std::push_heap(h.begin(), h.end(), std::greater<u64>());
        }
    }
};

}  // namespace

extern "C" int ts_search_cpu(const void* rows, int rows_dtype, int64_t n, int32_t d, int store_dtype, int metric, const void* queries,
                             int q_dtype, int32_t nq, int32_t k, float* out_scores, int64_t* out_idx, int32_t threads) {
    if ((!rows && n > 0) || (!queries && nq > 0) || !out_scores || !out_idx) return fail(TS_ERR_INVALID, "NULL argument");
    if ((rows_dtype != TS_F32 && rows_dtype != TS_BF16) || (q_dtype != TS_F32 && q_dtype != TS_BF16) ||
        (store_dtype != TS_F32 && store_dtype != TS_BF16))
        return fail(TS_ERR_INVALID, "dtype must be TS_F32 or TS_BF16");
    if (metric != TS_METRIC_IP && metric != TS_METRIC_COS) return fail(TS_ERR_INVALID, "metric %d", metric);
    if (n < 0 || n > 0xFFFFFFFFll || d < 1 || nq < 0 || k < 1 || k > TS_MAX_K) return fail(TS_ERR_INVALID, "bad shape");
    if (nq == 0) return TS_OK;
    const bool normalize = metric == TS_METRIC_COS, bf16 = store_dtype == TS_BF16;
    std::vector<float> q((size_t)nq * d);
    for (int b = 0; b < nq; ++b) prepare_row(queries, q_dtype, b, d, normalize, bf16, q.data() + (size_t)b * d);
    int T = threads > 0 ? threads : (int)std::thread::hardware_concurrency();
    T = (int)std::max<int64_t>(1, std::min<int64_t>(std::min(T, 256), (n + 1023) / 1024));
    std::vector<std::vector<KeyHeap>> heaps((size_t)T, std::vector<KeyHeap>((size_t)nq));
    auto work = [&](int t) {
        std::vector<float> row((size_t)d);
        for (auto& hp : heaps[t]) { hp.k = k; hp.h.reserve((size_t)k); }
        const int64_t lo = n * t / T, hi = n * (t + 1) / T;
        for (int64_t r = lo; r < hi; ++r) {
            prepare_row(rows, rows_dtype, r, d, normalize, bf16, row.data());
            for (int b = 0; b < nq; ++b) {
                const float s = dot_f32(q.data() + (size_t)b * d, row.data(), d);
                if (s == s) heaps[t][b].push(((u64)ord_host(s) << 32) | (u64)(0xFFFFFFFFu - (u32)r));
            }
        }
    };
    if (T == 1) {
        work(0);
    } else {
        std::vector<std::thread> pool;
        for (int t = 0; t < T; ++t) pool.emplace_back(work, t);
        for (auto& th : pool) th.join();
    }
    std::vector<u64> all;
    for (int b = 0; b < nq; ++b) {
        all.clear();
        for (int t = 0; t < T; ++t) all.insert(all.end(), heaps[t][b].h.begin(), heaps[t][b].h.end());
        std::sort(all.begin(), all.end(), std::greater<u64>());
        for (int j = 0; j < k; ++j) {
            const bool have = j < (int)all.size();
            out_scores[(size_t)b * k + j] = have ? unord_host((u32)(all[j] >> 32)) : -INFINITY;
            out_idx[(size_t)b * k + j] = have ? (int64_t)(0xFFFFFFFFu - (u32)all[j]) : -1;
        }
    }
    return TS_OK;
}
