// Streaming scan: scores of a few queries against every corpus row, with a per-wave running
// top-k in registers.  HBM-bound: each corpus byte is read once per pass with 16-byte-per-lane
// coalesced loads; one pass serves QB queries.
//
// Replaces, for small query batches (the Streamlit apps search one query at a time):
//   util.cos_sim(query_emb, embeddings_db)[0]            app_showcase_model.py:93, app_scratchpad.py:129
//   np.argsort(-cosine_scores)[:5] / torch.topk(.., 200) app_scratchpad.py:130, app_showcase_model.py:96
//   ORDER BY e.embedding <#> q ASC LIMIT k               streamlit_app.py:282-283
// and is the exact fall-back of the MFMA path when a query's candidate buffer overflows.
// In EMIT mode it writes the score matrix instead (util.cos_sim of compare_embeddings.py:61).
//
// Algorithmic traffic per pass: n * ld * sizeof(elem) bytes (+ QB * ld * 4 for the queries).
#pragma once
#include "common.h"

namespace ts {

struct ScanArgs {
    const void* corpus;   // [n_pad x ld] storage dtype
    int64_t ld;           // elements per row (multiple of 64)
    int64_t n;            // real rows
    const float* qbuf;    // prepared queries, fp32 [* x ld]; NULL = read qb16 instead
    const unsigned short* qb16;  // the same queries as bf16 bits [* x ld] (the caller's own matrix, used in place)
    const int* qlist;     // optional indirection: query ids to run (fallback list); NULL = 0..nq-1
    const int* qcount;    // optional device count of entries in qlist; NULL = nq
    int nq;
    int k;                // entries kept per wave / written per workgroup
    u64* partial;         // [slot][gridDim.x][k] keys, descending
    float* scores;        // EMIT: [nq x n]
    const u32* row_mask;  // optional filter: bit (row & 31) of word row >> 5 set = the row may be returned
    // optional additive per-row term (similarity + w * ln(citations), streamlit_app.py:351-358):
    // a row is ranked by fmaf(bias_w, bias[row], score); the keys - and so the results - carry that biased score
    const float* bias;
    float bias_w;
    // one-launch form (the exact re-run of the MFMA path): the workgroup that finishes last reduces the partial lists
    // and writes the results, instead of a second launch (scan_finish)
    unsigned* done_ctr;   // NULL = partial lists only; else a zeroed counter, left zeroed
    float* out_scores;    // [query][k]
    int64_t* out_idx;
    int64_t row_offset;
    const int64_t* id_map;
    // the launch behind a full pass of the 16x16 matrix kernel: block 0 first moves that pass's tile boundaries towards
    // equal finishing times for the NEXT search (rebalance_tiles, common.h) - this launch exists anyway and is almost
    // always empty, so the job costs no launch and no extra workgroup of the final select
    int64_t* part;                // [part_g + 1] tile boundaries, updated in place; NULL = nothing to move
    const unsigned* wg_ticks;     // [part_g] time of each workgroup of that pass (100 MHz ticks)
    int part_g;
    float part_gain;
};

__device__ __forceinline__ float scan_query_elem(const ScanArgs& a, int64_t off) {
    return a.qbuf ? a.qbuf[off] : bf16_to_f32(a.qb16[off]);
}

template <int DT> struct Elem;
template <> struct Elem<0> { static constexpr int VEC = 4; };
template <> struct Elem<1> { static constexpr int VEC = 8; };

// corpus rows are read once per pass: non-temporal 16-byte loads (measured on the DMA stream of the
// MFMA kernel: 7.1 vs 6.4 TB/s)
__device__ __forceinline__ uint4 stream_load(const uint4* p) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 v = __builtin_nontemporal_load((const u32x4*)p);
    return make_uint4(v.x, v.y, v.z, v.w);
}

// dot of one 16-byte chunk with VEC query values, accumulated left to right.
// bf16 rows: v_dot2c_f32_bf16 on the packed pairs as they lie (two products and the running sum per instruction) instead of
// two conversions and two fmaf - a pass of four queries over a bf16 index is bound by its vector work, not by HBM (3.35 ms
// for 10M x 768 against 2.3 for the stream).  The queries of a bf16 index are bf16 values held in fp32 (prep_rows_kernel
// rounds them to the storage type, scan_query_elem widens the caller's bf16 matrix), so their upper halves ARE the operands;
// the pairs are loop-invariant and the compiler packs them once per query, outside the row loop.
typedef __bf16 scan_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float dot2_bf16(u32 rows, float q_lo, float q_hi, float acc) {
    const u32 qp = (__float_as_uint(q_lo) >> 16) | (__float_as_uint(q_hi) & 0xFFFF0000u);
    return __builtin_amdgcn_fdot2_f32_bf16(*reinterpret_cast<const scan_bf16x2*>(&rows), *reinterpret_cast<const scan_bf16x2*>(&qp), acc,
                                           false);
}
template <int DT>
__device__ __forceinline__ float chunk_dot(const uint4& v, const float* q, float acc) {
    if (DT == 0) {
        acc = fmaf(__uint_as_float(v.x), q[0], acc);
        acc = fmaf(__uint_as_float(v.y), q[1], acc);
        acc = fmaf(__uint_as_float(v.z), q[2], acc);
        acc = fmaf(__uint_as_float(v.w), q[3], acc);
    } else {
        acc = dot2_bf16(v.x, q[0], q[1], acc);
        acc = dot2_bf16(v.y, q[2], q[3], acc);
        acc = dot2_bf16(v.z, q[4], q[5], acc);
        acc = dot2_bf16(v.w, q[6], q[7], acc);
    }
    return acc;
}

// Sum four per-lane partials over groups of G lanes.  Afterwards every lane holds the complete sum
// of row rho(lane) = 2*[(lane & G/2) != 0] + [(lane & G/4) != 0] of its group.  Fixed order.
template <int G>
__device__ __forceinline__ float reduce4(float v0, float v1, float v2, float v3, int lane) {
    const bool up = (lane & (G / 2)) != 0;
    float a = (up ? v2 : v0) + __shfl_xor(up ? v0 : v2, G / 2, 64);
    float b = (up ? v3 : v1) + __shfl_xor(up ? v1 : v3, G / 2, 64);
    const bool up2 = (lane & (G / 4)) != 0;
    float c = (up2 ? b : a) + __shfl_xor(up2 ? a : b, G / 4, 64);
#pragma unroll
    for (int off = G / 8; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
    return c;
}

template <int G>
__device__ __forceinline__ int rho(int lane) {
    return (((lane & (G / 2)) != 0) ? 2 : 0) + (((lane & (G / 4)) != 0) ? 1 : 0);
}

// Workgroup epilogue: merge the waves' lists through LDS and write the best k.
template <int KR>
__device__ __forceinline__ void wg_merge_store(WaveTopK<KR>& tk, int k, u64* lds, u64* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    int P = 1;
    while (P < nw * k) P <<= 1;
    for (int i = threadIdx.x; i < P; i += blockDim.x) lds[i] = 0;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < KR; ++r) {
        const int s = r * 64 + lane;
        if (s < k) lds[wave * k + s] = tk.key[r];
    }
    bitonic_sort_desc(lds, P, threadIdx.x, blockDim.x);
    for (int i = threadIdx.x; i < k; i += blockDim.x) out[i] = lds[i];
    __syncthreads();
}

// One-launch form: every workgroup publishes its lists (device-scope release), takes a ticket, and the workgroup that
// draws the last one reduces all of them: per query it streams the gridDim.x * k keys through per-wave running top-k
// lists and merges those - slower than the histogram select of a second launch, but this path runs only for the rare
// query the threshold estimate failed for, and what the common case pays is one empty launch instead of two or more.
template <int KR>
__device__ __forceinline__ void scan_finish(const ScanArgs& a, int count, u64* lds) {
    __shared__ int last_block;
    // hand-off between workgroups (MI355X_MICROARCH.md "inter-workgroup visibility"): every storing wave drains its
    // stores, the workgroup meets, ONE lane releases at agent scope (L2 write-back; the explicit vmcnt wait keeps the
    // ticket behind it - hipcc may drop its own), takes the ticket with an agent-scope atomic, and the lane that drew
    // the last ticket acquires; the keys themselves are then read with agent-scope (L1-bypassing) loads
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned ticket = __hip_atomic_fetch_add(a.done_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last_block = (ticket == gridDim.x - 1) ? 1 : 0;
        if (last_block) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    __syncthreads();
    if (!last_block) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int64_t m = (int64_t)gridDim.x * a.k;
    for (int s = 0; s < count; ++s) {
        const u64* src = a.partial + (int64_t)s * m;
        WaveTopK<KR> tk;
        tk.init();
        for (int64_t i0 = (int64_t)wave * 64; i0 < m; i0 += (int64_t)nw * 64) {
            const int64_t i = i0 + lane;
            const u64 key = (i < m) ? __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
            u64 mm = __ballot(key > tk.thr);
            while (mm) {
                const int from = __ffsll((long long)mm) - 1;
                mm &= mm - 1;
                const u64 K = shfl_u64(key, from);
                if (K > tk.thr) tk.insert(K, a.k, lane);
            }
        }
        int P = 1;
        while (P < nw * a.k) P <<= 1;
        __syncthreads();
        for (int i = threadIdx.x; i < P; i += blockDim.x) lds[i] = 0;
        __syncthreads();
#pragma unroll
        for (int r = 0; r < KR; ++r) {
            const int sl = r * 64 + lane;
            if (sl < a.k) lds[wave * a.k + sl] = tk.key[r];
        }
        bitonic_sort_desc(lds, P, threadIdx.x, blockDim.x);
        const int qid = a.qlist ? a.qlist[s] : s;
        for (int i = threadIdx.x; i < a.k; i += blockDim.x) {
            const u64 key = lds[i];
            a.out_scores[(int64_t)qid * a.k + i] = key ? key_score(key) : -INFINITY;
            a.out_idx[(int64_t)qid * a.k + i] = !key ? -1 : a.id_map ? a.id_map[key_row(key)] : (int64_t)key_row(key) + a.row_offset;
        }
    }
    if (threadIdx.x == 0) *a.done_ctr = 0;
}

constexpr int kScanRB = 4;  // rows per lane group per iteration

// Specialised: row = CH * G chunks of 16 bytes, queries in registers.
template <int DT, int CH, int G, int QB, int KR, bool EMIT>
__global__ void __launch_bounds__(256) scan_kernel(ScanArgs a) {
    constexpr int VEC = Elem<DT>::VEC;
    constexpr int GROUPS = 64 / G;
    constexpr int RW = kScanRB * GROUPS;  // rows per wave iteration
    __shared__ u64 lds_keys[1024];
    if (!EMIT && a.part && blockIdx.x == 0) {
        rebalance_tiles(a.part, a.wg_ticks, a.part_g, a.part_gain, (double*)lds_keys);
        __syncthreads();
    }
    const int lane = threadIdx.x & 63;
    const int gl = lane & (G - 1);
    const int grp = lane / G;
    const int64_t gw = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t W = (int64_t)gridDim.x * (blockDim.x >> 6);
    const int64_t nblocks = (a.n + RW - 1) / RW;
    const uint4* __restrict__ base = (const uint4*)a.corpus;
    const int64_t ld16 = a.ld / VEC;  // row stride in 16-byte chunks
    const int count = a.qcount ? *a.qcount : a.nq;

    for (int g0 = 0; g0 < count; g0 += QB) {
        float qv[QB][CH][VEC];
        int qid[QB];
#pragma unroll
        for (int q = 0; q < QB; ++q) {
            const int slot = (g0 + q < count) ? (g0 + q) : g0;
            qid[q] = a.qlist ? a.qlist[slot] : slot;
#pragma unroll
            for (int c = 0; c < CH; ++c)
#pragma unroll
                for (int e = 0; e < VEC; ++e) qv[q][c][e] = scan_query_elem(a, (int64_t)qid[q] * a.ld + (int64_t)(gl + c * G) * VEC + e);
        }
        WaveTopK<KR> tk[QB];
        if (!EMIT) {
#pragma unroll
            for (int q = 0; q < QB; ++q) tk[q].init();
        }
        for (int64_t blk = gw; blk < nblocks; blk += W) {
            const int64_t row0 = blk * RW + grp * kScanRB;
            uint4 v[kScanRB][CH];
#pragma unroll
            for (int r = 0; r < kScanRB; ++r)
#pragma unroll
                for (int c = 0; c < CH; ++c) v[r][c] = stream_load(base + (row0 + r) * ld16 + gl + c * G);  // rows < n_pad: in bounds
            const int myrow = rho<G>(lane);
            const int64_t row = row0 + myrow;
            const bool rep = (lane & (G / 4 - 1)) == 0;
            // metadata filter: one bit per row, N / 8 bytes per pass
            const bool allowed = !a.row_mask || (rep && row < a.n && ((a.row_mask[row >> 5] >> (row & 31)) & 1u));
#pragma unroll
            for (int q = 0; q < QB; ++q) {
                float acc[kScanRB];
#pragma unroll
                for (int r = 0; r < kScanRB; ++r) {
                    float s = 0.0f;
#pragma unroll
                    for (int c = 0; c < CH; ++c) s = chunk_dot<DT>(v[r][c], qv[q][c], s);
                    acc[r] = s;
                }
                float s = reduce4<G>(acc[0], acc[1], acc[2], acc[3], lane);
                if (EMIT) {
                    if (rep && row < a.n && g0 + q < count) a.scores[(int64_t)qid[q] * a.n + row] = s;
                } else {
                    if (a.bias && rep && row < a.n) s = fmaf(a.bias_w, a.bias[row], s);
                    const u64 key = (rep && row < a.n && s == s && allowed) ? make_key(s, (u32)row) : 0ull;
                    u64 m = __ballot(key > tk[q].thr);
                    while (m) {
                        const int src = __ffsll((long long)m) - 1;
                        m &= m - 1;
                        const u64 K = shfl_u64(key, src);
                        if (K > tk[q].thr) tk[q].insert(K, a.k, lane);
                    }
                }
            }
        }
        if (!EMIT) {
#pragma unroll
            for (int q = 0; q < QB; ++q) {
                if (g0 + q < count)  // uniform over the workgroup
                    wg_merge_store<KR>(tk[q], a.k, lds_keys, a.partial + ((int64_t)(g0 + q) * gridDim.x + blockIdx.x) * a.k);
            }
        }
    }
    if constexpr (!EMIT)
        if (a.done_ctr && count > 0) scan_finish<KR>(a, count, lds_keys);
}

// Generic: any ld (multiple of 64 elements), QB queries per pass staged in LDS (QB = 1 or 4).
template <int DT, int KR, bool EMIT, int QB>
__global__ void __launch_bounds__(256) scan_generic_kernel(ScanArgs a) {
    constexpr int VEC = Elem<DT>::VEC;
    constexpr int RW = kScanRB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u64* lds_keys = (u64*)smem;            // 1024 keys
    float* lds_q = (float*)(smem + 8192);  // QB x ld floats
    if (!EMIT && a.part && blockIdx.x == 0) {
        rebalance_tiles(a.part, a.wg_ticks, a.part_g, a.part_gain, (double*)lds_keys);
        __syncthreads();
    }
    const int lane = threadIdx.x & 63;
    const int64_t gw = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t W = (int64_t)gridDim.x * (blockDim.x >> 6);
    const int64_t nblocks = (a.n + RW - 1) / RW;
    const uint4* __restrict__ base = (const uint4*)a.corpus;
    const int64_t ld16 = a.ld / VEC;
    const int count = a.qcount ? *a.qcount : a.nq;

    for (int g0 = 0; g0 < count; g0 += QB) {
        int qid[QB];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < QB; ++q) {
            const int slot = (g0 + q < count) ? (g0 + q) : g0;
            qid[q] = a.qlist ? a.qlist[slot] : slot;
            for (int i = threadIdx.x; i < a.ld; i += blockDim.x) lds_q[(int64_t)q * a.ld + i] = scan_query_elem(a, (int64_t)qid[q] * a.ld + i);
        }
        __syncthreads();
        WaveTopK<KR> tk[QB];
#pragma unroll
        for (int q = 0; q < QB; ++q) tk[q].init();
        for (int64_t blk = gw; blk < nblocks; blk += W) {
            const int64_t row0 = blk * RW;
            float acc[QB][kScanRB];
#pragma unroll
            for (int q = 0; q < QB; ++q)
#pragma unroll
                for (int r = 0; r < kScanRB; ++r) acc[q][r] = 0.f;
            for (int64_t c = lane; c < ld16; c += 64) {
                uint4 v[kScanRB];
#pragma unroll
                for (int r = 0; r < kScanRB; ++r) v[r] = stream_load(base + (row0 + r) * ld16 + c);
#pragma unroll
                for (int q = 0; q < QB; ++q) {
                    float qreg[VEC];
#pragma unroll
                    for (int e = 0; e < VEC; ++e) qreg[e] = lds_q[(int64_t)q * a.ld + c * VEC + e];
#pragma unroll
                    for (int r = 0; r < kScanRB; ++r) acc[q][r] = chunk_dot<DT>(v[r], qreg, acc[q][r]);
                }
            }
            const int64_t row = row0 + rho<64>(lane);
            const bool rep = (lane & 15) == 0;
            const bool allowed = !a.row_mask || (rep && row < a.n && ((a.row_mask[row >> 5] >> (row & 31)) & 1u));
#pragma unroll
            for (int q = 0; q < QB; ++q) {
                float s = reduce4<64>(acc[q][0], acc[q][1], acc[q][2], acc[q][3], lane);
                if (EMIT) {
                    if (rep && row < a.n && g0 + q < count) a.scores[(int64_t)qid[q] * a.n + row] = s;
                } else {
                    if (a.bias && rep && row < a.n) s = fmaf(a.bias_w, a.bias[row], s);
                    const u64 key = (rep && row < a.n && s == s && allowed) ? make_key(s, (u32)row) : 0ull;
                    u64 m = __ballot(key > tk[q].thr);
                    while (m) {
                        const int src = __ffsll((long long)m) - 1;
                        m &= m - 1;
                        const u64 K = shfl_u64(key, src);
                        if (K > tk[q].thr) tk[q].insert(K, a.k, lane);
                    }
                }
            }
        }
        if (!EMIT) {
#pragma unroll
            for (int q = 0; q < QB; ++q)
                if (g0 + q < count)  // uniform over the workgroup
                    wg_merge_store<KR>(tk[q], a.k, lds_keys, a.partial + ((int64_t)(g0 + q) * gridDim.x + blockIdx.x) * a.k);
        }
    }
    if constexpr (!EMIT)
        if (a.done_ctr && count > 0) scan_finish<KR>(a, count, lds_keys);
}

// Raw similarity of biased results: sim = biased - w * bias[row] (one rounding away from the score the scan computed).
__global__ void unbias_kernel(const float* scores, const int64_t* idx, const float* bias, float w, int64_t row_offset, float* sims,
                              int64_t count) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const int64_t id = idx[i];
    sims[i] = id < 0 ? -INFINITY : fmaf(-w, bias[id - row_offset], scores[i]);
}

// ---- rank of a given row ("rank of gold") -----------------------------------------------------------------
// The evaluation script ranks the whole corpus per query and looks up where the relevant document landed
// (compare_embeddings.py:104-123 mrr_at_k with k=None: np.argsort(-sim) then the position of the exact doc).
// The position of row t in the canonical order is the number of rows whose key beats key(t): one streaming
// pass that counts, no [nq x N] matrix and no sort.  The score arithmetic is the scan kernel's (same chunk
// order, same reduce4), so the rank is consistent with what ts_search returns through the scan path.
struct RankArgs {
    const void* corpus;
    int64_t ld, n;
    const float* qbuf;            // prepared queries, fp32 [nq x ld]
    int nq;
    const int64_t* target;        // [nq] local row of each query's document; outside [0, n) = none
    const u64* tkey;              // optional [nq]: ready-made target keys (a document that lives on another shard);
                                  // `target` is not read then
    unsigned long long* counts;   // [nq] zeroed by the host; += rows ranked strictly before the target
    float* tscore;                // [nq] score of the target row (NaN when there is none)
};

template <int DT, int CH, int G, int QB>
__global__ void __launch_bounds__(256) rank_kernel(RankArgs a) {
    constexpr int VEC = Elem<DT>::VEC;
    constexpr int GROUPS = 64 / G;
    constexpr int RW = kScanRB * GROUPS;
    const int lane = threadIdx.x & 63;
    const int gl = lane & (G - 1);
    const int grp = lane / G;
    const int64_t gw = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t W = (int64_t)gridDim.x * (blockDim.x >> 6);
    const int64_t nblocks = (a.n + RW - 1) / RW;
    const uint4* __restrict__ base = (const uint4*)a.corpus;
    const int64_t ld16 = a.ld / VEC;
    const int myrow = rho<G>(lane);
    const bool rep = (lane & (G / 4 - 1)) == 0;
    __shared__ unsigned int wg_cnt[QB];

    for (int g0 = 0; g0 < a.nq; g0 += QB) {
        float qv[QB][CH][VEC];
        int qid[QB];
        u64 gkey[QB];
        unsigned int cnt[QB];
#pragma unroll
        for (int q = 0; q < QB; ++q) {
            qid[q] = (g0 + q < a.nq) ? (g0 + q) : g0;
#pragma unroll
            for (int c = 0; c < CH; ++c)
#pragma unroll
                for (int e = 0; e < VEC; ++e) qv[q][c][e] = a.qbuf[(int64_t)qid[q] * a.ld + (int64_t)(gl + c * G) * VEC + e];
            // key of the target row: its 4-row block goes through the same arithmetic as the pass below
            cnt[q] = 0;
            if (a.tkey) {  // uniform over the grid
                gkey[q] = a.tkey[qid[q]];
                continue;
            }
            const int64_t t = a.target[qid[q]];
            const bool has = t >= 0 && t < a.n;
            const int64_t tb = has ? (t & ~(int64_t)3) : 0;
            float acc[kScanRB];
#pragma unroll
            for (int r = 0; r < kScanRB; ++r) {
                float s = 0.0f;
#pragma unroll
                for (int c = 0; c < CH; ++c) s = chunk_dot<DT>(stream_load(base + (tb + r) * ld16 + gl + c * G), qv[q][c], s);
                acc[r] = s;
            }
            const float s4 = reduce4<G>(acc[0], acc[1], acc[2], acc[3], lane);
            const int tr = (int)(t & 3);
            const float st = __shfl(s4, ((tr & 2) ? G / 2 : 0) + ((tr & 1) ? G / 4 : 0), 64);
            gkey[q] = (has && st == st) ? make_key(st, (u32)t) : ~0ull;
            if (gw == 0 && lane == 0 && g0 + q < a.nq) a.tscore[qid[q]] = (has ? st : __uint_as_float(0x7FC00000u));
        }
        for (int64_t blk = gw; blk < nblocks; blk += W) {
            const int64_t row0 = blk * RW + grp * kScanRB;
            uint4 v[kScanRB][CH];
#pragma unroll
            for (int r = 0; r < kScanRB; ++r)
#pragma unroll
                for (int c = 0; c < CH; ++c) v[r][c] = stream_load(base + (row0 + r) * ld16 + gl + c * G);
            const int64_t row = row0 + myrow;
#pragma unroll
            for (int q = 0; q < QB; ++q) {
                float acc[kScanRB];
#pragma unroll
                for (int r = 0; r < kScanRB; ++r) {
                    float s = 0.0f;
#pragma unroll
                    for (int c = 0; c < CH; ++c) s = chunk_dot<DT>(v[r][c], qv[q][c], s);
                    acc[r] = s;
                }
                const float s = reduce4<G>(acc[0], acc[1], acc[2], acc[3], lane);
                const u64 key = (rep && row < a.n && s == s) ? make_key(s, (u32)row) : 0ull;
                cnt[q] += (unsigned int)__popcll(__ballot(key > gkey[q]));
            }
        }
        // one atomic per workgroup and query: thousands of waves adding to one address serialise in L2
        __syncthreads();
        if (threadIdx.x < QB) wg_cnt[threadIdx.x] = 0;
        __syncthreads();
#pragma unroll
        for (int q = 0; q < QB; ++q)
            if (lane == 0 && cnt[q]) atomicAdd(&wg_cnt[q], cnt[q]);
        __syncthreads();
        if (threadIdx.x < QB && wg_cnt[threadIdx.x] && g0 + threadIdx.x < a.nq)
            atomicAdd(&a.counts[g0 + threadIdx.x], (unsigned long long)wg_cnt[threadIdx.x]);
    }
}

// Any ld: one query per pass, query staged in LDS (the arithmetic of scan_generic_kernel).
template <int DT>
__global__ void __launch_bounds__(256) rank_generic_kernel(RankArgs a) {
    constexpr int VEC = Elem<DT>::VEC;
    constexpr int RW = kScanRB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* lds_q = (float*)smem;
    const int lane = threadIdx.x & 63;
    const int64_t gw = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t W = (int64_t)gridDim.x * (blockDim.x >> 6);
    const int64_t nblocks = (a.n + RW - 1) / RW;
    const uint4* __restrict__ base = (const uint4*)a.corpus;
    const int64_t ld16 = a.ld / VEC;
    const bool rep = (lane & 15) == 0;

    auto block_score = [&](int64_t row0) -> float {
        float acc[kScanRB] = {0.f, 0.f, 0.f, 0.f};
        for (int64_t c = lane; c < ld16; c += 64) {
            float qreg[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) qreg[e] = lds_q[c * VEC + e];
#pragma unroll
            for (int r = 0; r < kScanRB; ++r) acc[r] = chunk_dot<DT>(stream_load(base + (row0 + r) * ld16 + c), qreg, acc[r]);
        }
        return reduce4<64>(acc[0], acc[1], acc[2], acc[3], lane);
    };

    for (int qid = 0; qid < a.nq; ++qid) {
        __syncthreads();
        for (int i = threadIdx.x; i < a.ld; i += blockDim.x) lds_q[i] = a.qbuf[(int64_t)qid * a.ld + i];
        __syncthreads();
        u64 gkey;
        if (a.tkey) {
            gkey = a.tkey[qid];
        } else {
            const int64_t t = a.target[qid];
            const bool has = t >= 0 && t < a.n;
            const float s4 = block_score(has ? (t & ~(int64_t)3) : 0);
            const int tr = (int)(t & 3);
            const float st = __shfl(s4, ((tr & 2) ? 32 : 0) + ((tr & 1) ? 16 : 0), 64);
            gkey = (has && st == st) ? make_key(st, (u32)t) : ~0ull;
            if (gw == 0 && lane == 0) a.tscore[qid] = has ? st : __uint_as_float(0x7FC00000u);
        }
        unsigned int cnt = 0;
        for (int64_t blk = gw; blk < nblocks; blk += W) {
            const int64_t row0 = blk * RW;
            const float s = block_score(row0);
            const int64_t row = row0 + rho<64>(lane);
            const u64 key = (rep && row < a.n && s == s) ? make_key(s, (u32)row) : 0ull;
            cnt += (unsigned int)__popcll(__ballot(key > gkey));
        }
        __shared__ unsigned int wg_cnt;
        __syncthreads();
        if (threadIdx.x == 0) wg_cnt = 0;
        __syncthreads();
        if (lane == 0 && cnt) atomicAdd(&wg_cnt, cnt);
        __syncthreads();
        if (threadIdx.x == 0 && wg_cnt) atomicAdd(&a.counts[qid], (unsigned long long)wg_cnt);
    }
}

}  // namespace ts
