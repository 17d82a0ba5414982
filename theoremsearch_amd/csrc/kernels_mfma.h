// Batched search, bf16, d = 768: query x corpus contraction on the matrix cores with the top-k
// selection fused behind it.  The [nq x N] score matrix is never written.
//
// Replaces util.cos_sim(q_emb, s_emb) + np.argsort(-sim_matrix, axis=1) of the batched call
// sites (compare_embeddings.py:61,105) at the shapes of BASELINE.json configs[2..3].
//
// Shape of the work (one workgroup = 4 waves = one CU, one wave per SIMD, persistent):
//   * the 256 queries live in REGISTERS for the whole kernel: wave w owns queries 64w..64w+63 as two
//     B-operand groups of v_mfma_f32_32x32x16_bf16 (qa: 64w + r in VGPRs, qb: 64w + 32 + r in AGPRs;
//     2 x 48 k-steps x 4 registers = 384 of the wave's 512 registers);
//   * the corpus streams HBM -> LDS exactly once per CU by LDS-DMA (global_load_lds_dwordx4, non-
//     temporal) in UNITS of 32 rows x 384 k (24 KiB = half a tile), six units deep.  The barrier at
//     the top of unit u certifies unit u+1, so the operand reads of unit u+1 are issued while unit
//     u still computes (no bubble at the seam), and frees the slot of unit u-1 for unit u+5, whose
//     six DMA pieces per wave are issued between the MFMAs of unit u;
//   * every wave reads every unit from LDS as the A operand: one ds_read_b128 feeds two MFMAs (one
//     per query group), four k-steps ahead of use;
//   * D[i][j] = <corpus row i, query j>: a lane holds 16 corpus rows for ONE query per group, so
//     the epilogue is a per-lane compare against that query's threshold.  Scores that pass go to a
//     lane-private list in global memory (plain stores, no atomics: a returning atomic would drain
//     the DMA queue); a lane whose private list is full spills to the query's shared list.
//     Thresholds come from the previous, sparser level (mfma_search in tsearch_api.hip).
//   * every workgroup walks its own contiguous range of tiles.
//
// LDS image of a unit: 24 pieces of 1 KiB; piece (kb, p) = K-block kb (64 elements = 128 B per row)
// of rows 8p..8p+7 at (kb * 4 + p) * 1024, written by ONE wave-instruction whose lane l fetches row
// 8p + (l >> 3), 16-byte chunk (l & 7) ^ ((row >> 1) & 7) of that K-block: full 128-byte lines from
// HBM, and the XOR on the SOURCE side makes the MFMA operand reads (lane (r, h) reads chunk 2s' + h
// of row r) hit 16 distinct 16-byte slots per ds_read_b128 lane group: conflict-free.
//
// Algorithmic traffic: rows * 1536 bytes per launch; flops 2 * 256 * rows * 768.
#pragma once
#include "common.h"

// Cache policy of the corpus stream (read once per search): " nt" = non-temporal.
#ifndef TS_DMA_POLICY
#define TS_DMA_POLICY " nt"
#endif

namespace ts {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int kMfmaD = 768;
constexpr int kMfmaQ = 256;                                   // queries per launch
constexpr int kMfmaThreads = 256;
constexpr int kMfmaKSteps = kMfmaD / 16;                      // 48
constexpr int kMfmaUnitK = 384;
constexpr int kMfmaUnitSteps = kMfmaUnitK / 16;               // 24
constexpr int kMfmaUnitBytes = kTileRows * kMfmaUnitK * 2;    // 24576
constexpr int kMfmaSlots = 6;
constexpr int kMfmaLds = kMfmaSlots * kMfmaUnitBytes;         // 147456
constexpr int kMfmaPieces = 6;                                // DMA pieces per wave per unit
constexpr int kMfmaAhead = 4;                                 // A fragments (k-steps) in flight
constexpr int kMfmaPrivCap = 32;                              // entries of a lane-private candidate list

struct MfmaArgs {
    const unsigned short* corpus;  // bf16 [n_pad x 768]
    int64_t n;                     // real rows
    int64_t ntiles;                // tiles visited at this level
    int64_t tile_stride;           // visited tile j is global tile (j / run) * run * tile_stride + j % run:
    int run;                       //   runs of `run` consecutive tiles, `run * tile_stride` tiles apart
    const unsigned short* q;       // bf16 [256 x 768], zero rows past nq
    const float* thr;              // [256] pass threshold per query (+inf for absent queries)
    u64* priv;                     // [256][W][kMfmaPrivCap] lane-private lists, W = 2 * gridDim.x writers
    u32* pcount;                   // [256][W] entries each writer produced (may exceed kMfmaPrivCap: spilled)
    u64* cand;                     // [256][cap] shared spill lists
    u32* count;                    // [256]
    int cap;
    unsigned long long* dbg;       // VARIANT 3 only: per-wave cycle sums
};

// One LDS-DMA wave-instruction: 64 lanes x 16 bytes from per-lane global addresses to
// lds_dst + lane * 16 (lds_dst wave-uniform, in M0).  Inline asm on purpose: hipcc keeps no vmcnt
// bookkeeping for it, so it never drains the DMA queue behind our back (a compiler-visible LDS-DMA
// makes every following ds_read wait vmcnt(0)); the waits are the counted ones in the kernel.
// M0 is not saved: the generated code of this kernel never reads M0 (checked in the .s).
__device__ __forceinline__ void lds_dma16(const void* gsrc, unsigned lds_dst) {
    asm volatile(
        "s_mov_b32 m0, %1\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %0, off" TS_DMA_POLICY
        :
        : "v"(gsrc), "s"(lds_dst)
        : "memory");
}

// MFMA with pinned register classes (left alone the allocator shuttles operands between the VGPR and
// AGPR halves of the file): accumulators and the corpus fragment in AGPRs, query fragment in VGPRs
// (group A) or AGPRs (group B).  The *_first forms start a chain with C = 0.  hipcc knows nothing
// about what is inside: chains are back-to-back accumulations (no pad needed); before any other
// reader of an accumulator mfma_settle() supplies the wait states.
__device__ __forceinline__ void mfma_av_first(f32x16& acc, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&a"(acc) : "a"(a), "v"(b));
}
__device__ __forceinline__ void mfma_av(f32x16& acc, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "a"(a), "v"(b));
}
__device__ __forceinline__ void mfma_aa_first(f32x16& acc, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&a"(acc) : "a"(a), "a"(b));
}
__device__ __forceinline__ void mfma_aa(f32x16& acc, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "a"(a), "a"(b));
}
__device__ __forceinline__ void mfma_settle(f32x16& x, f32x16& y) {
    asm volatile("s_nop 15\n\ts_nop 7" : "+a"(x), "+a"(y));
}
__device__ __forceinline__ unsigned long long cycle_stamp() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

__device__ __forceinline__ float max16(const f32x16& acc) {
    float m = acc[0];
#pragma unroll
    for (int g = 1; g < 16; ++g) m = fmaxf(m, acc[g]);
    return m;
}

// Append this lane's passing scores of one accumulator tile.  cnt = entries this lane has produced
// for the query so far; the first kMfmaPrivCap go to its private list, later ones to the shared list.
// Rare path, so it is written for few instructions when few lanes hit: every accumulator register is
// tested wave-wide first, and only registers with a hit somewhere run the store code.
// FULL = every row of the tile is a real row (no row < n test).
template <bool FULL>
__device__ __forceinline__ void mfma_append(const f32x16& acc, float thr, int qid, int writer, int nwriters, u32& cnt,
                                            int64_t row_base, const MfmaArgs& a) {
    u64* mine = a.priv + ((int64_t)qid * nwriters + writer) * kMfmaPrivCap;
#pragma unroll
    for (int g = 0; g < 16; ++g) {
        const float s = acc[g];
        const int64_t row = row_base + (g & 3) + 8 * (g >> 2);
        const bool hit = (s >= thr) && (FULL || row < a.n);
        if (__any(hit)) {
            if (hit) {
                const u64 key = make_key(s, (u32)row);
                if (cnt < (u32)kMfmaPrivCap) {
                    mine[cnt] = key;
                } else {
                    const u32 pos = atomicAdd(&a.count[qid], 1u);
                    if (pos < (u32)a.cap) a.cand[(int64_t)qid * a.cap + pos] = key;
                }
                ++cnt;
            }
        }
    }
}

// VARIANT 0 = the product kernel.  Timing-only diagnostics (wrong results), selected with
// TS_MFMA_VARIANT: 1 = no epilogue, 2 = DMA stream only, 3 = product + cycle stamps into a.dbg.
// SPARSE only changes the symbol: the sample levels show up under their own name in kernel traces, so
// the statistics of the full-corpus pass are not mixed with them.
template <int VARIANT, bool SPARSE>
__global__ void __launch_bounds__(kMfmaThreads, 1) mfma_topk_kernel(MfmaArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int G = gridDim.x;
    const int nwriters = 2 * G;
    const int writer = 2 * blockIdx.x + h;
    const int qid_a = wave * 64 + r, qid_b = qid_a + 32;

    // this workgroup's contiguous share of the level's tiles (sequential pages: a round-robin deal of
    // tiles to workgroups measured 1.4x slower on the DMA stream)
    const int64_t t0 = (a.ntiles * (int64_t)blockIdx.x) / G;
    const int nt = (int)((a.ntiles * (int64_t)(blockIdx.x + 1)) / G - t0);
    if (nt <= 0) {
        a.pcount[(int64_t)qid_a * nwriters + writer] = 0;
        a.pcount[(int64_t)qid_b * nwriters + writer] = 0;
        return;
    }
    const int nu = 2 * nt;  // units

    bf16x8 qa[kMfmaKSteps], qb[kMfmaKSteps];
    {
        const bf16x8* pa = (const bf16x8*)(a.q + (int64_t)qid_a * kMfmaD + 8 * h);
        const bf16x8* pb = (const bf16x8*)(a.q + (int64_t)qid_b * kMfmaD + 8 * h);
#pragma unroll
        for (int s = 0; s < kMfmaKSteps; ++s) {
            qa[s] = pa[2 * s];
            qb[s] = pb[2 * s];
        }
    }
    float thr_a = a.thr[qid_a], thr_b = a.thr[qid_b];
    // pin: these loads (and the compiler's waits for them) complete here, outside the unit loop,
    // and the fragments stay in the register class the MFMA statements want
#pragma unroll
    for (int s = 0; s < kMfmaKSteps; ++s) {
        asm volatile("" : "+v"(qa[s]));
        asm volatile("" : "+a"(qb[s]));
    }
    asm volatile("" : "+v"(thr_a));
    asm volatile("" : "+v"(thr_b));

    // DMA source of this lane: row 8w + (lane >> 3) of the tile, swizzled chunk of K-block 0 of the unit
    const int drow = 8 * wave + (lane >> 3);
    const int dchunk = (lane & 7) ^ ((drow >> 1) & 7);
    const int64_t tile_bytes = (int64_t)kTileRows * kMfmaD * 2;
    const int64_t run_jump = tile_bytes * ((int64_t)a.run * a.tile_stride - a.run + 1);  // last tile of a run -> next run
    const int64_t g0 = (t0 / a.run) * a.run * a.tile_stride + t0 % a.run;               // global tile of level tile t0
    const unsigned char* tile_src = (const unsigned char*)a.corpus + (int64_t)drow * (kMfmaD * 2) + dchunk * 16 +
                                    g0 * tile_bytes;                              // tile of the next unit to issue
    int issue_run_pos = (int)(t0 % a.run);                                        // its position inside its run
    int issue_u = 0;  // next unit to issue
    int issue_slot = 0;
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem + wave * 1024;

    // per-lane LDS read offsets inside a unit image
    const int lane_off = (r >> 3) * 1024 + (r & 7) * 128;
    const int sw = (r >> 1) & 7;
    int xo[4];
#pragma unroll
    for (int sp = 0; sp < 4; ++sp) xo[sp] = lane_off + (((2 * sp + h) ^ sw) << 4);

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

#define TS_WAIT_KEEP(units)                                                     \
    do {                                                                        \
        const int keep_ = (units);                                              \
        if (keep_ >= 4) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");       \
        else if (keep_ == 3) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");  \
        else if (keep_ == 2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");  \
        else if (keep_ == 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   \
    } while (0)

    // prologue: five units in flight (slots 0..4); unit 0 must have landed before its fragments are read
    for (int i = 0; i < 5 && issue_u < nu; ++i) {
        const unsigned char* src = tile_src + (issue_u & 1) * (kMfmaUnitK * 2);
#pragma unroll
        for (int j = 0; j < kMfmaPieces; ++j) lds_dma16(src + j * 128, lds0 + issue_slot * kMfmaUnitBytes + j * 4096);
        if (issue_u & 1) {
            tile_src += (issue_run_pos + 1 == a.run) ? run_jump : tile_bytes;
            issue_run_pos = (issue_run_pos + 1 == a.run) ? 0 : issue_run_pos + 1;
        }
        ++issue_u;
        issue_slot = (issue_slot + 1 == kMfmaSlots) ? 0 : issue_slot + 1;
    }
    TS_WAIT_KEEP(issue_u - 1);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    bf16x8 af[kMfmaAhead];
#pragma unroll
    for (int s = 0; s < kMfmaAhead; ++s) af[s] = *(const bf16x8*)(smem + (s >> 2) * 4096 + xo[s & 3]);

    f32x16 acc_a, acc_b;
    u32 cnt_a = 0, cnt_b = 0;
    int slot = 0;
    int u = 0;
    unsigned long long t_vm = 0, t_bar = 0, t_begin = 0;
    if (VARIANT == 3) t_begin = cycle_stamp();

    // One unit = half a tile (compile-time HALF so that every qa/qb index is static).
#define TS_UNIT(HALF)                                                                                             \
    do {                                                                                                          \
        const int nslot = (slot + 1 == kMfmaSlots) ? 0 : slot + 1;                                                \
        const unsigned char* unit = smem + slot * kMfmaUnitBytes;                                                 \
        const unsigned char* next_unit = smem + nslot * kMfmaUnitBytes;                                           \
        /* certify unit u+1 (own pieces, then everyone's); every wave is past unit u-1: its slot is free */       \
        unsigned long long ts0 = 0, ts1 = 0;                                                                      \
        if (VARIANT == 3) ts0 = cycle_stamp();                                                                    \
        if (u + 1 < nu) TS_WAIT_KEEP(issue_u - (u + 2));                                                          \
        if (VARIANT == 3) ts1 = cycle_stamp();                                                                    \
        __builtin_amdgcn_s_barrier();                                                                             \
        asm volatile("" ::: "memory");                                                                            \
        if (VARIANT == 3) {                                                                                       \
            t_vm += ts1 - ts0;                                                                                    \
            t_bar += cycle_stamp() - ts1;                                                                         \
        }                                                                                                         \
        const bool do_issue = issue_u < nu;                                                                       \
        const unsigned char* isrc = tile_src + (issue_u & 1) * (kMfmaUnitK * 2);                                  \
        const unsigned idst = lds0 + issue_slot * kMfmaUnitBytes;                                                 \
        /* one A fragment feeds both query groups; the ring runs kMfmaAhead k-steps ahead and its tail */         \
        /* already fetches the head of unit u+1; the 6 DMA pieces of unit u+5 go out between MFMAs */             \
        _Pragma("unroll") for (int s = 0; s < kMfmaUnitSteps; ++s) {                                              \
            if (VARIANT == 2) {                                                                                   \
            } else if (HALF == 0 && s == 0) {                                                                     \
                mfma_av_first(acc_a, af[s % kMfmaAhead], qa[HALF * kMfmaUnitSteps + s]);                          \
                mfma_aa_first(acc_b, af[s % kMfmaAhead], qb[HALF * kMfmaUnitSteps + s]);                          \
            } else {                                                                                              \
                mfma_av(acc_a, af[s % kMfmaAhead], qa[HALF * kMfmaUnitSteps + s]);                                \
                mfma_aa(acc_b, af[s % kMfmaAhead], qb[HALF * kMfmaUnitSteps + s]);                                \
            }                                                                                                     \
            const int n = s + kMfmaAhead;                                                                         \
            if (VARIANT == 2) {                                                                                   \
            } else if (n < kMfmaUnitSteps)                                                                        \
                af[s % kMfmaAhead] = *(const bf16x8*)(unit + (n >> 2) * 4096 + xo[n & 3]);                        \
            else                                                                                                  \
                af[s % kMfmaAhead] = *(const bf16x8*)(next_unit + ((n - kMfmaUnitSteps) >> 2) * 4096 +            \
                                                      xo[(n - kMfmaUnitSteps) & 3]);                              \
            if ((s & 3) == 1 && do_issue) lds_dma16(isrc + (s >> 2) * 128, idst + (s >> 2) * 4096);               \
        }                                                                                                         \
        if (do_issue) {                                                                                           \
            if (issue_u & 1) {                                                                                    \
                tile_src += (issue_run_pos + 1 == a.run) ? run_jump : tile_bytes;                                 \
                issue_run_pos = (issue_run_pos + 1 == a.run) ? 0 : issue_run_pos + 1;                             \
            }                                                                                                     \
            ++issue_u;                                                                                            \
            issue_slot = (issue_slot + 1 == kMfmaSlots) ? 0 : issue_slot + 1;                                     \
        }                                                                                                         \
        slot = nslot;                                                                                             \
        ++u;                                                                                                      \
    } while (0)

    for (int t = 0; t < nt; ++t) {
        TS_UNIT(0);
        TS_UNIT(1);
        if (VARIANT == 2) continue;
        mfma_settle(acc_a, acc_b);
        if (VARIANT == 1) {
            asm volatile("" ::"a"(acc_a), "a"(acc_b));
            continue;
        }
        // lane holds rows (g & 3) + 8 (g >> 2) + 4 h of this tile for queries qid_a / qid_b
        const float m_a = max16(acc_a);
        const float m_b = max16(acc_b);
        const bool hit_a = __any(m_a >= thr_a), hit_b = __any(m_b >= thr_b);
        if (__builtin_expect(hit_a || hit_b, 0)) {
            const int64_t lt = t0 + t;  // level tile -> global tile -> first row
            const int64_t tile_row = ((lt / a.run) * a.run * a.tile_stride + lt % a.run) * kTileRows;
            const int64_t row_base = tile_row + 4 * h;
            if (tile_row + kTileRows <= a.n) {
                if (hit_a) mfma_append<true>(acc_a, thr_a, qid_a, writer, nwriters, cnt_a, row_base, a);
                if (hit_b) mfma_append<true>(acc_b, thr_b, qid_b, writer, nwriters, cnt_b, row_base, a);
            } else {
                if (hit_a) mfma_append<false>(acc_a, thr_a, qid_a, writer, nwriters, cnt_a, row_base, a);
                if (hit_b) mfma_append<false>(acc_b, thr_b, qid_b, writer, nwriters, cnt_b, row_base, a);
            }
        }
    }
#undef TS_UNIT
#undef TS_WAIT_KEEP
    a.pcount[(int64_t)qid_a * nwriters + writer] = cnt_a;
    a.pcount[(int64_t)qid_b * nwriters + writer] = cnt_b;
    if (VARIANT == 3 && lane == 0 && a.dbg) {
        unsigned long long* d = a.dbg + ((size_t)blockIdx.x * 4 + wave) * 4;
        d[0] = cycle_stamp() - t_begin;
        d[1] = t_vm;
        d[2] = t_bar;
        d[3] = (unsigned long long)nu;
    }
}

}  // namespace ts
