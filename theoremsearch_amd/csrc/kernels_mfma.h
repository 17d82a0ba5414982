// Batched search, bf16, d = 768 or 1024 (and the narrower 384 / 512): query x corpus contraction on the matrix cores with the top-k
// selection fused behind it.  The [nq x N] score matrix is never written.
//
// Replaces util.cos_sim(q_emb, s_emb) + np.argsort(-sim_matrix, axis=1) of the batched call
// sites (compare_embeddings.py:61,105) at the shapes of BASELINE.json configs[2..3].
//
// Shape of the work (one workgroup = 4 waves = one CU, one wave per SIMD, persistent):
//   * the queries live in REGISTERS for the whole kernel as B operands of v_mfma_f32_32x32x16_bf16:
//     wave w owns group A = queries 32w + r (VGPRs) and, with GROUPS = 2, group B = queries 128 + 32w + r
//     (AGPRs).  d = 768: 2 x 48 k-steps x 4 registers = 384 of the wave's 512 registers, 256 queries per
//     launch (GROUPS = 1 serves up to 128 queries with half the matrix work); d = 1024: one group of
//     64 k-steps = 256 registers, 128 queries per launch;
//   * the corpus streams HBM -> LDS exactly once per CU by LDS-DMA (global_load_lds_dwordx4, non-
//     temporal) in UNITS of 32 rows x 384 k (24 KiB = half a tile; d = 1024: 256 k, 16 KiB, a quarter),
//     six (eight) units deep.  The barrier at
//     the top of unit u certifies unit u+1, so the operand reads of unit u+1 are issued while unit
//     u still computes (no bubble at the seam), and frees the slot of unit u-1 for the next unit to
//     fetch, whose DMA pieces are issued between the MFMAs of unit u;
//   * every wave reads every unit from LDS as the A operand: one ds_read_b128 feeds two MFMAs (one
//     per query group), four k-steps ahead of use;
//   * D[i][j] = <corpus row i, query j>: a lane holds 16 corpus rows for ONE query per group, so
//     the epilogue is a per-lane compare against that query's threshold.  Scores that pass go to a
//     lane-private list in global memory (plain stores, no atomics: a returning atomic would drain
//     the DMA queue); a lane whose private list is full spills to the query's shared list.
//     Thresholds come from the previous, sparser level (mfma_search in search_mfma.hip).
//   * every workgroup walks its own contiguous range of tiles.
//
// LDS image of a unit: 24 (16) pieces of 1 KiB; piece (kb, p) = K-block kb (64 elements = 128 B per row)
// of rows 8p..8p+7 at (kb * 4 + p) * 1024, written by ONE wave-instruction whose lane l fetches row
// 8p + (l >> 3), 16-byte chunk (l & 7) ^ ((row >> 1) & 7) of that K-block: full 128-byte lines from
// HBM, and the XOR on the SOURCE side makes the MFMA operand reads (lane (r, h) reads chunk 2s' + h
// of row r) hit 16 distinct 16-byte slots per ds_read_b128 lane group: conflict-free.
//
// Algorithmic traffic: rows * 2 d bytes per launch; flops 2 * queries * rows * d.
#pragma once
#include "common.h"

// Cache policy of the corpus stream (read once per search): " nt" = non-temporal.
#ifndef TS_DMA_POLICY
#define TS_DMA_POLICY " nt"
#endif

namespace ts {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int kMfmaQ = 256;        // query slots of the per-search buffers (thr, private lists, ...)
constexpr int kMfmaThreads = 256;
constexpr int kMfmaPrivCap = 32;   // entries of a lane-private candidate list

// Per-dimension geometry.  A tile is 32 corpus rows; it streams in kUnits units of kUnitK columns.
template <int D> struct MfmaGeom;
template <> struct MfmaGeom<768> { static constexpr int kUnitK = 384, kSlots = 6; };
template <> struct MfmaGeom<1024> { static constexpr int kUnitK = 256, kSlots = 8; };
// the narrower common widths: same unit image (blocks of 64 columns), two units per tile, two query groups per wave
template <> struct MfmaGeom<512> { static constexpr int kUnitK = 256, kSlots = 8; };
template <> struct MfmaGeom<384> { static constexpr int kUnitK = 192, kSlots = 8; };
// fp32 rows seen as twice as many 2-byte elements (kernels_mfma16.h, F32): 1024 floats = 4096-byte rows, eight 16 KiB units
template <> struct MfmaGeom<2048> { static constexpr int kUnitK = 256, kSlots = 8; };
template <> struct MfmaGeom<1536> { static constexpr int kUnitK = 384, kSlots = 6; };   // 768 floats: 3072-byte rows, four 24 KiB units
// the 16x16x32 kernel (kernels_mfma16.h) walks units of 8 or 12 k-steps of 32: d = 384 streams as ONE unit per tile
// (12 k-steps, 24 KiB, six slots - the image of a half tile of d = 768); the other widths share the geometry above
template <int D> struct MfmaGeom16 : MfmaGeom<D> {};
template <> struct MfmaGeom16<384> { static constexpr int kUnitK = 384, kSlots = 6; };
// the k-split form of the paired d = 1024 pass (kernels_mfma16.h): the same ring as the other forms of that width; 16 KB of LDS
// behind it carry the partial sums the wave pairs exchange (160,016 bytes of the CU's 163,840 in all)
template <int D> struct MfmaGeomKsplit { static constexpr int kUnitK = 256, kSlots = 8; };
template <int D, class Geom = MfmaGeom<D>> struct MfmaDims {
    static constexpr int kKSteps = D / 16;
    static constexpr int kUnitK = Geom::kUnitK;
    static constexpr int kUnitSteps = kUnitK / 16;
    static constexpr int kUnits = D / kUnitK;                     // units per tile
    static constexpr int kUnitBytes = kTileRows * kUnitK * 2;
    static constexpr int kSlots = Geom::kSlots;
    static constexpr int kLds = kSlots * kUnitBytes;
    static constexpr int kPieces = kUnitBytes / 4096;            // DMA pieces per wave per unit
    static constexpr int kPieceEvery = kUnitSteps / kPieces;     // one piece every so many k-steps
    static constexpr int kAhead = 4;                             // A fragments (k-steps) in flight
};
template <int D> using Mfma16Dims = MfmaDims<D, MfmaGeom16<D>>;
constexpr int mfma_queries_per_launch(int d, int groups) { return 128 * groups; }

struct MfmaArgs {
    const unsigned short* corpus;  // bf16 [n_pad x D]
    int64_t n;                     // real rows
    int64_t ntiles;                // tiles visited at this level
    int64_t tile_stride;           // visited tile j is global tile (j / run) * run * tile_stride + j % run:
    int run;                       //   runs of `run` consecutive tiles, `run * tile_stride` tiles apart
    const unsigned short* q;       // bf16 [256 x D], zero rows past nq
    const float* thr;              // [256] pass threshold per query (+inf for absent queries)
    u64* priv;                     // [256][W][kMfmaPrivCap] lane-private lists, W = 2 * gridDim.x writers
    u32* pcount;                   // [256][W] entries each writer produced (may exceed kMfmaPrivCap: spilled)
    u64* cand;                     // [256][cap] shared spill lists
    u32* count;                    // [256]
    int cap;
    const u32* row_mask;           // optional filter: bit (row & 31) of word row >> 5 set = the row may be returned
    int ahead;                     // units kept in flight by the DMA ring (2 .. kSlots - 1; anything else = kSlots - 1)
    int nq;                        // real queries of this launch: waves / groups holding only padding skip the matrix work
    unsigned long long* dbg;       // VARIANT 3 only: per-wave cycle sums
    // first level of a search: thresholds are not read but set here (-inf for the nq_real real queries, +inf for padding)
    // and workgroup 0 resets the per-search counters - the job of a separate one-block launch before
    int first_level;
    int nq_real;
    int* fb_count;
    // full pass of the 16x16 kernel: tile ranges of the workgroups from a table instead of equal shares (part[w] .. part[w+1]),
    // and the time each workgroup took (100 MHz ticks) - the final select moves the boundaries for the next search
    const int64_t* part;
    unsigned* wg_ticks;
    // full pass of the 16x16 kernel, d = 1024 at more than 192 queries: PAIRS of workgroups share a tile range, each half of
    // the pair holding 64 * NB of the queries (workgroups w and w + 8 of a group of 16: the same XCD under round-robin
    // dispatch, so the second reader of a tile finds it in that XCD's L2 / the memory-side cache); part / wg_ticks are
    // then indexed by pair
    int pair;
    // ... and pace each other: pair_pos[2 * pair + half] = the tile that workgroup has reached (kernels_mfma16.h, PAIR);
    // pair_lag = 0 switches the pacing off (a workgroup then never looks at its partner)
    unsigned* pair_pos;
    int pair_lag;
};

__device__ __forceinline__ float mfma_level_thr(const MfmaArgs& a, int qid) {
    if (qid >= a.nq_real) return INFINITY;          // padding queries never produce candidates
    return a.first_level ? -INFINITY : a.thr[qid];
}
__device__ __forceinline__ void mfma_level_begin(const MfmaArgs& a) {
    if (a.first_level && blockIdx.x == 0 && threadIdx.x == 0) {
        *a.fb_count = 0;
    }
}

// One LDS-DMA wave-instruction: 64 lanes x 16 bytes from per-lane global addresses to
// lds_dst + lane * 16 (lds_dst wave-uniform, in M0).  Inline asm on purpose: hipcc keeps no vmcnt
// bookkeeping for it, so it never drains the DMA queue behind our back (a compiler-visible LDS-DMA
// makes every following ds_read wait vmcnt(0)); the waits are the counted ones in the kernel.
// M0 is not saved: the generated code of this kernel never reads M0 (checked in the .s).
// NT = false: default cache policy - the paired pass of kernels_mfma16.h, whose two workgroups read every tile one behind the
// other on one XCD: the first read must leave the lines in L2 for the second.
template <bool NT = true>
__device__ __forceinline__ void lds_dma16(const void* gsrc, unsigned lds_dst) {
    if constexpr (NT)
        asm volatile(
            "s_mov_b32 m0, %1\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %0, off" TS_DMA_POLICY
            :
            : "v"(gsrc), "s"(lds_dst)
            : "memory");
    else
        asm volatile(
            "s_mov_b32 m0, %1\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %0, off"
            :
            : "v"(gsrc), "s"(lds_dst)
            : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// Wait until at most `keep` units (of PIECES DMA instructions each) are still in flight for this wave.
template <int PIECES>
__device__ __forceinline__ void wait_keep_units(int keep) {
    if (keep >= 6) wait_vmcnt<6 * PIECES>();
    else if (keep == 5) wait_vmcnt<5 * PIECES>();
    else if (keep == 4) wait_vmcnt<4 * PIECES>();
    else if (keep == 3) wait_vmcnt<3 * PIECES>();
    else if (keep == 2) wait_vmcnt<2 * PIECES>();
    else if (keep == 1) wait_vmcnt<1 * PIECES>();
    else wait_vmcnt<0>();
}

// MFMA with pinned register classes (left alone the allocator shuttles operands between the VGPR and
// AGPR halves of the file): accumulators and the corpus fragment in AGPRs, query fragment in VGPRs
// (group A) or AGPRs (group B).  The *_first forms start a chain with C = 0.  hipcc knows nothing
// about what is inside, so each statement carries its own wait states: "s_nop 1" in front covers an
// operand the compiler has just written with a VALU move (v_accvgpr_write -> MFMA read needs 2 states;
// hipcc pads nothing for an asm consumer - this produced wrong scores before the pad was added; the
// nop issues in the shadow of the previous MFMA); chains are back-to-back accumulations (no pad
// needed); before any other reader of an accumulator mfma_settle() supplies the wait states.
__device__ __forceinline__ void mfma_av_first(f32x16& acc, const bf16x8& a, const bf16x8& b) {
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&a"(acc) : "a"(a), "v"(b));
}
__device__ __forceinline__ void mfma_av(f32x16& acc, const bf16x8& a, const bf16x8& b) {
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "a"(a), "v"(b));
}
__device__ __forceinline__ void mfma_aa_first(f32x16& acc, const bf16x8& a, const bf16x8& b) {
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&a"(acc) : "a"(a), "a"(b));
}
__device__ __forceinline__ void mfma_aa(f32x16& acc, const bf16x8& a, const bf16x8& b) {
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "a"(a), "a"(b));
}
__device__ __forceinline__ void mfma_settle(f32x16& x) { asm volatile("s_nop 15\n\ts_nop 7" : "+a"(x)); }
__device__ __forceinline__ void mfma_settle(f32x16& x, f32x16& y) { asm volatile("s_nop 15\n\ts_nop 7" : "+a"(x), "+a"(y)); }
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
            // metadata filter: tested only for scores that pass the threshold (a handful per query and pass)
            if (hit && (!a.row_mask || ((a.row_mask[row >> 5] >> (row & 31)) & 1u))) {
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

// D = 768 | 1024; GROUPS = query groups per wave (1: queries 0..127, 2: 0..255; D = 1024 has room for 1).
// VARIANT 0 = the product kernel.  Timing-only diagnostics (wrong results), selected with
// TS_MFMA_VARIANT: 1 = no epilogue, 2 = DMA stream only, 3 = product + cycle stamps into a.dbg.
// SPARSE only changes the symbol: the sample levels show up under their own name in kernel traces, so
// the statistics of the full-corpus pass are not mixed with them.
template <int D, int GROUPS, int VARIANT, bool SPARSE>
__global__ void __launch_bounds__(kMfmaThreads, 1) mfma_topk_kernel(MfmaArgs a) {
    using dims = MfmaDims<D>;
    // loop-attribution diagnostics (timing only): 4 = no epilogue, no barrier; 5 = no epilogue, no vmcnt wait;
    // 6 = neither; 7 = no epilogue, no DMA at all (MFMA + LDS reads + barrier)
    constexpr bool kNoEpi = VARIANT == 1 || (VARIANT >= 4 && VARIANT <= 7);
    constexpr bool kNoBar = VARIANT == 4 || VARIANT == 6;
    constexpr bool kNoVm = VARIANT == 5 || VARIANT == 6 || VARIANT == 7;
    constexpr bool kNoDma = VARIANT == 7;
    constexpr int kKSteps = dims::kKSteps, kUnitSteps = dims::kUnitSteps, kUnits = dims::kUnits;
    constexpr int kUnitBytes = dims::kUnitBytes, kSlots = dims::kSlots, kPieces = dims::kPieces;
    constexpr int kAhead = dims::kAhead, kUnitK = dims::kUnitK;
    static_assert(GROUPS == 1 || (GROUPS == 2 && D <= 768), "two query groups per wave only fit up to d = 768");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int G = gridDim.x;
    const int nwriters = 2 * G;
    const int writer = 2 * blockIdx.x + h;
    const int qid_a = wave * 32 + r, qid_b = 128 + qid_a;
    // small batches: a wave (or its second group) whose 32 queries are all padding keeps feeding the DMA ring and
    // the barriers but issues no MFMAs and no fragment reads - the pass is power-bound, idle matrix work costs time
    const bool idle_a = wave * 32 >= a.nq;
    const bool idle_b = 128 + wave * 32 >= a.nq;  // second query group of this wave (GROUPS == 2)

    mfma_level_begin(a);
    // this workgroup's contiguous share of the level's tiles (sequential pages: a round-robin deal of
    // tiles to workgroups measured 1.4x slower on the DMA stream)
    const int64_t t0 = (a.ntiles * (int64_t)blockIdx.x) / G;
    const int nt = (int)((a.ntiles * (int64_t)(blockIdx.x + 1)) / G - t0);
    if (nt <= 0) {
        a.pcount[(int64_t)qid_a * nwriters + writer] = 0;
        if (GROUPS == 2) a.pcount[(int64_t)qid_b * nwriters + writer] = 0;
        return;
    }
    const int nu = kUnits * nt;  // units

    // group A fragments of k-steps < kAV live in VGPRs, the rest in AGPRs (d = 1024: 256 registers of
    // one group do not fit the 256 architectural VGPRs next to everything else); group B all in AGPRs
    constexpr int kAV = (D <= 768) ? kKSteps : kKSteps / 2;
    bf16x8 qa[kAV], qa_hi[kKSteps > kAV ? kKSteps - kAV : 1], qb[GROUPS == 2 ? kKSteps : 1];
    {
        const bf16x8* pa = (const bf16x8*)(a.q + (int64_t)qid_a * D + 8 * h);
        const bf16x8* pb = (const bf16x8*)(a.q + (int64_t)qid_b * D + 8 * h);
#pragma unroll
        for (int s = 0; s < kKSteps; ++s) {
            if (s < kAV) qa[s] = pa[2 * s];
            else qa_hi[s < kAV ? 0 : s - kAV] = pa[2 * s];
            if (GROUPS == 2) qb[s] = pb[2 * s];
        }
    }
    float thr_a = mfma_level_thr(a, qid_a), thr_b = (GROUPS == 2) ? mfma_level_thr(a, qid_b) : INFINITY;
    // pin: these loads (and the compiler's waits for them) complete here, outside the unit loop,
    // and the fragments stay in the register class the MFMA statements want
#pragma unroll
    for (int s = 0; s < kKSteps; ++s) {
        if (s < kAV) asm volatile("" : "+v"(qa[s]));
        else asm volatile("" : "+a"(qa_hi[s < kAV ? 0 : s - kAV]));
        if (GROUPS == 2) asm volatile("" : "+a"(qb[s]));
    }
    asm volatile("" : "+v"(thr_a));
    asm volatile("" : "+v"(thr_b));

    // DMA source of this lane: row 8w + (lane >> 3) of the tile, swizzled chunk of K-block 0 of the unit
    const int drow = 8 * wave + (lane >> 3);
    const int dchunk = (lane & 7) ^ ((drow >> 1) & 7);
    const int64_t tile_bytes = (int64_t)kTileRows * D * 2;
    const int64_t run_jump = tile_bytes * ((int64_t)a.run * a.tile_stride - a.run + 1);  // last tile of a run -> next run
    const int64_t g0 = (t0 / a.run) * a.run * a.tile_stride + t0 % a.run;               // global tile of level tile t0
    const unsigned char* tile_src = (const unsigned char*)a.corpus + (int64_t)drow * (D * 2) + dchunk * 16 +
                                    g0 * tile_bytes;                              // tile of the next unit to issue
    int issue_run_pos = (int)(t0 % a.run);                                        // its position inside its run
    int issue_u = 0;   // next unit to issue
    int issue_ui = 0;  // its index inside its tile
    int issue_slot = 0;
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem + wave * 1024;

    // per-lane LDS read offsets inside a unit image
    const int lane_off = (r >> 3) * 1024 + (r & 7) * 128;
    const int sw = (r >> 1) & 7;
    int xo[4];
#pragma unroll
    for (int sp = 0; sp < 4; ++sp) xo[sp] = lane_off + (((2 * sp + h) ^ sw) << 4);

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // bookkeeping after the pieces of one unit have been issued
#define TS_ISSUED()                                                                   \
    do {                                                                              \
        if (++issue_ui == kUnits) {                                                   \
            issue_ui = 0;                                                             \
            tile_src += (issue_run_pos + 1 == a.run) ? run_jump : tile_bytes;         \
            issue_run_pos = (issue_run_pos + 1 == a.run) ? 0 : issue_run_pos + 1;     \
        }                                                                             \
        ++issue_u;                                                                    \
        issue_slot = (issue_slot + 1 == kSlots) ? 0 : issue_slot + 1;                 \
    } while (0)

    // prologue: kSlots - 1 units in flight; unit 0 must have landed before its fragments are read
    // at least two: the fragment reads at the end of unit u already fetch the head of unit u + 1, which is certified at the
    // start of unit u only if it was issued a unit earlier
    const int ahead = (a.ahead >= 2 && a.ahead < kSlots) ? a.ahead : kSlots - 1;
    for (int i = 0; i < ahead && issue_u < nu && !kNoDma; ++i) {
        const unsigned char* src = tile_src + issue_ui * (kUnitK * 2);
#pragma unroll
        for (int j = 0; j < kPieces; ++j) lds_dma16(src + j * 128, lds0 + issue_slot * kUnitBytes + j * 4096);
        TS_ISSUED();
    }
    wait_keep_units<kPieces>(issue_u - 1);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    bf16x8 af[kAhead];
#pragma unroll
    for (int s = 0; s < kAhead; ++s) af[s] = *(const bf16x8*)(smem + (s >> 2) * 4096 + xo[s & 3]);  // (idle waves: unused)

    f32x16 acc_a, acc_b;
    u32 cnt_a = 0, cnt_b = 0;
    int slot = 0;
    int u = 0;
    unsigned long long t_vm = 0, t_bar = 0, t_begin = 0;
    if (VARIANT >= 3) t_begin = cycle_stamp();

    // One unit (compile-time index UI inside the tile, so that every qa/qb index is static).
#define TS_UNIT(UI, IDLE, NOB)                                                                                            \
    do {                                                                                                          \
        const int nslot = (slot + 1 == kSlots) ? 0 : slot + 1;                                                    \
        const unsigned char* unit = smem + slot * kUnitBytes;                                                     \
        const unsigned char* next_unit = smem + nslot * kUnitBytes;                                               \
        /* certify unit u+1 (own pieces, then everyone's); every wave is past unit u-1: its slot is free */       \
        unsigned long long ts0 = 0, ts1 = 0;                                                                      \
        if (VARIANT == 3) ts0 = cycle_stamp();                                                                    \
        if (u + 1 < nu && !kNoVm) wait_keep_units<kPieces>(issue_u - (u + 2));                                    \
        if (VARIANT == 3) ts1 = cycle_stamp();                                                                    \
        if (!kNoBar) __builtin_amdgcn_s_barrier();                                                                \
        asm volatile("" ::: "memory");                                                                            \
        if (VARIANT == 3) {                                                                                       \
            t_vm += ts1 - ts0;                                                                                    \
            t_bar += cycle_stamp() - ts1;                                                                         \
        }                                                                                                         \
        const bool do_issue = issue_u < nu && !kNoDma;                                                            \
        const unsigned char* isrc = tile_src + issue_ui * (kUnitK * 2);                                           \
        const unsigned idst = lds0 + issue_slot * kUnitBytes;                                                     \
        /* one A fragment feeds both query groups; the ring runs kAhead k-steps ahead and its tail already */     \
        /* fetches the head of unit u+1; the DMA pieces of the next unit to fetch go out between MFMAs */         \
        _Pragma("unroll") for (int s = 0; s < kUnitSteps; ++s) {                                                  \
            if (VARIANT == 2 || (IDLE)) {                                                                         \
            } else if ((UI) == 0 && s == 0) {                                                                     \
                mfma_av_first(acc_a, af[s % kAhead], qa[0]);                                                      \
                if (GROUPS == 2 && !(NOB)) mfma_aa_first(acc_b, af[s % kAhead], qb[0]);                           \
            } else {                                                                                              \
                constexpr int ks_ = (UI) * kUnitSteps;                                                            \
                if (ks_ + s < kAV) mfma_av(acc_a, af[s % kAhead], qa[ks_ + s < kAV ? ks_ + s : 0]);               \
                else mfma_aa(acc_a, af[s % kAhead], qa_hi[ks_ + s >= kAV ? ks_ + s - kAV : 0]);                   \
                if (GROUPS == 2 && !(NOB)) mfma_aa(acc_b, af[s % kAhead], qb[GROUPS == 2 ? ks_ + s : 0]);         \
            }                                                                                                     \
            const int n = s + kAhead;                                                                             \
            if (VARIANT == 2 || (IDLE)) {                                                                         \
            } else if (n < kUnitSteps)                                                                            \
                af[s % kAhead] = *(const bf16x8*)(unit + (n >> 2) * 4096 + xo[n & 3]);                            \
            else                                                                                                  \
                af[s % kAhead] = *(const bf16x8*)(next_unit + ((n - kUnitSteps) >> 2) * 4096 + xo[(n - kUnitSteps) & 3]); \
            if (s % dims::kPieceEvery == 1 && do_issue)                                                           \
                lds_dma16(isrc + (s / dims::kPieceEvery) * 128, idst + (s / dims::kPieceEvery) * 4096);           \
        }                                                                                                         \
        if (do_issue) TS_ISSUED();                                                                                \
        slot = nslot;                                                                                             \
        ++u;                                                                                                      \
    } while (0)

    // After the units of a tile: settle the accumulators and pass on what beats the thresholds.
    // lane holds rows (g & 3) + 8 (g >> 2) + 4 h of this tile for queries qid_a / qid_b
#define TS_TILE_END(NOB)                                                                                          \
    do {                                                                                                          \
        if (VARIANT == 2) break;                                                                                  \
        if (GROUPS == 2 && !(NOB)) mfma_settle(acc_a, acc_b);                                                     \
        else mfma_settle(acc_a);                                                                                  \
        if (kNoEpi) {                                                                                             \
            if (GROUPS == 2 && !(NOB)) asm volatile("" ::"a"(acc_a), "a"(acc_b));                                 \
            else asm volatile("" ::"a"(acc_a));                                                                   \
            break;                                                                                                \
        }                                                                                                         \
        const bool hit_a = __any(max16(acc_a) >= thr_a);                                                          \
        bool hit_b = false;                                                                                       \
        if (GROUPS == 2 && !(NOB)) hit_b = __any(max16(acc_b) >= thr_b);                                          \
        if (__builtin_expect(hit_a || hit_b, 0)) {                                                                \
            const int64_t lt = t0 + t; /* level tile -> global tile -> first row */                               \
            const int64_t tile_row = (a.run == 1 ? lt * a.tile_stride : (lt / a.run) * a.run * a.tile_stride + lt % a.run) * kTileRows; \
            const int64_t row_base = tile_row + 4 * h;                                                            \
            if (tile_row + kTileRows <= a.n) {                                                                    \
                if (hit_a) mfma_append<true>(acc_a, thr_a, qid_a, writer, nwriters, cnt_a, row_base, a);          \
                if (GROUPS == 2 && !(NOB) && hit_b) mfma_append<true>(acc_b, thr_b, qid_b, writer, nwriters, cnt_b, row_base, a);  \
            } else {                                                                                              \
                if (hit_a) mfma_append<false>(acc_a, thr_a, qid_a, writer, nwriters, cnt_a, row_base, a);         \
                if (GROUPS == 2 && !(NOB) && hit_b) mfma_append<false>(acc_b, thr_b, qid_b, writer, nwriters, cnt_b, row_base, a); \
            }                                                                                                     \
        }                                                                                                         \
    } while (0)

#define TS_TILE_LOOP(IDLE, NOB)                      \
    for (int t = 0; t < nt; ++t) {                   \
        TS_UNIT(0, IDLE, NOB);                       \
        TS_UNIT(1, IDLE, NOB);                       \
        if (kUnits == 4) {                           \
            TS_UNIT(2 % kUnits, IDLE, NOB);          \
            TS_UNIT(3 % kUnits, IDLE, NOB);          \
        }                                            \
        if (!(IDLE)) TS_TILE_END(NOB);               \
    }

    // Three copies of the loop, chosen once per wave, so that the loop of a full batch compiles exactly as it
    // would alone (a runtime test inside the loop cost the full batch 30 %).
    if (idle_a) {
        TS_TILE_LOOP(1, 1)
    } else if (GROUPS == 2 && idle_b) {
        TS_TILE_LOOP(0, 1)
    } else {
        TS_TILE_LOOP(0, 0)
    }
#undef TS_TILE_LOOP
#undef TS_TILE_END
#undef TS_UNIT
#undef TS_ISSUED
    a.pcount[(int64_t)qid_a * nwriters + writer] = cnt_a;
    if (GROUPS == 2) a.pcount[(int64_t)qid_b * nwriters + writer] = cnt_b;
    if (VARIANT >= 3 && lane == 0 && a.dbg) {
        unsigned long long* d = a.dbg + ((size_t)blockIdx.x * 4 + wave) * 4;
        d[0] = cycle_stamp() - t_begin;
        d[1] = t_vm;
        d[2] = t_bar;
        d[3] = (unsigned long long)nu;
    }
}

}  // namespace ts
