// Batched search, bf16, d = 768: query x corpus contraction on the matrix cores with the top-k
// selection fused behind it.  The [nq x N] score matrix is never written.
//
// Replaces util.cos_sim(q_emb, s_emb) + np.argsort(-sim_matrix, axis=1) of the batched call
// sites (compare_embeddings.py:61,105) at the shapes of BASELINE.json configs[2..3].
//
// Shape of the work (one workgroup = 8 waves = one CU, persistent over a contiguous tile range):
//   * the 256 queries live in REGISTERS for the whole kernel: wave w owns queries 32w..32w+31 as
//     the B operand of v_mfma_f32_32x32x16_bf16 (48 k-steps x 4 VGPRs = 192 VGPRs per lane);
//   * the corpus streams HBM -> LDS exactly once per CU by LDS-DMA (global_load_lds_dwordx4) in
//     tiles of 32 rows x 768 (48 KiB), three tiles deep; every wave reads every tile from LDS as the
//     A operand (one ds_read_b128 per MFMA);
//   * D[i][j] = <corpus row i, query j>: a lane holds 16 corpus rows for ONE query (column = lane
//     & 31), so the epilogue is a per-lane compare against that query's threshold; scores that
//     pass are appended to the query's candidate list in global memory (rare: thresholds come
//     from the previous, sparser level - see mfma_search in tsearch_api.hip).
//
// LDS image of one tile: 48 pieces of 1 KiB; piece (kb, p) = K-block kb (64 elements = 128 B per
// row) of rows 8p..8p+7, written by ONE wave-instruction whose lane l fetches row 8p + (l >> 3),
// 16-byte chunk (l & 7) ^ ((row >> 1) & 7) of that K-block: full 128-byte lines from HBM, and the
// XOR on the SOURCE side makes the MFMA operand reads (lane (r, h) reads chunk 2s' + h of row r)
// hit 16 distinct 16-byte slots per ds_read_b128 lane group: conflict-free.
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
constexpr int kMfmaQ = 256;                        // queries per launch
constexpr int kMfmaThreads = 512;
constexpr int kMfmaKSteps = kMfmaD / 16;           // 48
constexpr int kMfmaTileBytes = kTileRows * kMfmaD * 2;  // 49152
constexpr int kMfmaStages = 3;
constexpr int kMfmaLds = kMfmaStages * kMfmaTileBytes;  // 147456
constexpr int kMfmaPiecesPerWave = 48 / 8;
constexpr int kMfmaAhead = 4;                      // A fragments in flight per wave

struct MfmaArgs {
    const unsigned short* corpus;  // bf16 [n_pad x 768]
    int64_t n;                     // real rows
    int64_t ntiles;                // tiles visited at this level
    int64_t tile_stride;           // visited tile j is global tile j * tile_stride
    const unsigned short* q;       // bf16 [256 x 768], zero rows past nq
    const float* thr;              // [256] pass threshold per query (+inf for absent queries)
    u64* cand;                     // [256][cap]
    u32* count;                    // [256]
    int cap;
    unsigned long long* dbg;       // diagnostics only (VARIANT 5 of the v2 kernel): per-wave cycle sums
};

// One LDS-DMA wave-instruction: 64 lanes x 16 bytes from per-lane global addresses to
// lds_dst + lane * 16 (lds_dst wave-uniform).  Written as inline asm on purpose: hipcc keeps no
// vmcnt bookkeeping for it, so it never drains the DMA queue behind our back (a compiler-visible
// LDS-DMA makes every following ds_read wait vmcnt(0)); the waits are the counted ones below.
__device__ __forceinline__ void lds_dma16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off" TS_DMA_POLICY "\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_dst)
        : "memory");
}

// The same without saving M0: for kernels whose generated code never touches M0 (checked in the .s).
__device__ __forceinline__ void lds_dma16_m0(const void* gsrc, unsigned lds_dst) {
    asm volatile(
        "s_mov_b32 m0, %1\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %0, off" TS_DMA_POLICY
        :
        : "v"(gsrc), "s"(lds_dst)
        : "memory");
}

// This wave's 6 of the 48 pieces of one tile.  Piece pi = wave + 8 j: K-block kb = pi >> 2 =
// (wave >> 2) + 2 j, row group p = wave & 3 - so the lane's row and chunk are the same for all j
// and the source address advances by 2 K-blocks = 256 bytes per piece.
__device__ __forceinline__ void mfma_issue_tile(const unsigned char* lane_src /* tile row 0 + this lane's row/chunk */,
                                                unsigned stage_lds /* LDS byte address of the stage + wave * 1024 */) {
#pragma unroll
    for (int j = 0; j < kMfmaPiecesPerWave; ++j) lds_dma16(lane_src + j * 256, stage_lds + j * 8192);
}

// VARIANT 0 = the product kernel.  1..3 are timing-only diagnostics (wrong results) selected with
// TS_MFMA_VARIANT: 1 = no LDS-DMA (MFMA + LDS reads on stale LDS), 2 = no MFMA/LDS reads (DMA stream
// only), 3 = no epilogue.
template <int VARIANT>
__global__ void __launch_bounds__(kMfmaThreads, 2) mfma_topk_kernel(MfmaArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;

    // this workgroup's contiguous share of the level's tiles
    const int64_t t0 = (a.ntiles * (int64_t)blockIdx.x) / gridDim.x;
    const int64_t t1 = (a.ntiles * (int64_t)(blockIdx.x + 1)) / gridDim.x;
    const int nt = (int)(t1 - t0);
    if (nt <= 0) return;

    // queries -> registers (B operand: lane (r, h), k-step s holds Q[32w + r][16s + 8h .. +8])
    const int qid = wave * 32 + r;
    bf16x8 qf[kMfmaKSteps];
    {
        const bf16x8* qp = (const bf16x8*)(a.q + (int64_t)qid * kMfmaD + 8 * h);
#pragma unroll
        for (int s = 0; s < kMfmaKSteps; ++s) qf[s] = qp[2 * s];
    }
    float thr = a.thr[qid];
    // Pin the query registers here: the compiler must finish these loads (and its own vmcnt waits for
    // them) BEFORE the tile loop, and may not re-load them inside it.
#pragma unroll
    for (int s = 0; s < kMfmaKSteps; ++s) asm volatile("" : "+v"(qf[s]));
    asm volatile("" : "+v"(thr));

    // DMA source of this lane inside a tile: row 8p + (lane >> 3), K-block (wave >> 2), swizzled chunk
    const int drow = 8 * (wave & 3) + (lane >> 3);
    const int dchunk = (lane & 7) ^ ((drow >> 1) & 7);
    const unsigned char* dma_src = (const unsigned char*)a.corpus + (int64_t)drow * (kMfmaD * 2) + (wave >> 2) * 128 + dchunk * 16;
    const int64_t tile_bytes = (int64_t)kTileRows * kMfmaD * 2 * a.tile_stride;  // distance between visited tiles
    const unsigned char* next_src = dma_src + t0 * tile_bytes;                   // tile to issue next
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem + wave * 1024;

    // per-lane LDS read offsets inside a tile image
    const int lane_off = (r >> 3) * 1024 + (r & 7) * 128;
    const int sw = (r >> 1) & 7;
    int xo[4];
#pragma unroll
    for (int sp = 0; sp < 4; ++sp) xo[sp] = lane_off + (((2 * sp + h) ^ sw) << 4);

    // wait until the Q loads above have landed before any DMA is in flight: from here on the only
    // compiler-visible vector-memory operations are the rare candidate appends
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // prologue: two tiles in flight
    if (VARIANT != 1) mfma_issue_tile(next_src, lds0);
    next_src += tile_bytes;
    if (nt > 1) {
        if (VARIANT != 1) mfma_issue_tile(next_src, lds0 + kMfmaTileBytes);
        next_src += tile_bytes;
    }

    int stage = 0;
    for (int it = 0; it < nt; ++it) {
        // tile `it` has landed for this wave's pieces; the barrier extends that to every wave's
        if (it + 1 < nt)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kMfmaPiecesPerWave) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // every wave has finished reading the stage used two iterations ago: refill it
        if (it + 2 < nt) {
            int s2 = stage + 2;
            if (s2 >= kMfmaStages) s2 -= kMfmaStages;
            if (VARIANT != 1) mfma_issue_tile(next_src, lds0 + s2 * kMfmaTileBytes);
            next_src += tile_bytes;
        }
        if (VARIANT == 2) {
            stage = (stage + 1 == kMfmaStages) ? 0 : stage + 1;
            continue;
        }

        const unsigned char* tile = smem + stage * kMfmaTileBytes;
        f32x16 acc;
#pragma unroll
        for (int g = 0; g < 16; ++g) acc[g] = 0.0f;
        // A fragments run kMfmaAhead k-steps ahead of the MFMA that consumes them
        bf16x8 af[kMfmaAhead];
#pragma unroll
        for (int s = 0; s < kMfmaAhead; ++s) af[s] = *(const bf16x8*)(tile + (s >> 2) * 4096 + xo[s & 3]);
#pragma unroll
        for (int s = 0; s < kMfmaKSteps; ++s) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s % kMfmaAhead], qf[s], acc, 0, 0, 0);
            if (s + kMfmaAhead < kMfmaKSteps)
                af[s % kMfmaAhead] = *(const bf16x8*)(tile + ((s + kMfmaAhead) >> 2) * 4096 + xo[(s + kMfmaAhead) & 3]);
        }

        // epilogue: lane holds rows (g & 3) + 8 (g >> 2) + 4 h of this tile for query qid
        float m = acc[0];
#pragma unroll
        for (int g = 1; g < 16; ++g) m = fmaxf(m, acc[g]);
        if (VARIANT == 3) {
            asm volatile("" ::"v"(m));
        } else if (__any(m >= thr)) {
            const int64_t row_base = (t0 + it) * a.tile_stride * kTileRows + 4 * h;
            int nhit = 0;
#pragma unroll
            for (int g = 0; g < 16; ++g) nhit += (acc[g] >= thr && row_base + (g & 3) + 8 * (g >> 2) < a.n) ? 1 : 0;
            if (nhit) {
                u32 pos = atomicAdd(&a.count[qid], (u32)nhit);
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const int64_t row = row_base + (g & 3) + 8 * (g >> 2);
                    if (acc[g] >= thr && row < a.n) {
                        if (pos < (u32)a.cap) a.cand[(int64_t)qid * a.cap + pos] = make_key(acc[g], (u32)row);
                        ++pos;
                    }
                }
            }
        }
        stage = (stage + 1 == kMfmaStages) ? 0 : stage + 1;
    }
}

// =================================================================================================
// v2: one wave per SIMD.  Same data layout idea, different work split:
//   * 4 waves per workgroup, 512 registers each; wave w owns queries 64w..64w+63 as TWO B-operand
//     groups (qa: 64w + r, qb: 64w + 32 + r), 2 x 192 registers;
//   * the corpus streams in UNITS of 32 rows x 384 k (24 KiB, half a tile), six units deep; a
//     unit's barrier certifies the NEXT unit, so the operand reads of unit u+1 can be issued while
//     unit u still computes: no bubble at the seam;
//   * per unit a wave runs 24 MFMAs into acc_a and then 24 into acc_b, re-reading the A fragments
//     from LDS; the epilogue of one group is placed in the other group's MFMA shadow.
// LDS image of a unit: 24 pieces of 1 KiB, piece (kb, p) at (kb * 4 + p) * 1024, same swizzle.
// =================================================================================================
constexpr int kV2Threads = 256;
constexpr int kV2UnitK = 384;
constexpr int kV2UnitSteps = kV2UnitK / 16;                 // 24
constexpr int kV2UnitBytes = kTileRows * kV2UnitK * 2;      // 24576
constexpr int kV2Slots = 6;
constexpr int kV2Lds = kV2Slots * kV2UnitBytes;             // 147456
constexpr int kV2Pieces = 6;                                // per wave per unit
#ifndef TS_V2_AHEAD
#define TS_V2_AHEAD 4
#endif
constexpr int kV2Ahead = TS_V2_AHEAD;                                 // A fragments (k-steps) in flight

__device__ __forceinline__ void v2_issue_unit(const unsigned char* lane_src, unsigned slot_lds) {
#pragma unroll
    for (int j = 0; j < kV2Pieces; ++j) lds_dma16(lane_src + j * 128, slot_lds + j * 4096);
}

__device__ __forceinline__ void v2_append(const f32x16& acc, float thr, int qid, int64_t row_base, const MfmaArgs& a) {
    int nhit = 0;
#pragma unroll
    for (int g = 0; g < 16; ++g) nhit += (acc[g] >= thr && row_base + (g & 3) + 8 * (g >> 2) < a.n) ? 1 : 0;
    if (nhit) {
        u32 pos = atomicAdd(&a.count[qid], (u32)nhit);
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const int64_t row = row_base + (g & 3) + 8 * (g >> 2);
            if (acc[g] >= thr && row < a.n) {
                if (pos < (u32)a.cap) a.cand[(int64_t)qid * a.cap + pos] = make_key(acc[g], (u32)row);
                ++pos;
            }
        }
    }
}

__device__ __forceinline__ float max16(const f32x16& acc) {
    float m = acc[0];
#pragma unroll
    for (int g = 1; g < 16; ++g) m = fmaxf(m, acc[g]);
    return m;
}

// MFMA with pinned register classes (the allocator otherwise shuttles operands between the VGPR and
// AGPR halves of the file): accumulators in AGPRs, corpus fragment in VGPRs, query fragment in VGPRs
// (group A) or AGPRs (group B).  The *_first forms start a chain with C = 0.  hipcc knows nothing
// about what is inside: chains are back-to-back accumulations (no pad needed); before any other
// reader of an accumulator v2_acc_settle() supplies the wait states.
__device__ __forceinline__ void mfma_vv_first(f32x16& acc, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&a"(acc) : "a"(a), "v"(b));
}
__device__ __forceinline__ void mfma_vv(f32x16& acc, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "a"(a), "v"(b));
}
__device__ __forceinline__ void mfma_va_first(f32x16& acc, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&a"(acc) : "a"(a), "a"(b));
}
__device__ __forceinline__ void mfma_va(f32x16& acc, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "a"(a), "a"(b));
}
typedef __attribute__((ext_vector_type(4))) float f32x4;
__device__ __forceinline__ void mfma16_vv(f32x4& acc, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "a"(a), "v"(b));
}
__device__ __forceinline__ void mfma16_va(f32x4& acc, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "a"(a), "a"(b));
}
__device__ __forceinline__ unsigned long long v2_stamp() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
__device__ __forceinline__ void v2_acc_settle(f32x16& x, f32x16& y) {
    asm volatile("s_nop 15\n\ts_nop 7" : "+a"(x), "+a"(y));
}

// VARIANT: 0 product; timing-only diagnostics (no epilogue): 1 no DMA, 2 DMA only, 3 no LDS reads, 4 MFMA only, 5 stamps, 6 half the MFMAs (group A only), 7 quarter (every other k-step of group A).
template <int VARIANT>
__global__ void __launch_bounds__(kV2Threads, 1) mfma_topk_v2_kernel(MfmaArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;

    const int64_t t0 = (a.ntiles * (int64_t)blockIdx.x) / gridDim.x;
    const int64_t t1 = (a.ntiles * (int64_t)(blockIdx.x + 1)) / gridDim.x;
    const int nt = (int)(t1 - t0);
    if (nt <= 0) return;
    const int nu = 2 * nt;  // units

    const int qid_a = wave * 64 + r, qid_b = qid_a + 32;
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
    // pin: loads (and the compiler's waits for them) complete here, outside the unit loop
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
    const unsigned char* tile_src = (const unsigned char*)a.corpus + (int64_t)drow * (kMfmaD * 2) + dchunk * 16;
    const int64_t tile_bytes = (int64_t)kTileRows * kMfmaD * 2 * a.tile_stride;
    tile_src += t0 * tile_bytes;  // tile of the next unit to issue
    int issue_u = 0;              // next unit to issue
    int issue_slot = 0;
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem + wave * 1024;

    const int lane_off = (r >> 3) * 1024 + (r & 7) * 128;
    const int sw = (r >> 1) & 7;
    int xo[4];
#pragma unroll
    for (int sp = 0; sp < 4; ++sp) xo[sp] = lane_off + (((2 * sp + h) ^ sw) << 4);

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

#define V2_ISSUE()                                                                                  \
    do {                                                                                            \
        v2_issue_unit(tile_src + (issue_u & 1) * (kV2UnitK * 2), lds0 + issue_slot * kV2UnitBytes); \
        if (issue_u & 1) tile_src += tile_bytes;                                                    \
        ++issue_u;                                                                                  \
        issue_slot = (issue_slot + 1 == kV2Slots) ? 0 : issue_slot + 1;                             \
    } while (0)
#define V2_WAIT_KEEP(units)                                                     \
    do {                                                                        \
        const int keep_ = (units);                                              \
        if (keep_ >= 4) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");       \
        else if (keep_ == 3) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");  \
        else if (keep_ == 2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");  \
        else if (keep_ == 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   \
    } while (0)

    // prologue: five units in flight (slots 0..4); unit 0 must have landed before its fragments are read
    if (VARIANT != 1 && VARIANT != 4)
        for (int i = 0; i < 5 && issue_u < nu; ++i) V2_ISSUE();
    else
        issue_u = nu < 5 ? nu : 5, issue_slot = issue_u % kV2Slots;
    V2_WAIT_KEEP(issue_u - 1);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    bf16x8 af[kV2Ahead];
#pragma unroll
    for (int s = 0; s < kV2Ahead; ++s) af[s] = *(const bf16x8*)(smem + (s >> 2) * 4096 + xo[s & 3]);

    f32x16 acc_a, acc_b;
    int slot = 0;
    int u = 0;
    unsigned long long t_vm = 0, t_bar = 0, t_begin = 0;
    f32x4 p0 = {0, 0, 0, 0}, p1 = {0, 0, 0, 0}, p2 = {0, 0, 0, 0}, p3 = {0, 0, 0, 0};
    if (VARIANT == 5) t_begin = v2_stamp();
    // One unit = half a tile (compile-time HALF so that every qa/qb index is static).
#define V2_UNIT(HALF)                                                                                             \
    do {                                                                                                          \
        const int nslot = (slot + 1 == kV2Slots) ? 0 : slot + 1;                                                  \
        const unsigned char* unit = smem + slot * kV2UnitBytes;                                                   \
        const unsigned char* next_unit = smem + nslot * kV2UnitBytes;                                             \
        /* certify unit u+1 (own pieces, then everyone's); every wave is past unit u-1: its slot is free */       \
        unsigned long long ts0 = 0, ts1 = 0, ts2 = 0;                                                             \
        if (VARIANT == 5) ts0 = v2_stamp();                                                                       \
        if (u + 1 < nu) V2_WAIT_KEEP(issue_u - (u + 2));                                                          \
        if (VARIANT == 5) ts1 = v2_stamp();                                                                       \
        __builtin_amdgcn_s_barrier();                                                                             \
        asm volatile("" ::: "memory");                                                                            \
        if (VARIANT == 5) {                                                                                       \
            ts2 = v2_stamp();                                                                                     \
            t_vm += ts1 - ts0;                                                                                    \
            t_bar += ts2 - ts1;                                                                                   \
        }                                                                                                         \
        const bool do_issue = issue_u < nu;                                                                       \
        const unsigned char* isrc = tile_src + (issue_u & 1) * (kV2UnitK * 2);                                    \
        const unsigned idst = lds0 + issue_slot * kV2UnitBytes;                                                   \
        /* one A fragment feeds both query groups; the ring runs kV2Ahead k-steps ahead and its tail */           \
        /* already fetches the head of unit u+1; the 6 DMA pieces of unit u+5 go out between MFMAs */             \
        _Pragma("unroll") for (int s = 0; s < kV2UnitSteps; ++s) {                                                \
            if (VARIANT == 2) {                                                                                   \
            } else if (VARIANT == 8) { /* power probe: same flops as two 32x32x16, issued as four 16x16x32 */     \
                mfma16_vv(p0, af[s % kV2Ahead], qa[HALF * kV2UnitSteps + s]);                                     \
                mfma16_va(p1, af[s % kV2Ahead], qb[HALF * kV2UnitSteps + s]);                                     \
                mfma16_vv(p2, af[s % kV2Ahead], qa[HALF * kV2UnitSteps + s]);                                     \
                mfma16_va(p3, af[s % kV2Ahead], qb[HALF * kV2UnitSteps + s]);                                     \
            } else if (HALF == 0 && s == 0) {                                                                     \
                mfma_vv_first(acc_a, af[s % kV2Ahead], qa[HALF * kV2UnitSteps + s]);                              \
                mfma_va_first(acc_b, af[s % kV2Ahead], qb[HALF * kV2UnitSteps + s]);                              \
            } else {                                                                                              \
                if (VARIANT != 7 || (s & 1)) mfma_vv(acc_a, af[s % kV2Ahead], qa[HALF * kV2UnitSteps + s]);       \
                if (VARIANT < 6) mfma_va(acc_b, af[s % kV2Ahead], qb[HALF * kV2UnitSteps + s]);                   \
            }                                                                                                     \
            const int n = s + kV2Ahead;                                                                           \
            if (VARIANT >= 2 && VARIANT != 5 && VARIANT != 9) {                                                                 \
            } else if (n < kV2UnitSteps)                                                                          \
                af[s % kV2Ahead] = *(const bf16x8*)(unit + (n >> 2) * 4096 + xo[n & 3]);                          \
            else                                                                                                  \
                af[s % kV2Ahead] =                                                                                \
                    *(const bf16x8*)(next_unit + ((n - kV2UnitSteps) >> 2) * 4096 + xo[(n - kV2UnitSteps) & 3]);  \
            if (VARIANT != 1 && VARIANT != 4 && (s & 3) == 1 && do_issue) lds_dma16_m0(isrc + (s >> 2) * 128, idst + (s >> 2) * 4096); \
        }                                                                                                         \
        if (do_issue) {                                                                                           \
            if (issue_u & 1) tile_src += tile_bytes;                                                              \
            ++issue_u;                                                                                            \
            issue_slot = (issue_slot + 1 == kV2Slots) ? 0 : issue_slot + 1;                                       \
        }                                                                                                         \
        slot = nslot;                                                                                             \
        ++u;                                                                                                      \
    } while (0)

    for (int t = 0; t < nt; ++t) {
        V2_UNIT(0);
        V2_UNIT(1);
        if (VARIANT == 2) continue;
        v2_acc_settle(acc_a, acc_b);
        if (VARIANT == 8) {
            asm volatile("" ::"a"(p0), "a"(p1), "a"(p2), "a"(p3));
            continue;
        }
        if (VARIANT != 0 && VARIANT != 5) {
            asm volatile("" ::"a"(acc_a), "a"(acc_b));
            continue;
        }
        const int64_t row_base = (t0 + t) * a.tile_stride * kTileRows + 4 * h;
        const float m_a = max16(acc_a);
        const float m_b = max16(acc_b);
        if (__builtin_expect(__any(m_a >= thr_a || m_b >= thr_b), 0)) {
            v2_append(acc_a, thr_a, qid_a, row_base, a);
            v2_append(acc_b, thr_b, qid_b, row_base, a);
        }
    }
#undef V2_UNIT
    if (VARIANT == 5 && lane == 0 && a.dbg) {
        unsigned long long* d = a.dbg + ((size_t)blockIdx.x * 4 + wave) * 4;
        d[0] = v2_stamp() - t_begin;
        d[1] = t_vm;
        d[2] = t_bar;
        d[3] = (unsigned long long)nu;
    }
#undef V2_ISSUE
#undef V2_WAIT_KEEP
}

}  // namespace ts
