// Batched search of an fp32 index (d = 768): the streaming structure of kernels_mfma.h on the exact-fp32 matrix
// instruction v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate: bit for bit an fmaf chain, no reduced precision;
// 64 flop / clock / SIMD = the fp32 vector rate, MI355X_MICROARCH.md "Matrix cores").
//
// Replaces, for batches, util.cos_sim(q_emb, s_emb) + np.argsort(-sim_matrix) of the reference's evaluation
// (compare_embeddings.py:61,105: an fp32 [Q x N] matrix, Q ~ 73) and of the fp32 apps (app_showcase_model.py:93-96),
// which the streaming scan serves 4 queries per pass: 128 fp32 queries per launch here.
//
//   * queries in registers for the whole kernel: wave w owns queries 32 w + r as B operands, one register per query and
//     k-pair: 768 / 2 = 384 registers per wave (the first kQV * 4 in VGPRs, the rest in AGPRs);
//   * the corpus streams HBM -> LDS once per CU through the same LDS-DMA ring and the same swizzled image as the bf16
//     kernel (an fp32 row is 3072 bytes = four units of 768 bytes per row; 32 rows x 768 B = 24 KiB per unit);
//   * one ds_read_b128 (lane (r, h): 16-byte chunk 2 j + h of row r's 128-byte K-block) feeds FOUR MFMAs: MFMA i of the
//     chunk takes float i of the chunk as its A operand and float i of the matching query chunk as B, i.e. it multiplies
//     k = 4 (2 j) + i (lanes 0-31) and k = 4 (2 j + 1) + i (lanes 32-63).  The order in which the products enter the
//     fp32 sum is therefore a fixed permutation of k - deterministic, and as legitimate as any BLAS's order;
//   * accumulator layout = the bf16 32x32 kernel's (C/D maps are dtype-independent), so the threshold epilogue and the
//     candidate lists are shared with it (mfma_append).
//
// The pass is bound by the matrix pipe, not by HBM: 2 * 128 * 768 flops per 3072-byte row = 64 flop/byte against a
// machine balance of 157 TF / 8 TB/s = 20.  Algorithmic traffic rows * 4 d bytes; flops 2 * queries * rows * d.
#pragma once
#include "kernels_mfma.h"

namespace ts {

typedef __attribute__((ext_vector_type(4))) float f32x4v;

constexpr int kMfmaF32Queries = 128;

__device__ __forceinline__ void mfma_f32_v_first(f32x16& acc, float a, float b) {
    asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, 0" : "=&a"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_f32_v(f32x16& acc, float a, float b) {
    asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_f32_a(f32x16& acc, float a, float b) {
    asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "a"(b));
}

struct MfmaF32Dims {
    static constexpr int kRowBytes = 768 * 4;
    static constexpr int kUnitRowBytes = 768;                       // bytes of one row in one unit
    static constexpr int kUnits = kRowBytes / kUnitRowBytes;        // 4 units per tile
    static constexpr int kUnitBytes = kTileRows * kUnitRowBytes;    // 24 KiB
    static constexpr int kSlots = 6;
    static constexpr int kLds = kSlots * kUnitBytes;
    static constexpr int kPieces = kUnitBytes / 4096;               // DMA pieces per wave per unit
    static constexpr int kUnitChunks = kUnitRowBytes / 32;          // chunk-steps (one b128 read, four MFMAs) per unit: 24
    static constexpr int kChunks = kRowBytes / 32;                  // chunk-steps per tile: 96
};

// VARIANT 0 = product; 1 = no epilogue (timing only).  SPARSE only changes the symbol.
template <int VARIANT, bool SPARSE>
__global__ void __launch_bounds__(kMfmaThreads, 1) mfma_f32_topk_kernel(MfmaArgs a) {
    using dims = MfmaF32Dims;
    constexpr int D = 768;
    constexpr int kUnits = dims::kUnits, kUnitBytes = dims::kUnitBytes, kSlots = dims::kSlots, kPieces = dims::kPieces;
    constexpr int kUnitChunks = dims::kUnitChunks, kChunks = dims::kChunks;
    constexpr int kPieceEvery = kUnitChunks / kPieces;              // 4
    constexpr int kQV = 44;                                         // query chunks held in VGPRs (176 registers)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int G = gridDim.x;
    const int nwriters = 2 * G;
    const int writer = 2 * blockIdx.x + h;
    const int qid = wave * 32 + r;

    mfma_level_begin(a);
    const int64_t t0 = (a.ntiles * (int64_t)blockIdx.x) / G;
    const int nt = (int)((a.ntiles * (int64_t)(blockIdx.x + 1)) / G - t0);
    if (nt <= 0) {
        a.pcount[(int64_t)qid * nwriters + writer] = 0;
        return;
    }
    const int nu = kUnits * nt;

    // query chunk c (0..95) = floats 8 c + 4 h .. + 4 of the query row
    f32x4v qv[kQV], qa[kChunks - kQV];
    {
        const f32x4v* pq = (const f32x4v*)((const float*)a.q + (int64_t)qid * D + 4 * h);
#pragma unroll
        for (int c = 0; c < kChunks; ++c) {
            if (c < kQV) qv[c < kQV ? c : 0] = pq[2 * c];
            else qa[c >= kQV ? c - kQV : 0] = pq[2 * c];
        }
    }
    float thr = mfma_level_thr(a, qid);
#pragma unroll
    for (int c = 0; c < kChunks; ++c) {
        if (c < kQV) asm volatile("" : "+v"(qv[c < kQV ? c : 0]));
        else asm volatile("" : "+a"(qa[c >= kQV ? c - kQV : 0]));
    }
    asm volatile("" : "+v"(thr));

    // DMA source of this lane: row 8 w + (lane >> 3) of the tile, swizzled 16-byte chunk of K-block 0 of the unit
    const int drow = 8 * wave + (lane >> 3);
    const int dchunk = (lane & 7) ^ ((drow >> 1) & 7);
    const int64_t tile_bytes = (int64_t)kTileRows * dims::kRowBytes;
    const int64_t run_jump = tile_bytes * ((int64_t)a.run * a.tile_stride - a.run + 1);
    const int64_t g0 = (t0 / a.run) * a.run * a.tile_stride + t0 % a.run;
    const unsigned char* tile_src = (const unsigned char*)a.corpus + (int64_t)drow * dims::kRowBytes + dchunk * 16 + g0 * tile_bytes;
    int issue_run_pos = (int)(t0 % a.run);
    int issue_u = 0, issue_ui = 0, issue_slot = 0;
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem + wave * 1024;

    // operand read offsets inside a unit image: chunk-step s -> K-block s >> 2, chunk 2 (s & 3) + h
    const int lane_off = (r >> 3) * 1024 + (r & 7) * 128;
    const int sw = (r >> 1) & 7;
    int xo[4];
#pragma unroll
    for (int sp = 0; sp < 4; ++sp) xo[sp] = lane_off + (((2 * sp + h) ^ sw) << 4);

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

#define TSF_ISSUED()                                                                  \
    do {                                                                              \
        if (++issue_ui == kUnits) {                                                   \
            issue_ui = 0;                                                             \
            tile_src += (issue_run_pos + 1 == a.run) ? run_jump : tile_bytes;         \
            issue_run_pos = (issue_run_pos + 1 == a.run) ? 0 : issue_run_pos + 1;     \
        }                                                                             \
        ++issue_u;                                                                    \
        issue_slot = (issue_slot + 1 == kSlots) ? 0 : issue_slot + 1;                 \
    } while (0)

    // at least two: the fragment reads at the end of unit u already fetch the head of unit u + 1, which is certified at the
    // start of unit u only if it was issued a unit earlier
    const int ahead = (a.ahead >= 2 && a.ahead < kSlots) ? a.ahead : kSlots - 1;
    for (int i = 0; i < ahead && issue_u < nu; ++i) {
        const unsigned char* src = tile_src + issue_ui * dims::kUnitRowBytes;
#pragma unroll
        for (int j = 0; j < kPieces; ++j) lds_dma16(src + j * 128, lds0 + issue_slot * kUnitBytes + j * 4096);
        TSF_ISSUED();
    }
    wait_keep_units<kPieces>(issue_u - 1);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    f32x4v af[2];
    af[0] = *(const f32x4v*)(smem + xo[0]);
    af[1] = *(const f32x4v*)(smem + xo[1]);

    f32x16 acc;
    u32 cnt = 0;
    int slot = 0, u = 0;

    // the four MFMAs of chunk-step S_ of unit UI (compile-time), then the reload of its ring register for step S_ + 2
#define TSF_STEP(UI, S_)                                                                                   \
    do {                                                                                                   \
        constexpr int c_ = (UI) * kUnitChunks + (S_);                                                      \
        constexpr int ri_ = (S_) & 1;                                                                      \
        if constexpr (c_ == 0) {                                                                           \
            mfma_f32_v_first(acc, af[ri_][0], qv[0][0]);                                                   \
            mfma_f32_v(acc, af[ri_][1], qv[0][1]);                                                         \
            mfma_f32_v(acc, af[ri_][2], qv[0][2]);                                                         \
            mfma_f32_v(acc, af[ri_][3], qv[0][3]);                                                         \
        } else if constexpr (c_ < kQV) {                                                                   \
            mfma_f32_v(acc, af[ri_][0], qv[c_ < kQV ? c_ : 0][0]);                                         \
            mfma_f32_v(acc, af[ri_][1], qv[c_ < kQV ? c_ : 0][1]);                                         \
            mfma_f32_v(acc, af[ri_][2], qv[c_ < kQV ? c_ : 0][2]);                                         \
            mfma_f32_v(acc, af[ri_][3], qv[c_ < kQV ? c_ : 0][3]);                                         \
        } else {                                                                                           \
            mfma_f32_a(acc, af[ri_][0], qa[c_ >= kQV ? c_ - kQV : 0][0]);                                  \
            mfma_f32_a(acc, af[ri_][1], qa[c_ >= kQV ? c_ - kQV : 0][1]);                                  \
            mfma_f32_a(acc, af[ri_][2], qa[c_ >= kQV ? c_ - kQV : 0][2]);                                  \
            mfma_f32_a(acc, af[ri_][3], qa[c_ >= kQV ? c_ - kQV : 0][3]);                                  \
        }                                                                                                  \
        constexpr int n_ = (S_) + 2;                                                                       \
        if constexpr (n_ < kUnitChunks) af[ri_] = *(const f32x4v*)(unit + (n_ >> 2) * 4096 + xo[n_ & 3]); \
        else af[ri_] = *(const f32x4v*)(next_unit + ((n_ - kUnitChunks) >> 2) * 4096 + xo[n_ & 3]);        \
        if constexpr ((S_) % kPieceEvery == 1)                                                             \
            if (do_issue) lds_dma16(isrc + ((S_) / kPieceEvery) * 128, idst + ((S_) / kPieceEvery) * 4096); \
    } while (0)

#define TSF_UNIT(UI)                                                                                       \
    do {                                                                                                   \
        const int nslot = (slot + 1 == kSlots) ? 0 : slot + 1;                                             \
        const unsigned char* unit = smem + slot * kUnitBytes;                                              \
        const unsigned char* next_unit = smem + nslot * kUnitBytes;                                        \
        if (u + 1 < nu) wait_keep_units<kPieces>(issue_u - (u + 2));                                       \
        __builtin_amdgcn_s_barrier();                                                                      \
        asm volatile("" ::: "memory");                                                                     \
        const bool do_issue = issue_u < nu;                                                                \
        const unsigned char* isrc = tile_src + issue_ui * dims::kUnitRowBytes;                             \
        const unsigned idst = lds0 + issue_slot * kUnitBytes;                                              \
        TSF_STEP(UI, 0); TSF_STEP(UI, 1); TSF_STEP(UI, 2); TSF_STEP(UI, 3); TSF_STEP(UI, 4); TSF_STEP(UI, 5);       \
        TSF_STEP(UI, 6); TSF_STEP(UI, 7); TSF_STEP(UI, 8); TSF_STEP(UI, 9); TSF_STEP(UI, 10); TSF_STEP(UI, 11);    \
        TSF_STEP(UI, 12); TSF_STEP(UI, 13); TSF_STEP(UI, 14); TSF_STEP(UI, 15); TSF_STEP(UI, 16); TSF_STEP(UI, 17); \
        TSF_STEP(UI, 18); TSF_STEP(UI, 19); TSF_STEP(UI, 20); TSF_STEP(UI, 21); TSF_STEP(UI, 22); TSF_STEP(UI, 23); \
        if (do_issue) TSF_ISSUED();                                                                        \
        slot = nslot;                                                                                      \
        ++u;                                                                                               \
    } while (0)

    static_assert(kUnitChunks == 24 && kUnits == 4, "unit = 24 chunk-steps, 4 units per tile");
    for (int t = 0; t < nt; ++t) {
        TSF_UNIT(0);
        TSF_UNIT(1);
        TSF_UNIT(2);
        TSF_UNIT(3);
        mfma_settle(acc);
        if (VARIANT == 1) {
            asm volatile("" ::"a"(acc));
            continue;
        }
        // lane holds rows (g & 3) + 8 (g >> 2) + 4 h of this tile for query qid
        if (__builtin_expect(__any(max16(acc) >= thr), 0)) {
            const int64_t lt = t0 + t;
            // (64-bit divisions are hundreds of instructions: runs of one tile - the default - take the short way)
            const int64_t tile_row = (a.run == 1 ? lt * a.tile_stride : (lt / a.run) * a.run * a.tile_stride + lt % a.run) * kTileRows;
            const int64_t row_base = tile_row + 4 * h;
            if (tile_row + kTileRows <= a.n) mfma_append<true>(acc, thr, qid, writer, nwriters, cnt, row_base, a);
            else mfma_append<false>(acc, thr, qid, writer, nwriters, cnt, row_base, a);
        }
    }
#undef TSF_UNIT
#undef TSF_STEP
#undef TSF_ISSUED
    a.pcount[(int64_t)qid * nwriters + writer] = cnt;
}

}  // namespace ts
