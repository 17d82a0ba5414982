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
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
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
    mfma_issue_tile(next_src, lds0);
    next_src += tile_bytes;
    if (nt > 1) {
        mfma_issue_tile(next_src, lds0 + kMfmaTileBytes);
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
            mfma_issue_tile(next_src, lds0 + s2 * kMfmaTileBytes);
            next_src += tile_bytes;
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
        if (__any(m >= thr)) {
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

}  // namespace ts
