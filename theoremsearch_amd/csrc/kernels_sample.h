// The threshold sample of the batched search: scores of every query against a sparse sample of the corpus (every
// `tile_stride`-th tile of 32 rows, at most 8192 rows), written as a dense [query][sample position] fp32 matrix that
// sample_select_kernel (kernels_select.h) turns into the pass thresholds of the full-corpus pass.
//
// Round 2 ran the full-pass kernel itself over the sample (SPARSE instantiation: 393 KB of query fragments per workgroup
// for ONE tile of work, every score appended to lane-private lists that the select then gathered from 1,024 writers):
// 30-36 us + 21-24 us per search, whatever the corpus size - a tenth of the step of an eighth-of-the-corpus shard.  This
// kernel is shaped for the sample instead, and for latency: the whole launch is two memory round trips deep.
//   * a workgroup takes RB = 2 blocks of 16 sample rows and brings them into LDS by LDS-DMA
//     (global_load_lds_dwordx4), all pieces in flight at once: every 128-byte line of a row is fetched once, in ONE round trip to HBM
//     (a first version read operand fragments straight from global memory, 64 bytes of each row per k-step: a chain of
//     six dependent HBM round trips, 22 us at 4,096 rows and 44 us at 8,192);
//   * the queries come from L2 (every workgroup reads the same 393 KB) as MFMA B fragments, one 16-byte load per k-step
//     and lane, all of a 64-query chunk in flight at once; wave w holds queries 16 w .. 16 w + 15 of the chunk;
//   * one workgroup per (32 rows, 64-query chunk): 4,096 rows x 256 queries = 128 row groups x 4 chunks = 512 workgroups,
//     two to a CU (a first cut with every chunk looped inside 64 workgroups of 64 rows took 36 us against 15 for this
//     shape: the chunks are independent work); the chunk's query fragments are requested before the rows (round 4).
//
// Arithmetic: v_mfma_f32_16x16x32_bf16 (bf16 rows) or v_mfma_f32_16x16x4_f32 (fp32 rows: float i of a 16-byte chunk times
// float i of the matching query chunk, as the full pass does).  The sums may differ from the full pass's in the order of
// the additions only; the thresholds they produce are estimates that the full pass verifies (a query with fewer than k
// candidates back is re-run exactly), so nothing downstream depends on bit-equal sample scores.
#pragma once
#include "kernels_mfma16.h"

namespace ts {

struct SampleArgs {
    const void* corpus;       // [n_pad x ld] storage dtype
    int64_t n;                // real rows
    int ld;                   // elements per row (multiple of 64, at most 1024)
    int64_t ntiles;           // sample tiles; sample position p = 32 * tile + row
    int64_t tile_stride;      // sample tile j is global tile (j / run) * run * tile_stride + j % run
    int run;
    const void* q;            // queries in the storage dtype, row stride ld, at least 64 * ceil(nq / 64) rows
    int nq;
    const u32* row_mask;      // optional filter: disallowed rows score -inf (the sample sees the allowed rows only)
    float* scores;            // [64 * ceil(nq / 64)][row_stride]
    int row_stride;           // sample positions per query row of `scores` (multiple of 64, >= 32 * ntiles)
    int* fb_count;            // per-search counters, reset here (this is the first launch of a search)
    // optional: tile boundaries of the previous search's full pass to rebalance (grid.y = chunks + 1 then)
    int64_t* part;
    const unsigned* wg_ticks;
    int part_g;
    float part_gain;
};

constexpr int kSampleRowPad = 16;                                   // bytes: rows land 4 banks apart, the 16-row reads spread out
constexpr int sample_lds_bytes(int rows, int row_bytes) { return rows * (row_bytes + kSampleRowPad); }

// RB = row blocks of 16 per workgroup (2: 32 rows).  grid = (row_stride / (16 RB), query chunks of 64 [+ 1]), 256 threads,
// dynamic LDS = sample_lds_bytes(16 RB, ld * elem).  Row y = chunks of the grid exists when `part` is set: its first
// workgroup moves the tile boundaries of the PREVIOUS search's full pass (rebalance_tiles, common.h) while the others score
// - the job used to ride on the empty re-run launch behind the pass, where it was that launch's whole duration.
template <bool F32, int RB>
__global__ void __launch_bounds__(256) sample_scores_kernel(SampleArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char srows[];
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        *a.fb_count = 0;
    }
    const int nchunks = (a.nq + 63) / 64;
    if ((int)blockIdx.y >= nchunks) {
        if (blockIdx.x == 0 && a.part) rebalance_tiles(a.part, a.wg_ticks, a.part_g, a.part_gain, (double*)srows);
        return;
    }
    constexpr int kRows = 16 * RB;
    constexpr int kElem = F32 ? 4 : 2;
    constexpr int kStepElems = F32 ? 16 : 32;                     // elements per k-step (16 bytes per lane and quarter)
    constexpr int kSeg = 32;                                      // k-steps whose query fragments are in flight together
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r16 = lane & 15, kq = lane >> 4;
    const int64_t p0 = (int64_t)blockIdx.x * kRows;               // first sample position of this workgroup
    const int64_t npos = a.ntiles * kTileRows;
    const int row_bytes = a.ld * kElem;
    const int pitch = row_bytes + kSampleRowPad;
    const int steps = a.ld / kStepElems;
    // global row of sample position p (rows of positions past the sample are clamped; their scores are never used)
    auto row_of = [&](int64_t p) -> int64_t {
        const int64_t j = min(p >> 5, a.ntiles - 1);
        const int64_t gt = (a.run == 1) ? j * a.tile_stride : (j / a.run) * a.run * a.tile_stride + j % a.run;
        return gt * kTileRows + (p & 31);
    };
    // the query fragments of the first segment (bf16: all of them up to d = 1024) are requested BEFORE the rows: they come
    // from L2 and do not depend on anything, so their round trip runs beside the rows' trip to HBM instead of behind it
    const int qrow = (int)blockIdx.y * 64 + wave * 16 + r16;      // this lane's query (B operand) and output column
    const unsigned char* brow = (const unsigned char*)a.q + ((int64_t)qrow * a.ld + (F32 ? 4 : 8) * kq) * kElem;
    uint4 bv[kSeg];
#pragma unroll
    for (int s = 0; s < kSeg; ++s)
        if (s < steps) bv[s] = *(const uint4*)(brow + (int64_t)s * 64);
    // rows -> LDS by LDS-DMA (no registers, no compiler-made waits): wave w moves rows w, w + 4, ... in pieces of 64 lanes x
    // 16 bytes - whole 128-byte lines per request, every piece of every row in flight before the single wait below
    {
        const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)srows;
        const int per_row = row_bytes / 16;
        for (int rr = 0; rr < kRows / 4; ++rr) {
            const int r = wv + 4 * rr;
            const unsigned char* src = (const unsigned char*)a.corpus + row_of(p0 + r) * row_bytes;
            for (int pc = 0; pc * 64 < per_row; ++pc) {
                const int c = pc * 64 + lane;
                if (c < per_row) lds_dma16(src + c * 16, lds_base + r * pitch + pc * 1024);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    const unsigned char* arow[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) arow[rb] = srows + (16 * rb + r16) * pitch + kq * 16;
    // validity of this lane's output positions
    bool ok[RB][4];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int64_t p = p0 + 16 * rb + 4 * kq + g;
            const int64_t row = row_of(p);
            ok[rb][g] = p < npos && row < a.n && (!a.row_mask || ((a.row_mask[row >> 5] >> (row & 31)) & 1u));
        }
    f32x4 acc[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) acc[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
    // k in segments of at most kSeg k-steps (bf16: one segment up to d = 1024; fp32: two): a segment's query fragments
    // are all requested before its first MFMA
    for (int s0 = 0; s0 < steps; s0 += kSeg) {
        if (s0 > 0) {
#pragma unroll
            for (int s = 0; s < kSeg; ++s)
                if (s0 + s < steps) bv[s] = *(const uint4*)(brow + (int64_t)(s0 + s) * 64);
        }
        // groups of 4 k-steps (every served width is a multiple): the 4 RB fragment reads of a group are issued before
        // its MFMAs, so the LDS latency is paid once per group, not once per MFMA
#pragma unroll
        for (int g4 = 0; g4 < kSeg; g4 += 4) {
            if (s0 + g4 < steps) {
                uint4 av[4][RB];
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) av[s][rb] = *(const uint4*)(arow[rb] + (s0 + g4 + s) * 64);
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) {
                        if constexpr (F32) {
                            const float* af = reinterpret_cast<const float*>(&av[s][rb]);
                            const float* bf = reinterpret_cast<const float*>(&bv[g4 + s]);
#pragma unroll
                            for (int i = 0; i < 4; ++i) acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[i], acc[rb], 0, 0, 0);
                        } else {
                            acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(reinterpret_cast<const bf16x8&>(av[s][rb]),
                                                                              reinterpret_cast<const bf16x8&>(bv[g4 + s]), acc[rb], 0, 0, 0);
                        }
                    }
            }
        }
    }
    // lane holds sample positions p0 + 16 rb + 4 kq + {0..3} for query qrow
    float* out = a.scores + (int64_t)qrow * a.row_stride + p0 + 4 * kq;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
        *(float4*)(out + 16 * rb) = make_float4(ok[rb][0] ? acc[rb][0] : -INFINITY, ok[rb][1] ? acc[rb][1] : -INFINITY,
                                                ok[rb][2] ? acc[rb][2] : -INFINITY, ok[rb][3] ? acc[rb][3] : -INFINITY);
}

}  // namespace ts
