// The threshold sample of the batched search: scores of every query against a sparse sample of the corpus (every
// `tile_stride`-th tile of 32 rows, at most 8192 rows), written as a dense [query][sample position] fp32 matrix that
// sample_select_kernel (kernels_select.h) turns into the pass thresholds of the full-corpus pass.
//
// Round 2 ran the full-pass kernel itself over the sample (SPARSE instantiation: 393 KB of query fragments per workgroup
// for ONE tile of work, every score appended to lane-private lists that the select then gathered from 1,024 writers):
// 30-36 us + 21-24 us per search, whatever the corpus size - a tenth of the step of an eighth-of-the-corpus shard.  This
// kernel is shaped for the sample instead: a workgroup takes 64 sample rows x 64 queries (one per CU at 4,096 rows x 256
// queries), each wave 16 queries x the 64 rows as four 16x16 accumulators, operands straight from global memory (the
// sample and the queries are L2-resident after the first touch), no LDS, no ring.
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
    int ld;                   // elements per row (multiple of 64)
    int64_t ntiles;           // sample tiles; sample position p = 32 * tile + row
    int64_t tile_stride;      // sample tile j is global tile (j / run) * run * tile_stride + j % run
    int run;
    const void* q;            // queries in the storage dtype, row stride ld, at least 64 * ceil(nq / 64) rows
    int nq;
    const u32* row_mask;      // optional filter: disallowed rows score -inf (the sample sees the allowed rows only)
    float* scores;            // [64 * ceil(nq / 64)][row_stride]
    int row_stride;           // sample positions per query row of `scores` (multiple of 64, >= 32 * ntiles)
    int* fb_count;            // per-search counters, reset here (this is the first launch of a search)
    unsigned long long* stat;
};

// grid = (row_stride / 64, ceil(nq / 64)), 256 threads
template <bool F32>
__global__ void __launch_bounds__(256) sample_scores_kernel(SampleArgs a) {
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        *a.fb_count = 0;
        *a.stat = 0ull;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r16 = lane & 15, kq = lane >> 4;
    const int64_t p0 = (int64_t)blockIdx.x * 64;                  // first sample position of this workgroup
    const int qrow = blockIdx.y * 64 + wave * 16 + r16;           // this lane's query (B operand) and output column
    const int64_t npos = a.ntiles * kTileRows;
    constexpr int kElem = F32 ? 4 : 2;
    constexpr int kStepElems = F32 ? 16 : 32;                     // elements per k-step (16 bytes per lane and quarter)
    const int steps = a.ld / kStepElems;
    // global row of sample position p (rows of positions past the sample are clamped; their scores are never used)
    auto row_of = [&](int64_t p) -> int64_t {
        const int64_t j = min(p >> 5, a.ntiles - 1);
        const int64_t gt = (a.run == 1) ? j * a.tile_stride : (j / a.run) * a.run * a.tile_stride + j % a.run;
        return gt * kTileRows + (p & 31);
    };
    const unsigned char* arow[4];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
        arow[rb] = (const unsigned char*)a.corpus + (row_of(p0 + 16 * rb + r16) * a.ld + (F32 ? 4 : 8) * kq) * kElem;
    const unsigned char* brow = (const unsigned char*)a.q + ((int64_t)qrow * a.ld + (F32 ? 4 : 8) * kq) * kElem;

    f32x4 acc[4];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) acc[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
    // chunks of 4 k-steps (steps is a multiple of 4 for every served width), double-buffered: the 20 loads of chunk c + 1
    // are in flight while the 16 (fp32: 64) MFMAs of chunk c issue - the kernel is a chain of L2 round trips otherwise
    uint4 av[2][4][4], bv[2][4];
    auto load_chunk = [&](int buf, int s0) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int64_t off = (int64_t)(s0 + s) * kStepElems * kElem;
            bv[buf][s] = *(const uint4*)(brow + off);
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) av[buf][s][rb] = *(const uint4*)(arow[rb] + off);
        }
    };
    auto mma_chunk = [&](int buf) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) {
                if constexpr (F32) {
                    const float* af = reinterpret_cast<const float*>(&av[buf][s][rb]);
                    const float* bf = reinterpret_cast<const float*>(&bv[buf][s]);
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[i], acc[rb], 0, 0, 0);
                } else {
                    acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(reinterpret_cast<const bf16x8&>(av[buf][s][rb]),
                                                                      reinterpret_cast<const bf16x8&>(bv[buf][s]), acc[rb], 0, 0, 0);
                }
            }
        }
    };
    load_chunk(0, 0);
    for (int s0 = 0; s0 < steps; s0 += 8) {                       // two chunks per trip: the buffer index stays static
        if (s0 + 4 < steps) load_chunk(1, s0 + 4);
        mma_chunk(0);
        if (s0 + 4 < steps) {
            if (s0 + 8 < steps) load_chunk(0, s0 + 8);
            mma_chunk(1);
        }
    }
    // lane holds sample positions p0 + 16 rb + 4 kq + {0..3} for query qrow
    float* out = a.scores + (int64_t)qrow * a.row_stride + p0 + 4 * kq;
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
        float v[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int64_t p = p0 + 16 * rb + 4 * kq + g;
            const int64_t row = row_of(p);
            const bool ok = p < npos && row < a.n && (!a.row_mask || ((a.row_mask[row >> 5] >> (row & 31)) & 1u));
            v[g] = ok ? acc[rb][g] : -INFINITY;
        }
        *(float4*)(out + 16 * rb) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

}  // namespace ts
