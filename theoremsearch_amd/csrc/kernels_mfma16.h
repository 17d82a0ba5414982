// Batched search, bf16, d = 768 or 1024: the same streaming structure as kernels_mfma.h (queries in registers, corpus
// HBM -> LDS once per CU by LDS-DMA, fused threshold epilogue, one wave per SIMD with all 512 registers) built on
// v_mfma_f32_16x16x32_bf16 instead of v_mfma_f32_32x32x16_bf16.
//
// Why a second shape: the full pass at batch 256 is bound by the package power limit, not by issue slots or HBM
// (DESIGN.md section 3.2).  The 16x16x32 instruction does twice the K-depth per accumulator access (4 accumulator
// registers per 16,384 flops instead of 16 per 32,768), the chip holds a higher clock on it for the same flops
// (MI355X_MICROARCH.md "DVFS give-back" item 7: 1.12-1.15x), and its accumulators are small enough (NB * 8 registers
// per tile) to live in architectural VGPRs, so the threshold test reads them directly (no v_accvgpr_read per register).
//
// Work split (one workgroup = 4 waves = one CU):
//   * a tile is 32 corpus rows = two row blocks of 16; the batch is cut into blocks of 16 queries, query block j
//     belongs to wave j % 4, slot j / 4: every wave holds NB = ceil(blocks / 4) blocks (64 * NB queries per launch:
//     64, 128, 192 or 256), so partial batches load all four SIMDs evenly and there is no per-wave special case;
//   * per 32-deep k-step a wave reads its two A fragments (16 rows x 32 k, one ds_read_b128 each) and issues 2 * NB
//     MFMAs; a query fragment is 4 registers per (block, k-step): NB * 24 * 4 = 384 registers at NB = 4, the first
//     kQV fragments in VGPRs, the rest in AGPRs (MFMA B operands may be either);
//   * D[i][j] = <row i, query j>: lane l holds rows 4 (l >> 4) + {0..3} of each row block for query (l & 15) of each of
//     its blocks, so thresholds are per lane and block.  Passing scores of the FULL pass are staged in LDS and written
//     to the queries' shared lists after the tile loop; the sample levels (every score is a candidate) write lane-private
//     lists in global memory (one writer per workgroup and lane quarter: 4 * gridDim.x writers, 16 entries each).
//
// The LDS image, the DMA ring, its counted waits and the barrier protocol are those of kernels_mfma.h (the 16-row
// operand read of this shape is conflict-free on the same swizzled image: lane (r, q) reads chunk 4 (s & 1) + q of row r).
//
// Instruction stream of a k-step (measured): a 16x16x32 MFMA holds the SIMD's vector issue for 8
// of its 16 cycles, so fillers hide two per MFMA gap and not in a cluster.  The fragment reads are asm statements with
// fixed places between the MFMAs, behind explicit lgkmcnt waits (tools/audit_ring.py checks the compiled ISA); the
// steady part of the tile loop has no branches, an immediate vmcnt wait and DMA pieces addressed by a scalar base + a
// lane offset + an immediate.  2,419 -> 1,865 cycles per unit of 96 MFMAs per wave (1,536 MFMA cycles).
//
// Algorithmic traffic: rows * 2 d bytes per launch; flops 2 * queries * rows * d.
#pragma once
#include <type_traits>

#include "kernels_mfma.h"

namespace ts {

typedef __attribute__((ext_vector_type(4))) float f32x4;
// a corpus fragment in flight (one ds_read_b128): four dwords, so that copies of it are four plain register moves
typedef __attribute__((ext_vector_type(4))) unsigned frag16;

constexpr int kMfma16PrivCap = 16;   // entries of a lane-private candidate list (4 * gridDim.x writers per query)
// Full pass: candidates are staged in LDS (16 bytes each: key, query) and written to the queries' shared lists when the
// workgroup has finished its tiles: no vector-memory operation of the append path joins the vmcnt queue of the DMA ring
// (its counted waits would ask for one piece more than intended), and the final select gathers one list per query
// instead of 4 * gridDim.x private ones.  What an appended candidate still costs is the path's own ~150 instructions on a
// wave the other three wait for at the next barrier: ~0.3 us of one CU per candidate, whatever N - 10 % of a 1.25M-row
// shard's pass at 160 candidates per query, which is why the threshold estimate aims at 6 k of them (search_mfma.hip).
constexpr int kMfma16StageCap = 192;                                 // entries per wave (~40 expected at k = 10)
constexpr int kMfma16StageBytes = 4 * kMfma16StageCap * 16 + 16;     // + one counter per wave
// paired full pass: 64 dwords behind the staged candidates, where the partner workgroup's position lands (one LDS-DMA dword
// per lane of wave 0, all from the same address)
constexpr int kMfma16PaceBytes = 256;

// MFMA statements with pinned register classes: accumulator and corpus fragment in VGPRs, query fragment in a VGPR
// ("v" forms) or an AGPR ("a" forms) quadruple.  No pads inside: the A fragment comes from a ds_read behind the k-step's
// explicit lgkmcnt wait (lds_read16 below), the query fragments are written once before the loop, accumulators chain
// MFMA -> MFMA; the only non-MFMA reader of an accumulator is the epilogue, behind mfma16_settle().
__device__ __forceinline__ void mfma16_v_first(f32x4& acc, const frag16& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma16_v(f32x4& acc, const frag16& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma16_a_first(f32x4& acc, const frag16& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "a"(b));
}
__device__ __forceinline__ void mfma16_a(f32x4& acc, const frag16& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(b));
}
// fp32 rows (F32): v_mfma_f32_16x16x4_f32, one float of the corpus chunk x one float of the query chunk per instruction
__device__ __forceinline__ void mfma16f_v_first(f32x4& acc, float a, float b) {
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma16f_v(f32x4& acc, float a, float b) {
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma16f_a(f32x4& acc, float a, float b) {
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(b));
}
__device__ __forceinline__ void mfma16f_a_first(f32x4& acc, float a, float b) {
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "a"(b));
}
// A-fragment read with a fixed place in the instruction stream (asm volatile statements keep their order among themselves):
// the compiler does not know the result is asynchronous - every consumer sits behind an explicit s_waitcnt lgkmcnt below.
template <int OFF>
__device__ __forceinline__ void lds_read16(frag16& dst, unsigned addr) {
    static_assert(OFF >= 0 && OFF < 65536, "ds_read offset field");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
// LDS-DMA piece with a wave-uniform base in SGPRs, a 32-bit per-lane offset and an immediate: no vector arithmetic per
// piece.  The immediate is added to the global AND to the LDS address (LDS address = M0 + immediate + 16 * lane), so the
// caller passes lds_dst - IMM.
#ifndef TS16_DMA_IMM_LDS
#define TS16_DMA_IMM_LDS 1
#endif
template <int IMM, bool NT = true>
__device__ __forceinline__ void lds_dma16s(unsigned voff, const void* sbase, unsigned lds_dst) {
    static_assert(IMM >= 0 && IMM < 4096, "13-bit signed immediate");
    if constexpr (NT)
        asm volatile(
            "s_mov_b32 m0, %2\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %0, %1 offset:%3" TS_DMA_POLICY
            :
            : "v"(voff), "s"(sbase), "s"(lds_dst), "n"(IMM)
            : "memory");
    else
        asm volatile(
            "s_mov_b32 m0, %2\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %0, %1 offset:%3"
            :
            : "v"(voff), "s"(sbase), "s"(lds_dst), "n"(IMM)
            : "memory");
}
// Every outstanding fragment read has landed; names the whole ring, so that no copy of a ring register the compiler may
// need where control flow merges (end of a tile, steady / general branch) is placed above it.
template <int N>
__device__ __forceinline__ void lds_ring_landed(frag16 (&af)[N]) {
    static_assert(N == 6 || N == 8, "ring of 3 or 4 k-steps, two row blocks");
    if constexpr (N == 6)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[0]), "+v"(af[1]), "+v"(af[2]), "+v"(af[3]), "+v"(af[4]), "+v"(af[5]));
    else
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[0]), "+v"(af[1]), "+v"(af[2]), "+v"(af[3]), "+v"(af[4]), "+v"(af[5]),
                     "+v"(af[6]), "+v"(af[7]));
}
// wait states between the last MFMA writing an accumulator and its first VALU reader (hipcc pads nothing for asm)
template <int NB>
__device__ __forceinline__ void mfma16_settle(f32x4 (&acc)[2][NB]) {
    if constexpr (NB == 4)
        asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[0][2]), "+v"(acc[0][3]), "+v"(acc[1][0]),
                     "+v"(acc[1][1]), "+v"(acc[1][2]), "+v"(acc[1][3]));
    else if constexpr (NB == 3)
        asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[0][2]), "+v"(acc[1][0]), "+v"(acc[1][1]),
                     "+v"(acc[1][2]));
    else if constexpr (NB == 2)
        asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]));
    else
        asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc[0][0]), "+v"(acc[1][0]));
}

// Threshold test of one query block: the lane's 8 scores (4 rows of each row block) against the block's threshold,
// in 5 instructions with a fixed order (asm volatile statements keep their order among themselves and the MFMAs).
// Returns the wave mask of lanes with a passing score; `m` = the lane's best score of the block.
__device__ __forceinline__ u64 mfma16_block_test(const f32x4& a0, const f32x4& a1, float thr, float& m) {
    u64 mask;
    asm volatile(
        "v_max3_f32 %0, %2, %3, %4\n\t"
        "v_max3_f32 %0, %0, %5, %6\n\t"
        "v_max3_f32 %0, %0, %7, %8\n\t"
        "v_max_f32 %0, %0, %9\n\t"
        "v_cmp_ge_f32 %1, %0, %10"
        : "=&v"(m), "=s"(mask)
        : "v"(a0[0]), "v"(a0[1]), "v"(a0[2]), "v"(a0[3]), "v"(a1[0]), "v"(a1[1]), "v"(a1[2]), "v"(a1[3]), "v"(thr));
    return mask;
}

// The same test in three parts and the sum in front of it, for the k-split: placed one by one in the gaps behind the MFMAs
// of a tile's last unit (see TS16_FILL), where the matrix pipe leaves the vector issue free.
__device__ __forceinline__ void ksplit_add4(f32x4& d, const f32x4& a, const f32x4& b) {
    ts_f32x2 lo, hi;
    const ts_f32x2 a0 = {a[0], a[1]}, a1 = {a[2], a[3]}, b0 = {b[0], b[1]}, b1 = {b[2], b[3]};
    asm volatile("v_pk_add_f32 %0, %2, %4\n\tv_pk_add_f32 %1, %3, %5" : "=&v"(lo), "=v"(hi) : "v"(a0), "v"(a1), "v"(b0), "v"(b1));
    d = f32x4{lo[0], lo[1], hi[0], hi[1]};
}
__device__ __forceinline__ void ksplit_test_a(float& m, const f32x4& a0, const f32x4& a1) {
    asm volatile("v_max3_f32 %0, %1, %2, %3\n\tv_max3_f32 %0, %0, %4, %5" : "=&v"(m) : "v"(a0[0]), "v"(a0[1]), "v"(a0[2]), "v"(a0[3]), "v"(a1[0]));
}
__device__ __forceinline__ void ksplit_test_b(float& m, const f32x4& a1) {
    asm volatile("v_max3_f32 %0, %0, %1, %2\n\tv_max_f32 %0, %0, %3" : "+v"(m) : "v"(a1[1]), "v"(a1[2]), "v"(a1[3]));
}
__device__ __forceinline__ void ksplit_test_c(u64& mask, float m, float thr) {
    asm volatile("v_cmp_ge_f32 %0, %1, %2" : "=s"(mask) : "v"(m), "v"(thr));
}

// Append the passing scores of one query block (rare path: entered for a block only when some lane passed).  Written
// for few instructions when ONE lane holds ONE passing score - the usual case: a slow wave holds up the other three
// at the next barrier, so this path is paid four-fold.  STAGED: into the wave's LDS list (stage / stage_cnt); else into
// the lane-private global list (sample levels: every score is a candidate there).
template <bool STAGED>
__device__ __forceinline__ void mfma16_append_block(const f32x4& a0, const f32x4& a1, float thr, float m, int qid, int writer,
                                                    int nwriters, u32& cnt, int64_t row_base, const MfmaArgs& a,
                                                    uint4* stage, u32* stage_cnt) {
    if (m >= thr) {
        u64* mine = a.priv + ((int64_t)qid * nwriters + writer) * kMfma16PrivCap;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const float s = (g < 4) ? a0[g & 3] : a1[g & 3];
            if (s >= thr) {
                const int64_t row = row_base + (g & 3) + 16 * (g >> 2);
                // padding rows of the last tile, and the metadata filter: tested only for scores that pass the threshold
                if (row < a.n && (!a.row_mask || ((a.row_mask[row >> 5] >> (row & 31)) & 1u))) {
                    const u64 key = make_key(s, (u32)row);
                    bool direct = !STAGED;
                    if (STAGED) {
                        const u32 at = atomicAdd(stage_cnt, 1u);                   // LDS atomic: lgkmcnt, not vmcnt
                        if (at < (u32)kMfma16StageCap) stage[at] = make_uint4((u32)key, (u32)(key >> 32), (u32)qid, 0u);
                        else direct = true;                                        // list full: the slow way, still exact
                    }
                    if (direct) {
                        if (!STAGED && cnt < (u32)kMfma16PrivCap) {
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
    }
}

// Pacing of a workgroup pair (PAIR).  Both halves of a pair stream the same tiles; the second reader of a tile is served by
// the XCD's L2 only while the two stay within what that L2 holds of the stream - 4 MB turn over in ~7 us at 535 GB/s per XCD,
// about 15 units of 16 KB per pair.  Left alone they drift (an append, a slow barrier) and a pair that is further apart reads
// every tile twice from the fabric: 24.9 GB per launch for a 20.5 GB corpus (profiles/r04_traffic_notes.txt).  So once per
// tile wave 0 of each workgroup publishes the tile it has reached and looks at its partner's word; whoever is AHEAD by more
// than the allowed lag sleeps a few hundred cycles before its next barrier.  No spin, no hard wait: a partner that is not
// resident yet, or whose word never changes, slows nobody.  Everything is asynchronous and joins the counted queues on the
// strict side only: the publish is a store and the look an LDS-DMA dword (both older than the DMA pieces that follow: a counted
// vmcnt wait certifies MORE than before), the LDS word is read by a ds_read one unit before its value is used.
// Cache scope: the two workgroups sit on ONE XCD (w and w + 8 under round-robin dispatch) and meet in its L2 - a plain store
// (the vector L1 writes through) and an `sc0` look (past the L1).  Agent scope (`sc1`) sends both to memory: microseconds
// each, and vmcnt retires in order, so every DMA piece behind them waits that long to be counted - the first cut, 4-6 %
// slower than no pacing at all (profiles/r05_c3q_pair_pacing.txt).  On another dispatch order the word simply never changes.
__device__ __forceinline__ void pair_publish_and_fetch(unsigned* mine, const unsigned* partner, unsigned tile, unsigned lds_word) {
    const unsigned zero = 0;
    asm volatile(
        "global_store_dword %0, %1, %2\n\t"
        "s_mov_b32 m0, %4\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dword %0, %3 sc0"
        :
        : "v"(zero), "v"(tile), "s"(mine), "s"(partner), "s"(lds_word)
        : "memory");
}
__device__ __forceinline__ void pair_read_word(unsigned& dst, unsigned lds_word) {
    asm volatile("ds_read_b32 %0, %1" : "=v"(dst) : "v"(lds_word));
}

// NB = query blocks (of 16) per wave: the launch serves 64 * NB queries.
// VARIANT 0 = the product kernel.  Timing-only diagnostics (wrong results): 1 = no epilogue, 2 = DMA stream only,
// 7 = no DMA (MFMA + LDS reads), 4 = threshold test without the append path, 5 = product + per-unit cycle stamps around
// the vmcnt wait, the barrier and each DMA issue (sums per wave into a.dbg; the stamps drain the LDS queue: read the
// SHARES, not the length).  3 = product + clock probe: s_memtime / s_memrealtime around the tile loop into a.dbg
// (4 words per workgroup: shader cycles, 100 MHz ticks, units, 0) - MI355X_MICROARCH.md "DVFS give-back" item 6.
// 6 = product + s_sleep of ~256 cycles per unit (how much of an added idle cycle shows up as time under the power cap).
// SPARSE only changes the symbol (sample levels show up under their own name in kernel traces).
//
// F32: the same kernel over an fp32 index (exact fp32: v_mfma_f32_16x16x4_f32, bit for bit an fmaf chain).  A row of D
// floats is treated as 2 D two-byte elements by the DMA ring and the LDS image; a "k-step" is still one 16-byte chunk per
// lane and row block, now four floats that feed FOUR MFMAs per (row block, query block): MFMA i multiplies float i of the
// corpus chunk by float i of the matching query chunk, i.e. k = 16 s + 4 (lane >> 4) + i - a fixed permutation of k in the
// fp32 sum.  The four MFMAs of a chunk go to the accumulators in turn (i-major), so that no accumulator is used twice in a
// row (16x16x4: 32-cycle issue, 40-cycle dependent latency).  d = 1024: one block of 16 queries per wave (256 registers),
// 64 queries per launch - the Qwen-sized fp32 tables; d = 768: one or two blocks, 64 or 128 queries per launch (the
// 32x32x2 kernel of kernels_mfma_f32.h stays selectable: TS_MFMA_F32=32).
// PAIR: the paired full pass (MfmaArgs::pair; a template parameter, not a run-time branch: the headline instantiation has no
// register to spare for the pair's bookkeeping).
// KSPLIT (the paired pass of d = 1024; profiles/r05k_c3q_ksplit.txt): the four waves of a workgroup are a 2 x 2 grid - wave w
// holds the queries of column w & 1 (64 of the workgroup's 128: NB = 4 blocks) for the k-steps of HALF w >> 1 of every unit (4
// of its 8 k-steps).  Each corpus fragment a wave reads from LDS then feeds four MFMAs, as at d = 768, instead of two - at NB = 2
// the pass spends 2,048 cycles of LDS reads per tile beside 2,048 cycles of MFMAs.  The two partial sums of a (row, query) meet
// through LDS once per tile: a wave KEEPS its blocks 0, 1 (in accK[tile parity]: the loop runs two tiles per trip, nothing is
// copied) and hands its blocks 2, 3 to wave ^ 2, whose kept blocks they are (local block b of half h = block (b + 2 h) & 3 of
// the column): four ds_write_b128 at the end of a tile - those blocks are multiplied FIRST in every k-step, so they are four
// MFMAs old by then - and four asynchronous ds_read_b128 behind the third unit barrier of the NEXT tile (TS16_KSPLIT_FETCH:
// one 16 KB buffer; a barrier between every write and its read, and one between a read's landing and the next write).  Sum and
// threshold test of tile t - 1 are ten two-instruction fillers behind the MFMAs of the kept blocks in the last three k-steps of
// tile t (TS16_FILL), where the matrix pipe leaves the vector issue free; what is left at the end of a tile is the branch on the
// two masks.  So a tile's candidates are appended one tile late, the last tile's after the loop (one more barrier).
// Registers: 24 query fragments in VGPRs, 40 in AGPRs.  (With 32 in VGPRs hipcc parked two in AGPRs and copied them back one
// instruction ahead of their MFMA - asm text, no hazard padding: wrong k-steps.  tools/audit_ring.py reports that pattern.)
template <int D, int NB, int VARIANT, bool SPARSE, bool F32 = false, bool PAIR = false, bool KSPLIT = false>
__global__ void __launch_bounds__(kMfmaThreads, 1) mfma16_topk_kernel(MfmaArgs a) {
    static_assert(!PAIR || (!SPARSE && !F32), "pairs exist for the bf16 full pass");
    static_assert(!KSPLIT || (PAIR && NB == 4 && (VARIANT == 0 || VARIANT == 1 || VARIANT == 2 || VARIANT == 7) && D == 1024), "the k-split is the paired pass of d = 1024 with four query blocks per wave");
    constexpr int Deq = F32 ? 2 * D : D;                 // row length in 2-byte elements
    using dims = typename std::conditional<KSPLIT, MfmaDims<Deq, MfmaGeomKsplit<Deq>>, Mfma16Dims<Deq>>::type;
    constexpr bool kNoEpi = VARIANT == 1 || VARIANT == 7;
    constexpr bool kNoDma = VARIANT == 7;
    constexpr bool kNoMma = VARIANT == 2;
    constexpr int kStepsAll = Deq / 32;                  // k-steps (16-byte chunks per lane) per tile
    constexpr int kUnitStepsAll = dims::kUnitK / 32;     // k-steps per unit
    constexpr int kSteps = KSPLIT ? kStepsAll / 2 : kStepsAll;               // ... of them, this wave's
    constexpr int kUnitSteps = KSPLIT ? kUnitStepsAll / 2 : kUnitStepsAll;
    constexpr int kUnits = dims::kUnits, kUnitBytes = dims::kUnitBytes, kSlots = dims::kSlots, kPieces = dims::kPieces;
    constexpr int kUnitK = dims::kUnitK;
    constexpr int kPieceEvery = kUnitSteps / kPieces;    // one DMA piece every so many k-steps
    static_assert(kUnitSteps % kPieces == 0 && kPieceEvery >= 1, "DMA pieces must spread evenly over the k-steps");
    static_assert(!KSPLIT || (kUnitSteps == 4 && kPieces == 4), "k-split: four k-steps and four DMA pieces per unit and wave");
    static_assert(NB >= 1 && NB * kSteps * 4 <= 384, "query fragments must fit the register file");
    static_assert(!F32 || NB <= 2, "the fp32 issue order is written for one or two query blocks per wave");
    static_assert(kUnits == 1 || kUnits == 2 || kUnits == 4 || kUnits == 8, "units per tile");
    constexpr bool kStaged = !SPARSE;                    // full pass: candidates through LDS (see kMfma16StageCap)
    constexpr int kFrags = NB * kSteps;                  // query fragments of this wave
    // A-fragment ring of kA k-steps: during k-step s the two reads of k-step s + kA - 1 are issued into the slot k-step s - 1
    // has just left, right behind the first two MFMAs of s (two k-steps = 16 MFMAs of latency cover at NB = 4).  The ring
    // position is s % kA inside every unit, so kA divides the unit: 3 for units of 12 k-steps, 4 for units of 8.
    constexpr int kA = (kUnitSteps % 3 == 0) ? 3 : 4;
    // (k-split: the exchange holds 32 registers more - the kept blocks and the partner's sums; with the queries' share left at 32
    // fragments hipcc parks two of them in AGPRs and copies them back right in front of their MFMA inside the asm stream, where
    // its hazard recognizer sees no MFMA: the k-steps of those two fragments came out wrong on the GPU)
    constexpr int kQVmax = KSPLIT ? 24 : 36 - 2 * (kA - 2);   // the ring's registers come out of the VGPR share of the queries
    constexpr int kQV = kFrags < kQVmax ? kFrags : kQVmax;   // ... the first kQV of them in VGPRs, the rest in AGPRs
    // cache policy of the corpus stream: non-temporal (read once per search) - except in the paired pass, where the first
    // of a pair's two reads of a tile must leave its lines in the XCD's L2 for the second (with nt on both the fabric
    // carried 35.6 GB per launch of 10M x 1024 x 256 queries, 1.74 x the corpus: profiles/r04_traffic_notes.txt)
    constexpr bool kStreamNT = !PAIR;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r16 = lane & 15, kq = lane >> 4;
    // paired full pass (MfmaArgs::pair): workgroups w and w + 8 of every group of 16 walk the same tiles with the two halves
    // of the query batch; `wg` numbers the pairs
    const int wg = PAIR ? (int)(((blockIdx.x >> 4) << 3) | (blockIdx.x & 7)) : (int)blockIdx.x;
    const int qhalf = PAIR ? (int)((blockIdx.x >> 3) & 1) : 0;
    const int G = PAIR ? (int)(gridDim.x >> 1) : (int)gridDim.x;
    const int nwriters = 4 * gridDim.x;
    const int writer = 4 * blockIdx.x + kq;
    const int khalf = KSPLIT ? (wave >> 1) : 0;          // k-split: which half of every unit's k-steps this wave multiplies
    int qid[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b)
        qid[b] = KSPLIT ? qhalf * 128 + (((b + 2 * khalf) & 3) * 2 + (wave & 1)) * 16 + r16 : qhalf * 64 * NB + (b * 4 + wave) * 16 + r16;

    mfma_level_begin(a);
    uint4* stage = (uint4*)(smem + dims::kLds) + wave * kMfma16StageCap;          // this wave's staged candidates
    u32* stage_cnt = (u32*)(smem + dims::kLds + 4 * kMfma16StageCap * 16) + wave;
    if (kStaged && lane == 0) *stage_cnt = 0;
    // k-split: the partial sums a wave hands to its partner, [wave][4][lane] x 16 bytes = 16 KB behind the pair's word; written
    // at the end of a tile, fetched by the partner behind the third unit barrier of the next tile (into `got`: an asynchronous
    // read like the ring's, long landed when the tile ends) - a barrier between every write and its read, and one between
    // that read's wait and the next write
    uint4* const xbuf = (uint4*)(smem + dims::kLds + kMfma16StageBytes + kMfma16PaceBytes);
    frag16 got[4] = {};
    // pair pacing (see pair_publish_and_fetch): wave 0 only; `pace_word` = LDS address of the partner's position
    const bool pace = PAIR && a.pair_pos != nullptr && a.pair_lag > 0 && wave == 0;
    unsigned* const pace_mine = PAIR ? a.pair_pos + 2 * wg + qhalf : nullptr;
    const unsigned* const pace_partner = PAIR ? a.pair_pos + 2 * wg + (1 - qhalf) : nullptr;
    unsigned pace_seen = 0, pace_prev = 0, pace_v;
    if (PAIR && wave == 0) ((u32*)(smem + dims::kLds + kMfma16StageBytes))[lane] = 0;   // visible behind the first barrier
    // Tile range of this workgroup: equal shares, or the table the previous search's final select left (the XCDs of one
    // device run this pass at rates 3-4 % apart; the launch ends with its slowest workgroup)
    const unsigned long long wg_start = (!SPARSE && a.wg_ticks) ? __builtin_amdgcn_s_memrealtime() : 0ull;
    const int64_t t0 = (!SPARSE && a.part) ? a.part[wg] : (a.ntiles * (int64_t)wg) / G;
    const int nt = (int)(((!SPARSE && a.part) ? a.part[wg + 1] : (a.ntiles * (int64_t)(wg + 1)) / G) - t0);
    if (nt <= 0) {
        if (!SPARSE && a.wg_ticks && threadIdx.x == 0 && qhalf == 0) a.wg_ticks[wg] = 0;
        if (!kStaged) {
#pragma unroll
            for (int b = 0; b < NB; ++b) a.pcount[(int64_t)qid[b] * nwriters + writer] = 0;
        }
        return;
    }
    const int nu = kUnits * nt;
    // The stream starts first: the DMA prologue goes out before the query fragments are fetched, so that the first units
    // cross the chip while 393 KB of fragments per workgroup come out of L2 (both are vector-memory operations: they
    // return in order, the one vmcnt(0) below certifies this wave's pieces of every prologue unit and its fragments).
    // DMA source of this lane (as kernels_mfma.h): row 8w + (lane >> 3) of the tile, swizzled chunk of K-block 0 of the unit
    const int drow = 8 * wave + (lane >> 3);
    const int dchunk = (lane & 7) ^ ((drow >> 1) & 7);
    const int64_t tile_bytes = (int64_t)kTileRows * Deq * 2;
    const int64_t run_jump = tile_bytes * ((int64_t)a.run * a.tile_stride - a.run + 1);
    const int64_t g0 = (t0 / a.run) * a.run * a.tile_stride + t0 % a.run;
    const unsigned char* tile_src = (const unsigned char*)a.corpus + (int64_t)drow * (Deq * 2) + dchunk * 16 + g0 * tile_bytes;
    int issue_run_pos = (int)(t0 % a.run);
    int issue_u = 0, issue_ui = 0, issue_slot = 0;
    const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem;
    const unsigned lds0 = lds_base + wave * 1024;
    const unsigned pace_word = __builtin_amdgcn_readfirstlane(lds_base + dims::kLds + kMfma16StageBytes);
    const unsigned xfetch = lds_base + dims::kLds + kMfma16StageBytes + kMfma16PaceBytes + (((wave ^ 2) * 256 + lane) << 4);

    // operand read offsets inside a unit image: row block rb, k-step s -> (s >> 1) * 4096 + rb * 2048 + xo[s & 1]
    const int lane_off = (r16 >> 3) * 1024 + (r16 & 7) * 128;
    const int sw = (r16 >> 1) & 7;
    int xo[2];
#pragma unroll
    for (int sp = 0; sp < 2; ++sp) xo[sp] = lane_off + (((4 * sp + kq) ^ sw) << 4);
    const unsigned koff = KSPLIT ? (unsigned)(khalf * (kUnitSteps / 2) * 4096) : 0u;     // this wave's k-steps inside a unit image

#define TS16_ISSUED()                                                                 \
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
    for (int i = 0; i < ahead && issue_u < nu && !kNoDma; ++i) {
        const unsigned char* src = tile_src + issue_ui * (kUnitK * 2);
#pragma unroll
        for (int j = 0; j < kPieces; ++j) lds_dma16<kStreamNT>(src + j * 128, lds0 + issue_slot * kUnitBytes + j * 4096);
        TS16_ISSUED();
    }
    // query fragments: fragment f = b * kSteps + ks holds q[qid[b]][32 ks + 8 kq .. + 8]
    bf16x8 qv[kQV], qa[kFrags > kQV ? kFrags - kQV : 1];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const bf16x8* pq = F32 ? (const bf16x8*)((const float*)a.q + (int64_t)qid[b] * D + 4 * kq)
                               : (const bf16x8*)(a.q + (int64_t)qid[b] * D + 8 * kq);
        if (KSPLIT) pq += 4 * kUnitSteps * khalf;          // this wave's k-steps of a unit: 4 khalf .. 4 khalf + 3
#pragma unroll
        for (int ks = 0; ks < kSteps; ++ks) {
            const int f = b * kSteps + ks;
            // the k-step of the row this fragment multiplies (k-split: local step ks = unit ks / 4, step ks % 4 of this wave's half)
            const int kabs = KSPLIT ? (ks / kUnitSteps) * kUnitStepsAll + (ks % kUnitSteps) : ks;
            if (f < kQV) qv[f < kQV ? f : 0] = pq[4 * kabs];
            else qa[f >= kQV ? f - kQV : 0] = pq[4 * kabs];
        }
    }
    float thr[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) thr[b] = mfma_level_thr(a, qid[b]);
    // pin: the loads above complete here, outside the unit loop, in the register class the MFMA statements want
#pragma unroll
    for (int f = 0; f < kFrags; ++f) {
        if (f < kQV) asm volatile("" : "+v"(qv[f < kQV ? f : 0]));
        else asm volatile("" : "+a"(qa[f >= kQV ? f - kQV : 0]));
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) asm volatile("" : "+v"(thr[b]));

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    // A-fragment ring: af[2 (s % kA) + rb]; k-steps 0 .. kA - 2 of the first unit are fetched here
    frag16 af[2 * kA] = {};
    {
        const unsigned p0 = lds_base + xo[0] + (KSPLIT ? khalf * (kUnitSteps / 2) * 4096 : 0);
        const unsigned p1 = lds_base + xo[1] + (KSPLIT ? khalf * (kUnitSteps / 2) * 4096 : 0);
        lds_read16<0>(af[0], p0);
        lds_read16<2048>(af[1], p0);
        lds_read16<0>(af[2], p1);
        lds_read16<2048>(af[3], p1);
        if constexpr (kA > 3) {
            lds_read16<4096>(af[4], p0);
            lds_read16<4096 + 2048>(af[5], p0);
        }
        static_assert(kA == 3 || kA == 4, "prologue written for rings of 3 or 4 k-steps");
    }
    // (ring registers not fetched yet are named too: they hold nothing anyone reads before their first fetch)
    lds_ring_landed(af);

    f32x4 acc[2][NB];
    f32x4 accK[2][2][2] = {};                            // k-split: [tile parity][row block][kept block]
    u32 cnt[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) cnt[b] = 0;
    int slot = 0, u = 0;
    unsigned long long t_vm = 0, t_bar = 0, t_dma = 0, t_all0 = 0;   // VARIANT 5: cycle sums of the waits / DMA issue
    if (VARIANT == 5) t_all0 = cycle_stamp();
    unsigned long long c_begin = 0, r_begin = 0;
    if (VARIANT == 3) {
        c_begin = __builtin_amdgcn_s_memtime();
        r_begin = __builtin_amdgcn_s_memrealtime();
        __builtin_amdgcn_s_waitcnt(0xC07F);
    }

    // the accumulator of (row block, query block); k-split: the kept blocks 0, 1 live in the tile parity's half of accK
#define TS16_ACC(RB_, B_) (*((KSPLIT && (B_) < 2) ? &accK[par_][RB_][(B_) & 1] : &acc[RB_][B_]))
    // one MFMA of block B_, row block RB_, global k-step KS_ (all compile-time); F32: MFMA I_ of the chunk's four
#define TS16_MMA(RB_, B_, KS_, AF_)                                                                        \
    do {                                                                                                   \
        constexpr int f_ = (B_) * kSteps + (KS_);                                                          \
        if constexpr ((KS_) == 0) {                                                                        \
            if constexpr (f_ < kQV) mfma16_v_first(TS16_ACC(RB_, B_), AF_, qv[f_ < kQV ? f_ : 0]);         \
            else mfma16_a_first(TS16_ACC(RB_, B_), AF_, qa[f_ >= kQV ? f_ - kQV : 0]);                     \
        } else {                                                                                           \
            if constexpr (f_ < kQV) mfma16_v(TS16_ACC(RB_, B_), AF_, qv[f_ < kQV ? f_ : 0]);               \
            else mfma16_a(TS16_ACC(RB_, B_), AF_, qa[f_ >= kQV ? f_ - kQV : 0]);                           \
        }                                                                                                  \
    } while (0)
#define TS16_MMAF(RB_, B_, KS_, AF_, I_)                                                                   \
    do {                                                                                                   \
        constexpr int f_ = (B_) * kSteps + (KS_);                                                          \
        const f32x4& af4_ = reinterpret_cast<const f32x4&>(AF_);                                           \
        if constexpr ((KS_) == 0 && (I_) == 0) {                                                           \
            if constexpr (f_ < kQV) mfma16f_v_first(acc[RB_][B_], af4_[I_], reinterpret_cast<const f32x4&>(qv[f_ < kQV ? f_ : 0])[I_]); \
            else mfma16f_a_first(acc[RB_][B_], af4_[I_], reinterpret_cast<const f32x4&>(qa[f_ >= kQV ? f_ - kQV : 0])[I_]); \
        } else if constexpr (f_ < kQV) {                                                                   \
            mfma16f_v(acc[RB_][B_], af4_[I_], reinterpret_cast<const f32x4&>(qv[f_ < kQV ? f_ : 0])[I_]);  \
        } else {                                                                                           \
            mfma16f_a(acc[RB_][B_], af4_[I_], reinterpret_cast<const f32x4&>(qa[f_ >= kQV ? f_ - kQV : 0])[I_]); \
        }                                                                                                  \
    } while (0)

    /* the two reads of k-step N_ (this unit's or the next one's image) into the ring slot of N_ */
#define TS16_LOAD(RB_, N_)                                                                                 \
    do {                                                                                                   \
        constexpr int w_ = 2 * ((N_) % kA) + (RB_);                                                        \
        if constexpr ((N_) < kUnitSteps) lds_read16<((N_) >> 1) * 4096 + (RB_) * 2048>(af[w_], ua[(N_) & 1]); \
        else lds_read16<(((N_) - kUnitSteps) >> 1) * 4096 + (RB_) * 2048>(af[w_], na[(N_) & 1]);          \
    } while (0)
#define TS16_PIECE(S_)                                                                                     \
    do {                                                                                                   \
        if constexpr ((S_) % kPieceEvery == kPieceEvery - 1)                                               \
            if (do_issue) {                                                                                \
                constexpr int j_ = (S_) / kPieceEvery;                                                     \
                unsigned long long d0_ = 0;                                                                \
                if (VARIANT == 5) d0_ = cycle_stamp();                                                     \
                if constexpr (steady_) lds_dma16s<j_ * 128, kStreamNT>(dma_voff, ssrc, idst + j_ * (4096 - 128 * TS16_DMA_IMM_LDS)); \
                else lds_dma16<kStreamNT>(isrc + j_ * 128, idst + j_ * 4096);                                         \
                if (VARIANT == 5) t_dma += cycle_stamp() - d0_;                                            \
            }                                                                                              \
    } while (0)
    // k-split: the blocks that go to the partner (2, 3) first - at the end of a tile their sums are four MFMAs old when they are
    // stored, no wait states to add - and one filler behind each MFMA of the kept blocks: in the tile's last unit the sum and the
    // test of the PREVIOUS tile's kept blocks (held + the partner's half, long landed), ten pieces of two instructions
#define TS16_FILL(KS_, G_)                                                                                 \
    do {                                                                                                   \
        constexpr int fi_ = ((KS_) - (kSteps - 3)) * 4 + (G_);                                             \
        if constexpr (KSPLIT && !kNoEpi && (KS_) >= kSteps - 3) {                                          \
            if constexpr (fi_ == 0) {                                                                      \
                asm volatile("" : "+v"(got[0]), "+v"(got[1]), "+v"(got[2]), "+v"(got[3]));                 \
                ksplit_add4(sum_[0][0], accK[par_ ^ 1][0][0], reinterpret_cast<const f32x4&>(got[0]));               \
            }                                                                                              \
            if constexpr (fi_ == 1) ksplit_add4(sum_[1][0], accK[par_ ^ 1][1][0], reinterpret_cast<const f32x4&>(got[2])); \
            if constexpr (fi_ == 2) ksplit_add4(sum_[0][1], accK[par_ ^ 1][0][1], reinterpret_cast<const f32x4&>(got[1])); \
            if constexpr (fi_ == 3) ksplit_add4(sum_[1][1], accK[par_ ^ 1][1][1], reinterpret_cast<const f32x4&>(got[3])); \
            if constexpr (fi_ == 4) ksplit_test_a(best_[0], sum_[0][0], sum_[1][0]);                       \
            if constexpr (fi_ == 5) ksplit_test_b(best_[0], sum_[1][0]);                                   \
            if constexpr (fi_ == 6) ksplit_test_c(hit_[0], best_[0], thr[0]);                              \
            if constexpr (fi_ == 7) ksplit_test_a(best_[1], sum_[0][1], sum_[1][1]);                       \
            if constexpr (fi_ == 8) ksplit_test_b(best_[1], sum_[1][1]);                                   \
            if constexpr (fi_ == 9) ksplit_test_c(hit_[1], best_[1], thr[1]);                              \
        }                                                                                                  \
    } while (0)
#ifndef TS16_ORDER_A
#define TS16_BF16_ORDER(KS_, R0_, N_, S_)                                                                  \
    do {                                                                                                   \
        if constexpr (KSPLIT) {                                                                            \
            TS16_MMA(0, 2, KS_, af[R0_]); TS16_LOAD(0, N_);                                                \
            TS16_MMA(1, 2, KS_, af[R0_ + 1]); TS16_LOAD(1, N_);                                            \
            TS16_MMA(0, 3, KS_, af[R0_]); TS16_MMA(1, 3, KS_, af[R0_ + 1]);                                \
            TS16_PIECE(S_);                                                                                \
            TS16_MMA(0, 0, KS_, af[R0_]); TS16_FILL(KS_, 0);                                               \
            TS16_MMA(1, 0, KS_, af[R0_ + 1]); TS16_FILL(KS_, 1);                                           \
            TS16_MMA(0, 1, KS_, af[R0_]); TS16_FILL(KS_, 2);                                               \
            TS16_MMA(1, 1, KS_, af[R0_ + 1]); TS16_FILL(KS_, 3);                                           \
            break;                                                                                         \
        }                                                                                                  \
        TS16_MMA(0, 0, KS_, af[R0_]); TS16_LOAD(0, N_);                                                    \
        TS16_MMA(1, 0, KS_, af[R0_ + 1]); TS16_LOAD(1, N_);                                                \
        if constexpr (NB > 1) { TS16_MMA(0, 1, KS_, af[R0_]); TS16_MMA(1, 1, KS_, af[R0_ + 1]); }          \
        TS16_PIECE(S_);                                                                                    \
        if constexpr (NB > 2) { TS16_MMA(0, 2, KS_, af[R0_]); TS16_MMA(1, 2, KS_, af[R0_ + 1]); }          \
        if constexpr (NB > 3) { TS16_MMA(0, 3, KS_, af[R0_]); TS16_MMA(1, 3, KS_, af[R0_ + 1]); }          \
    } while (0)
#else   /* A/B build: row-block major - the corpus fragment stays put for NB consecutive MFMAs */
#define TS16_BF16_ORDER(KS_, R0_, N_, S_)                                                                  \
    do {                                                                                                   \
        TS16_MMA(0, 0, KS_, af[R0_]); TS16_LOAD(0, N_);                                                    \
        if constexpr (NB > 1) { TS16_MMA(0, 1, KS_, af[R0_]); }                                            \
        TS16_LOAD(1, N_);                                                                                  \
        if constexpr (NB > 2) { TS16_MMA(0, 2, KS_, af[R0_]); }                                            \
        if constexpr (NB > 3) { TS16_MMA(0, 3, KS_, af[R0_]); }                                            \
        TS16_PIECE(S_);                                                                                    \
        TS16_MMA(1, 0, KS_, af[R0_ + 1]);                                                                  \
        if constexpr (NB > 1) { TS16_MMA(1, 1, KS_, af[R0_ + 1]); }                                        \
        if constexpr (NB > 2) { TS16_MMA(1, 2, KS_, af[R0_ + 1]); }                                        \
        if constexpr (NB > 3) { TS16_MMA(1, 3, KS_, af[R0_ + 1]); }                                        \
    } while (0)
#endif
    // One k-step.  The fillers have fixed places between the MFMAs (a 16x16x32 MFMA leaves 8 of its 16 cycles of vector
    // issue free: two simple instructions per gap hide, a cluster behind the last MFMA does not): the wait for this
    // k-step's fragments, MFMA, read, MFMA, read, two MFMAs, the DMA piece when one is due, the remaining MFMAs.
#define TS16_STEP(UI, S_)                                                                                  \
    do {                                                                                                   \
        constexpr int ks_ = (UI) * kUnitSteps + (S_);                                                      \
        constexpr int r0_ = 2 * ((S_) % kA);                                                               \
        constexpr int n_ = (S_) + kA - 1;                                                                  \
        if constexpr (!kNoMma) {                                                                           \
            /* all but the reads of the kA - 2 k-steps after this one have landed (LDS returns in order); the wait */ \
            /* names this k-step's fragments, so that no consumer of them is placed above it */             \
            if constexpr (kA == 3) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(af[r0_]), "+v"(af[r0_ + 1])); \
            else asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(af[r0_]), "+v"(af[r0_ + 1]));                  \
        }                                                                                                  \
        if constexpr (!kNoMma && F32) {                                                                    \
            /* i-major: the 2 NB accumulators in turn for each of the chunk's four floats */               \
            TS16_MMAF(0, 0, ks_, af[r0_], 0); TS16_LOAD(0, n_);                                            \
            TS16_MMAF(1, 0, ks_, af[r0_ + 1], 0); TS16_LOAD(1, n_);                                        \
            if constexpr (NB > 1) { TS16_MMAF(0, 1, ks_, af[r0_], 0); TS16_MMAF(1, 1, ks_, af[r0_ + 1], 0); } \
            TS16_MMAF(0, 0, ks_, af[r0_], 1); TS16_MMAF(1, 0, ks_, af[r0_ + 1], 1);                        \
            TS16_PIECE(S_);                                                                                \
            if constexpr (NB > 1) { TS16_MMAF(0, 1, ks_, af[r0_], 1); TS16_MMAF(1, 1, ks_, af[r0_ + 1], 1); } \
            TS16_MMAF(0, 0, ks_, af[r0_], 2); TS16_MMAF(1, 0, ks_, af[r0_ + 1], 2);                        \
            if constexpr (NB > 1) { TS16_MMAF(0, 1, ks_, af[r0_], 2); TS16_MMAF(1, 1, ks_, af[r0_ + 1], 2); } \
            TS16_MMAF(0, 0, ks_, af[r0_], 3); TS16_MMAF(1, 0, ks_, af[r0_ + 1], 3);                        \
            if constexpr (NB > 1) { TS16_MMAF(0, 1, ks_, af[r0_], 3); TS16_MMAF(1, 1, ks_, af[r0_ + 1], 3); } \
        }                                                                                                  \
        if constexpr (!kNoMma && !F32) {                                                                   \
            /* query-block major (consecutive MFMAs share the query fragment; the epilogue's block order relies on it) */ \
            TS16_BF16_ORDER(ks_, r0_, n_, S_);                                                             \
        }                                                                                                  \
        if constexpr (kNoMma) TS16_PIECE(S_);                                                              \
    } while (0)

    // k-split: the partner's partial sums of the previous tile (its blocks 2, 3 = this wave's blocks 0, 1), four asynchronous reads
#define TS16_KSPLIT_FETCH()                                                                                \
    do {                                                                                                   \
        lds_read16<0>(got[0], xfetch); lds_read16<1024>(got[1], xfetch);                                   \
        lds_read16<2048>(got[2], xfetch); lds_read16<3072>(got[3], xfetch);                                \
    } while (0)
    // A unit of the steady part of the tile loop: the unit kSlots - 1 ahead is still to be issued (so every piece goes out,
    // no branch), exactly kSlots - 3 units stay in flight behind the counted wait (an immediate, no ladder), the unit in
    // the tile that is issued is a compile-time constant, and the ring positions are three running byte offsets
    // (this unit's slot, the next one's, the one unit u - 1 has left = the issue slot).
#define TS16_UNIT_S(UI)                                                                                    \
    do {                                                                                                   \
        constexpr int cui_ = ((UI) + kSlots - 1) % kUnits;                                                 \
        const unsigned noff = (soff + kUnitBytes == (unsigned)dims::kLds) ? 0u : soff + kUnitBytes;        \
        unsigned ua[2], na[2];                                                                             \
        ua[0] = lds_base + soff + xo[0] + koff; ua[1] = lds_base + soff + xo[1] + koff;                    \
        na[0] = lds_base + noff + xo[0] + koff; na[1] = lds_base + noff + xo[1] + koff;                    \
        if constexpr (!kNoDma) wait_vmcnt<(kSlots - 3) * kPieces>();   /* own pieces of unit u + 1 */       \
        if constexpr (PAIR && (UI) == (2 % kUnits)) {                   /* ahead of the partner: give way */ \
            /* (the read and its use are unconditional: a value defined under a branch would reach this point through a */ \
            /* copy the compiler is free to place right behind the asynchronous ds_read) */                \
            asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(pace_seen) : "v"(pace_v));                   \
            const int ahead_ = (int)((unsigned)t - pace_seen);                                             \
            if (pace && pace_seen != pace_prev && ahead_ > a.pair_lag && ahead_ < (1 << 20))               \
                for (int z_ = 0; z_ < ahead_ - a.pair_lag && z_ < 4; ++z_) __builtin_amdgcn_s_sleep(5);    \
            pace_prev = pace_seen;                                                                         \
        }                                                                                                  \
        __builtin_amdgcn_s_barrier();                                   /* ... and everyone's */            \
        asm volatile("" ::: "memory");                                                                     \
        if constexpr (KSPLIT && (UI) == 2) TS16_KSPLIT_FETCH();                                            \
        if constexpr (PAIR && (UI) == 0) {                                                                 \
            if (pace) pair_publish_and_fetch(pace_mine, pace_partner, (unsigned)t, pace_word);            \
        }                                                                                                  \
        if constexpr (PAIR && (UI) == (1 % kUnits)) pair_read_word(pace_v, pace_word);                     \
        if (VARIANT == 6) __builtin_amdgcn_s_sleep(4);                                                     \
        constexpr bool do_issue = !kNoDma;                                                                 \
        constexpr bool steady_ = true;                                                                     \
        const unsigned char* isrc = nullptr;                                                               \
        const unsigned char* ssrc = s_tile + cui_ * (kUnitK * 2);       /* wave-uniform: SGPRs */           \
        const unsigned idst = lds0 + poff;                                                                 \
        TS16_STEP(UI, 0); TS16_STEP(UI, 1); TS16_STEP(UI, 2); TS16_STEP(UI, 3);                            \
        if constexpr (kUnitSteps > 4) { TS16_STEP(UI, 4 % kUnitSteps); TS16_STEP(UI, 5 % kUnitSteps); TS16_STEP(UI, 6 % kUnitSteps); TS16_STEP(UI, 7 % kUnitSteps); } \
        if constexpr (kUnitSteps > 8) { TS16_STEP(UI, 8 % kUnitSteps); TS16_STEP(UI, 9 % kUnitSteps); TS16_STEP(UI, 10 % kUnitSteps); TS16_STEP(UI, 11 % kUnitSteps); } \
        if constexpr (cui_ == kUnits - 1 && !kNoDma) s_tile += steady_jump;                                \
        poff = soff;                                                                                       \
        soff = noff;                                                                                       \
    } while (0)

#define TS16_UNIT(UI)                                                                                      \
    do {                                                                                                   \
        const int nslot = (slot + 1 == kSlots) ? 0 : slot + 1;                                             \
        unsigned ua[2], na[2];                                                                             \
        ua[0] = lds_base + slot * kUnitBytes + xo[0] + koff; ua[1] = lds_base + slot * kUnitBytes + xo[1] + koff;   \
        na[0] = lds_base + nslot * kUnitBytes + xo[0] + koff; na[1] = lds_base + nslot * kUnitBytes + xo[1] + koff; \
        /* certify unit u + 1 (own pieces, then everyone's); every wave is past unit u - 1: its slot is free */ \
        unsigned long long s0_ = 0, s1_ = 0;                                                               \
        if (VARIANT == 5) s0_ = cycle_stamp();                                                             \
        if (u + 1 < nu && !kNoDma) wait_keep_units<kPieces>(issue_u - (u + 2));                            \
        if (VARIANT == 5) s1_ = cycle_stamp();                                                             \
        __builtin_amdgcn_s_barrier();                                                                      \
        asm volatile("" ::: "memory");                                                                     \
        if constexpr (KSPLIT && (UI) == 2) TS16_KSPLIT_FETCH();                                            \
        if (VARIANT == 5) { t_vm += s1_ - s0_; t_bar += cycle_stamp() - s1_; }                             \
        if (VARIANT == 6) __builtin_amdgcn_s_sleep(4);   /* ~256 idle cycles per unit: elasticity of time to cycles */ \
        const bool do_issue = issue_u < nu && !kNoDma;                                                     \
        constexpr bool steady_ = false;                                                                    \
        const unsigned char* isrc = tile_src + issue_ui * (kUnitK * 2);                                    \
        const unsigned char* ssrc = nullptr;                                                               \
        /* (wave-uniform by construction; with one unit per tile hipcc loses track of that and would hand M0 a VGPR) */ \
        const unsigned idst = __builtin_amdgcn_readfirstlane(lds0 + issue_slot * kUnitBytes);              \
        TS16_STEP(UI, 0); TS16_STEP(UI, 1); TS16_STEP(UI, 2); TS16_STEP(UI, 3);                            \
        if constexpr (kUnitSteps > 4) { TS16_STEP(UI, 4 % kUnitSteps); TS16_STEP(UI, 5 % kUnitSteps); TS16_STEP(UI, 6 % kUnitSteps); TS16_STEP(UI, 7 % kUnitSteps); } \
        if constexpr (kUnitSteps > 8) { TS16_STEP(UI, 8 % kUnitSteps); TS16_STEP(UI, 9 % kUnitSteps); TS16_STEP(UI, 10 % kUnitSteps); TS16_STEP(UI, 11 % kUnitSteps); } \
        if (do_issue) TS16_ISSUED();                                                                       \
        slot = nslot;                                                                                      \
        ++u;                                                                                               \
    } while (0)

    // k-split: the candidates of tile TP (one tile late), from the sums and masks the fillers of the next tile's last unit left
#define TS16_KSPLIT_APPEND(TP)                                                                             \
    do {                                                                                                   \
        if (__builtin_expect((hit_[0] | hit_[1]) != 0, 0)) {                                               \
            const int64_t lt_ = t0 + (TP);                                                                 \
            const int64_t tile_row_ = (a.run == 1 ? lt_ * a.tile_stride : (lt_ / a.run) * a.run * a.tile_stride + lt_ % a.run) * kTileRows; \
            const int64_t row_base_ = tile_row_ + 4 * kq;                                                  \
            _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_)                                               \
                if (hit_[i_] != 0)                                                                         \
                    mfma16_append_block<kStaged>(sum_[0][i_], sum_[1][i_], thr[i_], best_[i_], qid[i_], writer, nwriters, cnt[i_], \
                                                 row_base_, a, stage, stage_cnt);                          \
        }                                                                                                  \
    } while (0)
    static_assert(kUnitSteps == 8 || kUnitSteps == 12 || (KSPLIT && kUnitSteps == 4), "unit = 8 or 12 k-steps of 32 (k-split: this wave's 4 of 8)");
    static_assert(kSlots >= 3, "the steady part keeps kSlots - 3 units in flight behind its wait");
    // Steady tiles: every unit of the tile still has a unit to issue kSlots - 1 ahead.  (Runs of tiles, a shallower
    // ring and the stamped build go through the general units below.)
    const int nt_steady = (a.run == 1 && ahead == kSlots - 1 && VARIANT != 5 && nu > ahead) ? (nu - ahead) / kUnits : 0;
    const int64_t steady_jump = tile_bytes * a.tile_stride;
    unsigned soff = 0, poff = (unsigned)(kSlots - 1) * kUnitBytes;
    // the steady part addresses the stream as (uniform tile base in SGPRs) + (this lane's offset inside a tile); the
    // prologue has issued `ahead` units: the next one lies in visited tile t0 + ahead / kUnits
    const unsigned dma_voff = (unsigned)(drow * (Deq * 2) + dchunk * 16);
    const unsigned char* s_tile = (const unsigned char*)a.corpus + (t0 + ahead / kUnits) * steady_jump;
    f32x4 sum_[2][2] = {};                                // k-split: what the fillers of a tile's last unit leave for its end
    float best_[2] = {};
    u64 hit_[2] = {};
    // One tile.  (k-split: PAR = the tile's parity - the kept blocks of a tile stay where they are, in accK[PAR], while the next
    // tile's sums grow in accK[PAR ^ 1], instead of being copied: the loop below runs two tiles per trip.)
    auto tile = [&](auto par_c, const int t) __attribute__((always_inline)) {
        constexpr int par_ = decltype(par_c)::value;
        (void)par_;
        if (t < nt_steady) {
            TS16_UNIT_S(0);
            if constexpr (kUnits >= 2) TS16_UNIT_S(1 % kUnits);
            if constexpr (kUnits >= 4) {
                TS16_UNIT_S(2 % kUnits);
                TS16_UNIT_S(3 % kUnits);
            }
            if constexpr (kUnits == 8) {
                TS16_UNIT_S(4 % kUnits);
                TS16_UNIT_S(5 % kUnits);
                TS16_UNIT_S(6 % kUnits);
                TS16_UNIT_S(7 % kUnits);
            }
            if constexpr (!kNoMma) lds_ring_landed(af);      // before any branch: see lds_ring_landed
            if (t + 1 == nt_steady) {
                // hand over to the general units: the same ring, counted in units
                u = kUnits * nt_steady;
                slot = u % kSlots;
                issue_u = u + ahead;
                issue_ui = issue_u % kUnits;
                issue_slot = issue_u % kSlots;
                issue_run_pos = 0;
                tile_src = s_tile + dma_voff;
            }
        } else {
            TS16_UNIT(0);
            if constexpr (kUnits >= 2) TS16_UNIT(1 % kUnits);
            if constexpr (kUnits >= 4) {
                TS16_UNIT(2 % kUnits);
                TS16_UNIT(3 % kUnits);
            }
            if constexpr (kUnits == 8) {
                TS16_UNIT(4 % kUnits);
                TS16_UNIT(5 % kUnits);
                TS16_UNIT(6 % kUnits);
                TS16_UNIT(7 % kUnits);
            }
            if constexpr (!kNoMma) lds_ring_landed(af);
        }
        if constexpr (kNoMma) return;
        if constexpr (KSPLIT) {
            if constexpr (kNoEpi) {
                asm volatile("s_nop 15\n\ts_nop 3" : "+v"(accK[par_][0][0]), "+v"(accK[par_][0][1]), "+v"(accK[par_][1][0]), "+v"(accK[par_][1][1]),
                             "+v"(acc[0][2]), "+v"(acc[0][3]), "+v"(acc[1][2]), "+v"(acc[1][3]));
                return;
            }
            if (t > 0) TS16_KSPLIT_APPEND(t - 1);        // the previous tile, summed and tested by this tile's fillers
            // blocks 2, 3 go to the partner (their last MFMAs are four MFMAs back: TS16_BF16_ORDER); blocks 0, 1 are kept
            asm volatile("" : "+v"(acc[0][2]), "+v"(acc[0][3]), "+v"(acc[1][2]), "+v"(acc[1][3]));
            uint4* mine = xbuf + wave * 256 + lane;
            mine[0] = *reinterpret_cast<const uint4*>(&acc[0][2]); mine[64] = *reinterpret_cast<const uint4*>(&acc[0][3]);
            mine[128] = *reinterpret_cast<const uint4*>(&acc[1][2]); mine[192] = *reinterpret_cast<const uint4*>(&acc[1][3]);
            return;
        }
        // The last k-step issued its MFMAs in block order, so with NB = 4 the results of block b are at least 6 MFMAs
        // old when its test (5 VALU instructions per block, in order) reads them; fewer blocks need explicit wait states.
#ifdef TS16_ORDER_A
        mfma16_settle<NB>(acc);
#else
        if constexpr (NB < 4 || kNoEpi) mfma16_settle<NB>(acc);
#endif
        if constexpr (kNoEpi) {
#pragma unroll
            for (int b = 0; b < NB; ++b) asm volatile("" ::"v"(acc[0][b]), "v"(acc[1][b]));
            return;
        }
        // lane holds rows 4 kq + {0..3} of both row blocks for query qid[b]
        float best[NB];
        u64 hit[NB], any_hit = 0;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            hit[b] = mfma16_block_test(acc[0][b], acc[1][b], thr[b], best[b]);
            any_hit |= hit[b];
        }
        if (VARIANT == 4) {                      // diagnostic: the test without the append path
            asm volatile("" ::"s"(any_hit));
            return;
        }
        if (__builtin_expect(any_hit != 0, 0)) {
            const int64_t lt = t0 + t;
            // (64-bit divisions are hundreds of instructions: runs of one tile - the default - take the short way)
            const int64_t tile_row = (a.run == 1 ? lt * a.tile_stride : (lt / a.run) * a.run * a.tile_stride + lt % a.run) * kTileRows;
            const int64_t row_base = tile_row + 4 * kq;
#pragma unroll
            for (int b = 0; b < NB; ++b)
                if (hit[b] != 0)
                    mfma16_append_block<kStaged>(acc[0][b], acc[1][b], thr[b], best[b], qid[b], writer, nwriters, cnt[b], row_base, a,
                                                 stage, stage_cnt);
        }
    };
    if constexpr (KSPLIT) {
        for (int t = 0; t < nt; t += 2) {
            tile(std::integral_constant<int, 0>{}, t);
            if (t + 1 < nt) tile(std::integral_constant<int, 1>{}, t + 1);
        }
    } else {
        for (int t = 0; t < nt; ++t) tile(std::integral_constant<int, 0>{}, t);
    }
    if constexpr (KSPLIT && !kNoEpi && !kNoMma) {
        // the last tile's halves: one more barrier (every wave has written), then the same test
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        TS16_KSPLIT_FETCH();
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(got[0]), "+v"(got[1]), "+v"(got[2]), "+v"(got[3]));
        // the last tile's kept blocks (its parity's half; the tile's last MFMAs wrote them: wait states first)
        asm volatile("s_nop 15\n\ts_nop 3" : "+v"(accK[0][0][0]), "+v"(accK[0][0][1]), "+v"(accK[0][1][0]), "+v"(accK[0][1][1]),
                     "+v"(accK[1][0][0]), "+v"(accK[1][0][1]), "+v"(accK[1][1][0]), "+v"(accK[1][1][1]));
        f32x4 held[2][2];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int i = 0; i < 2; ++i) held[rb][i] = ((nt - 1) & 1) ? accK[1][rb][i] : accK[0][rb][i];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int i = 0; i < 2; ++i) sum_[rb][i] = held[rb][i] + reinterpret_cast<const f32x4&>(got[rb * 2 + i]);
        hit_[0] = mfma16_block_test(sum_[0][0], sum_[1][0], thr[0], best_[0]);
        hit_[1] = mfma16_block_test(sum_[0][1], sum_[1][1], thr[1], best_[1]);
        TS16_KSPLIT_APPEND(nt - 1);
    }
#undef TS16_KSPLIT_APPEND
#undef TS16_KSPLIT_FETCH
#undef TS16_UNIT
#undef TS16_UNIT_S
#undef TS16_STEP
#undef TS16_MMA
#undef TS16_ACC
#undef TS16_MMAF
#undef TS16_LOAD
#undef TS16_BF16_ORDER
#undef TS16_FILL
#undef TS16_PIECE
#undef TS16_ISSUED
    if (kStaged) {
        // the tile loop is over (no DMA in flight that a counted wait still watches): staged candidates -> shared lists
        const u32 n = min(*stage_cnt, (u32)kMfma16StageCap);
        for (u32 e = lane; e < n; e += 64) {
            const uint4 v = stage[e];
            const u32 pos = atomicAdd(&a.count[v.z], 1u);
            if (pos < (u32)a.cap) a.cand[(int64_t)v.z * a.cap + pos] = ((u64)v.y << 32) | v.x;
        }
    } else {
#pragma unroll
        for (int b = 0; b < NB; ++b) a.pcount[(int64_t)qid[b] * nwriters + writer] = cnt[b];
    }
    if (!SPARSE && a.wg_ticks && threadIdx.x == 0 && qhalf == 0)
        a.wg_ticks[wg] = (unsigned)(__builtin_amdgcn_s_memrealtime() - wg_start);
    if (VARIANT == 5 && a.dbg && lane == 0) {
        unsigned long long* d = a.dbg + ((size_t)blockIdx.x * 4 + wave) * 4;
        d[0] = cycle_stamp() - t_all0;
        d[1] = t_vm;
        d[2] = t_bar;
        d[3] = t_dma;
    }
    if (VARIANT == 3 && a.dbg && threadIdx.x == 0) {
        const unsigned long long c_end = __builtin_amdgcn_s_memtime();
        const unsigned long long r_end = __builtin_amdgcn_s_memrealtime();
        unsigned long long* d = a.dbg + (size_t)blockIdx.x * 4;
        d[0] = c_end - c_begin;
        d[1] = r_end - r_begin;
        d[2] = (unsigned long long)nu;
        d[3] = 0;
    }
}

}  // namespace ts
