// The encoder's linear layers (BertSelfAttention / BertSelfOutput / BertIntermediate / BertOutput of the
// sentence-transformer the reference loads, compare_embeddings.py:11-12; `model.encode` at app_showcase_model.py:92):
//     y[M x N] = act(x[M x K] . w[N x K]^T + bias[N])        bf16 in, fp32 accumulate, bf16 out
// `w` is a torch Linear weight as it lies ([out][in], K contiguous), so both operands are K-contiguous rows - the very
// contraction of the search kernels (rows x queries), with both sides streamed.
//
// Why a kernel of our own: the encoder-in-loop step (BASELINE configs[4]) spends 2.04 of its 5.6 ms in four GEMM shapes
// (8,192 x {2304, 768, 3072, 768} x {768, 768, 768, 3072}) that the BLAS library runs at 550-860 TF/s inside that step.  The
// starting premise - its 192 x 128 / 192 x 256 tiles leaving a tail (258 / 516 tiles for 256 CUs) - holds only if the 192 runs
// along the tokens; along the output features they divide the shapes exactly (profiles/r03_linear_gemm_ab.txt).  Here the tile is chosen per shape so that the grid is a whole number of rounds (256 x 96 -> 256 tiles for
// N = 768, 256 x 288 -> 256 tiles for N = 2304, 256 x 192 -> 512 tiles for N = 3072), and the exact erf GELU of the
// intermediate layer is applied where the accumulators are (one elementwise launch and a 100 MB round trip less per
// layer).
//
// Structure (one workgroup = 4 waves = one CU, one wave per SIMD with the whole register file):
//   * a unit = 64 of K for the tile (two 32-deep k-steps): BN rows of w and BM rows of x, 128 bytes each - whole cache
//     lines: a first cut with 64-byte rows (one k-step per stage, deeper ring) ran at a third of the MFMA rate, every
//     line fetched twice as two half-line requests - brought into LDS by LDS-DMA (global_load_lds_dwordx4) in pieces of
//     8 rows x 128 bytes (one wave instruction = 1 KB, lane-linear); ring of S units; the XOR swizzle is on the SOURCE
//     address (chunk ^ ((row >> 1) & 7)), the 16-row fragment reads (ds_read_b128) apply the same involution and are
//     conflict-free (the LDS image of the search kernels, kernels_mfma.h);
//   * wave (wm, wn) of the WM x WN grid owns TN x TM accumulator tiles of 16 x 16 (v_mfma_f32_16x16x32_bf16 with w as the
//     A operand and x as the B operand: lane l then holds y[m = l & 15][n = 4 (l >> 4) .. + 3] - four consecutive output
//     columns, one 8-byte store);
//   * per unit: counted `s_waitcnt vmcnt` (this wave's pieces of unit u have landed) -> raw s_barrier A -> fragment reads
//     of the first k-step -> half of its TN x TM MFMAs -> fragment reads of the second k-step -> the other half ->
//     lgkmcnt(0) -> raw s_barrier B (every wave holds the whole unit in registers) -> DMA of unit u + S into the slot
//     just read -> the MFMAs of the second k-step.  Nothing in the loop waits for vmcnt(0); the DMA instructions are inline asm, so the
//     compiler's own waits do not see them.
// XCD-aware tile order: workgroup ids are dealt round-robin over the 8 XCDs, so XCD x takes the x-th eighth of the tiles
// (consecutive row blocks of x, all of w through its own L2).
#pragma once
#include "../../../theoremsearch_amd/csrc/kernels_mfma16.h"

// timing experiments (wrong results; tools/linear_dbg.py builds them as libraries of their own): 1 = no stores, 2 = no DMA,
// 4 = no MFMA, 8 = fragment reads of the first unit only
#ifndef TS_LIN_DBG
#define TS_LIN_DBG 0
#endif

namespace ts {

struct LinearArgs {
    const unsigned char* x;      // [M][K] bf16
    const unsigned char* w;      // [N][K] bf16
    const unsigned short* bias;  // [N] bf16 or NULL
    unsigned short* y;           // [M][N] bf16
    int M, N, K;
    int mt, nt;                  // tiles along M and N (grid = mt * nt)
};

// DMA piece with default cache policy: these operands are re-read from L2 by every workgroup of a tile row / column
__device__ __forceinline__ void lin_dma16(unsigned voff, const void* sbase, unsigned lds_dst) {
    asm volatile(
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %0, %1"
        :
        : "v"(voff), "s"(sbase), "s"(lds_dst)
        : "memory");
}

// MFMA with pinned register classes (left alone the allocator shuttles accumulators between the VGPR and AGPR halves of
// the file and spills fragments, whose scratch loads then wait vmcnt(0) and drain the DMA ring): accumulators in AGPRs
// (AG) or, for what does not fit there (the 256 x 288 tile has 288), in VGPRs; fragments in VGPRs.  Accumulations chain
// MFMA -> MFMA; the epilogue reads the accumulators behind lin_settle().
template <bool AG>
__device__ __forceinline__ void lin_mfma(f32x4& acc, const bf16x8& wfrag, const bf16x8& xfrag) {
    if constexpr (AG) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(wfrag), "v"(xfrag));
    else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(wfrag), "v"(xfrag));
}
__device__ __forceinline__ void lin_settle() { asm volatile("s_nop 15\n\ts_nop 3" ::: "memory"); }

__device__ __forceinline__ float gelu_erf(float v) { return v * 0.5f * (1.0f + erff(v * 0.70710678118654752440f)); }

constexpr int linear_stage_pitch(int BN) { return BN * 2 + 16; }   // bytes per row of the epilogue's staging tile
constexpr int linear_lds_bytes(int BM, int BN, int S) {
    const int ring = S * (((BM + BN) / 8 + 3) / 4) * 4096, stage = BM * linear_stage_pitch(BN);
    return ring > stage ? ring : stage;
}

// ACT: 0 = none, 1 = exact erf GELU (of the bf16-rounded sum, as an elementwise GELU behind a bf16 GEMM sees it)
template <int BM, int BN, int WM, int WN, int S, int ACT>
__global__ void __launch_bounds__(256, 1) linear_bf16_kernel(LinearArgs a) {
    static_assert(WM * WN == 4, "four waves");
    static_assert(BM % (16 * WM) == 0 && BN % (16 * WN) == 0, "whole 16 x 16 tiles per wave");
    static_assert(S >= 2, "ring: one unit being read, one landing");
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
    constexpr int kPiecesW = BN / 8, kPiecesX = BM / 8, kPieces = kPiecesW + kPiecesX;
    constexpr int PPW = (kPieces + 3) / 4;                 // DMA pieces per wave and unit (the last ones may be fillers)
    constexpr int kUnitBytes = PPW * 4 * 1024;
    static_assert(S * kUnitBytes <= 160 * 1024, "LDS");
    static_assert((S - 1) * PPW <= 63, "vmcnt field");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int tiles = a.mt * a.nt;
    int t = blockIdx.x;
    if ((tiles & 7) == 0) t = (t & 7) * (tiles >> 3) + (t >> 3);
    const int m0 = (t / a.nt) * BM, n0 = (t % a.nt) * BN;

    // DMA sources of this lane: piece p = wave + 4 i covers rows 8 p .. 8 p + 7 of the unit (w rows first, then x rows), whole
    // 128-byte lines; lane l moves chunk (l & 7) ^ ((row >> 1) & 7) of row 8 p + (l >> 3) to position l of the piece
    unsigned voff[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        int p = wave + 4 * i;
        if (p >= kPieces) p = wave;                        // filler piece: lands in the unit's padding
        const int pp = p < kPiecesW ? p : p - kPiecesW;     // piece within its operand
        const int chunk = (lane & 7) ^ ((4 * pp + (lane >> 4)) & 7);
        int row;
        if (p < kPiecesW) row = n0 + 8 * pp + (lane >> 3);
        else row = min(m0 + 8 * pp + (lane >> 3), a.M - 1);
        voff[i] = (unsigned)row * (unsigned)(2 * a.K) + (unsigned)chunk * 16u;
    }
    const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem;
    const unsigned dma_dst0 = lds_base + wave * 1024;
    auto issue = [&](int unit, int slot) {
        const unsigned char* wk = a.w + (int64_t)unit * 128;
        const unsigned char* xk = a.x + (int64_t)unit * 128;
        const unsigned dst = dma_dst0 + slot * kUnitBytes;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int p = wave + 4 * i;
            const bool is_w = (p < kPiecesW) || (p >= kPieces && wave < kPiecesW);
            lin_dma16(voff[i], is_w ? wk : xk, dst + i * 4096);
        }
    };

    // fragment reads: lane (r16, q) reads, for k-step h of the unit, source chunk 4 h + q of row r16 of a 16-row block
    // (two pieces) = position (4 h + q) ^ ((r16 >> 1) & 7) of that row
    const int r16 = lane & 15, q = lane >> 4;
    const int lane_off = (r16 >> 3) * 1024 + (r16 & 7) * 128 + ((q ^ ((r16 >> 1) & 7)) << 4);
    typedef __attribute__((address_space(3))) unsigned char lds_u8;
    typedef __attribute__((address_space(3))) const bf16x8 lds_frag;
    lds_u8* const lbase = (lds_u8*)smem;
    const unsigned fw = (wn * TN) * 2048 + lane_off;
    const unsigned fx = kPiecesW * 1024 + (wm * TM) * 2048 + lane_off;
    bf16x8 wf0[TN], xf0[TM], wf1[TN], xf1[TM];
    auto load = [&](bf16x8 (&wf)[TN], bf16x8 (&xf)[TM], int slot, int h) {
        // h toggles bit 2 of the chunk index: byte 64 of the position (no other term of the address has that bit)
        const unsigned pw = (fw ^ (unsigned)(h * 64)) + slot * kUnitBytes;
        const unsigned px = (fx ^ (unsigned)(h * 64)) + slot * kUnitBytes;
#pragma unroll
        for (int i = 0; i < TN; ++i) wf[i] = *(lds_frag*)(lbase + pw + i * 2048);
#pragma unroll
        for (int j = 0; j < TM; ++j) xf[j] = *(lds_frag*)(lbase + px + j * 2048);
    };
    f32x4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // the MFMAs of a k-step in two halves (x blocks [0, TM / 2) and [TM / 2, TM)): the fragment reads of the unit's second
    // k-step go between them, so that they have half a k-step to land
    auto mma = [&](const bf16x8 (&wf)[TN], const bf16x8 (&xf)[TM], int j0, int j1) {
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                if (j < j0 || j >= j1) continue;
                if ((i * TM + j) * 4 < 256) lin_mfma<true>(acc[i][j], wf[i], xf[j]);
                else lin_mfma<false>(acc[i][j], wf[i], xf[j]);
            }
    };

    // the MFMAs of the unit's second k-step with the DMA pieces of the refill between them (one piece every kEvery MFMAs):
    // issued in one burst, a wave's PPW DMA instructions hold its in-order issue while the memory pipe takes them, and
    // with one wave per SIMD nothing else runs - DMA time and MFMA time then ADD (measured: 8 + 18 = 26 us on the qkv shape)
    constexpr int kEvery = (TM * TN) / PPW >= 1 ? (TM * TN) / PPW : 1;
    static_assert((TM * TN) / kEvery >= PPW, "every piece finds its place between the MFMAs");
    auto mma_issue = [&](const bf16x8 (&wf)[TN], const bf16x8 (&xf)[TM], int unit, int slot) {
        const unsigned char* wk = a.w + (int64_t)unit * 128;
        const unsigned char* xk = a.x + (int64_t)unit * 128;
        const unsigned dst = dma_dst0 + slot * kUnitBytes;
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                if ((i * TM + j) * 4 < 256) lin_mfma<true>(acc[i][j], wf[i], xf[j]);
                else lin_mfma<false>(acc[i][j], wf[i], xf[j]);
                const int c = j * TN + i;
                if (c % kEvery == kEvery - 1 && c / kEvery < PPW && !(TS_LIN_DBG & 2)) {
                    const int pi = c / kEvery;
                    const int p = wave + 4 * pi;
                    const bool is_w = (p < kPiecesW) || (p >= kPieces && wave < kPiecesW);
                    lin_dma16(voff[pi], is_w ? wk : xk, dst + pi * 4096);
                }
            }
    };

    const int nu = a.K / 64;
    int issued = 0;
    for (; issued < S; ++issued)                                          // every slot of the ring is filled (units past
        if (!(TS_LIN_DBG & 2)) issue(min(issued, nu - 1), issued);        // the end: the last one again, never read)
    int slot = 0;
    for (int u = 0; u < nu; ++u) {
        // unit u has landed (this wave's pieces; barrier A makes it everyone's); S - 1 later refills may be in flight - every
        // unit issues exactly PPW pieces (past the end of K the last unit again, into a slot nobody reads any more), so the
        // count is the same in every iteration
        if (!(TS_LIN_DBG & 2)) wait_vmcnt<(S - 1) * PPW>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (!(TS_LIN_DBG & 8) || u == 0) load(wf0, xf0, slot, 0);
        if (!(TS_LIN_DBG & 4)) mma(wf0, xf0, 0, TM / 2);
        asm volatile("" ::: "memory");
        if (!(TS_LIN_DBG & 8) || u == 0) load(wf1, xf1, slot, 1);
        if (!(TS_LIN_DBG & 4)) mma(wf0, xf0, TM / 2, TM);
        // every fragment of the unit is in registers a quarter of the way through it: behind barrier B the slot takes unit
        // u + S, which then has S - 1/4 units of MFMA time to land
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (!(TS_LIN_DBG & 4)) mma_issue(wf1, xf1, min(issued, nu - 1), slot);
        // wait states between the last MFMA and whatever hipcc places behind the loop: it knows nothing of what the MFMA
        // statements are, and read three accumulator registers of the 256 x 288 tile right behind the last of them (spilled
        // straight from the accumulation: 0.2 % of that tile's outputs wrong).  Inside the loop body nothing can be put
        // between the MFMAs and these; 20 cycles per unit of 1,500-2,300.
        lin_settle();
        ++issued;
        slot = (slot + 1 == S) ? 0 : slot + 1;
    }
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();                          // every wave's DMA has landed and every fragment read is done: the
    asm volatile("" ::: "memory");                         // ring's LDS becomes the staging tile of the epilogue

    // epilogue: lane holds y[m][n .. n + 3] of every accumulator tile; bias (+ GELU), round, 8 bytes into the staging tile
    // [BM][BN] (row pitch + 16 bytes); then the workgroup writes the tile out in whole rows, 16 bytes per lane (a first cut
    // stored the 8 bytes straight from the accumulator layout - 16 rows x 32 bytes per instruction: 19-22 us for the 38-50 MB
    // outputs, ~2 TB/s)
    constexpr int kPitch = linear_stage_pitch(BN);
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#pragma unroll
    for (int i = 0; i < TN; ++i) {
        const int nl = (wn * TN + i) * 16 + 4 * q;
        float b[4] = {0.f, 0.f, 0.f, 0.f};
        if (a.bias) {
            const uint2 bb = *(const uint2*)(a.bias + n0 + nl);
            b[0] = bf16_lo(bb.x); b[1] = bf16_hi(bb.x); b[2] = bf16_lo(bb.y); b[3] = bf16_hi(bb.y);
        }
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const int ml = (wm * TM + j) * 16 + r16;
            unsigned short o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                unsigned short h = f32_to_bf16(acc[i][j][r] + b[r]);
                if (ACT == 1) h = f32_to_bf16(gelu_erf(bf16_to_f32(h)));
                o[r] = h;
            }
            *(__attribute__((address_space(3))) u32x2*)(lbase + ml * kPitch + nl * 2) =
                u32x2{(u32)o[0] | ((u32)o[1] << 16), (u32)o[2] | ((u32)o[3] << 16)};
        }
    }
    __syncthreads();
    constexpr int kCPR = BN / 8;                           // 16-byte chunks per row
    for (int idx = threadIdx.x; idx < BM * kCPR; idx += 256) {
        const int row = idx / kCPR, c = idx - row * kCPR;
        const u32x4 v = *(__attribute__((address_space(3))) const u32x4*)(lbase + row * kPitch + c * 16);
        if (m0 + row < a.M && !(TS_LIN_DBG & 1))
            *(u32x4*)((unsigned char*)a.y + ((int64_t)(m0 + row) * a.N + n0) * 2 + c * 16) = v;
    }
}

}  // namespace ts
