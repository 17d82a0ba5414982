// Self-attention of the encoder for short sequences (BertSelfAttention of the sentence-transformer the reference loads,
// compare_embeddings.py:11-12; queries are one sentence each, `model.encode` at app_showcase_model.py:92): softmax(Q K^T / 8 +
// key mask) V for head size 64 and at most 64 tokens, bf16 in / out, one WAVE per (sequence, head), straight from the fused
// projection's output [B][S][3][H][64] into the context layout [B][S][H * 64] the output projection reads.
//
// Why a kernel of our own: at 256 sequences x 32 tokens the library's flash-attention launch takes 18.7 us per layer for 50 MB
// of traffic and 0.8 GFLOP - it is built for long sequences (tiles of 64-128 queries per workgroup); here the whole problem of
// a (sequence, head) is 12 KB and sixteen MFMAs, and the launch is bound by how many of them are in flight.
//
//   * scores: v_mfma_f32_16x16x32_bf16 with Q rows as the A operand and K rows as the B operand, both read from global memory as
//     the fragments they are (16 bytes per lane along the head dimension): D[q][key], lane l holds q = 4 (l >> 4) + r, key = l & 15
//     of each 16 x 16 tile;
//   * softmax over the keys in fp32: the 16 lanes of a quarter-wave hold one row's keys of a tile (xor-shuffles 1, 2, 4, 8), the
//     tiles are registers; padded keys (attention_mask = 0) and keys past the sequence are -inf;
//   * P goes to LDS as bf16 [q][key] (wave-private) and comes back as the B operand of O^T = V^T P^T, whose A operand (V^T:
//     lane l holds V[8 (l >> 4) + j][l & 15]) comes from a transposed LDS image of V (16-byte loads of the rows, 4-byte writes
//     of key pairs);
//   * O^T leaves the accumulators as [d = 4 (l >> 4) + r][q = l & 15]: four consecutive d per lane, staged through the same LDS and
//     written as whole 128-byte rows.
#pragma once
#include "kernels_mfma16.h"

namespace ts {

constexpr int kAttnMaxSeq = 64;
constexpr int attn_wave_lds(int T) {
    const int sp = 16 * T, ks = (sp + 31) / 32;
    const int p_bytes = sp * (32 * ks + 8) * 2, o_bytes = sp * (64 + 8) * 2;
    return (p_bytes > o_bytes ? p_bytes : o_bytes) + 64 * (32 * ks + 8) * 2;      // + V^T [64][keys]
}
constexpr int attn_vt_offset(int T) {
    const int sp = 16 * T, ks = (sp + 31) / 32;
    const int p_bytes = sp * (32 * ks + 8) * 2, o_bytes = sp * (64 + 8) * 2;
    return p_bytes > o_bytes ? p_bytes : o_bytes;
}

// T = ceil(S / 16).  grid = ceil(B * H / 4), 256 threads; no workgroup barrier (every wave works in its own LDS slice).
template <int T>
__global__ void __launch_bounds__(256) attention_short_kernel(const unsigned short* __restrict__ qkv, const int64_t* __restrict__ mask,
                                                               int B, int S, int H, unsigned short* __restrict__ out) {
    constexpr int SP = 16 * T;                       // padded sequence
    constexpr int KS = (SP + 31) / 32;               // 32-key steps of the second product
    constexpr int PP = 32 * KS + 8;                  // pitch of P rows (elements): 16 bytes of padding spread the banks
    constexpr int OP = 64 + 8;                       // pitch of O rows
    __shared__ __attribute__((aligned(16))) unsigned char sbuf[4][attn_wave_lds(T)];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int bh = blockIdx.x * 4 + wave;
    if (bh >= B * H) return;
    const int b = bh / H, h = bh - b * H;
    const int r16 = lane & 15, g = lane >> 4;
    const int64_t tok = (int64_t)3 * H * 64;         // elements per token of the projection's output
    const unsigned short* base = qkv + (int64_t)b * S * tok + h * 64;
    unsigned short* sP = (unsigned short*)sbuf[wave];

    // every load of the problem is requested before the first use
    bf16x8 qf[T][2], kf[T][2];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int row = min(16 * t + r16, S - 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            qf[t][ks] = *(const bf16x8*)(base + row * tok + 32 * ks + 8 * g);
            kf[t][ks] = *(const bf16x8*)(base + row * tok + H * 64 + 32 * ks + 8 * g);
        }
    }
    // V rows as they lie (16 bytes per lane: lane -> key pair lane / 8 + 8 i, columns 8 (lane & 7) ..); transposed through LDS below
    constexpr int KP = SP / 2;                       // key pairs
    constexpr int VI = (KP + 7) / 8;                 // pairs per lane
    uint4 v0[VI], v1[VI];
#pragma unroll
    for (int i = 0; i < VI; ++i) {
        const int kp = (lane >> 3) + 8 * i;
        const int k0 = min(2 * kp, S - 1), k1 = min(2 * kp + 1, S - 1);
        v0[i] = *(const uint4*)(base + k0 * tok + 2 * H * 64 + 8 * (lane & 7));
        v1[i] = *(const uint4*)(base + k1 * tok + 2 * H * 64 + 8 * (lane & 7));
    }
    bool keyok[T];
#pragma unroll
    for (int kj = 0; kj < T; ++kj) {
        const int key = 16 * kj + r16;
        keyok[kj] = key < S && (!mask || mask[(int64_t)b * S + key] != 0);
    }
    // P's LDS image starts as zeros: keys past the padded sequence inside the last 32-key step contribute nothing
    for (int i = lane; i < SP * PP / 8; i += 64) ((uint4*)sP)[i] = make_uint4(0u, 0u, 0u, 0u);
    // V^T [d][key] in LDS: a lane holds 8 columns of two consecutive keys and writes 8 dwords (column d, keys 2 kp and 2 kp + 1);
    // the second product reads its A fragments (column d = 16 dj + r16, 8 consecutive keys) as 16 bytes.  A first cut gathered
    // those fragments from global memory with 2-byte loads: 32 instructions of 128 bytes each per wave.
    unsigned short* sVT = (unsigned short*)(sbuf[wave] + attn_vt_offset(T));
    if (KS * 32 > SP)                                // keys past the padded sequence: zeros (0 x P = 0, never NaN)
        for (int i = lane; i < 64 * PP / 8; i += 64) ((uint4*)sVT)[i] = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
    for (int i = 0; i < VI; ++i) {
        const int kp = (lane >> 3) + 8 * i;
        if (kp < KP) {
            const u32 a[4] = {v0[i].x, v0[i].y, v0[i].z, v0[i].w}, c[4] = {v1[i].x, v1[i].y, v1[i].z, v1[i].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                u32* dst = (u32*)(sVT + (8 * (lane & 7) + 2 * e) * PP + 2 * kp);
                dst[0] = (a[e] & 0xFFFFu) | (c[e] << 16);                     // column 2 e: (key 2 kp, key 2 kp + 1)
                *(u32*)((unsigned short*)dst + PP) = (a[e] >> 16) | (c[e] & 0xFFFF0000u);   // column 2 e + 1
            }
        }
    }

    // scores D[q][key]
    f32x4 sc[T][T];
#pragma unroll
    for (int qi = 0; qi < T; ++qi)
#pragma unroll
        for (int kj = 0; kj < T; ++kj) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[qi][ks], kf[kj][ks], a, 0, 0, 0);
            sc[qi][kj] = a;
        }
    // softmax over the keys of row q = 16 qi + 4 g + r: tiles kj are registers, the 16 keys of a tile are the lanes of this quarter
    constexpr float kScaleLog2e = 0.125f * 1.4426950408889634f;     // 1 / sqrt(64), exp through exp2
#pragma unroll
    for (int qi = 0; qi < T; ++qi)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float m = -INFINITY;
#pragma unroll
            for (int kj = 0; kj < T; ++kj) m = fmaxf(m, keyok[kj] ? sc[qi][kj][r] : -INFINITY);
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
            float e[T], sum = 0.0f;
#pragma unroll
            for (int kj = 0; kj < T; ++kj) {
                e[kj] = keyok[kj] ? exp2f((sc[qi][kj][r] - m) * kScaleLog2e) : 0.0f;
                sum += e[kj];
            }
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) sum += __shfl_xor(sum, off, 64);
            const float inv = sum > 0.0f ? 1.0f / sum : 0.0f;        // a row without a single allowed key: zeros
#pragma unroll
            for (int kj = 0; kj < T; ++kj) sP[(16 * qi + 4 * g + r) * PP + 16 * kj + r16] = (unsigned short)pack_bf16_hw(e[kj] * inv, 0.0f);
        }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // O^T = V^T P^T: A = V^T fragment (d = 16 dj + r16, keys 32 ks + 8 g ..), B = P^T fragment (keys .., q = 16 qi + r16)
    bf16x8 pf[T][KS];
#pragma unroll
    for (int qi = 0; qi < T; ++qi)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) pf[qi][ks] = *(const bf16x8*)(sP + (16 * qi + r16) * PP + 32 * ks + 8 * g);
    bf16x8 vf[4][KS];
#pragma unroll
    for (int dj = 0; dj < 4; ++dj)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) vf[dj][ks] = *(const bf16x8*)(sVT + (16 * dj + r16) * PP + 32 * ks + 8 * g);
    f32x4 oc[4][T];
#pragma unroll
    for (int dj = 0; dj < 4; ++dj)
#pragma unroll
        for (int qi = 0; qi < T; ++qi) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[dj][ks], pf[qi][ks], a, 0, 0, 0);
            oc[dj][qi] = a;
        }
    // the P image has been read (the fragments are in registers): the same LDS takes O as [q][d]
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    unsigned short* sO = sP;
#pragma unroll
    for (int dj = 0; dj < 4; ++dj)
#pragma unroll
        for (int qi = 0; qi < T; ++qi) {
            const u32 lo = pack_bf16_hw(oc[dj][qi][0], oc[dj][qi][1]);
            const u32 hi = pack_bf16_hw(oc[dj][qi][2], oc[dj][qi][3]);
            *(uint2*)(sO + (16 * qi + r16) * OP + 16 * dj + 4 * g) = make_uint2(lo, hi);
        }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    unsigned short* obase = out + (int64_t)b * S * H * 64 + h * 64;
    for (int i = lane; i < SP * 8; i += 64) {
        const int q = i >> 3, c = i & 7;
        if (q < S) *(uint4*)(obase + (int64_t)q * H * 64 + 8 * c) = *(const uint4*)(sO + q * OP + 8 * c);
    }
}

// ---- 65 .. 128 tokens: the same wave-per-(sequence, head) form, one 16-query tile at a time --------------------------------
// Real inputs of the reference are `global_context + statement` (app_create_embeddings.py:48-70), which its own notes call long
// (notes.md:3); SURVEY.md section 8d names 128 tokens as the second length of the encoder-in-loop measurement.  Holding every
// score tile of a (sequence, head) at once would take T x T x 4 registers (256 at T = 8); here the K fragments (T x 2) and the
// V^T fragments (4 x KS) stay in registers for the whole problem and the wave walks the query tiles: scores of ONE tile against
// all keys (T x 4 registers) -> softmax over registers and quarter-wave shuffles, exactly as above (every key of a row is in
// hand: no running maximum, no rescaling) -> P tile through wave-private LDS -> O^T tile -> whole 128-byte rows out.
// LDS per wave: V^T [64][PP] bf16 (17 KB at T = 8) - and nothing else: once the V^T fragments are in registers the image is
// dead, and the P tile [16][PP] and the O tile [16][72] of the query loop live in its place.  Two workgroups per CU instead of
// one (the first cut kept all three: 22.5 KB per wave, one wave per SIMD, every latency of the serial chain exposed).
constexpr int kAttnRowsMaxSeq = 128;
constexpr int attn_rows_wave_lds(int T) {
    const int sp = 16 * T, ks = (sp + 31) / 32, pp = 32 * ks + 8;
    return 64 * pp * 2;
}

template <int T>
__global__ void __launch_bounds__(256) attention_rows_kernel(const unsigned short* __restrict__ qkv, const int64_t* __restrict__ mask,
                                                              int B, int S, int H, unsigned short* __restrict__ out) {
    constexpr int SP = 16 * T;
    constexpr int KS = (SP + 31) / 32;
    constexpr int PP = 32 * KS + 8;
    constexpr int OP = 64 + 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char attn_smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int bh = blockIdx.x * 4 + wave;
    if (bh >= B * H) return;
    const int b = bh / H, h = bh - b * H;
    const int r16 = lane & 15, g = lane >> 4;
    const int64_t tok = (int64_t)3 * H * 64;
    const unsigned short* base = qkv + (int64_t)b * S * tok + h * 64;
    unsigned short* sVT = (unsigned short*)(attn_smem + (size_t)wave * attn_rows_wave_lds(T));
    unsigned short* sP = sVT;                      // after the V^T fragments have been read (below)
    unsigned short* sO = sVT + 16 * PP;
    static_assert(16 * PP + 16 * OP <= 64 * PP, "P and O tiles fit the V^T image");

    // K fragments of every key tile, the first query tile's fragments and the V rows: all requested before the first use
    bf16x8 kf[T][2], qf[2];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int row = min(16 * t + r16, S - 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) kf[t][ks] = *(const bf16x8*)(base + row * tok + H * 64 + 32 * ks + 8 * g);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) qf[ks] = *(const bf16x8*)(base + min(r16, S - 1) * tok + 32 * ks + 8 * g);
    constexpr int KP = SP / 2;
    constexpr int VI = (KP + 7) / 8;
    uint4 v0[VI], v1[VI];
#pragma unroll
    for (int i = 0; i < VI; ++i) {
        const int kp = (lane >> 3) + 8 * i;
        const int k0 = min(2 * kp, S - 1), k1 = min(2 * kp + 1, S - 1);
        v0[i] = *(const uint4*)(base + k0 * tok + 2 * H * 64 + 8 * (lane & 7));
        v1[i] = *(const uint4*)(base + k1 * tok + 2 * H * 64 + 8 * (lane & 7));
    }
    bool keyok[T];
#pragma unroll
    for (int kj = 0; kj < T; ++kj) {
        const int key = 16 * kj + r16;
        keyok[kj] = key < S && (!mask || mask[(int64_t)b * S + key] != 0);
    }
    // keys past the padded sequence inside the last 32-key step: zeros in V^T (and in P, below)
    if (KS * 32 > SP)
        for (int i = lane; i < 64 * PP / 8; i += 64) ((uint4*)sVT)[i] = make_uint4(0u, 0u, 0u, 0u);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int i = 0; i < VI; ++i) {
        const int kp = (lane >> 3) + 8 * i;
        if (kp < KP) {
            const u32 a[4] = {v0[i].x, v0[i].y, v0[i].z, v0[i].w}, c[4] = {v1[i].x, v1[i].y, v1[i].z, v1[i].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                u32* dst = (u32*)(sVT + (8 * (lane & 7) + 2 * e) * PP + 2 * kp);
                dst[0] = (a[e] & 0xFFFFu) | (c[e] << 16);
                *(u32*)((unsigned short*)dst + PP) = (a[e] >> 16) | (c[e] & 0xFFFF0000u);
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    bf16x8 vf[4][KS];
#pragma unroll
    for (int dj = 0; dj < 4; ++dj)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) vf[dj][ks] = *(const bf16x8*)(sVT + (16 * dj + r16) * PP + 32 * ks + 8 * g);
    // the image is dead from here on (the wave's LDS operations execute in order: the reads above come first); the P tile
    // takes its place, zeroed once: the tile's stores stop at SP, the columns past it stay zero
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int i = lane; i < 16 * PP / 8; i += 64) ((uint4*)sP)[i] = make_uint4(0u, 0u, 0u, 0u);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    constexpr float kScaleLog2e = 0.125f * 1.4426950408889634f;
    unsigned short* obase = out + (int64_t)b * S * H * 64 + h * 64;
    const int ntile = (S + 15) / 16;
#pragma unroll 1
    for (int qi = 0; qi < ntile; ++qi) {
        // the next tile's query fragments are on their way while this one is worked on
        bf16x8 qn[2];
        const int nrow = min(16 * (qi + 1) + r16, S - 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) qn[ks] = *(const bf16x8*)(base + nrow * tok + 32 * ks + 8 * g);
        f32x4 sc[T];
#pragma unroll
        for (int kj = 0; kj < T; ++kj) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[ks], kf[kj][ks], a, 0, 0, 0);
            sc[kj] = a;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float m = -INFINITY;
#pragma unroll
            for (int kj = 0; kj < T; ++kj) m = fmaxf(m, keyok[kj] ? sc[kj][r] : -INFINITY);
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
            float e[T], sum = 0.0f;
#pragma unroll
            for (int kj = 0; kj < T; ++kj) {
                e[kj] = keyok[kj] ? exp2f((sc[kj][r] - m) * kScaleLog2e) : 0.0f;
                sum += e[kj];
            }
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) sum += __shfl_xor(sum, off, 64);
            const float inv = sum > 0.0f ? 1.0f / sum : 0.0f;
#pragma unroll
            for (int kj = 0; kj < T; ++kj) sP[(4 * g + r) * PP + 16 * kj + r16] = (unsigned short)pack_bf16_hw(e[kj] * inv, 0.0f);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        bf16x8 pf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) pf[ks] = *(const bf16x8*)(sP + r16 * PP + 32 * ks + 8 * g);
#pragma unroll
        for (int dj = 0; dj < 4; ++dj) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[dj][ks], pf[ks], a, 0, 0, 0);
            const u32 lo = pack_bf16_hw(a[0], a[1]);
            const u32 hi = pack_bf16_hw(a[2], a[3]);
            *(uint2*)(sO + r16 * OP + 16 * dj + 4 * g) = make_uint2(lo, hi);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int i = lane; i < 16 * 8; i += 64) {
            const int q = 16 * qi + (i >> 3), c = i & 7;
            if (q < S) *(uint4*)(obase + (int64_t)q * H * 64 + 8 * c) = *(const uint4*)(sO + (i >> 3) * OP + 8 * c);
        }
        // the O tile has been read and the P tile's fragments are in registers: the next tile may overwrite both
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        qf[0] = qn[0];
        qf[1] = qn[1];
    }
}

// ---- head size 128, grouped-query, causal: the attention of the Qwen3-family encoder ---------------------------------------
// Qwen/Qwen3-Embedding-0.6B is what the production app embeds with (streamlit_app.py:55, ec2/generate_embeddings/embedders.py:1-4):
// 16 query heads over 8 key / value heads of 128, causal, queries of one sentence each.  At 256 sequences x 32 tokens the
// library's flash-attention launch takes 202 us per layer (profiles/r04_c5_qwen_kernel_stats.csv: 5.7 of the step's 20.7 ms);
// the whole problem of a (sequence, query head) is 24 KB and 32 MFMAs.  The same wave-per-problem form as attention_short_kernel:
//   qkv [tokens][(hq + 2 hkv) * 128]: query heads, key heads, value heads of each token (the output of ONE GEMM over the
//   stacked projection weights, after ts_qk_norm_rope);  query head h reads key / value head h / (hq / hkv);
//   scores D[q][key] = Q K^T over four 32-deep k-steps, allowed = key <= q (causal) and not a padding key, softmax in fp32
//   (scale 1 / sqrt(128)), P -> LDS (bf16) -> O^T = V^T P^T with V^T [128][keys] transposed through LDS;
//   out [tokens][hq * 128].
// (The per-head RMSNorm of q / k and the rotary embedding were also applied to the fragments on the way in, in place of
// ts_qk_norm_rope's launch: the transform took the kernel from two or three waves per SIMD to one, and the step of the
// Qwen3-shaped encoder did not move - 14.60 against 14.65 ms.  Removed; profiles/HISTORY.md, round 4.)
// LDS per wave: V^T image + P image; the O tile reuses the V^T image once its fragments are in registers.
constexpr int kAttnGqaMaxSeq = 64;
constexpr int attn_gqa_wave_lds(int T) {
    const int sp = 16 * T, ks = (sp + 31) / 32, pp = 32 * ks + 8;
    return 128 * pp * 2 + sp * pp * 2;
}

template <int T, bool CAUSAL>
__global__ void __launch_bounds__(256, (T <= 2 ? 2 : 1)) attention_gqa_kernel(const unsigned short* __restrict__ qkv,
                                                                               const int64_t* __restrict__ mask, int B, int S, int HQ, int HKV,
                                                                               unsigned short* __restrict__ out) {
    constexpr int HD = 128;
    constexpr int SP = 16 * T;
    constexpr int KS = (SP + 31) / 32;
    constexpr int PP = 32 * KS + 8;
    constexpr int OP = HD + 8;
    static_assert(SP * OP <= HD * PP, "the O tile fits the V^T image");
    extern __shared__ __attribute__((aligned(16))) unsigned char attn_smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int bh = blockIdx.x * 4 + wave;
    if (bh >= B * HQ) return;
    const int b = bh / HQ, h = bh - b * HQ;
    const int kvh = h / (HQ / HKV);
    const int r16 = lane & 15, g = lane >> 4;
    const int64_t tok = (int64_t)(HQ + 2 * HKV) * HD;
    const unsigned short* qb = qkv + (int64_t)b * S * tok + (int64_t)h * HD;
    const unsigned short* kb = qkv + (int64_t)b * S * tok + (int64_t)(HQ + kvh) * HD;
    const unsigned short* vb = qkv + (int64_t)b * S * tok + (int64_t)(HQ + HKV + kvh) * HD;
    unsigned short* sVT = (unsigned short*)(attn_smem + (size_t)wave * attn_gqa_wave_lds(T));
    unsigned short* sP = sVT + HD * PP;
    unsigned short* sO = sVT;                        // after the V^T fragments have been read

    bf16x8 qf[T][4], kf[T][4];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int row = min(16 * t + r16, S - 1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            qf[t][ks] = *(const bf16x8*)(qb + row * tok + 32 * ks + 8 * g);
            kf[t][ks] = *(const bf16x8*)(kb + row * tok + 32 * ks + 8 * g);
        }
    }
    // V rows as they lie: 16 lanes x 16 bytes per row, four key pairs per pass (lane >> 4)
    constexpr int KP = SP / 2;
    constexpr int VI = (KP + 3) / 4;
    uint4 v0[VI], v1[VI];
#pragma unroll
    for (int i = 0; i < VI; ++i) {
        const int kp = g + 4 * i;
        const int k0 = min(2 * kp, S - 1), k1 = min(2 * kp + 1, S - 1);
        v0[i] = *(const uint4*)(vb + k0 * tok + 8 * r16);
        v1[i] = *(const uint4*)(vb + k1 * tok + 8 * r16);
    }
    bool keyok[T];
#pragma unroll
    for (int kj = 0; kj < T; ++kj) {
        const int key = 16 * kj + r16;
        keyok[kj] = key < S && (!mask || mask[(int64_t)b * S + key] != 0);
    }
    for (int i = lane; i < SP * PP / 8; i += 64) ((uint4*)sP)[i] = make_uint4(0u, 0u, 0u, 0u);
    if (KS * 32 > SP)
        for (int i = lane; i < HD * PP / 8; i += 64) ((uint4*)sVT)[i] = make_uint4(0u, 0u, 0u, 0u);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int i = 0; i < VI; ++i) {
        const int kp = g + 4 * i;
        if (kp < KP) {
            const u32 a[4] = {v0[i].x, v0[i].y, v0[i].z, v0[i].w}, c[4] = {v1[i].x, v1[i].y, v1[i].z, v1[i].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                u32* dst = (u32*)(sVT + (8 * r16 + 2 * e) * PP + 2 * kp);
                dst[0] = (a[e] & 0xFFFFu) | (c[e] << 16);
                *(u32*)((unsigned short*)dst + PP) = (a[e] >> 16) | (c[e] & 0xFFFF0000u);
            }
        }
    }

    f32x4 sc[T][T];
#pragma unroll
    for (int qi = 0; qi < T; ++qi)
#pragma unroll
        for (int kj = 0; kj < T; ++kj) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
            if (!CAUSAL || kj <= qi) {               // a key tile past the query tile holds no allowed key
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[qi][ks], kf[kj][ks], a, 0, 0, 0);
            }
            sc[qi][kj] = a;
        }
    constexpr float kScaleLog2e = 0.08838834764831845f * 1.4426950408889634f;     // 1 / sqrt(128), exp through exp2
#pragma unroll
    for (int qi = 0; qi < T; ++qi)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int qrow = 16 * qi + 4 * g + r;
            bool ok[T];
#pragma unroll
            for (int kj = 0; kj < T; ++kj) ok[kj] = keyok[kj] && (!CAUSAL || 16 * kj + r16 <= qrow);
            float m = -INFINITY;
#pragma unroll
            for (int kj = 0; kj < T; ++kj) m = fmaxf(m, ok[kj] ? sc[qi][kj][r] : -INFINITY);
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
            float e[T], sum = 0.0f;
#pragma unroll
            for (int kj = 0; kj < T; ++kj) {
                e[kj] = ok[kj] ? exp2f((sc[qi][kj][r] - m) * kScaleLog2e) : 0.0f;
                sum += e[kj];
            }
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) sum += __shfl_xor(sum, off, 64);
            const float inv = sum > 0.0f ? 1.0f / sum : 0.0f;        // a row without a single allowed key: zeros
#pragma unroll
            for (int kj = 0; kj < T; ++kj) sP[qrow * PP + 16 * kj + r16] = (unsigned short)pack_bf16_hw(e[kj] * inv, 0.0f);
        }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    bf16x8 pf[T][KS];
#pragma unroll
    for (int qi = 0; qi < T; ++qi)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) pf[qi][ks] = *(const bf16x8*)(sP + (16 * qi + r16) * PP + 32 * ks + 8 * g);
    // O^T tile by tile of 16 head dimensions: the V^T fragments of a tile are read, used and dropped (8 x KS x 4 registers
    // would not fit beside the rest at T = 4)
    f32x4 oc[HD / 16][T];
#pragma unroll
    for (int dj = 0; dj < HD / 16; ++dj) {
        bf16x8 vf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) vf[ks] = *(const bf16x8*)(sVT + (16 * dj + r16) * PP + 32 * ks + 8 * g);
#pragma unroll
        for (int qi = 0; qi < T; ++qi) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[ks], pf[qi][ks], a, 0, 0, 0);
            oc[dj][qi] = a;
        }
    }
    // the V^T image has been read: the same LDS takes O as [q][d]
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int dj = 0; dj < HD / 16; ++dj)
#pragma unroll
        for (int qi = 0; qi < T; ++qi) {
            const u32 lo = pack_bf16_hw(oc[dj][qi][0], oc[dj][qi][1]);
            const u32 hi = pack_bf16_hw(oc[dj][qi][2], oc[dj][qi][3]);
            *(uint2*)(sO + (16 * qi + r16) * OP + 16 * dj + 4 * g) = make_uint2(lo, hi);
        }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    unsigned short* obase = out + (int64_t)b * S * HQ * HD + (int64_t)h * HD;
    for (int i = lane; i < SP * 16; i += 64) {
        const int q = i >> 4, c = i & 15;
        if (q < S) *(uint4*)(obase + (int64_t)q * HQ * HD + 8 * c) = *(const uint4*)(sO + q * OP + 8 * c);
    }
}

// ---- head size 128, grouped-query, causal, 65 .. 128 tokens: one 16-query tile at a time ------------------------------------------
// The production embedder's real inputs are `global_context + statement` (app_create_embeddings.py:48-70): longer than 64 tokens
// more often than not, and torch's flash-attention launch takes 504 us per layer at 256 sequences x 128 tokens (14 of the 48 ms of
// the Qwen3-shaped encoder-in-loop step).  The form of attention_rows_kernel with attention_gqa_kernel's layout: the K fragments of
// the whole sequence (T x 4 x 4 registers) stay in registers, V^T [128][keys] stays in LDS (its fragments are read per query
// tile: 128 more registers do not exist), the wave walks the query tiles; causal: tile qi multiplies key tiles 0 .. qi only.
// LDS: one V^T image per workgroup + per wave ONE tile that is the P tile first and the O tile after P's fragments have been read.
constexpr int kAttnGqaRowsMaxSeq = 128;
constexpr int attn_gqa_rows_tile_bytes(int T) {
    const int sp = 16 * T, ks = (sp + 31) / 32, pp = 32 * ks + 8;
    return 16 * (pp > 136 ? pp : 136) * 2;
}
constexpr int attn_gqa_rows_lds(int T, int R) {            // one V^T image per workgroup + one tile per wave
    const int sp = 16 * T, ks = (sp + 31) / 32, pp = 32 * ks + 8;
    return 128 * pp * 2 + R * attn_gqa_rows_tile_bytes(T);
}

// R = query heads per key / value head served by ONE workgroup of R waves (1, 2 or 4; grid = B * HKV * (HQ / HKV / R)): the
// waves share the V^T image of their key / value head - each transposes its share of the rows - instead of building one each
// (R = 2 for Qwen3's 16 / 8 heads: 43 KB of LDS per workgroup instead of 78, three workgroups per CU instead of two waves' worth).
template <int T, bool CAUSAL, int R>
__global__ void __launch_bounds__(64 * R) attention_gqa_rows_kernel(const unsigned short* __restrict__ qkv, const int64_t* __restrict__ mask,
                                                                     int B, int S, int HQ, int HKV, unsigned short* __restrict__ out) {
    constexpr int HD = 128;
    constexpr int SP = 16 * T;
    constexpr int KS = (SP + 31) / 32;
    constexpr int PP = 32 * KS + 8;
    constexpr int OP = HD + 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char attn_smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int per_kv = HQ / HKV;                                  // query heads per key / value head (a multiple of R)
    const int groups = per_kv / R;                                // workgroups per (sequence, key / value head)
    const int bk = blockIdx.x / groups, sub = blockIdx.x - bk * groups;
    const int b = bk / HKV, kvh = bk - b * HKV;
    const int h = kvh * per_kv + sub * R + wave;
    const int r16 = lane & 15, g = lane >> 4;
    const int64_t tok = (int64_t)(HQ + 2 * HKV) * HD;
    const unsigned short* qb = qkv + (int64_t)b * S * tok + (int64_t)h * HD;
    const unsigned short* kb = qkv + (int64_t)b * S * tok + (int64_t)(HQ + kvh) * HD;
    const unsigned short* vb = qkv + (int64_t)b * S * tok + (int64_t)(HQ + HKV + kvh) * HD;
    unsigned short* sVT = (unsigned short*)attn_smem;
    unsigned short* sP = (unsigned short*)(attn_smem + HD * PP * 2 + (size_t)wave * attn_gqa_rows_tile_bytes(T));   // P tile, then the O tile
    unsigned short* sO = sP;

    bf16x8 kf[T][4], qf[4];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int row = min(16 * t + r16, S - 1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) kf[t][ks] = *(const bf16x8*)(kb + row * tok + 32 * ks + 8 * g);
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const bf16x8*)(qb + min(r16, S - 1) * tok + 32 * ks + 8 * g);
    // V rows -> the shared V^T image: 16 lanes x 16 bytes per row, four key pairs per pass, the passes dealt over the waves
    constexpr int KP = SP / 2;
    constexpr int VI = (KP + 3) / 4;
    if (KS * 32 > SP) {
        for (int i = threadIdx.x; i < HD * PP / 8; i += 64 * R) ((uint4*)sVT)[i] = make_uint4(0u, 0u, 0u, 0u);
        __syncthreads();
    }
    constexpr int VW = (VI + R - 1) / R;                          // passes per wave
#pragma unroll
    for (int i0 = 0; i0 < VW; i0 += 4) {                          // four passes of loads in flight at a time
        uint4 v0[4], v1[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int kp = g + 4 * ((i0 + i) * R + wave);
            const int k0 = min(2 * kp, S - 1), k1 = min(2 * kp + 1, S - 1);
            v0[i] = *(const uint4*)(vb + k0 * tok + 8 * r16);
            v1[i] = *(const uint4*)(vb + k1 * tok + 8 * r16);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int kp = g + 4 * ((i0 + i) * R + wave);
            if (i0 + i < VW && kp < KP) {
                const u32 a[4] = {v0[i].x, v0[i].y, v0[i].z, v0[i].w}, c[4] = {v1[i].x, v1[i].y, v1[i].z, v1[i].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    u32* dst = (u32*)(sVT + (8 * r16 + 2 * e) * PP + 2 * kp);
                    dst[0] = (a[e] & 0xFFFFu) | (c[e] << 16);
                    *(u32*)((unsigned short*)dst + PP) = (a[e] >> 16) | (c[e] & 0xFFFF0000u);
                }
            }
        }
    }
    __syncthreads();                                              // the image is complete; from here on the waves go their own ways
    bool keyok[T];
#pragma unroll
    for (int kj = 0; kj < T; ++kj) {
        const int key = 16 * kj + r16;
        keyok[kj] = key < S && (!mask || mask[(int64_t)b * S + key] != 0);
    }
    constexpr float kScaleLog2e = 0.08838834764831845f * 1.4426950408889634f;
    unsigned short* obase = out + (int64_t)b * S * HQ * HD + (int64_t)h * HD;
    const int ntile = (S + 15) / 16;
#pragma unroll 1
    for (int qi = 0; qi < ntile; ++qi) {
        bf16x8 qn[4];
        const int nrow = min(16 * (qi + 1) + r16, S - 1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qn[ks] = *(const bf16x8*)(qb + nrow * tok + 32 * ks + 8 * g);
        // the tile (P now) starts as zeros: the columns this query tile does not write (keys past it, keys past the padded sequence) stay zero
        for (int i = lane; i < 16 * PP / 8; i += 64) ((uint4*)sP)[i] = make_uint4(0u, 0u, 0u, 0u);
        f32x4 sc[T];
#pragma unroll
        for (int kj = 0; kj < T; ++kj) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
            if (!CAUSAL || kj <= qi) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[ks], kf[kj][ks], a, 0, 0, 0);
            }
            sc[kj] = a;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int qrow = 16 * qi + 4 * g + r;
            bool ok[T];
#pragma unroll
            for (int kj = 0; kj < T; ++kj) ok[kj] = keyok[kj] && (!CAUSAL || 16 * kj + r16 <= qrow);
            float m = -INFINITY;
#pragma unroll
            for (int kj = 0; kj < T; ++kj) m = fmaxf(m, ok[kj] ? sc[kj][r] : -INFINITY);
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
            float e[T], sum = 0.0f;
#pragma unroll
            for (int kj = 0; kj < T; ++kj) {
                e[kj] = ok[kj] ? exp2f((sc[kj][r] - m) * kScaleLog2e) : 0.0f;
                sum += e[kj];
            }
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) sum += __shfl_xor(sum, off, 64);
            const float inv = sum > 0.0f ? 1.0f / sum : 0.0f;
#pragma unroll
            for (int kj = 0; kj < T; ++kj)
                if (!CAUSAL || kj <= qi) sP[(4 * g + r) * PP + 16 * kj + r16] = (unsigned short)pack_bf16_hw(e[kj] * inv, 0.0f);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        bf16x8 pf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) pf[ks] = *(const bf16x8*)(sP + r16 * PP + 32 * ks + 8 * g);
        const int ks_end = CAUSAL ? min(KS, (16 * (qi + 1) + 31) / 32) : KS;       // 32-key steps that hold an allowed key
        f32x4 oc[HD / 16];
#pragma unroll
        for (int dj = 0; dj < HD / 16; ++dj) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                if (ks < ks_end) {
                    const bf16x8 vf = *(const bf16x8*)(sVT + (16 * dj + r16) * PP + 32 * ks + 8 * g);
                    a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[ks], a, 0, 0, 0);
                }
            oc[dj] = a;
        }
        // P's fragments are in registers: the tile takes O as [q][d]
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int dj = 0; dj < HD / 16; ++dj)
            *(uint2*)(sO + r16 * OP + 16 * dj + 4 * g) = make_uint2(pack_bf16_hw(oc[dj][0], oc[dj][1]), pack_bf16_hw(oc[dj][2], oc[dj][3]));
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int i = lane; i < 16 * 16; i += 64) {
            const int q = 16 * qi + (i >> 4), c = i & 15;
            if (q < S) *(uint4*)(obase + (int64_t)q * HQ * HD + 8 * c) = *(const uint4*)(sO + (i >> 4) * OP + 8 * c);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = qn[ks];
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// fp32 attention (the encoder at the reference's fp32 storage: SentenceTransformer(name) without a dtype, streamlit_app.py:55,173)
// for sequences of at most 512 (head 64) / 256 (head 128) / 128 (head 256) tokens - the corpus texts of app_create_embeddings.py:48-70
// are global context + statement, up to BERT's 512 positions - any of the three families: softmax(Q K^T * scale + key mask [+ causal]) V with every
// product on v_mfma_f32_16x16x4_f32 - exact fp32 multiplies, fp32 accumulation - straight from the stacked projection's output
// [B][S][(HQ + 2 HKV) HD] into the context layout [B][S][HQ HD] (and, for the fp32-class GEMM behind it, its bf16 pieces).
// torch's attention on this layout first copies q, k, v and the context (four launches of 34 us each per BERT layer at 256 x 128
// tokens, r05_c5 kernel stats) around a 317 us flash launch; here a workgroup per (sequence, query head) holds V^T in LDS and
// each wave walks query tiles of 16:
//   * S^T = K Q^T per key tile: A = K rows (lane (key r16, g): K[key][16 s + 4 g .. + 4), 16 bytes from global / L2), B = Q rows of
//     the tile (same chunks, loaded once per tile); MFMA i of a chunk multiplies float i of both: k = 16 s + 4 g + i - a fixed
//     permutation of the head dimension in the fp32 sum;  D[key 4 g + r][query r16];
//   * softmax over the keys of a query = over r, the key tiles (registers) and the four lane groups (xor 16, 32), in fp32;
//   * O^T = V^T P^T: B = P^T is the score registers as they lie (MFMA r of key tile kj takes keys 16 kj + 4 g + r), A = V^T
//     [d r16][those four keys] = ONE ds_read_b128 of the transposed image (pitch SP + 4 floats: conflict-free);
//     D[d 4 g + r][query r16]: four consecutive d per lane, stored as 16 bytes.
// HD = head size (64 BERT, 128 Qwen3, 256 Gemma3); CAUSAL skips key tiles past the query tile; grouped-query: KV head = h / (HQ / HKV).
constexpr int kAttnF32MaxSeq = 128;                                  // every head size; smaller heads go further:
constexpr int attn_f32_max_seq(int HD) { return HD == 64 ? 512 : HD == 128 ? 256 : 128; }   // V^T must fit the CU's LDS (<= 133 KB)
constexpr int attn_f32_lds(int HD, int T) { return HD * (16 * T + 4) * 4; }

// The key tiles are walked as a RUNNING softmax (one tile's scores in registers at a time: running maximum m, running sum l,
// O rescaled by 2^(m - m') when the maximum moves), so the loop over key tiles is a plain run-time loop over fixed registers: the
// K fragments of tile kj + 1 are requested before the MFMAs of tile kj, the V^T fragments of a tile before its softmax arithmetic.
// (A first cut kept all T score tiles in registers behind `if (kj < kt_end)` branches: no load could move above a branch, every
// key tile paid an L2 round trip in front of its 16 MFMAs, and the launch took as long as torch's attention with its four copies.)
template <int HD, bool CAUSAL>
__global__ void __launch_bounds__(256) attention_f32_kernel(const float* __restrict__ qkv, const int64_t* __restrict__ mask, int B, int S,
                                                             int HQ, int HKV, float scale_log2e, float* __restrict__ out,
                                                             unsigned short* __restrict__ pieces, const float* __restrict__ bias) {
    // bias (may be NULL): the stacked projection's bias [(HQ + 2 HKV) HD], added to q, k and v as they are loaded - the GEMM in
    // front then runs without one (torch.addmm with an output type first copies the broadcast bias into the result: 35 us per
    // GEMM at 256 x 128 tokens, r05 kernel stats)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_attn[];
    float* sVT = (float*)smem_attn;
    const int T = (S + 15) / 16, SP = 16 * T, PP = SP + 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const int b = blockIdx.x / HQ, h = blockIdx.x - b * HQ;
    const int hk = h / (HQ / HKV);
    const int r16 = lane & 15, g = lane >> 4;
    const int64_t tok = (int64_t)(HQ + 2 * HKV) * HD;                   // floats per token of the projection's output
    const float* base = qkv + (int64_t)b * S * tok;
    const float* qb = base + (int64_t)h * HD;
    const float* kb = base + (int64_t)(HQ + hk) * HD;
    const float* vb = base + (int64_t)(HQ + HKV + hk) * HD;
    // V^T [d][key] in LDS (keys past the sequence: zeros - 0 x P = 0): 16 bytes of a V row per thread, four 4-byte stores
    for (int i = threadIdx.x; i < SP * (HD / 4); i += blockDim.x) {
        const int key = i / (HD / 4), c = i - key * (HD / 4);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (key < S) {
            v = *(const float4*)(vb + (int64_t)key * tok + 4 * c);
            if (bias) {
                const float4 bv = *(const float4*)(bias + (int64_t)(HQ + HKV + hk) * HD + 4 * c);
                v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
            }
        }
        sVT[(4 * c + 0) * PP + key] = v.x;
        sVT[(4 * c + 1) * PP + key] = v.y;
        sVT[(4 * c + 2) * PP + key] = v.z;
        sVT[(4 * c + 3) * PP + key] = v.w;
    }
    __syncthreads();
    constexpr int KC = HD / 16;                                          // 16-float chunks of a row
    // query tiles of this wave: wave, wave + nwaves, ...  CAUSAL: tile qi walks qi + 1 key tiles, so every second round runs from
    // the far end (rounds of tiles {w, 2 nw - 1 - w}: with eight tiles every wave walks nine key tiles instead of six to twelve)
    for (int it = 0;; ++it) {
        const int lo_ = it * nwaves + wave, hi_ = (it + 1) * nwaves - 1 - wave;
        const int qi = (CAUSAL && (it & 1)) ? hi_ : lo_;
        if (it * nwaves >= T) break;
        if (qi >= T) continue;
        const int qrow = min(16 * qi + r16, S - 1);
        const int qpos = 16 * qi + r16;
        float4 qf[KC], kf[KC], kbias[KC];
#pragma unroll
        for (int s = 0; s < KC; ++s) {
            qf[s] = *(const float4*)(qb + (int64_t)qrow * tok + 16 * s + 4 * g);
            kbias[s] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (bias) {
                const float4 bq = *(const float4*)(bias + (int64_t)h * HD + 16 * s + 4 * g);
                qf[s].x += bq.x; qf[s].y += bq.y; qf[s].z += bq.z; qf[s].w += bq.w;
                kbias[s] = *(const float4*)(bias + (int64_t)(HQ + hk) * HD + 16 * s + 4 * g);
            }
        }
        const int kt_end = CAUSAL ? qi + 1 : T;                          // key tiles that hold an allowed key
        {
            const int krow = min(r16, S - 1);
#pragma unroll
            for (int s = 0; s < KC; ++s) kf[s] = *(const float4*)(kb + (int64_t)krow * tok + 16 * s + 4 * g);
        }
        f32x4 o[KC];
#pragma unroll
        for (int dj = 0; dj < KC; ++dj) o[dj] = f32x4{0.f, 0.f, 0.f, 0.f};
        float m = -INFINITY, l = 0.0f;                                   // running maximum / sum of query r16 (the same in its four lane groups)
        for (int kj = 0; kj < kt_end; ++kj) {
            // scores of this tile: two accumulator chains (a dependent 16x16x4 MFMA waits 40 cycles, an independent one 32)
            f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
            if (bias) {
#pragma unroll
                for (int s = 0; s < KC; ++s) { kf[s].x += kbias[s].x; kf[s].y += kbias[s].y; kf[s].z += kbias[s].z; kf[s].w += kbias[s].w; }
            }
#pragma unroll
            for (int s = 0; s < KC; s += 2) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[s].x, qf[s].x, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[s + 1].x, qf[s + 1].x, a1, 0, 0, 0);
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[s].y, qf[s].y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[s + 1].y, qf[s + 1].y, a1, 0, 0, 0);
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[s].z, qf[s].z, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[s + 1].z, qf[s + 1].z, a1, 0, 0, 0);
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[s].w, qf[s].w, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[s + 1].w, qf[s + 1].w, a1, 0, 0, 0);
            }
            // the next tile's K fragments and this tile's V^T fragments are on their way while the softmax arithmetic runs
            if (kj + 1 < kt_end) {
                const int krow = min(16 * (kj + 1) + r16, S - 1);
#pragma unroll
                for (int s = 0; s < KC; ++s) kf[s] = *(const float4*)(kb + (int64_t)krow * tok + 16 * s + 4 * g);
            }
            float4 vf[KC];
#pragma unroll
            for (int dj = 0; dj < KC; ++dj) vf[dj] = *(const float4*)(sVT + (16 * dj + r16) * PP + 16 * kj + 4 * g);
            float sc[4], tmax = -INFINITY;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = 16 * kj + 4 * g + r;
                const bool ok = key < S && (!mask || mask[(int64_t)b * S + key] != 0) && (!CAUSAL || key <= qpos);
                sc[r] = ok ? a0[r] + a1[r] : -INFINITY;
                tmax = fmaxf(tmax, sc[r]);
            }
            tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
            const float m_new = fmaxf(m, tmax);
            // a tile without a single allowed key so far leaves m = -inf: nothing to rescale, nothing to add
            const float alpha = m_new > -INFINITY ? __builtin_amdgcn_exp2f((m - m_new) * scale_log2e) : 1.0f;
            float psum = 0.0f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                sc[r] = sc[r] > -INFINITY ? __builtin_amdgcn_exp2f((sc[r] - m_new) * scale_log2e) : 0.0f;
                psum += sc[r];
            }
            psum += __shfl_xor(psum, 16, 64);
            psum += __shfl_xor(psum, 32, 64);
            l = l * alpha + psum;
            m = m_new;
#pragma unroll
            for (int dj = 0; dj < KC; ++dj) {
                o[dj][0] *= alpha; o[dj][1] *= alpha; o[dj][2] *= alpha; o[dj][3] *= alpha;
            }
            // O^T += V^T P^T: KC independent accumulators, MFMA r of the tile takes keys 16 kj + 4 g + r
#pragma unroll
            for (int dj = 0; dj < KC; ++dj) o[dj] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[dj].x, sc[0], o[dj], 0, 0, 0);
#pragma unroll
            for (int dj = 0; dj < KC; ++dj) o[dj] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[dj].y, sc[1], o[dj], 0, 0, 0);
#pragma unroll
            for (int dj = 0; dj < KC; ++dj) o[dj] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[dj].z, sc[2], o[dj], 0, 0, 0);
#pragma unroll
            for (int dj = 0; dj < KC; ++dj) o[dj] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[dj].w, sc[3], o[dj], 0, 0, 0);
        }
        const float inv = l > 0.0f ? 1.0f / l : 0.0f;                    // a query without a single allowed key: zeros
        if (qpos < S) {
            float* orow = out ? out + ((int64_t)b * S + qpos) * HQ * HD + (int64_t)h * HD : nullptr;
#pragma unroll
            for (int dj = 0; dj < KC; ++dj) {
                const float y[4] = {o[dj][0] * inv, o[dj][1] * inv, o[dj][2] * inv, o[dj][3] * inv};
                if (orow) *(float4*)(orow + 16 * dj + 4 * g) = make_float4(y[0], y[1], y[2], y[3]);   // (not needed when only the pieces are read)
                if (pieces) {
                    const int dtot = HQ * HD;
                    store_pieces4(pieces + ((int64_t)b * S + qpos) * 3 * dtot, dtot, (h * HD + 16 * dj + 4 * g) / 4, y);
                }
            }
        }
    }
}

}  // namespace ts
