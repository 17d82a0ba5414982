// Shared device helpers for libtsearch (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ts {

typedef unsigned long long u64;
typedef unsigned int u32;

constexpr int kWave = 64;
constexpr int kTileRows = 32;      // corpus rows per MFMA tile; index allocations are padded to kRowPad rows
constexpr int kRowPad = 256;
constexpr int kLdPad = 64;         // row stride is a multiple of 64 elements, zero padded

// ---- ordered keys -------------------------------------------------------------------------
// A candidate (score, row) is one 64-bit key whose unsigned order is the result order:
// larger key = better = (higher score, then lower row).  Key 0 is "empty" (below every real key).
// NaN scores never become keys (callers test s == s first); -0.0 is folded into +0.0.
__device__ __forceinline__ u32 ord_f32(float s) {
    s = s + 0.0f;
    u32 u = __float_as_uint(s);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unord_f32(u32 o) {
    u32 u = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
    return __uint_as_float(u);
}
__device__ __forceinline__ u64 make_key(float s, u32 row) {
    return ((u64)ord_f32(s) << 32) | (u64)(0xFFFFFFFFu - row);
}
__device__ __forceinline__ float key_score(u64 k) { return unord_f32((u32)(k >> 32)); }
__device__ __forceinline__ u32 key_row(u64 k) { return 0xFFFFFFFFu - (u32)k; }

// ---- bf16 ------------------------------------------------------------------------------------
__device__ __forceinline__ float bf16_lo(u32 packed) { return __uint_as_float(packed << 16); }
__device__ __forceinline__ float bf16_hi(u32 packed) { return __uint_as_float(packed & 0xFFFF0000u); }
__device__ __forceinline__ float bf16_to_f32(unsigned short b) { return __uint_as_float(((u32)b) << 16); }
// round-to-nearest-even; NaN stays NaN (quiet bit forced) - same rule as oracle.f32_to_bf16_bits
__device__ __forceinline__ unsigned short f32_to_bf16(float f) {
    u32 u = __float_as_uint(f);
    if (f != f) return (unsigned short)((u >> 16) | 0x0040u);
    return (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}

// The same rounding by the hardware (gfx950: v_cvt_pk_bf16_f32, two values per instruction; RNE, NaN stays NaN - the payload
// may differ from the software form's).  For the kernels around the encoder's GEMMs, whose roundings are emulated many
// times per element; the index's own storage (prep_rows_kernel) keeps the software form the oracle restates bit for bit.
typedef __bf16 ts_bf16x2 __attribute__((ext_vector_type(2)));
typedef float ts_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u32 pack_bf16_hw(float lo, float hi) {
    const ts_f32x2 v = {lo, hi};
    const ts_bf16x2 r = __builtin_convertvector(v, ts_bf16x2);
    return *reinterpret_cast<const u32*>(&r);
}
__device__ __forceinline__ void round2_bf16_hw(float& a, float& b) {
    const u32 p = pack_bf16_hw(a, b);
    a = bf16_lo(p);
    b = bf16_hi(p);
}

// Four consecutive fp32 values of a row as bf16 pieces for the fp32-class GEMM that reads them (split3_kernel below; activation
// pattern [hi | lo | hi]): `prow` = the row's 3 d bf16 elements, `c` = the values' 4-element chunk.  fp32 producers write them
// next to (or instead of) their fp32 output, so the GEMM's operand never makes a round trip through ts_split_pieces.
__device__ __forceinline__ void store_pieces4(unsigned short* prow, int d, int c, const float* y) {
    float hi[4], lo[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        hi[e] = bf16_to_f32(f32_to_bf16(y[e]));
        lo[e] = y[e] - hi[e];
    }
    const uint2 ph = make_uint2(pack_bf16_hw(hi[0], hi[1]), pack_bf16_hw(hi[2], hi[3]));
    const uint2 pl = make_uint2(pack_bf16_hw(lo[0], lo[1]), pack_bf16_hw(lo[2], lo[3]));
    uint2* o = (uint2*)prow + c;
    o[0] = ph;
    o[d / 4] = pl;
    o[2 * (d / 4)] = ph;
}

// ---- wave helpers ---------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__device__ __forceinline__ u64 shfl_u64(u64 v, int src) {
    u32 lo = __shfl((int)(u32)v, src, 64);
    u32 hi = __shfl((int)(u32)(v >> 32), src, 64);
    return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u64 shfl_up_u64(u64 v, int delta) {
    u32 lo = __shfl_up((int)(u32)v, delta, 64);
    u32 hi = __shfl_up((int)(u32)(v >> 32), delta, 64);
    return ((u64)hi << 32) | lo;
}

// Per-wave running top-k: slot s = r * 64 + lane holds the s-th best key (descending).
template <int KR>
struct WaveTopK {
    u64 key[KR];
    u64 thr;  // wave-uniform: key of slot k-1 (0 until k entries are present)

    __device__ __forceinline__ void init() {
#pragma unroll
        for (int r = 0; r < KR; ++r) key[r] = 0;
        thr = 0;
    }
    // K must be wave-uniform and non-zero.
    __device__ __forceinline__ void insert(u64 K, int k, int lane) {
        int pos = 0;
#pragma unroll
        for (int r = 0; r < KR; ++r) pos += __popcll(__ballot(key[r] > K));
#pragma unroll
        for (int r = KR - 1; r >= 0; --r) {
            u64 up = shfl_up_u64(key[r], 1);
            if (r > 0) {
                u64 carry = shfl_u64(key[r - 1], 63);
                if (lane == 0) up = carry;
            }
            const int s = r * 64 + lane;
            key[r] = (s < pos) ? key[r] : (s == pos ? K : up);
        }
        const int last = k - 1;
        u64 t = 0;
#pragma unroll
        for (int r = 0; r < KR; ++r)
            if ((last >> 6) == r) t = shfl_u64(key[r], last & 63);
        thr = t;
    }
};

// Descending bitonic sort of n (power of two) keys in LDS by `nthreads` threads of one workgroup.
__device__ __forceinline__ void bitonic_sort_desc(u64* keys, int n, int tid, int nthreads) {
    for (int size = 2; size <= n; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int i = tid; i < (n >> 1); i += nthreads) {
                const int lo = 2 * i - (i & (stride - 1));
                const int hi = lo + stride;
                const bool desc = ((lo & size) == 0);
                const u64 a = keys[lo], b = keys[hi];
                if ((a < b) == desc) {
                    keys[lo] = b;
                    keys[hi] = a;
                }
            }
        }
    }
    __syncthreads();
}

// One workgroup, after a full pass (block 0 of the launch that follows it - the exact re-run's, search.hip): the workgroups of the pass are dispatched round-robin over the 8 XCDs, and the XCDs
// of one device run it at rates a few per cent apart (per-XCD clock under the shared power budget), so with equal shares
// the launch waits for the slowest XCD.  Shares move towards equal finishing times: share_w *= 1 + 0.7 (T / t_x(w) - 1),
// t_x = mean time of the workgroups with w % 8 = x, T = mean of the t_x; boundaries are rounded, monotone, and keep a
// minimum share.  The answers do not depend on the partition.  `tmp`: 2 part_g + 16 doubles of LDS; part_g <= blockDim.x.
__device__ __forceinline__ void rebalance_tiles(int64_t* part, const unsigned* wg_ticks, int G, float gain, double* tmp) {
    // G <= blockDim.x: one thread per workgroup of the pass
    double* sh[2] = {tmp, tmp + G};               // shares, then their running sum (two buffers for the scan)
    double* tx = tmp + 2 * G;                     // [8] summed time per XCD, [8] counts
    const int w = threadIdx.x;
    if (w < 16) tx[w] = 0.0;
    __syncthreads();
    const int64_t ntiles = part[G];
    bool ok = ntiles > 0;
    double share = 0.0;
    if (w < G) {
        const unsigned t = wg_ticks[w];
        share = (double)(part[w + 1] - part[w]);
        if (t == 0 || share <= 0.0) ok = false;
        atomicAdd(&tx[w & 7], (double)t);
        atomicAdd(&tx[8 + (w & 7)], 1.0);
    }
    if (!__syncthreads_and(ok ? 1 : 0)) return;   // a workgroup without work or time: leave the table alone
    double T = 0.0;
    int nx = 0;
    for (int x = 0; x < 8; ++x)
        if (tx[8 + x] > 0.0) { T += tx[x] / tx[8 + x]; ++nx; }
    T /= (double)(nx > 0 ? nx : 1);
    if (w < G) {
        double f = 1.0 + (double)gain * (T / (tx[w & 7] / tx[8 + (w & 7)]) - 1.0);
        f = f < 0.9 ? 0.9 : (f > 1.1 ? 1.1 : f);
        const double floor_share = (double)ntiles / (4.0 * G);
        share *= f;
        sh[0][w] = share > floor_share ? share : floor_share;
    }
    __syncthreads();
    int cur = 0;                                  // inclusive scan of the shares (Hillis-Steele)
    for (int d = 1; d < G; d <<= 1) {
        if (w < G) sh[cur ^ 1][w] = sh[cur][w] + (w >= d ? sh[cur][w - d] : 0.0);
        __syncthreads();
        cur ^= 1;
    }
    if (w < G) {
        const double total = sh[cur][G - 1];
        const int64_t b = (w + 1 == G) ? ntiles : (int64_t)(sh[cur][w] * ((double)ntiles / total) + 0.5);
        part[w + 1] = b;                        // shares >= ntiles / (4.2 G) >= 7 tiles: boundaries stay strictly increasing
    }
}


}  // namespace ts
