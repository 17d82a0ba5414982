// Row preparation: dtype conversion, one-time L2 normalisation, zero padding to the row stride.
//
// Replaces the per-call re-normalisation inside sentence_transformers.util.cos_sim (reference
// call sites compare_embeddings.py:24,61, app_showcase_model.py:93): rows are normalised ONCE
// when they enter the index, queries once per search.  Rule (same as oracle.l2_normalize):
//   norm  = (float) sqrt( sum_i (double)x_i^2 )       - fp64 accumulation, one rounding
//   x_n   = x / max(norm, 1e-12f)                     - correctly rounded fp32 division
// then, for a bf16 destination, round-to-nearest-even.
#pragma once
#include "common.h"

namespace ts {

// One wave per row.  SRC/DST: 0 = f32, 1 = bf16 bits.  Writes dst[row][0..ld) (zeros past d).
// dst_f32_copy (optional): the same prepared values widened to fp32, stride ld (query buffers).
template <int SRC, int DST, bool NORMALIZE>
__global__ void __launch_bounds__(256) prep_rows_kernel(const void* __restrict__ src, int64_t src_ld, void* __restrict__ dst,
                                                         float* __restrict__ dst_f32_copy, int64_t ld, int d, int64_t nrows,
                                                         int64_t rows_total /* rows to write incl. zero rows past nrows */) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    for (int64_t row = wave; row < rows_total; row += nwaves) {
        const bool real = row < nrows;
        float denom = 1.0f;
        if (NORMALIZE && real) {
            double ss = 0.0;
            for (int c = lane; c < d; c += 64) {
                float x = (SRC == 0) ? ((const float*)src)[row * src_ld + c]
                                     : bf16_to_f32(((const unsigned short*)src)[row * src_ld + c]);
                ss += (double)x * (double)x;
            }
            ss = wave_sum_f64(ss);
            const float norm = (float)sqrt(ss);
            denom = fmaxf(norm, 1e-12f);
        }
        for (int c = lane; c < ld; c += 64) {
            float x = 0.0f;
            if (real && c < d) {
                x = (SRC == 0) ? ((const float*)src)[row * src_ld + c]
                               : bf16_to_f32(((const unsigned short*)src)[row * src_ld + c]);
                if (NORMALIZE) x = (float)((double)x / (double)denom);
            }
            if (DST == 0) {
                ((float*)dst)[row * ld + c] = x;
                if (dst_f32_copy) dst_f32_copy[row * ld + c] = x;
            } else {
                const unsigned short b = f32_to_bf16(x);
                ((unsigned short*)dst)[row * ld + c] = b;
                if (dst_f32_copy) dst_f32_copy[row * ld + c] = bf16_to_f32(b);
            }
        }
    }
}

// Dense device rows -> dense host-layout rows of the storage dtype (for ts_index_download).
template <int DT>
__global__ void __launch_bounds__(256) unpad_rows_kernel(const void* __restrict__ src, int64_t ld, void* __restrict__ dst, int d,
                                                          int64_t nrows) {
    const int64_t total = nrows * d;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / d;
        const int c = (int)(i - r * d);
        if (DT == 0)
            ((float*)dst)[i] = ((const float*)src)[r * ld + c];
        else
            ((unsigned short*)dst)[i] = ((const unsigned short*)src)[r * ld + c];
    }
}

// Subset index: copy the stored (already prepared) rows src[ids[i] - id_base] -> dst[i], one wave per row,
// 16 bytes per lane.  row_bytes is a multiple of 128 (ld is a multiple of 64 elements).
__global__ void __launch_bounds__(256) gather_rows_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst,
                                                           const int64_t* __restrict__ ids, int64_t id_base, int64_t nrows,
                                                           int64_t row_bytes) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    for (int64_t i = wave; i < nrows; i += nwaves) {
        const uint4* s = (const uint4*)(src + (ids[i] - id_base) * row_bytes);
        uint4* d = (uint4*)(dst + i * row_bytes);
        for (int64_t c = lane; c < row_bytes / 16; c += 64) d[c] = s[c];
    }
}

}  // namespace ts
