// Encoder-side kernels: what a sentence-transformer does around its GEMMs, each as one launch -
// pooling + L2 normalisation + cast of the last hidden state (the query buffer the search reads in place), residual add +
// LayerNorm / RMSNorm, the input layer (embedding gathers + LayerNorm).  Reference call sites: SentenceTransformer.encode at
// parsed_papers_to_vector_rds/embeddings.py:31-37, ec2/generate_embeddings/embeddings.py:24-30, streamlit_app.py:173.
#pragma once
#include "common.h"

namespace ts {

// Encoder epilogue: pooling over the sequence + L2 normalisation + cast, one workgroup per sequence.
// Columns are spread over the threads (coalesced reads of every token row); the squared norm is reduced
// in fp64 through LDS like the index rows.
template <int HDT, int ODT>
__global__ void __launch_bounds__(256) pool_normalize_kernel(const void* __restrict__ hidden, const int64_t* __restrict__ mask,
                                                              int seq, int d, int pooling, int normalize, void* __restrict__ out,
                                                              int64_t out_ld) {
    __shared__ double red[4];
    __shared__ int s_count, s_last;
    const int64_t row = blockIdx.x;
    const int64_t* m = mask + row * seq;
    if (threadIdx.x == 0) {
        int cnt = 0, last = 0;
        for (int t = 0; t < seq; ++t)
            if (m[t] != 0) {
                ++cnt;
                last = t;
            }
        s_count = cnt;
        s_last = last;
    }
    __syncthreads();
    const int count = s_count, last = s_last;
    constexpr int kMaxPerThread = 16;  // d <= 4096
    float val[kMaxPerThread];
    double ss = 0.0;
#pragma unroll
    for (int j = 0; j < kMaxPerThread; ++j) {
        const int c = threadIdx.x + j * 256;
        float v = 0.0f;
        if (c < d) {
            auto load = [&](int t) -> float {
                const int64_t off = (row * seq + t) * (int64_t)d + c;
                return (HDT == 0) ? ((const float*)hidden)[off] : bf16_to_f32(((const unsigned short*)hidden)[off]);
            };
            if (pooling == 0) {
                float acc = 0.0f;
                for (int t = 0; t < seq; ++t)
                    if (m[t] != 0) acc += load(t);
                v = acc / fmaxf((float)count, 1e-9f);
            } else {
                v = load(pooling == 1 ? last : 0);
            }
            ss += (double)v * (double)v;
        }
        val[j] = v;
    }
    float denom = 1.0f;
    if (normalize) {
        ss = wave_sum_f64(ss);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
        __syncthreads();
        const double total = red[0] + red[1] + red[2] + red[3];
        denom = fmaxf((float)sqrt(total), 1e-12f);
    }
#pragma unroll
    for (int j = 0; j < kMaxPerThread; ++j) {
        const int c = threadIdx.x + j * 256;
        if (c < d) {
            const float v = normalize ? (float)((double)val[j] / (double)denom) : val[j];
            if (ODT == 0) ((float*)out)[row * out_ld + c] = v;
            else ((unsigned short*)out)[row * out_ld + c] = f32_to_bf16(v);
        }
    }
}

// The same epilogue for the shapes the encoders have (d a multiple of the 16-byte vector, at most 256 vectors per row; seq <=
// kPoolVecSeq): 16-byte loads, the tokens dealt over the G = 256 / (vectors per row) thread groups so that every thread has
// seq / G independent loads in flight, the mask counted by the whole workgroup.  The kernel above walks the tokens with one
// 2-byte load per thread and token behind a serial scan of the mask by thread 0: 32 us for 256 x 32 x 768 bf16 (12.6 MB),
// this one is bound by the read.  Partial sums are combined in group order (deterministic).
constexpr int kPoolVecSeq = 1024;
template <int HDT, int ODT>
__global__ void __launch_bounds__(256) pool_normalize_vec_kernel(const void* __restrict__ hidden, const int64_t* __restrict__ mask,
                                                                  int seq, int d, int pooling, int normalize, void* __restrict__ out,
                                                                  int64_t out_ld) {
    constexpr int VEC = HDT == 0 ? 4 : 8;
    __shared__ float part[256 * VEC];
    __shared__ unsigned char s_mask[kPoolVecSeq];
    __shared__ double red[4];
    __shared__ int s_count, s_last;
    const int64_t row = blockIdx.x;
    const int64_t* m = mask + row * seq;
    if (threadIdx.x == 0) {
        s_count = 0;
        s_last = 0;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < seq; t += 256) {
        const bool on = m[t] != 0;
        s_mask[t] = on ? 1 : 0;
        if (on) {
            atomicAdd(&s_count, 1);
            atomicMax(&s_last, t);
        }
    }
    __syncthreads();
    const int count = s_count, last = s_last;
    const int cw = d / VEC;                       // vectors per row
    const int G = 256 / cw;                       // token groups
    const int g = threadIdx.x / cw, c = threadIdx.x - g * cw;
    float acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] = 0.0f;
    auto add = [&](int t) {
        const uint4 v = *((const uint4*)((const unsigned char*)hidden + ((row * seq + t) * (int64_t)d) * (HDT == 0 ? 4 : 2)) + c);
        if (HDT == 0) {
            acc[0] += __uint_as_float(v.x); acc[1] += __uint_as_float(v.y); acc[2] += __uint_as_float(v.z); acc[3] += __uint_as_float(v.w);
        } else {
            acc[0] += bf16_lo(v.x); acc[1] += bf16_hi(v.x); acc[2] += bf16_lo(v.y); acc[3] += bf16_hi(v.y);
            acc[4 % VEC] += bf16_lo(v.z); acc[5 % VEC] += bf16_hi(v.z); acc[6 % VEC] += bf16_lo(v.w); acc[7 % VEC] += bf16_hi(v.w);
        }
    };
    if (g < G) {
        if (pooling == 0) {
            for (int t = g; t < seq; t += G)
                if (s_mask[t]) add(t);
        } else if (g == 0) {
            add(pooling == 1 ? last : 0);
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) part[(g * cw + c) * VEC + e] = acc[e];
    }
    __syncthreads();
    float val[VEC];
    double ss = 0.0;
    const bool owner = g == 0;                    // threads 0 .. cw - 1 own one vector of the result each
    if (owner) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            float v = part[c * VEC + e];
            if (pooling == 0) {
                for (int gg = 1; gg < G; ++gg) v += part[(gg * cw + c) * VEC + e];
                v = v / fmaxf((float)count, 1e-9f);
            }
            val[e] = v;
            ss += (double)v * (double)v;
        }
    }
    float denom = 1.0f;
    if (normalize) {
        ss = wave_sum_f64(ss);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
        __syncthreads();
        const double total = red[0] + red[1] + red[2] + red[3];
        denom = fmaxf((float)sqrt(total), 1e-12f);
    }
    if (owner) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const float v = normalize ? (float)((double)val[e] / (double)denom) : val[e];
            if (ODT == 0) ((float*)out)[row * out_ld + c * VEC + e] = v;
            else ((unsigned short*)out)[row * out_ld + c * VEC + e] = f32_to_bf16(v);
        }
    }
}

// Residual add + LayerNorm of the encoder (BertSelfOutput / BertOutput: LayerNorm(dense_out + input)), one kernel
// instead of torch's add and layer_norm launches (25 of each per BERT-base forward): out = (x - mean) * rstd * gamma + beta
// with x = a + b taken in fp32 (torch rounds the sum to the storage type first; this keeps it in fp32), mean and variance
// over the row in fp32 (two passes over registers: mean first, then the centred squares).  One wave per row, 8 elements
// (bf16) or 4 (fp32) per 16-byte access; d a multiple of that, at most 64 * kLnMax accesses per row.
// LN = 16-byte accesses per lane (1, 2 or 4, the smallest that covers the row): the BERT-base width (96 accesses per row)
// takes 2 and a third of the registers of the 4-access form, so every row of a launch is resident at once.
constexpr int kLnMax = 4;
template <int DT, int LN = kLnMax>
__global__ void __launch_bounds__(256) add_layernorm_kernel(const void* a, const void* b,      // `out` may alias a or b: no restrict
                                                             const void* __restrict__ gamma, const void* __restrict__ beta, float eps,
                                                             int64_t rows, int d, void* out, unsigned short* pieces = nullptr,
                                                             const float* __restrict__ a_bias = nullptr) {   // fp32 only: a + a_bias[col]
    constexpr int VEC = DT == 0 ? 4 : 8;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nchunk = d / VEC;
    float x[LN][VEC];
    auto unpack = [](const uint4& v, float* f) {
        if (DT == 0) {
            f[0] = __uint_as_float(v.x); f[1] = __uint_as_float(v.y); f[2] = __uint_as_float(v.z); f[3] = __uint_as_float(v.w);
        } else {
            f[0] = bf16_lo(v.x); f[1] = bf16_hi(v.x); f[2] = bf16_lo(v.y); f[3] = bf16_hi(v.y);
            f[4] = bf16_lo(v.z); f[5] = bf16_hi(v.z); f[6] = bf16_lo(v.w); f[7] = bf16_hi(v.w);
        }
    };
    const uint4* pa = (const uint4*)a + row * nchunk;
    const uint4* pb = (const uint4*)b + row * nchunk;
    // every load of the row - the two operands, gamma and beta - is requested before the first use: one memory round trip
    uint4 ra[LN], rb[LN], rg[LN], re[LN];
#pragma unroll
    for (int j = 0; j < LN; ++j) {
        const int c = lane + 64 * j;
        if (c < nchunk) {
            ra[j] = pa[c];
            rb[j] = pb[c];
            rg[j] = ((const uint4*)gamma)[c];
            re[j] = ((const uint4*)beta)[c];
        }
    }
    float sum = 0.0f;
#pragma unroll
    for (int j = 0; j < LN; ++j) {
        const int c = lane + 64 * j;
        if (c < nchunk) {
            float fa[VEC], fb[VEC];
            unpack(ra[j], fa);
            unpack(rb[j], fb);
            if (DT == 0 && a_bias) {                 // the bias of the GEMM that produced `a` (it ran without one)
                const float4 bb = ((const float4*)a_bias)[c];
                fa[0] += bb.x; fa[1] += bb.y; fa[2] += bb.z; fa[3] += bb.w;
            }
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                x[j][e] = fa[e] + fb[e];
                sum += x[j][e];
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
    const float mean = sum / (float)d;
    float sq = 0.0f;
#pragma unroll
    for (int j = 0; j < LN; ++j) {
        const int c = lane + 64 * j;
        if (c < nchunk) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const float t = x[j][e] - mean;
                sq += t * t;
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sq += __shfl_xor(sq, off, 64);
    const float rstd = rsqrtf(sq / (float)d + eps);
    uint4* po = (uint4*)out + row * nchunk;
#pragma unroll
    for (int j = 0; j < LN; ++j) {
        const int c = lane + 64 * j;
        if (c < nchunk) {
            float g[VEC], be[VEC], y[VEC];
            unpack(rg[j], g);
            unpack(re[j], be);
#pragma unroll
            for (int e = 0; e < VEC; ++e) y[e] = (x[j][e] - mean) * rstd * g[e] + be[e];
            uint4 o;
            if (DT == 0) {
                o = make_uint4(__float_as_uint(y[0]), __float_as_uint(y[1]), __float_as_uint(y[2]), __float_as_uint(y[3]));
            } else {
                o.x = (u32)f32_to_bf16(y[0]) | ((u32)f32_to_bf16(y[1]) << 16);
                o.y = (u32)f32_to_bf16(y[2]) | ((u32)f32_to_bf16(y[3]) << 16);
                o.z = (u32)f32_to_bf16(y[4]) | ((u32)f32_to_bf16(y[5]) << 16);
                o.w = (u32)f32_to_bf16(y[6]) | ((u32)f32_to_bf16(y[7]) << 16);
            }
            po[c] = o;
            if (DT == 0 && pieces) store_pieces4(pieces + row * 3 * (int64_t)d, d, c, y);
        }
    }
}

// The encoder's input layer (BertEmbeddings: word + token-type + position embedding, LayerNorm) as one kernel instead of
// three gathers, two adds and a layer_norm launch (66 us per forward of 8,192 tokens in PyTorch): one wave per token, the
// three table rows, gamma and beta requested before the first use, the sum, the mean and the variance in fp32 (torch rounds
// each partial sum to the storage type; this keeps them in fp32).  Token i has position i % seq.  Ids are clamped to the
// tables (torch raises on an id outside its table; a kernel cannot).
template <int DT, int LN>
__global__ void __launch_bounds__(256) embed_layernorm_kernel(const int64_t* __restrict__ ids, const int64_t* __restrict__ type_ids,
                                                               const void* __restrict__ word, const void* __restrict__ pos,
                                                               const void* __restrict__ type, int64_t n_word, int64_t n_pos,
                                                               int64_t n_type, const void* __restrict__ gamma,
                                                               const void* __restrict__ beta, float eps, int64_t tokens, int seq, int d,
                                                               void* __restrict__ out) {
    constexpr int VEC = DT == 0 ? 4 : 8;
    const int lane = threadIdx.x & 63;
    const int64_t tok = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (tok >= tokens) return;
    const int nchunk = d / VEC;
    auto unpack = [](const uint4& v, float* f) {
        if (DT == 0) {
            f[0] = __uint_as_float(v.x); f[1] = __uint_as_float(v.y); f[2] = __uint_as_float(v.z); f[3] = __uint_as_float(v.w);
        } else {
            f[0] = bf16_lo(v.x); f[1] = bf16_hi(v.x); f[2] = bf16_lo(v.y); f[3] = bf16_hi(v.y);
            f[4 % VEC] = bf16_lo(v.z); f[5 % VEC] = bf16_hi(v.z); f[6 % VEC] = bf16_lo(v.w); f[7 % VEC] = bf16_hi(v.w);
        }
    };
    const int64_t wid = min(max(ids[tok], (int64_t)0), n_word - 1);
    const int64_t tid = type_ids ? min(max(type_ids[tok], (int64_t)0), n_type - 1) : 0;
    const int64_t pid = min(tok % seq, n_pos - 1);
    const uint4* pw = (const uint4*)word + wid * nchunk;
    const uint4* pt = (const uint4*)type + tid * nchunk;
    const uint4* pp = (const uint4*)pos + pid * nchunk;
    uint4 rw[LN], rt[LN], rp[LN], rg[LN], re[LN];
#pragma unroll
    for (int j = 0; j < LN; ++j) {
        const int c = lane + 64 * j;
        if (c < nchunk) {
            rw[j] = pw[c];
            rt[j] = pt[c];
            rp[j] = pp[c];
            rg[j] = ((const uint4*)gamma)[c];
            re[j] = ((const uint4*)beta)[c];
        }
    }
    float x[LN][VEC];
    float sum = 0.0f;
#pragma unroll
    for (int j = 0; j < LN; ++j) {
        const int c = lane + 64 * j;
        if (c < nchunk) {
            float fw[VEC], ft[VEC], fp[VEC];
            unpack(rw[j], fw);
            unpack(rt[j], ft);
            unpack(rp[j], fp);
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                x[j][e] = (fw[e] + ft[e]) + fp[e];          // BertEmbeddings' order: inputs + token type, then + position
                sum += x[j][e];
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
    const float mean = sum / (float)d;
    float sq = 0.0f;
#pragma unroll
    for (int j = 0; j < LN; ++j) {
        const int c = lane + 64 * j;
        if (c < nchunk) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const float t = x[j][e] - mean;
                sq += t * t;
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sq += __shfl_xor(sq, off, 64);
    const float rstd = rsqrtf(sq / (float)d + eps);
    uint4* po = (uint4*)out + tok * nchunk;
#pragma unroll
    for (int j = 0; j < LN; ++j) {
        const int c = lane + 64 * j;
        if (c < nchunk) {
            float g[VEC], be[VEC], y[VEC];
            unpack(rg[j], g);
            unpack(re[j], be);
#pragma unroll
            for (int e = 0; e < VEC; ++e) y[e] = (x[j][e] - mean) * rstd * g[e] + be[e];
            uint4 o;
            if (DT == 0) {
                o = make_uint4(__float_as_uint(y[0]), __float_as_uint(y[1]), __float_as_uint(y[2]), __float_as_uint(y[3]));
            } else {
                o.x = (u32)f32_to_bf16(y[0]) | ((u32)f32_to_bf16(y[1]) << 16);
                o.y = (u32)f32_to_bf16(y[2]) | ((u32)f32_to_bf16(y[3]) << 16);
                o.z = (u32)f32_to_bf16(y[4 % VEC]) | ((u32)f32_to_bf16(y[5 % VEC]) << 16);
                o.w = (u32)f32_to_bf16(y[6 % VEC]) | ((u32)f32_to_bf16(y[7 % VEC]) << 16);
            }
            po[c] = o;
        }
    }
}

// ---- the decoder-style encoder (Qwen3-Embedding: RMSNorm, rotary positions, grouped-query attention, gated MLP) ------------
// The production embedder of the reference is Qwen/Qwen3-Embedding-0.6B (streamlit_app.py:55, ec2/generate_embeddings/
// embedders.py:1-4: 28 layers, 1024 wide, 16 query / 8 key-value heads of 128, SwiGLU 3072).  Around its GEMMs PyTorch runs,
// per layer: two RMSNorms of six launches each (to fp32, pow, mean, add + rsqrt, mul, cast + mul), two residual adds, the
// per-head RMSNorm of queries and keys (twelve launches), the rotary embedding (ten) and SiLU x up (two).  Three kernels
// take their place; each follows the roundings of the modules it replaces (noted per kernel), so the hidden states agree
// with the model's own forward to the last bf16 bit wherever fp32 sums agree.

template <int DT>
__device__ __forceinline__ void enc_unpack(const uint4& v, float* f) {
    if (DT == 0) {
        f[0] = __uint_as_float(v.x); f[1] = __uint_as_float(v.y); f[2] = __uint_as_float(v.z); f[3] = __uint_as_float(v.w);
    } else {
        f[0] = bf16_lo(v.x); f[1] = bf16_hi(v.x); f[2] = bf16_lo(v.y); f[3] = bf16_hi(v.y);
        f[4] = bf16_lo(v.z); f[5] = bf16_hi(v.z); f[6] = bf16_lo(v.w); f[7] = bf16_hi(v.w);
    }
}
template <int DT>
__device__ __forceinline__ uint4 enc_pack(const float* y) {
    if (DT == 0) return make_uint4(__float_as_uint(y[0]), __float_as_uint(y[1]), __float_as_uint(y[2]), __float_as_uint(y[3]));
    // hardware conversion (common.h): two values per instruction
    return make_uint4(pack_bf16_hw(y[0], y[1]), pack_bf16_hw(y[2], y[3]), pack_bf16_hw(y[4 % (DT == 0 ? 4 : 8)], y[5 % (DT == 0 ? 4 : 8)]),
                      pack_bf16_hw(y[6 % (DT == 0 ? 4 : 8)], y[7 % (DT == 0 ? 4 : 8)]));
}
// round to the storage type and back: where the replaced module chain materialises a tensor of that type
template <int DT>
__device__ __forceinline__ float enc_round(float v) { return DT == 0 ? v : bf16_lo(pack_bf16_hw(v, 0.0f)); }

// Residual add + RMSNorm (Qwen3DecoderLayer: `hidden = residual + sublayer(hidden)` followed by the next RMSNorm):
//   s    = a + b          rounded to the storage type (torch materialises the sum)       -> out_sum (optional: the new residual)
//   y    = s * rsqrt(mean(s^2) + eps)   in fp32, rounded to the storage type (Qwen3RMSNorm: `hidden_states.to(input_dtype)`)
//   out  = gamma * y      rounded to the storage type                                      -> out_norm
// b may be NULL (a plain RMSNorm: the first norm of the first layer).  One wave per row, LN 16-byte accesses per lane.
template <int DT, int LN>
__global__ void __launch_bounds__(256) add_rmsnorm_kernel(const void* a, const void* b, const void* __restrict__ gamma, float eps,
                                                           int64_t rows, int d, void* out_sum, void* out_norm, unsigned short* pieces = nullptr) {
    constexpr int VEC = DT == 0 ? 4 : 8;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nchunk = d / VEC;
    const uint4* pa = (const uint4*)a + row * nchunk;
    const uint4* pb = b ? (const uint4*)b + row * nchunk : nullptr;
    uint4 ra[LN], rb[LN], rg[LN];
#pragma unroll
    for (int j = 0; j < LN; ++j) {
        const int c = lane + 64 * j;
        if (c < nchunk) {
            ra[j] = pa[c];
            if (pb) rb[j] = pb[c];
            rg[j] = ((const uint4*)gamma)[c];
        }
    }
    float x[LN][VEC];
    float sq = 0.0f;
#pragma unroll
    for (int j = 0; j < LN; ++j) {
        const int c = lane + 64 * j;
        if (c < nchunk) {
            float fa[VEC], fb[VEC];
            enc_unpack<DT>(ra[j], fa);
            if (pb) enc_unpack<DT>(rb[j], fb);
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                x[j][e] = pb ? enc_round<DT>(fa[e] + fb[e]) : fa[e];
                sq += x[j][e] * x[j][e];
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sq += __shfl_xor(sq, off, 64);
    const float rstd = rsqrtf(sq / (float)d + eps);
    uint4* ps = out_sum ? (uint4*)out_sum + row * nchunk : nullptr;
    uint4* po = (uint4*)out_norm + row * nchunk;
#pragma unroll
    for (int j = 0; j < LN; ++j) {
        const int c = lane + 64 * j;
        if (c < nchunk) {
            float g[VEC], y[VEC];
            enc_unpack<DT>(rg[j], g);
#pragma unroll
            for (int e = 0; e < VEC; ++e) y[e] = g[e] * enc_round<DT>(x[j][e] * rstd);
            if (ps) ps[c] = enc_pack<DT>(x[j]);
            po[c] = enc_pack<DT>(y);
            if (DT == 0 && pieces) store_pieces4(pieces + row * 3 * (int64_t)d, d, c, y);
        }
    }
}

// Per-head RMSNorm of queries and keys + rotary position embedding, in place on the fused projection's output
// (Qwen3Attention.forward: q_norm(q_proj(x).view(.., heads, 128)), k_norm(..), apply_rotary_pos_emb):
//   qkv   [tokens][(hq + 2 hkv) * 128]: query heads, key heads, value heads (values untouched)
//   y     = w * round(x * rsqrt(mean_128(x^2) + eps))                      (Qwen3RMSNorm over the head, fp32 inside)
//   out_i = round(round(y_i * cos_i) + round(rot_i * sin_i)),  rot = (-y[64..127], y[0..63])     (rotate_half; torch rounds
//           each product and the sum to the storage type)
// cos / sin: [seq][128] of the storage type (Qwen3RotaryEmbedding: fp32 angles, cast to x.dtype), token t has position
// t % seq.  Head size 128 only.  A group of 128 / VEC lanes per (token, head), 16 bytes per lane (VEC = 8 bf16 / 4 fp32
// elements): four (bf16) or two (fp32) heads per wave.  Element i rotates with element i +- 64 = the same slot of the lane half
// a group away: one exchange of the lane's packed, already rounded y.  (The first cut gave a whole wave to a head, two bytes
// per lane: 68 us for 8,192 tokens x 24 heads in the Qwen3-shaped step - the launch moves 100 MB.)
// HD = head size (128: Qwen3; 256: Gemma3).  GEMMA: Gemma3RMSNorm - x * rsqrt(mean(x^2) + eps) * (1 + w), all in fp32, rounded
// ONCE to the storage type - instead of Qwen3RMSNorm's w * round(x * rsqrt(..)).
template <int DT, int HD = 128, bool GEMMA = false>
__global__ void __launch_bounds__(256) qk_norm_rope_kernel(void* qkv, const void* __restrict__ wq, const void* __restrict__ wk,
                                                            const void* __restrict__ cos_t, const void* __restrict__ sin_t, float eps,
                                                            int64_t tokens, int seq, int hq, int hkv) {
    constexpr int VEC = DT == 0 ? 4 : 8;
    constexpr int LPH = HD / VEC;                        // lanes per head: 16 / 32 (bf16 128 / 256) or 32 / 64 (fp32)
    constexpr int HPW = 64 / LPH;                        // heads per wave
    static_assert(LPH <= 64 && HPW >= 1, "a head fits a wave");
    const int lane = threadIdx.x & 63;
    const int j = lane & (LPH - 1);
    const int64_t wave_id = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int heads = hq + hkv;                          // the heads this kernel touches (queries, then keys)
    const int64_t item_raw = wave_id * HPW + lane / LPH;
    const bool live = item_raw < tokens * heads;
    const int64_t item = live ? item_raw : 0;            // idle groups of the last wave follow along (shuffles below) and store nothing
    const int64_t tok = item / heads;
    const int h = (int)(item - tok * heads);
    const int pos = (int)(tok % seq);
    const int64_t width = (int64_t)(hq + 2 * hkv) * HD;
    uint4* px = (uint4*)((char*)qkv + (tok * width + (int64_t)h * HD) * (DT == 0 ? 4 : 2)) + j;
    // (idle groups load nothing: item 0 is being rewritten in place by the wave that owns it)
    const uint4 rx = live ? *px : make_uint4(0u, 0u, 0u, 0u);
    const uint4 rw = ((const uint4*)(h < hq ? wq : wk))[j];
    const uint4 rc = ((const uint4*)cos_t)[(int64_t)pos * LPH + j];
    const uint4 rs = ((const uint4*)sin_t)[(int64_t)pos * LPH + j];
    float x[VEC], w[VEC], c[VEC], sn[VEC], y[VEC], yp[VEC], o[VEC];
    enc_unpack<DT>(rx, x);
    enc_unpack<DT>(rw, w);
    enc_unpack<DT>(rc, c);
    enc_unpack<DT>(rs, sn);
    float sq = 0.0f;
#pragma unroll
    for (int e = 0; e < VEC; ++e) sq += x[e] * x[e];
#pragma unroll
    for (int off = LPH / 2; off > 0; off >>= 1) sq += __shfl_xor(sq, off, 64);
    const float rstd = rsqrtf(sq / (float)HD + eps);
#pragma unroll
    for (int e = 0; e < VEC; ++e)
        y[e] = GEMMA ? enc_round<DT>((x[e] * rstd) * (1.0f + w[e])) : enc_round<DT>(w[e] * enc_round<DT>(x[e] * rstd));
    // the partner's y (exact in the storage type: it was just rounded to it), packed: four shuffles
    const uint4 py = enc_pack<DT>(y);
    uint4 pp;
    pp.x = (u32)__shfl_xor((int)py.x, LPH / 2, 64);
    pp.y = (u32)__shfl_xor((int)py.y, LPH / 2, 64);
    pp.z = (u32)__shfl_xor((int)py.z, LPH / 2, 64);
    pp.w = (u32)__shfl_xor((int)py.w, LPH / 2, 64);
    enc_unpack<DT>(pp, yp);
    const float sgn = j < LPH / 2 ? -1.0f : 1.0f;       // rot = (-y[HD/2 ..], y[.. HD/2])
#pragma unroll
    for (int e = 0; e < VEC; ++e) o[e] = enc_round<DT>(y[e] * c[e]) + enc_round<DT>(sgn * yp[e] * sn[e]);
    if (live) *px = enc_pack<DT>(o);
}

// Gated MLP activation (Qwen3MLP: act_fn(gate_proj(x)) * up_proj(x), act_fn = SiLU) on the fused projection's output:
//   gate_up [rows][2 * inter]: gate columns, then up columns;   out [rows][inter] = round(round(silu(gate)) * up)
// (torch rounds SiLU's result to the storage type before the product).  16 bytes per lane.
template <int DT>
__global__ void __launch_bounds__(256) swiglu_kernel(const void* __restrict__ gate_up, int64_t rows, int inter, void* __restrict__ out) {
    constexpr int VEC = DT == 0 ? 4 : 8;
    const int per_row = inter / VEC;
    const int64_t total = rows * per_row;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / per_row;
        const int c = (int)(i - r * per_row);
        const uint4 g4 = ((const uint4*)gate_up)[r * 2 * per_row + c];
        const uint4 u4 = ((const uint4*)gate_up)[r * 2 * per_row + per_row + c];
        float g[VEC], u[VEC], y[VEC];
        enc_unpack<DT>(g4, g);
        enc_unpack<DT>(u4, u);
#pragma unroll
        for (int e = 0; e < VEC; ++e) y[e] = enc_round<DT>(g[e] / (1.0f + expf(-g[e]))) * u[e];   // expf, not the fast intrinsic: torch's SiLU
        ((uint4*)out)[i] = enc_pack<DT>(y);
    }
}

// fp32 values as bf16 PIECES for an fp32-class GEMM on the bf16 matrix pipe: hi = bf16(x), lo = bf16(x - hi) - sixteen bits of
// significand between them - laid out so that ONE bf16 GEMM with fp32 accumulation over the three-fold depth computes
//     x . w  ~  x_hi w_hi + x_lo w_hi + x_hi w_lo        (what is dropped: x_lo w_lo and the pieces' own residuals, ~2^-17 |x||w|)
//   PATTERN 0 (activations): out[r] = [ hi(x_r) | lo(x_r) | hi(x_r) ]      PATTERN 1 (weights): out[r] = [ hi | hi | lo ]
// x [rows][k] fp32, out [rows][3 k] bf16; k a multiple of 4; 16 bytes in, 3 x 8 bytes out per thread.
template <int PATTERN>
__global__ void __launch_bounds__(256) split3_kernel(const float* __restrict__ x, int64_t rows, int k, unsigned short* __restrict__ out) {
    const int per_row = k / 4;
    const int64_t total = rows * per_row;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / per_row;
        const int c = (int)(i - r * per_row);
        const float4 v = ((const float4*)x)[i];
        const float f[4] = {v.x, v.y, v.z, v.w};
        float hi[4], lo[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            hi[e] = bf16_to_f32(f32_to_bf16(f[e]));
            lo[e] = f[e] - hi[e];                       // exact in fp32: hi is f rounded to 8 significant bits
        }
        const uint2 ph = make_uint2(pack_bf16_hw(hi[0], hi[1]), pack_bf16_hw(hi[2], hi[3]));
        const uint2 pl = make_uint2(pack_bf16_hw(lo[0], lo[1]), pack_bf16_hw(lo[2], lo[3]));
        uint2* o = (uint2*)(out + r * 3 * (int64_t)k) + c;
        o[0] = ph;
        o[per_row] = PATTERN == 0 ? pl : ph;
        o[2 * per_row] = PATTERN == 0 ? ph : pl;
    }
}

// Gemma3's "sandwich" norms around a sublayer (Gemma3DecoderLayer.forward; google/embeddinggemma-300m is the reference's second
// embedder, ec2/generate_embeddings/embedders.py:1-4):   s = x + post_norm(y),   h = pre_norm_of_what_follows(s)
// with Gemma3RMSNorm(v) = v * rsqrt(mean(v^2) + eps) * (1 + w) in fp32, rounded once; the residual add in the storage type.
// y may be NULL (the first norm of the first layer: s = x).  out_sum (may be NULL) receives s.  One wave per row.
template <int DT, int LN>
__global__ void __launch_bounds__(256) gemma_norm_kernel(const void* y, const void* x, const void* __restrict__ w_post,
                                                          const void* __restrict__ w_next, float eps, int64_t rows, int d, void* out_sum,
                                                          void* out_norm, unsigned short* pieces = nullptr) {
    constexpr int VEC = DT == 0 ? 4 : 8;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nchunk = d / VEC;
    const uint4* px = (const uint4*)x + row * nchunk;
    const uint4* py = y ? (const uint4*)y + row * nchunk : nullptr;
    uint4 rx[LN], ry[LN], rp[LN], rn[LN];
#pragma unroll
    for (int j = 0; j < LN; ++j) {
        const int c = lane + 64 * j;
        if (c < nchunk) {
            rx[j] = px[c];
            rn[j] = ((const uint4*)w_next)[c];
            if (py) {
                ry[j] = py[c];
                rp[j] = ((const uint4*)w_post)[c];
            }
        }
    }
    float s[LN][VEC];
    if (py) {
        float t[LN][VEC];
        float sq = 0.0f;
#pragma unroll
        for (int j = 0; j < LN; ++j)
            if (lane + 64 * j < nchunk) {
                enc_unpack<DT>(ry[j], t[j]);
#pragma unroll
                for (int e = 0; e < VEC; ++e) sq += t[j][e] * t[j][e];
            }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sq += __shfl_xor(sq, off, 64);
        const float rstd = rsqrtf(sq / (float)d + eps);
#pragma unroll
        for (int j = 0; j < LN; ++j)
            if (lane + 64 * j < nchunk) {
                float fx[VEC], wp[VEC];
                enc_unpack<DT>(rx[j], fx);
                enc_unpack<DT>(rp[j], wp);
#pragma unroll
                for (int e = 0; e < VEC; ++e) s[j][e] = enc_round<DT>(fx[e] + enc_round<DT>((t[j][e] * rstd) * (1.0f + wp[e])));
            }
    } else {
#pragma unroll
        for (int j = 0; j < LN; ++j)
            if (lane + 64 * j < nchunk) enc_unpack<DT>(rx[j], s[j]);
    }
    float sq = 0.0f;
#pragma unroll
    for (int j = 0; j < LN; ++j)
        if (lane + 64 * j < nchunk) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) sq += s[j][e] * s[j][e];
        }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sq += __shfl_xor(sq, off, 64);
    const float rstd = rsqrtf(sq / (float)d + eps);
    uint4* ps = out_sum ? (uint4*)out_sum + row * nchunk : nullptr;
    uint4* po = (uint4*)out_norm + row * nchunk;
#pragma unroll
    for (int j = 0; j < LN; ++j) {
        const int c = lane + 64 * j;
        if (c < nchunk) {
            float wn[VEC], h[VEC];
            enc_unpack<DT>(rn[j], wn);
#pragma unroll
            for (int e = 0; e < VEC; ++e) h[e] = (s[j][e] * rstd) * (1.0f + wn[e]);
            if (ps) ps[c] = enc_pack<DT>(s[j]);
            po[c] = enc_pack<DT>(h);
            if (DT == 0 && pieces) store_pieces4(pieces + row * 3 * (int64_t)d, d, c, h);
        }
    }
}

// Gemma3MLP's activation: act_fn(gate_proj(x)) * up_proj(x), act_fn = gelu_pytorch_tanh, on the fused projection's output
//   gate_up [rows][2 * inter]: gate columns, then up columns;   out [rows][inter] = round(round(gelu_tanh(gate)) * up)
template <int DT>
__global__ void __launch_bounds__(256) geglu_kernel(const void* __restrict__ gate_up, int64_t rows, int inter, void* __restrict__ out) {
    constexpr int VEC = DT == 0 ? 4 : 8;
    const int per_row = inter / VEC;
    const int64_t total = rows * per_row;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / per_row;
        const int c = (int)(i - r * per_row);
        const uint4 g4 = ((const uint4*)gate_up)[r * 2 * per_row + c];
        const uint4 u4 = ((const uint4*)gate_up)[r * 2 * per_row + per_row + c];
        float g[VEC], u[VEC], y[VEC];
        enc_unpack<DT>(g4, g);
        enc_unpack<DT>(u4, u);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const float v = g[e];
            const float inner = 0.7978845608028654f * (v + 0.044715f * v * v * v);       // sqrt(2 / pi) (v + 0.044715 v^3)
            y[e] = enc_round<DT>(0.5f * v * (1.0f + tanhf(inner))) * u[e];
        }
        ((uint4*)out)[i] = enc_pack<DT>(y);
    }
}

// fp32 activations straight into the pieces of the GEMM that follows (fp32_gemm = "bf16x3"): one pass instead of the activation's
// own (read + write fp32) and ts_split_pieces' (read fp32 + write pieces).
//   KIND 0: y = gelu(x) (exact erf form: BertIntermediate with hidden_act = "gelu"), x [rows][n]
//   KIND 1: y = silu(gate) * up (Qwen3MLP),  KIND 2: y = gelu_tanh(gate) * up (Gemma3MLP), x = [rows][2 n]: gate columns, then up
// out [rows][3 n] bf16 = [hi | lo | hi] of y.
template <int KIND>
__global__ void __launch_bounds__(256) act_pieces_kernel(const float* __restrict__ x, int64_t rows, int n, unsigned short* __restrict__ out,
                                                          const float* __restrict__ bias) {    // bias (may be NULL): [n] (KIND 0) / [2 n], added to x
    const int per_row = n / 4;
    const int64_t total = rows * per_row;
    const int in_row = (KIND == 0 ? 1 : 2) * per_row;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / per_row;
        const int c = (int)(i - r * per_row);
        const float4 g4 = ((const float4*)x)[r * in_row + c];
        float g[4] = {g4.x, g4.y, g4.z, g4.w};
        if (bias) {
            const float4 bg = ((const float4*)bias)[c];
            g[0] += bg.x; g[1] += bg.y; g[2] += bg.z; g[3] += bg.w;
        }
        float y[4];
        if (KIND == 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = 0.5f * g[e] * (1.0f + erff(g[e] * 0.7071067811865476f));
        } else {
            const float4 u4 = ((const float4*)x)[r * in_row + per_row + c];
            float u[4] = {u4.x, u4.y, u4.z, u4.w};
            if (bias) {
                const float4 bu = ((const float4*)bias)[per_row + c];
                u[0] += bu.x; u[1] += bu.y; u[2] += bu.z; u[3] += bu.w;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (KIND == 1) {
                    y[e] = (g[e] / (1.0f + expf(-g[e]))) * u[e];
                } else {
                    const float inner = 0.7978845608028654f * (g[e] + 0.044715f * g[e] * g[e] * g[e]);
                    y[e] = (0.5f * g[e] * (1.0f + tanhf(inner))) * u[e];
                }
            }
        }
        store_pieces4(out + r * 3 * (int64_t)n, n, c, y);
    }
}

}  // namespace ts
