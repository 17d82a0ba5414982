// libtsearch.so - C ABI (include/tsearch.h), part 3: the encoder-side kernels: pooling + L2
// normalisation + cast, residual add + LayerNorm, the input layer, short-sequence attention.
#include "host.h"
#include "kernels_attention.h"
#include "kernels_encoder.h"

extern "C" int ts_pool_normalize(int device, const void* hidden, int h_dtype, const int64_t* attention_mask, int64_t n,
                                 int32_t seq, int32_t d, int pooling, int normalize, void* out, int out_dtype, int64_t out_ld,
                                 void* stream) {
    if (!hidden || !attention_mask || !out) return fail(TS_ERR_INVALID, "NULL argument");
    if ((h_dtype != TS_F32 && h_dtype != TS_BF16) || (out_dtype != TS_F32 && out_dtype != TS_BF16))
        return fail(TS_ERR_INVALID, "dtype");
    if (n < 0 || seq < 1 || d < 1 || d > 4096 || out_ld < d) return fail(TS_ERR_INVALID, "bad shape (d must be <= 4096)");
    if (pooling < TS_POOL_MEAN || pooling > TS_POOL_CLS) return fail(TS_ERR_INVALID, "pooling %d", pooling);
    if (n == 0) return TS_OK;
    TS_TRY(check_device(device));
    HIP_TRY(hipSetDevice(device));
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)n);
    // the encoders' shapes take the vector form (16-byte loads, tokens dealt over thread groups); anything else the general one
    const int vec = h_dtype == TS_BF16 ? 8 : 4;
    const bool vform = d % vec == 0 && d / vec <= 256 && seq <= kPoolVecSeq && ((uintptr_t)hidden & 15) == 0;
#define TS_POOL_LAUNCH(H, O)                                                                                                  \
    do {                                                                                                                      \
        if (vform) pool_normalize_vec_kernel<H, O><<<grid, 256, 0, st>>>(hidden, attention_mask, seq, d, pooling, normalize, out, out_ld); \
        else pool_normalize_kernel<H, O><<<grid, 256, 0, st>>>(hidden, attention_mask, seq, d, pooling, normalize, out, out_ld); \
    } while (0)
    if (h_dtype == TS_F32 && out_dtype == TS_F32) TS_POOL_LAUNCH(0, 0);
    else if (h_dtype == TS_F32) TS_POOL_LAUNCH(0, 1);
    else if (out_dtype == TS_F32) TS_POOL_LAUNCH(1, 0);
    else TS_POOL_LAUNCH(1, 1);
#undef TS_POOL_LAUNCH
    HIP_TRY(hipGetLastError());
    return TS_OK;
}

static int add_layernorm_impl(int device, const void* a, const void* b, const void* gamma, const void* beta, float eps, int64_t rows,
                              int32_t d, int dtype, void* out, unsigned short* pieces, void* stream, const float* a_bias = nullptr) {
    if (a_bias && (dtype != TS_F32 || ((uintptr_t)a_bias & 15) != 0)) return fail(TS_ERR_INVALID, "a_bias: fp32 rows only, 16-byte aligned");
    if (!a || !b || !gamma || !beta || !out) return fail(TS_ERR_INVALID, "NULL argument");
    if (pieces && (dtype != TS_F32 || ((uintptr_t)pieces & 7) != 0)) return fail(TS_ERR_INVALID, "pieces come from fp32 rows, 8-byte aligned");
    if (dtype != TS_F32 && dtype != TS_BF16) return fail(TS_ERR_INVALID, "dtype %d", dtype);
    const int vec = dtype == TS_BF16 ? 8 : 4;
    if (rows < 0 || d < vec || d % vec || d > 64 * kLnMax * vec)
        return fail(TS_ERR_INVALID, "d = %d must be a multiple of %d and at most %d", d, vec, 64 * kLnMax * vec);
    if ((((uintptr_t)a | (uintptr_t)b | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)out) & 15) != 0)
        return fail(TS_ERR_INVALID, "buffers must be 16-byte aligned");
    if (rows == 0) return TS_OK;
    TS_TRY(check_device(device));
    HIP_TRY(hipSetDevice(device));
    const unsigned grid = (unsigned)((rows + 3) / 4);
    const int per_lane = (d / vec + 63) / 64;              // 16-byte accesses per lane
    hipStream_t st = (hipStream_t)stream;
#define TS_LN_LAUNCH(DT_)                                                                                        \
    do {                                                                                                         \
        if (per_lane <= 1) add_layernorm_kernel<DT_, 1><<<grid, 256, 0, st>>>(a, b, gamma, beta, eps, rows, d, out, pieces, a_bias); \
        else if (per_lane <= 2) add_layernorm_kernel<DT_, 2><<<grid, 256, 0, st>>>(a, b, gamma, beta, eps, rows, d, out, pieces, a_bias); \
        else add_layernorm_kernel<DT_, 4><<<grid, 256, 0, st>>>(a, b, gamma, beta, eps, rows, d, out, pieces, a_bias); \
    } while (0)
    if (dtype == TS_F32) TS_LN_LAUNCH(0);
    else TS_LN_LAUNCH(1);
#undef TS_LN_LAUNCH
    HIP_TRY(hipGetLastError());
    return TS_OK;
}

extern "C" int ts_add_layernorm(int device, const void* a, const void* b, const void* gamma, const void* beta, float eps, int64_t rows,
                                int32_t d, int dtype, void* out, void* stream) {
    return add_layernorm_impl(device, a, b, gamma, beta, eps, rows, d, dtype, out, nullptr, stream);
}

extern "C" int ts_add_layernorm_pieces(int device, const void* a, const void* a_bias, const void* b, const void* gamma, const void* beta,
                                       float eps, int64_t rows, int32_t d, void* out, void* pieces, void* stream) {
    if (!pieces) return fail(TS_ERR_INVALID, "NULL argument");
    return add_layernorm_impl(device, a, b, gamma, beta, eps, rows, d, TS_F32, out, (unsigned short*)pieces, stream, (const float*)a_bias);
}

extern "C" int ts_embed_layernorm(int device, const int64_t* ids, const int64_t* type_ids, const void* word, const void* pos,
                                  const void* type, int64_t n_word, int64_t n_pos, int64_t n_type, const void* gamma, const void* beta,
                                  float eps, int64_t tokens, int32_t seq, int32_t d, int dtype, void* out, void* stream) {
    if (!ids || !word || !pos || !type || !gamma || !beta || !out) return fail(TS_ERR_INVALID, "NULL argument");
    if (dtype != TS_F32 && dtype != TS_BF16) return fail(TS_ERR_INVALID, "dtype %d", dtype);
    const int vec = dtype == TS_BF16 ? 8 : 4;
    if (tokens < 0 || seq < 1 || d < vec || d % vec || d > 64 * kLnMax * vec)
        return fail(TS_ERR_INVALID, "d = %d must be a multiple of %d and at most %d; seq >= 1", d, vec, 64 * kLnMax * vec);
    if (n_word < 1 || n_pos < 1 || n_type < 1) return fail(TS_ERR_INVALID, "empty embedding table");
    if ((((uintptr_t)word | (uintptr_t)pos | (uintptr_t)type | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)out) & 15) != 0)
        return fail(TS_ERR_INVALID, "tables and output must be 16-byte aligned");
    if (tokens == 0) return TS_OK;
    TS_TRY(check_device(device));
    HIP_TRY(hipSetDevice(device));
    const unsigned grid = (unsigned)((tokens + 3) / 4);
    const int per_lane = (d / vec + 63) / 64;
    hipStream_t st = (hipStream_t)stream;
#define TS_EMB_LAUNCH(DT_, LN_)                                                                                              \
    embed_layernorm_kernel<DT_, LN_><<<grid, 256, 0, st>>>(ids, type_ids, word, pos, type, n_word, n_pos, n_type, gamma, beta, eps, \
                                                          tokens, seq, d, out)
    if (dtype == TS_F32) {
        if (per_lane <= 1) TS_EMB_LAUNCH(0, 1);
        else if (per_lane <= 2) TS_EMB_LAUNCH(0, 2);
        else TS_EMB_LAUNCH(0, 4);
    } else {
        if (per_lane <= 1) TS_EMB_LAUNCH(1, 1);
        else if (per_lane <= 2) TS_EMB_LAUNCH(1, 2);
        else TS_EMB_LAUNCH(1, 4);
    }
#undef TS_EMB_LAUNCH
    HIP_TRY(hipGetLastError());
    return TS_OK;
}

extern "C" int ts_attention_short(int device, const void* qkv, const int64_t* attention_mask, int32_t batch, int32_t seq, int32_t heads,
                                 int32_t head_dim, void* out, void* stream) {
    if (!qkv || !out) return fail(TS_ERR_INVALID, "NULL argument");
    if (batch < 0 || seq < 1 || heads < 1) return fail(TS_ERR_INVALID, "batch = %d, seq = %d, heads = %d", batch, seq, heads);
    if (head_dim != 64 || seq > kAttnRowsMaxSeq)
        return fail(TS_ERR_UNSUPPORTED, "head size %d / %d tokens: this kernel serves head size 64 and at most %d tokens", head_dim, seq,
                    kAttnRowsMaxSeq);
    if ((((uintptr_t)qkv | (uintptr_t)out) & 15) != 0) return fail(TS_ERR_INVALID, "qkv and out must be 16-byte aligned");
    if (batch == 0) return TS_OK;
    TS_TRY(check_device(device));
    HIP_TRY(hipSetDevice(device));
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = (unsigned)(((int64_t)batch * heads + 3) / 4);
    const unsigned short* in = (const unsigned short*)qkv;
    unsigned short* o = (unsigned short*)out;
    // 65 .. 128 tokens: one query tile at a time (attention_rows_kernel, dynamic LDS: 4 waves x up to 17 KB, two workgroups per CU)
#define TS_ATTN_ROWS(T_)                                                                                                     \
    do {                                                                                                                     \
        constexpr int lds_ = 4 * attn_rows_wave_lds(T_);                                                                     \
        static std::atomic<unsigned long long> attr_{0};                                                                     \
        const unsigned long long bit_ = 1ull << (device & 63);                                                               \
        if (!(attr_.load(std::memory_order_acquire) & bit_)) {                                                               \
            HIP_TRY(hipFuncSetAttribute((const void*)attention_rows_kernel<T_>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_)); \
            attr_.fetch_or(bit_, std::memory_order_release);                                                                 \
        }                                                                                                                    \
        attention_rows_kernel<T_><<<grid, 256, lds_, st>>>(in, attention_mask, batch, seq, heads, o);                        \
    } while (0)
    switch ((seq + 15) / 16) {
        case 1: attention_short_kernel<1><<<grid, 256, 0, st>>>(in, attention_mask, batch, seq, heads, o); break;
        case 2: attention_short_kernel<2><<<grid, 256, 0, st>>>(in, attention_mask, batch, seq, heads, o); break;
        case 3: attention_short_kernel<3><<<grid, 256, 0, st>>>(in, attention_mask, batch, seq, heads, o); break;
        case 4: attention_short_kernel<4><<<grid, 256, 0, st>>>(in, attention_mask, batch, seq, heads, o); break;
        case 5: TS_ATTN_ROWS(5); break;
        case 6: TS_ATTN_ROWS(6); break;
        case 7: TS_ATTN_ROWS(7); break;
        default: TS_ATTN_ROWS(8); break;
    }
#undef TS_ATTN_ROWS
    HIP_TRY(hipGetLastError());
    return TS_OK;
}


static int add_rmsnorm_impl(int device, const void* a, const void* b, const void* gamma, float eps, int64_t rows, int32_t d, int dtype,
                            void* out_sum, void* out_norm, unsigned short* pieces, void* stream) {
    if (!a || !gamma || !out_norm) return fail(TS_ERR_INVALID, "NULL argument");
    if (pieces && (dtype != TS_F32 || ((uintptr_t)pieces & 7) != 0)) return fail(TS_ERR_INVALID, "pieces come from fp32 rows, 8-byte aligned");
    if (dtype != TS_F32 && dtype != TS_BF16) return fail(TS_ERR_INVALID, "dtype %d", dtype);
    const int vec = dtype == TS_BF16 ? 8 : 4;
    if (rows < 0 || d < vec || d % vec || d > 64 * kLnMax * vec)
        return fail(TS_ERR_INVALID, "d = %d must be a multiple of %d and at most %d", d, vec, 64 * kLnMax * vec);
    if ((((uintptr_t)a | (uintptr_t)b | (uintptr_t)gamma | (uintptr_t)out_sum | (uintptr_t)out_norm) & 15) != 0)
        return fail(TS_ERR_INVALID, "buffers must be 16-byte aligned");
    if (rows == 0) return TS_OK;
    TS_TRY(check_device(device));
    HIP_TRY(hipSetDevice(device));
    const unsigned grid = (unsigned)((rows + 3) / 4);
    const int per_lane = (d / vec + 63) / 64;
    hipStream_t st = (hipStream_t)stream;
#define TS_RMS_LAUNCH(DT_)                                                                                               \
    do {                                                                                                                 \
        if (per_lane <= 1) add_rmsnorm_kernel<DT_, 1><<<grid, 256, 0, st>>>(a, b, gamma, eps, rows, d, out_sum, out_norm, pieces); \
        else if (per_lane <= 2) add_rmsnorm_kernel<DT_, 2><<<grid, 256, 0, st>>>(a, b, gamma, eps, rows, d, out_sum, out_norm, pieces); \
        else add_rmsnorm_kernel<DT_, 4><<<grid, 256, 0, st>>>(a, b, gamma, eps, rows, d, out_sum, out_norm, pieces);     \
    } while (0)
    if (dtype == TS_F32) TS_RMS_LAUNCH(0);
    else TS_RMS_LAUNCH(1);
#undef TS_RMS_LAUNCH
    HIP_TRY(hipGetLastError());
    return TS_OK;
}

extern "C" int ts_add_rmsnorm(int device, const void* a, const void* b, const void* gamma, float eps, int64_t rows, int32_t d, int dtype,
                              void* out_sum, void* out_norm, void* stream) {
    return add_rmsnorm_impl(device, a, b, gamma, eps, rows, d, dtype, out_sum, out_norm, nullptr, stream);
}

extern "C" int ts_add_rmsnorm_pieces(int device, const void* a, const void* b, const void* gamma, float eps, int64_t rows, int32_t d,
                                     void* out_sum, void* out_norm, void* pieces, void* stream) {
    if (!pieces) return fail(TS_ERR_INVALID, "NULL argument");
    return add_rmsnorm_impl(device, a, b, gamma, eps, rows, d, TS_F32, out_sum, out_norm, (unsigned short*)pieces, stream);
}

extern "C" int ts_attention_float(int device, const void* qkv, const void* qkv_bias, const int64_t* attention_mask, int32_t batch, int32_t seq,
                                  int32_t q_heads, int32_t kv_heads, int32_t head_dim, int causal, float scale, void* out, void* pieces,
                                  void* stream) {
    if (!qkv || (!out && !pieces)) return fail(TS_ERR_INVALID, "NULL argument");
    if (((uintptr_t)qkv_bias & 15) != 0) return fail(TS_ERR_INVALID, "qkv_bias must be 16-byte aligned");
    if (batch < 0 || seq < 1 || q_heads < 1 || kv_heads < 1 || q_heads % kv_heads != 0)
        return fail(TS_ERR_INVALID, "batch = %d, seq = %d, heads = %d over %d", batch, seq, q_heads, kv_heads);
    if ((head_dim != 64 && head_dim != 128 && head_dim != 256) || seq > attn_f32_max_seq(head_dim))
        return fail(TS_ERR_UNSUPPORTED, "head size %d / %d tokens: this kernel serves head sizes 64 / 128 / 256 up to 512 / 256 / 128 tokens",
                    head_dim, seq);
    if ((((uintptr_t)qkv | (uintptr_t)out) & 15) != 0 || (((uintptr_t)pieces) & 7) != 0)
        return fail(TS_ERR_INVALID, "qkv and out must be 16-byte aligned, pieces 8-byte");
    if (!(scale > 0.0f)) return fail(TS_ERR_INVALID, "scale must be positive");
    if (batch == 0) return TS_OK;
    TS_TRY(check_device(device));
    HIP_TRY(hipSetDevice(device));
    hipStream_t st = (hipStream_t)stream;
    const int T = (seq + 15) / 16;
    const unsigned grid = (unsigned)((int64_t)batch * q_heads);
    const unsigned threads = 64u * (unsigned)std::min(T, 4);
    const float scale_log2e = scale * 1.4426950408889634f;
#define TS_ATTN_F32(HD_, C_)                                                                                                  \
    do {                                                                                                                      \
        const int lds_ = attn_f32_lds(HD_, T);                                                                                \
        static std::atomic<unsigned long long> attr_{0};                                                                      \
        const unsigned long long bit_ = 1ull << (device & 63);                                                                \
        if (!(attr_.load(std::memory_order_acquire) & bit_)) {                                                                \
            HIP_TRY(hipFuncSetAttribute((const void*)attention_f32_kernel<HD_, C_>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                        attn_f32_lds(HD_, attn_f32_max_seq(HD_) / 16)));                                      \
            attr_.fetch_or(bit_, std::memory_order_release);                                                                  \
        }                                                                                                                     \
        attention_f32_kernel<HD_, C_><<<grid, threads, lds_, st>>>((const float*)qkv, attention_mask, batch, seq, q_heads, kv_heads, \
                                                                     scale_log2e, (float*)out, (unsigned short*)pieces,      \
                                                                     (const float*)qkv_bias);                                \
    } while (0)
    if (head_dim == 64) { if (causal) TS_ATTN_F32(64, true); else TS_ATTN_F32(64, false); }
    else if (head_dim == 128) { if (causal) TS_ATTN_F32(128, true); else TS_ATTN_F32(128, false); }
    else { if (causal) TS_ATTN_F32(256, true); else TS_ATTN_F32(256, false); }
#undef TS_ATTN_F32
    HIP_TRY(hipGetLastError());
    return TS_OK;
}

extern "C" int ts_attention_gqa(int device, const void* qkv, const int64_t* attention_mask, int32_t batch, int32_t seq, int32_t q_heads,
                               int32_t kv_heads, int32_t head_dim, int causal, void* out, void* stream) {
    if (!qkv || !out) return fail(TS_ERR_INVALID, "NULL argument");
    if (batch < 0 || seq < 1 || q_heads < 1 || kv_heads < 1 || q_heads % kv_heads != 0)
        return fail(TS_ERR_INVALID, "batch = %d, seq = %d, heads = %d over %d", batch, seq, q_heads, kv_heads);
    if (head_dim != 128 || seq > kAttnGqaRowsMaxSeq)
        return fail(TS_ERR_UNSUPPORTED, "head size %d / %d tokens: this kernel serves head size 128 and at most %d tokens", head_dim, seq,
                    kAttnGqaRowsMaxSeq);
    if ((((uintptr_t)qkv | (uintptr_t)out) & 15) != 0) return fail(TS_ERR_INVALID, "qkv and out must be 16-byte aligned");
    if (batch == 0) return TS_OK;
    TS_TRY(check_device(device));
    HIP_TRY(hipSetDevice(device));
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = (unsigned)(((int64_t)batch * q_heads + 3) / 4);
    const unsigned short* in = (const unsigned short*)qkv;
    unsigned short* o = (unsigned short*)out;
#define TS_ATTN_GQA(T_, C_)                                                                                                  \
    do {                                                                                                                     \
        constexpr int lds_ = 4 * attn_gqa_wave_lds(T_);                                                                      \
        static std::atomic<unsigned long long> attr_{0};                                                                     \
        const unsigned long long bit_ = 1ull << (device & 63);                                                               \
        if (!(attr_.load(std::memory_order_acquire) & bit_)) {                                                               \
            HIP_TRY(hipFuncSetAttribute((const void*)attention_gqa_kernel<T_, C_>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_)); \
            attr_.fetch_or(bit_, std::memory_order_release);                                                                 \
        }                                                                                                                    \
        attention_gqa_kernel<T_, C_><<<grid, 256, lds_, st>>>(in, attention_mask, batch, seq, q_heads, kv_heads, o);         \
    } while (0)
    // 65 .. 128 tokens: one query tile at a time against K fragments in registers and a V^T image in LDS that the R query heads
    // of a key / value group (R waves of one workgroup) share
    const int per_kv = q_heads / kv_heads;
    const int R = per_kv % 4 == 0 ? 4 : (per_kv % 2 == 0 ? 2 : 1);
    const unsigned rows_grid = (unsigned)((int64_t)batch * kv_heads * (per_kv / R));
#define TS_ATTN_GQA_ROWS_R(T_, C_, R_)                                                                                       \
    do {                                                                                                                     \
        constexpr int lds_ = attn_gqa_rows_lds(T_, R_);                                                                      \
        static_assert(lds_ <= 160 * 1024, "the image and the waves' tiles fit the CU's LDS");                                \
        static std::atomic<unsigned long long> attr_{0};                                                                     \
        const unsigned long long bit_ = 1ull << (device & 63);                                                               \
        if (!(attr_.load(std::memory_order_acquire) & bit_)) {                                                               \
            HIP_TRY(hipFuncSetAttribute((const void*)attention_gqa_rows_kernel<T_, C_, R_>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_)); \
            attr_.fetch_or(bit_, std::memory_order_release);                                                                 \
        }                                                                                                                    \
        attention_gqa_rows_kernel<T_, C_, R_><<<rows_grid, 64 * R_, lds_, st>>>(in, attention_mask, batch, seq, q_heads, kv_heads, o); \
    } while (0)
#define TS_ATTN_GQA_ROWS(T_, C_)                                                                                             \
    do {                                                                                                                     \
        if (R == 4) TS_ATTN_GQA_ROWS_R(T_, C_, 4);                                                                           \
        else if (R == 2) TS_ATTN_GQA_ROWS_R(T_, C_, 2);                                                                      \
        else TS_ATTN_GQA_ROWS_R(T_, C_, 1);                                                                                  \
    } while (0)
    const int tiles = (seq + 15) / 16;
    if (causal) {
        switch (tiles) {
            case 1: TS_ATTN_GQA(1, true); break;
            case 2: TS_ATTN_GQA(2, true); break;
            case 3: TS_ATTN_GQA(3, true); break;
            case 4: TS_ATTN_GQA(4, true); break;
            case 5: TS_ATTN_GQA_ROWS(5, true); break;
            case 6: TS_ATTN_GQA_ROWS(6, true); break;
            case 7: TS_ATTN_GQA_ROWS(7, true); break;
            default: TS_ATTN_GQA_ROWS(8, true); break;
        }
    } else {
        switch (tiles) {
            case 1: TS_ATTN_GQA(1, false); break;
            case 2: TS_ATTN_GQA(2, false); break;
            case 3: TS_ATTN_GQA(3, false); break;
            case 4: TS_ATTN_GQA(4, false); break;
            case 5: TS_ATTN_GQA_ROWS(5, false); break;
            case 6: TS_ATTN_GQA_ROWS(6, false); break;
            case 7: TS_ATTN_GQA_ROWS(7, false); break;
            default: TS_ATTN_GQA_ROWS(8, false); break;
        }
    }
#undef TS_ATTN_GQA_ROWS
#undef TS_ATTN_GQA_ROWS_R
#undef TS_ATTN_GQA
    HIP_TRY(hipGetLastError());
    return TS_OK;
}

static int qk_norm_rope_launch(int device, void* qkv, const void* q_weight, const void* k_weight, const void* cos_table,
                               const void* sin_table, float eps, int64_t tokens, int32_t seq, int32_t q_heads, int32_t kv_heads,
                               int32_t head_dim, int dtype, bool gemma, void* stream) {
    if (!qkv || !q_weight || !k_weight || !cos_table || !sin_table) return fail(TS_ERR_INVALID, "NULL argument");
    if (dtype != TS_F32 && dtype != TS_BF16) return fail(TS_ERR_INVALID, "dtype %d", dtype);
    if (tokens < 0 || seq < 1 || q_heads < 1 || kv_heads < 1) return fail(TS_ERR_INVALID, "bad shape");
    if (head_dim != (gemma ? 256 : 128))
        return fail(TS_ERR_UNSUPPORTED, "head size %d: this kernel serves head size %d", head_dim, gemma ? 256 : 128);
    if ((((uintptr_t)qkv | (uintptr_t)q_weight | (uintptr_t)k_weight | (uintptr_t)cos_table | (uintptr_t)sin_table) & 15) != 0)
        return fail(TS_ERR_INVALID, "buffers must be 16-byte aligned");
    if (tokens == 0) return TS_OK;
    TS_TRY(check_device(device));
    HIP_TRY(hipSetDevice(device));
    const int64_t items = tokens * (q_heads + kv_heads);
    const int vec = dtype == TS_BF16 ? 8 : 4;
    const int per_wg = 4 * (64 / (head_dim / vec));                    // 4 waves x heads per wave
    const unsigned grid = (unsigned)((items + per_wg - 1) / per_wg);
    hipStream_t st = (hipStream_t)stream;
    if (gemma) {
        if (dtype == TS_F32) qk_norm_rope_kernel<0, 256, true><<<grid, 256, 0, st>>>(qkv, q_weight, k_weight, cos_table, sin_table, eps, tokens, seq, q_heads, kv_heads);
        else qk_norm_rope_kernel<1, 256, true><<<grid, 256, 0, st>>>(qkv, q_weight, k_weight, cos_table, sin_table, eps, tokens, seq, q_heads, kv_heads);
    } else {
        if (dtype == TS_F32) qk_norm_rope_kernel<0, 128, false><<<grid, 256, 0, st>>>(qkv, q_weight, k_weight, cos_table, sin_table, eps, tokens, seq, q_heads, kv_heads);
        else qk_norm_rope_kernel<1, 128, false><<<grid, 256, 0, st>>>(qkv, q_weight, k_weight, cos_table, sin_table, eps, tokens, seq, q_heads, kv_heads);
    }
    HIP_TRY(hipGetLastError());
    return TS_OK;
}

extern "C" int ts_qk_norm_rope(int device, void* qkv, const void* q_weight, const void* k_weight, const void* cos_table,
                               const void* sin_table, float eps, int64_t tokens, int32_t seq, int32_t q_heads, int32_t kv_heads,
                               int32_t head_dim, int dtype, void* stream) {
    return qk_norm_rope_launch(device, qkv, q_weight, k_weight, cos_table, sin_table, eps, tokens, seq, q_heads, kv_heads, head_dim, dtype,
                               false, stream);
}

extern "C" int ts_gemma_qk_norm_rope(int device, void* qkv, const void* q_weight, const void* k_weight, const void* cos_table,
                                     const void* sin_table, float eps, int64_t tokens, int32_t seq, int32_t q_heads, int32_t kv_heads,
                                     int32_t head_dim, int dtype, void* stream) {
    return qk_norm_rope_launch(device, qkv, q_weight, k_weight, cos_table, sin_table, eps, tokens, seq, q_heads, kv_heads, head_dim, dtype,
                               true, stream);
}

static int gemma_norm_impl(int device, const void* y, const void* x, const void* w_post, const void* w_next, float eps, int64_t rows,
                           int32_t d, int dtype, void* out_sum, void* out_norm, unsigned short* pieces, void* stream) {
    if (!x || !w_next || !out_norm || (y && !w_post)) return fail(TS_ERR_INVALID, "NULL argument");
    if (pieces && (dtype != TS_F32 || ((uintptr_t)pieces & 7) != 0)) return fail(TS_ERR_INVALID, "pieces come from fp32 rows, 8-byte aligned");
    if (dtype != TS_F32 && dtype != TS_BF16) return fail(TS_ERR_INVALID, "dtype %d", dtype);
    const int vec = dtype == TS_BF16 ? 8 : 4;
    if (rows < 0 || d < vec || d % vec || d > 64 * kLnMax * vec)
        return fail(TS_ERR_INVALID, "d = %d must be a multiple of %d and at most %d", d, vec, 64 * kLnMax * vec);
    if ((((uintptr_t)y | (uintptr_t)x | (uintptr_t)w_post | (uintptr_t)w_next | (uintptr_t)out_sum | (uintptr_t)out_norm) & 15) != 0)
        return fail(TS_ERR_INVALID, "buffers must be 16-byte aligned");
    if (rows == 0) return TS_OK;
    TS_TRY(check_device(device));
    HIP_TRY(hipSetDevice(device));
    const unsigned grid = (unsigned)((rows + 3) / 4);
    const int per_lane = (d / vec + 63) / 64;
    hipStream_t st = (hipStream_t)stream;
#define TS_GN_LAUNCH(DT_)                                                                                                      \
    do {                                                                                                                       \
        if (per_lane <= 1) gemma_norm_kernel<DT_, 1><<<grid, 256, 0, st>>>(y, x, w_post, w_next, eps, rows, d, out_sum, out_norm, pieces); \
        else if (per_lane <= 2) gemma_norm_kernel<DT_, 2><<<grid, 256, 0, st>>>(y, x, w_post, w_next, eps, rows, d, out_sum, out_norm, pieces); \
        else gemma_norm_kernel<DT_, 4><<<grid, 256, 0, st>>>(y, x, w_post, w_next, eps, rows, d, out_sum, out_norm, pieces);   \
    } while (0)
    if (dtype == TS_F32) TS_GN_LAUNCH(0);
    else TS_GN_LAUNCH(1);
#undef TS_GN_LAUNCH
    HIP_TRY(hipGetLastError());
    return TS_OK;
}

extern "C" int ts_gemma_norm(int device, const void* y, const void* x, const void* w_post, const void* w_next, float eps, int64_t rows,
                             int32_t d, int dtype, void* out_sum, void* out_norm, void* stream) {
    return gemma_norm_impl(device, y, x, w_post, w_next, eps, rows, d, dtype, out_sum, out_norm, nullptr, stream);
}

extern "C" int ts_gemma_norm_pieces(int device, const void* y, const void* x, const void* w_post, const void* w_next, float eps,
                                    int64_t rows, int32_t d, void* out_sum, void* out_norm, void* pieces, void* stream) {
    if (!pieces) return fail(TS_ERR_INVALID, "NULL argument");
    return gemma_norm_impl(device, y, x, w_post, w_next, eps, rows, d, TS_F32, out_sum, out_norm, (unsigned short*)pieces, stream);
}

extern "C" int ts_act_pieces(int device, const void* x, const void* bias, int64_t rows, int32_t n, int kind, void* pieces, void* stream) {
    if (!x || !pieces) return fail(TS_ERR_INVALID, "NULL argument");
    if (((uintptr_t)bias & 15) != 0) return fail(TS_ERR_INVALID, "bias must be 16-byte aligned");
    if (rows < 0 || n < 4 || n % 4 || kind < 0 || kind > 2) return fail(TS_ERR_INVALID, "n = %d must be a multiple of 4, kind 0 / 1 / 2", n);
    if ((((uintptr_t)x) & 15) != 0 || (((uintptr_t)pieces) & 7) != 0) return fail(TS_ERR_INVALID, "x must be 16-byte, pieces 8-byte aligned");
    if (rows == 0) return TS_OK;
    TS_TRY(check_device(device));
    HIP_TRY(hipSetDevice(device));
    const int64_t total = rows * (n / 4);
    const unsigned grid = (unsigned)std::min<int64_t>((total + 255) / 256, 16384);
    hipStream_t st = (hipStream_t)stream;
    if (kind == 0) act_pieces_kernel<0><<<grid, 256, 0, st>>>((const float*)x, rows, n, (unsigned short*)pieces, (const float*)bias);
    else if (kind == 1) act_pieces_kernel<1><<<grid, 256, 0, st>>>((const float*)x, rows, n, (unsigned short*)pieces, (const float*)bias);
    else act_pieces_kernel<2><<<grid, 256, 0, st>>>((const float*)x, rows, n, (unsigned short*)pieces, (const float*)bias);
    HIP_TRY(hipGetLastError());
    return TS_OK;
}

static int gated_act_launch(int device, const void* gate_up, int64_t rows, int32_t inter, int dtype, bool gelu_tanh, void* out, void* stream) {
    if (!gate_up || !out) return fail(TS_ERR_INVALID, "NULL argument");
    if (dtype != TS_F32 && dtype != TS_BF16) return fail(TS_ERR_INVALID, "dtype %d", dtype);
    const int vec = dtype == TS_BF16 ? 8 : 4;
    if (rows < 0 || inter < vec || inter % vec) return fail(TS_ERR_INVALID, "inter = %d must be a multiple of %d", inter, vec);
    if ((((uintptr_t)gate_up | (uintptr_t)out) & 15) != 0) return fail(TS_ERR_INVALID, "buffers must be 16-byte aligned");
    if (rows == 0) return TS_OK;
    TS_TRY(check_device(device));
    HIP_TRY(hipSetDevice(device));
    const int64_t total = rows * (inter / vec);
    const unsigned grid = (unsigned)std::min<int64_t>((total + 255) / 256, 16384);
    hipStream_t st = (hipStream_t)stream;
    if (gelu_tanh) {
        if (dtype == TS_F32) geglu_kernel<0><<<grid, 256, 0, st>>>(gate_up, rows, inter, out);
        else geglu_kernel<1><<<grid, 256, 0, st>>>(gate_up, rows, inter, out);
    } else {
        if (dtype == TS_F32) swiglu_kernel<0><<<grid, 256, 0, st>>>(gate_up, rows, inter, out);
        else swiglu_kernel<1><<<grid, 256, 0, st>>>(gate_up, rows, inter, out);
    }
    HIP_TRY(hipGetLastError());
    return TS_OK;
}

extern "C" int ts_split_pieces(int device, const void* x, int64_t rows, int32_t k, int pattern, void* out, void* stream) {
    if (!x || !out) return fail(TS_ERR_INVALID, "NULL argument");
    if (rows < 0 || k < 4 || k % 4 || (pattern != 0 && pattern != 1)) return fail(TS_ERR_INVALID, "k = %d must be a multiple of 4, pattern 0 or 1", k);
    if ((((uintptr_t)x) & 15) != 0 || (((uintptr_t)out) & 7) != 0) return fail(TS_ERR_INVALID, "x must be 16-byte, out 8-byte aligned");
    if (rows == 0) return TS_OK;
    TS_TRY(check_device(device));
    HIP_TRY(hipSetDevice(device));
    const int64_t total = rows * (k / 4);
    const unsigned grid = (unsigned)std::min<int64_t>((total + 255) / 256, 16384);
    hipStream_t st = (hipStream_t)stream;
    if (pattern == 0) split3_kernel<0><<<grid, 256, 0, st>>>((const float*)x, rows, k, (unsigned short*)out);
    else split3_kernel<1><<<grid, 256, 0, st>>>((const float*)x, rows, k, (unsigned short*)out);
    HIP_TRY(hipGetLastError());
    return TS_OK;
}

extern "C" int ts_swiglu(int device, const void* gate_up, int64_t rows, int32_t inter, int dtype, void* out, void* stream) {
    return gated_act_launch(device, gate_up, rows, inter, dtype, false, out, stream);
}

extern "C" int ts_geglu(int device, const void* gate_up, int64_t rows, int32_t inter, int dtype, void* out, void* stream) {
    return gated_act_launch(device, gate_up, rows, inter, dtype, true, out, stream);
}
