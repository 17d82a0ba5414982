// What the full pass of configs[2] (10M x 768 bf16, 256 queries) could reach on THIS device, measured, not quoted:
//
//   leg 0  stream + matrix   the product kernel's tile loop without its epilogue and candidate path: the same 256 workgroups
//                            x 4 waves, the same LDS-DMA ring, the same ds_read_b128 and v_mfma_f32_16x16x32_bf16 count per
//                            unit (mfma16_topk_kernel<768, 4, VARIANT 1>)
//   leg 1  stream only       the DMA ring alone (VARIANT 2): what HBM -> LDS delivers with nothing else on the chip
//   leg 2  matrix + LDS      the MFMAs and the fragment reads without the DMA stream (VARIANT 7)
//   leg 3  bare matrix       the same number of v_mfma_f32_16x16x32_bf16 per wave with every operand in registers: the bf16
//                            matrix rate this device sustains on random operands under its power cap - the number
//                            MI355X_MICROARCH.md "DVFS give-back" quotes as 1,247 TF/s for a GEMM on another device
//
// Built two ways from this one file:
//   libts_ceiling.so   ts_ceiling_run(): the legs over a corpus that is ALREADY resident (bench.py calls it on its own
//                      corpus after the timed region: same data, same device, same run)
//   mfma_stream_ceiling (-DTS_CEILING_MAIN)  standalone: generates its own random bf16 corpus, interleaved rounds, JSON out
//
// Measurement infrastructure: nothing in libtsearch.so depends on it.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../theoremsearch_amd/csrc/kernels_mfma16.h"

using namespace ts;

namespace {

constexpr int kD = 768, kNB = 4;
constexpr int kLds = MfmaDims<kD>::kLds + kMfma16StageBytes;

// bare matrix loop: per tile of 32 rows 24 k-steps x (2 row blocks x 4 query blocks) MFMAs, accumulators restarted per
// tile as in the product; query fragments as in the product (96 per wave), corpus fragments from a ring of 8 registers
// quadruples that hold random bf16 values
__global__ void __launch_bounds__(kMfmaThreads, 1) bare_mfma_kernel(const unsigned short* q, const unsigned short* corpus,
                                                                     int64_t ntiles, float* sink) {
    constexpr int kSteps = kD / 32;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r16 = lane & 15, kq = lane >> 4;
    const int G = gridDim.x;
    const int64_t t0 = (ntiles * (int64_t)blockIdx.x) / G;
    const int nt = (int)((ntiles * (int64_t)(blockIdx.x + 1)) / G - t0);
    bf16x8 qf[kNB * kSteps];
#pragma unroll
    for (int b = 0; b < kNB; ++b) {
        const int qid = (b * 4 + wave) * 16 + r16;
        const bf16x8* pq = (const bf16x8*)(q + (int64_t)qid * kD + 8 * kq);
#pragma unroll
        for (int ks = 0; ks < kSteps; ++ks) qf[b * kSteps + ks] = pq[4 * ks];
    }
    bf16x8 af[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
        af[i] = *(const bf16x8*)(corpus + ((int64_t)(blockIdx.x * 64 + lane) * kD + (wave * 8 + i) * 8));
#pragma unroll
    for (int i = 0; i < kNB * kSteps; ++i) asm volatile("" : "+v"(qf[i]));
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(af[i]));
    f32x4 acc[2][kNB];
    float keep = 0.0f;
    for (int t = 0; t < nt; ++t) {
#pragma unroll
        for (int ks = 0; ks < kSteps; ++ks) {
#pragma unroll
            for (int b = 0; b < kNB; ++b)
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) {
                    const f32x4 c0 = {0.f, 0.f, 0.f, 0.f};
                    acc[rb][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[2 * (ks & 3) + rb], qf[b * kSteps + ks],
                                                                          ks == 0 ? c0 : acc[rb][b], 0, 0, 0);
                }
        }
#pragma unroll
        for (int b = 0; b < kNB; ++b) asm volatile("" ::"a"(acc[0][b]), "a"(acc[1][b]));
        if (t == nt - 1) keep = acc[0][0][0] + acc[1][kNB - 1][3];
    }
    if (sink && keep == 123456.789f) sink[blockIdx.x] = keep;      // never true: keeps the chain alive
}

// normal-ish random bf16 rows of unit length in expectation (what the bench's corpus looks like to the matrix pipe)
__global__ void fill_random_bf16(unsigned short* dst, int64_t count, unsigned long long seed) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        unsigned long long x = (unsigned long long)i * 0x9E3779B97F4A7C15ull + seed;
        x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;
        const float u1 = ((unsigned)(x >> 40) + 1.0f) * (1.0f / 16777217.0f);
        const float u2 = (unsigned)((x >> 8) & 0xFFFFFF) * (1.0f / 16777216.0f);
        const float g = sqrtf(-2.0f * __logf(u1)) * __cosf(6.2831853f * u2) * 0.0360844f;   // 1 / sqrt(768)
        dst[i] = f32_to_bf16(g);
    }
}

struct Scratch {
    float* thr = nullptr;
    u32* count = nullptr;
    int* fb = nullptr;
    unsigned long long* stat = nullptr;
    float* sink = nullptr;
    bool attr = false;
};

int run_leg(int leg, const unsigned short* corpus, int64_t rows, const unsigned short* q, int grid, Scratch& s, hipStream_t st) {
    MfmaArgs a;
    memset(&a, 0, sizeof(a));
    a.corpus = corpus;
    a.n = rows;
    a.ntiles = rows / kTileRows;
    a.tile_stride = 1;
    a.run = 1;
    a.q = q;
    a.thr = s.thr;
    a.count = s.count;
    a.cap = 0;
    a.nq = 256;
    a.first_level = 1;
    a.nq_real = 256;
    a.fb_count = s.fb;
    if (!s.attr) {
        if (hipFuncSetAttribute((const void*)mfma16_topk_kernel<kD, kNB, 1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds) != hipSuccess) return -1;
        if (hipFuncSetAttribute((const void*)mfma16_topk_kernel<kD, kNB, 2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds) != hipSuccess) return -1;
        if (hipFuncSetAttribute((const void*)mfma16_topk_kernel<kD, kNB, 7, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds) != hipSuccess) return -1;
        s.attr = true;
    }
    if (leg == 0) mfma16_topk_kernel<kD, kNB, 1, false><<<grid, kMfmaThreads, kLds, st>>>(a);
    else if (leg == 1) mfma16_topk_kernel<kD, kNB, 2, false><<<grid, kMfmaThreads, kLds, st>>>(a);
    else if (leg == 2) mfma16_topk_kernel<kD, kNB, 7, false><<<grid, kMfmaThreads, kLds, st>>>(a);
    else bare_mfma_kernel<<<grid, kMfmaThreads, 0, st>>>(q, corpus, a.ntiles, s.sink);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace

// Average milliseconds per launch of the four legs over `rows` resident bf16 rows of 768 (rows a multiple of 32, at least
// 64 * grid rows; queries: 256 x 768 bf16), `rounds` interleaved rounds of `launches` launches each; out_ms[4] as listed at
// the top.  Returns 0, or -1 with nothing written.
extern "C" int ts_ceiling_run(int device, const void* corpus_dev, int64_t rows, const void* queries_dev, int rounds, int launches,
                              double* out_ms, void* stream) {
    if (!corpus_dev || !queries_dev || !out_ms || rows < 32 || rows % 32 || rounds < 1 || launches < 1) return -1;
    if (hipSetDevice(device) != hipSuccess) return -1;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return -1;
    const int grid = prop.multiProcessorCount;
    if (rows < 64ll * grid) return -1;
    hipStream_t st = (hipStream_t)stream;
    Scratch s;
    bool ok = hipMalloc((void**)&s.thr, 256 * 4) == hipSuccess && hipMalloc((void**)&s.count, 256 * 4) == hipSuccess &&
              hipMalloc((void**)&s.fb, 16) == hipSuccess && hipMalloc((void**)&s.stat, 16) == hipSuccess &&
              hipMalloc((void**)&s.sink, 4096 * 4) == hipSuccess;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ok = ok && hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess;
    if (ok) ok = hipMemsetAsync(s.count, 0, 256 * 4, st) == hipSuccess;
    double sum[4] = {0, 0, 0, 0};
    for (int leg = 0; ok && leg < 4; ++leg) ok = run_leg(leg, (const unsigned short*)corpus_dev, rows, (const unsigned short*)queries_dev, grid, s, st) == 0;   // warm-up
    for (int r = 0; ok && r < rounds; ++r)
        for (int leg = 0; ok && leg < 4; ++leg) {
            ok = hipEventRecord(e0, st) == hipSuccess;
            for (int i = 0; ok && i < launches; ++i)
                ok = run_leg(leg, (const unsigned short*)corpus_dev, rows, (const unsigned short*)queries_dev, grid, s, st) == 0;
            ok = ok && hipEventRecord(e1, st) == hipSuccess && hipEventSynchronize(e1) == hipSuccess;
            float ms = 0.f;
            ok = ok && hipEventElapsedTime(&ms, e0, e1) == hipSuccess;
            sum[leg] += ms;
        }
    if (ok)
        for (int leg = 0; leg < 4; ++leg) out_ms[leg] = sum[leg] / ((double)rounds * launches);
    if (e0) hipEventDestroy(e0);
    if (e1) hipEventDestroy(e1);
    hipFree(s.thr); hipFree(s.count); hipFree(s.fb); hipFree(s.stat); hipFree(s.sink);
    return ok ? 0 : -1;
}

#ifdef TS_CEILING_MAIN
// mfma_stream_ceiling [rows] [rounds] [launches] [seconds]: own random corpus; `seconds` > 0 keeps the product-like leg 0
// running back to back for that long first (settles the clock; sample rocm-smi meanwhile) and reports its sustained time
int main(int argc, char** argv) {
    const int64_t rows = argc > 1 ? atoll(argv[1]) / 32 * 32 : 10000000;
    const int rounds = argc > 2 ? atoi(argv[2]) : 5, launches = argc > 3 ? atoi(argv[3]) : 20;
    const double seconds = argc > 4 ? atof(argv[4]) : 0.0;
    unsigned short *corpus = nullptr, *q = nullptr;
    if (hipMalloc((void**)&corpus, (size_t)rows * kD * 2) != hipSuccess || hipMalloc((void**)&q, 256 * kD * 2) != hipSuccess) {
        fprintf(stderr, "allocation failed\n");
        return 1;
    }
    fill_random_bf16<<<4096, 256>>>(corpus, rows * kD, 1234);
    fill_random_bf16<<<64, 256>>>(q, 256 * kD, 5678);
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    double sustained_ms = 0.0;
    if (seconds > 0) {
        double one[4];
        if (ts_ceiling_run(0, corpus, rows, q, 1, 5, one, nullptr) != 0) return 1;
        const int n = std::max(1, (int)(seconds * 1e3 / one[0]));
        Scratch s;
        hipMalloc((void**)&s.thr, 1024); hipMalloc((void**)&s.count, 1024); hipMalloc((void**)&s.fb, 16); hipMalloc((void**)&s.stat, 16);
        hipMemset(s.count, 0, 1024);
        hipDeviceProp_t prop;
        hipGetDeviceProperties(&prop, 0);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, nullptr);
        for (int i = 0; i < n; ++i) run_leg(0, corpus, rows, q, prop.multiProcessorCount, s, nullptr);
        hipEventRecord(e1, nullptr);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        sustained_ms = ms / n;
    }
    double ms[4];
    if (ts_ceiling_run(0, corpus, rows, q, rounds, launches, ms, nullptr) != 0) {
        fprintf(stderr, "ts_ceiling_run failed: %s\n", hipGetErrorString(hipGetLastError()));
        return 1;
    }
    const double bytes = (double)rows * kD * 2, flops = 2.0 * 256 * rows * kD;
    printf("{\"rows\": %lld, \"rounds\": %d, \"launches_per_round\": %d, "
           "\"stream_plus_mfma_ms\": %.4f, \"stream_plus_mfma_hbm_frac\": %.4f, \"stream_plus_mfma_tflops\": %.1f, "
           "\"stream_only_ms\": %.4f, \"stream_only_tbs\": %.3f, "
           "\"mfma_lds_only_ms\": %.4f, \"mfma_lds_only_tflops\": %.1f, "
           "\"bare_mfma_ms\": %.4f, \"measured_gemm_tflops\": %.1f, \"sustained_stream_plus_mfma_ms\": %.4f}\n",
           (long long)rows, rounds, launches, ms[0], bytes / (ms[0] * 1e-3) / 8e12, flops / (ms[0] * 1e-3) / 1e12, ms[1],
           bytes / (ms[1] * 1e-3) / 1e12, ms[2], flops / (ms[2] * 1e-3) / 1e12, ms[3], flops / (ms[3] * 1e-3) / 1e12, sustained_ms);
    return 0;
}
#endif
