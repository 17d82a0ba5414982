// Launches of the 32x32x16 bf16 matrix kernel (kernels_mfma.h) and of the 32x32x2 fp32 one (kernels_mfma_f32.h).
#include "host.h"
#include "kernels_mfma.h"
#include "kernels_mfma_f32.h"

template <int D, int GROUPS>
static int launch_mfma(bool full_pass, int variant, int grid, hipStream_t st, const MfmaArgs& a) {
    constexpr int lds = MfmaDims<D>::kLds;
    // the dynamic-LDS limit is set per function AND per device: one bit per device, per instantiation
    static std::atomic<unsigned long long> attr_done{0};
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(attr_done.load(std::memory_order_acquire) & bit)) {
        HIP_TRY(hipFuncSetAttribute((const void*)mfma_topk_kernel<D, GROUPS, 0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        HIP_TRY(hipFuncSetAttribute((const void*)mfma_topk_kernel<D, GROUPS, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
#ifdef TS_DIAG
        HIP_TRY(hipFuncSetAttribute((const void*)mfma_topk_kernel<D, GROUPS, 1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        HIP_TRY(hipFuncSetAttribute((const void*)mfma_topk_kernel<D, GROUPS, 2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        HIP_TRY(hipFuncSetAttribute((const void*)mfma_topk_kernel<D, GROUPS, 3, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        HIP_TRY(hipFuncSetAttribute((const void*)mfma_topk_kernel<D, GROUPS, 4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        HIP_TRY(hipFuncSetAttribute((const void*)mfma_topk_kernel<D, GROUPS, 5, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        HIP_TRY(hipFuncSetAttribute((const void*)mfma_topk_kernel<D, GROUPS, 6, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        HIP_TRY(hipFuncSetAttribute((const void*)mfma_topk_kernel<D, GROUPS, 7, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
#endif
        attr_done.fetch_or(bit, std::memory_order_release);
    }
    if (!full_pass) mfma_topk_kernel<D, GROUPS, 0, true><<<grid, kMfmaThreads, lds, st>>>(a);
#ifdef TS_DIAG
    else if (variant == 1) mfma_topk_kernel<D, GROUPS, 1, false><<<grid, kMfmaThreads, lds, st>>>(a);
    else if (variant == 2) mfma_topk_kernel<D, GROUPS, 2, false><<<grid, kMfmaThreads, lds, st>>>(a);
    else if (variant == 3) mfma_topk_kernel<D, GROUPS, 3, false><<<grid, kMfmaThreads, lds, st>>>(a);
    else if (variant == 4) mfma_topk_kernel<D, GROUPS, 4, false><<<grid, kMfmaThreads, lds, st>>>(a);
    else if (variant == 5) mfma_topk_kernel<D, GROUPS, 5, false><<<grid, kMfmaThreads, lds, st>>>(a);
    else if (variant == 6) mfma_topk_kernel<D, GROUPS, 6, false><<<grid, kMfmaThreads, lds, st>>>(a);
    else if (variant == 7) mfma_topk_kernel<D, GROUPS, 7, false><<<grid, kMfmaThreads, lds, st>>>(a);
#endif
    else mfma_topk_kernel<D, GROUPS, 0, false><<<grid, kMfmaThreads, lds, st>>>(a);
    (void)variant;
    HIP_TRY(hipGetLastError());
    return TS_OK;
}

static int launch_mfma_f32(bool full_pass, int variant, int grid, hipStream_t st, const MfmaArgs& a) {
    constexpr int lds = MfmaF32Dims::kLds;
    static std::atomic<unsigned long long> attr_done{0};
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(attr_done.load(std::memory_order_acquire) & bit)) {
        HIP_TRY(hipFuncSetAttribute((const void*)mfma_f32_topk_kernel<0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        HIP_TRY(hipFuncSetAttribute((const void*)mfma_f32_topk_kernel<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
#ifdef TS_DIAG
        HIP_TRY(hipFuncSetAttribute((const void*)mfma_f32_topk_kernel<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
#endif
        attr_done.fetch_or(bit, std::memory_order_release);
    }
    if (!full_pass) mfma_f32_topk_kernel<0, true><<<grid, kMfmaThreads, lds, st>>>(a);
#ifdef TS_DIAG
    else if (variant == 1) mfma_f32_topk_kernel<1, false><<<grid, kMfmaThreads, lds, st>>>(a);
#endif
    else mfma_f32_topk_kernel<0, false><<<grid, kMfmaThreads, lds, st>>>(a);
    (void)variant;
    HIP_TRY(hipGetLastError());
    return TS_OK;
}


// groups = 128-query groups per launch (d = 1024: one)
int launch_pass_mfma32(int d, int groups, bool full_pass, int variant, int grid, hipStream_t st, const MfmaArgs& a) {
    if (d == 1024) return launch_mfma<1024, 1>(full_pass, variant, grid, st, a);
    if (d == 512) return groups == 1 ? launch_mfma<512, 1>(full_pass, variant, grid, st, a) : launch_mfma<512, 2>(full_pass, variant, grid, st, a);
    if (d == 384) return groups == 1 ? launch_mfma<384, 1>(full_pass, variant, grid, st, a) : launch_mfma<384, 2>(full_pass, variant, grid, st, a);
    if (d == 768) return groups == 1 ? launch_mfma<768, 1>(full_pass, variant, grid, st, a) : launch_mfma<768, 2>(full_pass, variant, grid, st, a);
    return fail(TS_ERR_INTERNAL, "no 32x32x16 kernel for d = %d", d);
}

int launch_pass_mfma32_f32(bool full_pass, int variant, int grid, hipStream_t st, const MfmaArgs& a) {
    return launch_mfma_f32(full_pass, variant, grid, st, a);
}
