// Launches of the 16x16x32 bf16 matrix kernel (kernels_mfma16.h): one instantiation per (width, query blocks per wave).
#include "host.h"
#include "kernels_mfma16.h"

template <int D, int NB>
static int launch_mfma16(bool full_pass, int variant, int grid, hipStream_t st, const MfmaArgs& a) {
    constexpr int lds = Mfma16Dims<D>::kLds + kMfma16StageBytes;
    static_assert(lds <= 160 * 1024, "DMA ring + staged candidates must fit the CU's LDS");
    // d = 384 / 512: the full pass only (the threshold sample has its own kernel; the thresholded sparse levels of the
    // guaranteed chain run the 32x32 kernel for these widths: use_shape16)
    constexpr bool kSparseToo = (D == 768 || D == 1024);
#ifdef TS_DIAG
    constexpr bool kDiag = (D == 768 && NB == 4);     // the timing-only variants exist for the headline shape only
#else
    constexpr bool kDiag = false;                     // ... and in the diagnostic build only (make diag)
#endif
    static std::atomic<unsigned long long> attr_done{0};
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(attr_done.load(std::memory_order_acquire) & bit)) {
        HIP_TRY(hipFuncSetAttribute((const void*)mfma16_topk_kernel<D, NB, 0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        if constexpr (kSparseToo)
            HIP_TRY(hipFuncSetAttribute((const void*)mfma16_topk_kernel<D, NB, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        if constexpr (kDiag) {
            HIP_TRY(hipFuncSetAttribute((const void*)mfma16_topk_kernel<D, NB, 1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            HIP_TRY(hipFuncSetAttribute((const void*)mfma16_topk_kernel<D, NB, 2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            HIP_TRY(hipFuncSetAttribute((const void*)mfma16_topk_kernel<D, NB, 3, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            HIP_TRY(hipFuncSetAttribute((const void*)mfma16_topk_kernel<D, NB, 4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            HIP_TRY(hipFuncSetAttribute((const void*)mfma16_topk_kernel<D, NB, 7, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            HIP_TRY(hipFuncSetAttribute((const void*)mfma16_topk_kernel<D, NB, 5, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            HIP_TRY(hipFuncSetAttribute((const void*)mfma16_topk_kernel<D, NB, 6, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        }
        attr_done.fetch_or(bit, std::memory_order_release);
    }
    if (!full_pass) {
        if constexpr (kSparseToo) mfma16_topk_kernel<D, NB, 0, true><<<grid, kMfmaThreads, lds, st>>>(a);
        else return fail(TS_ERR_INTERNAL, "no sparse level of the 16x16 kernel at d = %d", D);
    } else if (kDiag && variant != 0) {
        if constexpr (kDiag) {
            if (variant == 1) mfma16_topk_kernel<D, NB, 1, false><<<grid, kMfmaThreads, lds, st>>>(a);
            else if (variant == 2) mfma16_topk_kernel<D, NB, 2, false><<<grid, kMfmaThreads, lds, st>>>(a);
            else if (variant == 3) mfma16_topk_kernel<D, NB, 3, false><<<grid, kMfmaThreads, lds, st>>>(a);
            else if (variant == 4) mfma16_topk_kernel<D, NB, 4, false><<<grid, kMfmaThreads, lds, st>>>(a);
            else if (variant == 7) mfma16_topk_kernel<D, NB, 7, false><<<grid, kMfmaThreads, lds, st>>>(a);
            else if (variant == 5) mfma16_topk_kernel<D, NB, 5, false><<<grid, kMfmaThreads, lds, st>>>(a);
            else if (variant == 6) mfma16_topk_kernel<D, NB, 6, false><<<grid, kMfmaThreads, lds, st>>>(a);
            else mfma16_topk_kernel<D, NB, 0, false><<<grid, kMfmaThreads, lds, st>>>(a);
        }
    } else {
        mfma16_topk_kernel<D, NB, 0, false><<<grid, kMfmaThreads, lds, st>>>(a);
    }
    HIP_TRY(hipGetLastError());
    return TS_OK;
}


// The paired full pass of d = 1024 (MfmaArgs::pair): 128 queries per workgroup, two workgroups per tile range.
static int launch_mfma16_pair(int variant, int grid, hipStream_t st, const MfmaArgs& a) {
    constexpr int lds = Mfma16Dims<1024>::kLds + kMfma16StageBytes + kMfma16PaceBytes;
    static_assert(lds <= 160 * 1024, "DMA ring + staged candidates + the pair's word must fit the CU's LDS");
    static std::atomic<unsigned long long> attr_done{0};
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(attr_done.load(std::memory_order_acquire) & bit)) {
        HIP_TRY(hipFuncSetAttribute((const void*)mfma16_topk_kernel<1024, 2, 0, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_done.fetch_or(bit, std::memory_order_release);
    }
    if (grid % 16 != 0) return fail(TS_ERR_INTERNAL, "the paired pass needs a grid of whole groups of 16 workgroups, not %d", grid);
    if (a.pair == 2) {
        // the k-split form: 2 x 2 waves (query column x k half), eight ring slots, one 16 KB exchange buffer for the partial sums
        constexpr int lds_k = MfmaDims<1024, MfmaGeomKsplit<1024>>::kLds + kMfma16StageBytes + kMfma16PaceBytes + 16384;
        static_assert(lds_k <= 160 * 1024, "ring + staged candidates + the pair's word + the exchange buffers must fit the CU's LDS");
        static std::atomic<unsigned long long> attr_k{0};
        if (!(attr_k.load(std::memory_order_acquire) & bit)) {
            HIP_TRY(hipFuncSetAttribute((const void*)mfma16_topk_kernel<1024, 4, 0, false, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_k));
            attr_k.fetch_or(bit, std::memory_order_release);
        }
#ifdef TS_DIAG
        if (variant == 1 || variant == 2 || variant == 7) {
            HIP_TRY(hipFuncSetAttribute((const void*)mfma16_topk_kernel<1024, 4, 1, false, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_k));
            HIP_TRY(hipFuncSetAttribute((const void*)mfma16_topk_kernel<1024, 4, 2, false, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_k));
            HIP_TRY(hipFuncSetAttribute((const void*)mfma16_topk_kernel<1024, 4, 7, false, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_k));
            if (variant == 1) mfma16_topk_kernel<1024, 4, 1, false, false, true, true><<<grid, kMfmaThreads, lds_k, st>>>(a);
            else if (variant == 2) mfma16_topk_kernel<1024, 4, 2, false, false, true, true><<<grid, kMfmaThreads, lds_k, st>>>(a);
            else mfma16_topk_kernel<1024, 4, 7, false, false, true, true><<<grid, kMfmaThreads, lds_k, st>>>(a);
            HIP_TRY(hipGetLastError());
            return TS_OK;
        }
#endif
        mfma16_topk_kernel<1024, 4, 0, false, false, true, true><<<grid, kMfmaThreads, lds_k, st>>>(a);
        HIP_TRY(hipGetLastError());
        return TS_OK;
    }
#ifdef TS_DIAG
    if (variant == 1 || variant == 2 || variant == 7) {
        HIP_TRY(hipFuncSetAttribute((const void*)mfma16_topk_kernel<1024, 2, 1, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        HIP_TRY(hipFuncSetAttribute((const void*)mfma16_topk_kernel<1024, 2, 2, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        HIP_TRY(hipFuncSetAttribute((const void*)mfma16_topk_kernel<1024, 2, 7, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        if (variant == 1) mfma16_topk_kernel<1024, 2, 1, false, false, true><<<grid, kMfmaThreads, lds, st>>>(a);
        else if (variant == 2) mfma16_topk_kernel<1024, 2, 2, false, false, true><<<grid, kMfmaThreads, lds, st>>>(a);
        else mfma16_topk_kernel<1024, 2, 7, false, false, true><<<grid, kMfmaThreads, lds, st>>>(a);
        HIP_TRY(hipGetLastError());
        return TS_OK;
    }
#endif
    mfma16_topk_kernel<1024, 2, 0, false, false, true><<<grid, kMfmaThreads, lds, st>>>(a);
    HIP_TRY(hipGetLastError());
    return TS_OK;
}

// d = 384 / 512 / 768 / 1024, nb = query blocks of 16 per wave (64 * nb queries per launch; d = 1024: at most 3)
int launch_pass_mfma16(int d, int nb, bool full_pass, int variant, int grid, hipStream_t st, const MfmaArgs& a) {
    if (a.pair) {
        if (d != 1024 || nb != 2 || !full_pass) return fail(TS_ERR_INTERNAL, "paired pass asked for d = %d, %d blocks per wave", d, nb);   // (a.pair == 2: the k-split form, same queries per workgroup)
        return launch_mfma16_pair(variant, grid, st, a);
    }
#define TS_NB_SWITCH(D_)                                                          \
    switch (nb) {                                                                 \
        case 1: return launch_mfma16<D_, 1>(full_pass, variant, grid, st, a);     \
        case 2: return launch_mfma16<D_, 2>(full_pass, variant, grid, st, a);     \
        case 3: return launch_mfma16<D_, 3>(full_pass, variant, grid, st, a);     \
        case 4: if constexpr (D_ != 1024) return launch_mfma16<D_, 4>(full_pass, variant, grid, st, a); break; \
        default: break;                                                           \
    }                                                                             \
    break
    switch (d) {
        case 384: TS_NB_SWITCH(384);
        case 512: TS_NB_SWITCH(512);
        case 768: TS_NB_SWITCH(768);
        case 1024: TS_NB_SWITCH(1024);
        default: break;
    }
#undef TS_NB_SWITCH
    return fail(TS_ERR_INTERNAL, "no 16x16x32 kernel for d = %d with %d query blocks per wave", d, nb);
}
