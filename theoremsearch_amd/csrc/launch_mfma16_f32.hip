// Launches of the 16x16x4 fp32 form of the matrix kernel (kernels_mfma16.h, F32 mode).
#include "host.h"
#include "kernels_mfma16.h"

// d = 768: two blocks, 128 per launch (one block when the batch has at most 64 queries)
template <int D, int NB>
static int launch_mfma16_f32(bool full_pass, int grid, hipStream_t st, const MfmaArgs& a) {
    constexpr int lds = Mfma16Dims<2 * D>::kLds + kMfma16StageBytes;
    static_assert(lds <= 160 * 1024, "DMA ring + staged candidates must fit the CU's LDS");
    constexpr bool kSparseToo = (D == 768 || D == 1024);        // d = 384 / 512: the full pass only (as launch_mfma16)
    static std::atomic<unsigned long long> attr_done{0};
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(attr_done.load(std::memory_order_acquire) & bit)) {
        HIP_TRY(hipFuncSetAttribute((const void*)mfma16_topk_kernel<D, NB, 0, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        if constexpr (kSparseToo)
            HIP_TRY(hipFuncSetAttribute((const void*)mfma16_topk_kernel<D, NB, 0, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_done.fetch_or(bit, std::memory_order_release);
    }
    if (!full_pass) {
        if constexpr (kSparseToo) mfma16_topk_kernel<D, NB, 0, true, true><<<grid, kMfmaThreads, lds, st>>>(a);
        else return fail(TS_ERR_INTERNAL, "no sparse level of the 16x16 kernel at d = %d", D);
    } else {
        mfma16_topk_kernel<D, NB, 0, false, true><<<grid, kMfmaThreads, lds, st>>>(a);
    }
    HIP_TRY(hipGetLastError());
    return TS_OK;
}


// fp32 rows: d = 1024 holds one block of 16 queries per wave, the other widths one or two
int launch_pass_mfma16_f32(int d, int nb, bool full_pass, int grid, hipStream_t st, const MfmaArgs& a) {
    if (d == 1024 && nb == 1) return launch_mfma16_f32<1024, 1>(full_pass, grid, st, a);
    if (d == 512) return nb == 1 ? launch_mfma16_f32<512, 1>(full_pass, grid, st, a) : launch_mfma16_f32<512, 2>(full_pass, grid, st, a);
    if (d == 384) return nb == 1 ? launch_mfma16_f32<384, 1>(full_pass, grid, st, a) : launch_mfma16_f32<384, 2>(full_pass, grid, st, a);
    if (d == 768) return nb == 1 ? launch_mfma16_f32<768, 1>(full_pass, grid, st, a) : launch_mfma16_f32<768, 2>(full_pass, grid, st, a);
    return fail(TS_ERR_INTERNAL, "no 16x16x4 kernel for d = %d with %d query blocks per wave", d, nb);
}
