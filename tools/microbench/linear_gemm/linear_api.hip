// Experiment, NOT part of libtsearch (profiles/r03_linear_gemm_ab.txt): the encoder's linear layers as a
// hand-written gfx950 kernel (kernels_linear.h) behind ts_linear_bf16, built as libts_linear.so for tools/linear_ab.py.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#define TS_OK 0
#define TS_ERR_INVALID (-1)
#define TS_ERR_HIP (-2)
#define TS_ERR_NODEVICE (-4)
#define TS_ERR_UNSUPPORTED (-5)
#define TS_ACT_NONE 0
#define TS_ACT_GELU 1
#include "kernels_linear.h"

using namespace ts;

static thread_local char g_err[400] = "";
extern "C" const char* ts_linear_last_error(void) { return g_err; }

static int lfail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define LHIP_TRY(expr)                                                                                     \
    do {                                                                                                   \
        hipError_t e_ = (expr);                                                                            \
        if (e_ != hipSuccess)                                                                              \
            return lfail(TS_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

template <int BM, int BN, int WM, int WN, int S>
static int launch_linear(const LinearArgs& a0, int act, hipStream_t st) {
    LinearArgs a = a0;
    a.mt = (a.M + BM - 1) / BM;
    a.nt = a.N / BN;
    constexpr int lds = linear_lds_bytes(BM, BN, S);
    static bool ready[2] = {false, false};                 // per process: the attribute is per function, not per device state we track
    if (act == 0) {
        if (!ready[0]) {
            LHIP_TRY(hipFuncSetAttribute((const void*)linear_bf16_kernel<BM, BN, WM, WN, S, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            ready[0] = true;
        }
        linear_bf16_kernel<BM, BN, WM, WN, S, 0><<<a.mt * a.nt, 256, lds, st>>>(a);
    } else if constexpr (BN != 288) {
        if (!ready[1]) {
            LHIP_TRY(hipFuncSetAttribute((const void*)linear_bf16_kernel<BM, BN, WM, WN, S, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            ready[1] = true;
        }
        linear_bf16_kernel<BM, BN, WM, WN, S, 1><<<a.mt * a.nt, 256, lds, st>>>(a);
    }
    LHIP_TRY(hipGetLastError());
    return TS_OK;
}

extern "C" int ts_linear_bf16(int device, const void* x, const void* w, const void* bias, void* y, int64_t m, int32_t n, int32_t k,
                              int act, int tile, void* stream) {
    if (!x || !w || !y) return lfail(TS_ERR_INVALID, "NULL argument");
    if (act != TS_ACT_NONE && act != TS_ACT_GELU) return lfail(TS_ERR_INVALID, "act %d", act);
    if (m < 0 || n <= 0 || k <= 0) return lfail(TS_ERR_INVALID, "m = %lld, n = %d, k = %d", (long long)m, n, k);
    if (k % 64 != 0 || n % 96 != 0)
        return lfail(TS_ERR_UNSUPPORTED, "k = %d must be a multiple of 64 and n = %d a multiple of 96", k, n);
    if (m * (int64_t)k * 2 >= (1ll << 31) || (int64_t)n * k * 2 >= (1ll << 31) || m * (int64_t)n >= (1ll << 31))
        return lfail(TS_ERR_UNSUPPORTED, "operands of 2 GiB or more");
    if ((((uintptr_t)x | (uintptr_t)w | (uintptr_t)y) & 15) != 0 || ((uintptr_t)bias & 7) != 0)
        return lfail(TS_ERR_INVALID, "x, w, y must be 16-byte aligned (bias: 8)");
    if (m == 0) return TS_OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return lfail(TS_ERR_NODEVICE, "no HIP device %d", device);
    LHIP_TRY(hipSetDevice(device));
    LinearArgs a{};
    a.x = (const unsigned char*)x;
    a.w = (const unsigned char*)w;
    a.bias = (const unsigned short*)bias;
    a.y = (unsigned short*)y;
    a.M = (int)m;
    a.N = n;
    a.K = k;
    hipStream_t st = (hipStream_t)stream;
    // tile width: the one whose grid wastes the least of its last round of 256 workgroups (time ~ rounds x width), the
    // wider one on a tie (fewer fragment reads per MFMA)
    static const int widths[3] = {288, 192, 96};
    int bn = tile;
    if (bn == 0) {
        const int64_t mt = (m + 255) / 256;
        int64_t best = -1;
        for (int wdt : widths) {
            if (n % wdt) continue;
            const int64_t rounds = (mt * (n / wdt) + 255) / 256;
            const int64_t cost = rounds * wdt;
            if (best < 0 || cost < best) { best = cost; bn = wdt; }
        }
    }
    if (bn == 288 && act != TS_ACT_NONE) bn = 96;          // the widest tile has no activation form (registers)
    switch (bn) {
        case 288: if (n % 288) break; return launch_linear<256, 288, 2, 2, 2>(a, TS_ACT_NONE, st);
        case 192: if (n % 192) break; return launch_linear<256, 192, 2, 2, 2>(a, act, st);
        case 96: return launch_linear<256, 96, 4, 1, 3>(a, act, st);
        default: break;
    }
    return lfail(TS_ERR_INVALID, "tile width %d does not divide n = %d (288, 192, 96 or 0 = choose)", tile, n);
}
