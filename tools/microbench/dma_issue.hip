// Microbenchmark: what one LDS-DMA piece costs the issuing wave inside an MFMA + ds_read loop (one wave per SIMD),
// for the two instruction forms:  global_load_lds_dwordx4 (64-bit per-lane address)  and
// buffer_load_dwordx4 ... offen lds (SGPR descriptor + 32-bit per-lane offset + SGPR offset).
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/dma_issue.hip -o /tmp/dma_issue && /tmp/dma_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ void mfma(f32x16& acc, const bf16x8& a, const bf16x8& b) {
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void dma_global(const void* src, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off nt" : : "v"(src), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void dma_buffer(unsigned voff, i32x4 rsrc, unsigned soff, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen nt lds" : : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_dst) : "memory");
}

// FORM 0: no DMA; 1: global form; 2: buffer form.  PER = MFMAs per DMA piece.
template <int FORM, int PER, int GROUPS>
__global__ void __launch_bounds__(256, 1) loop_kernel(const unsigned char* src, size_t span, int iters, unsigned long long* out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 32768 / 4; i += 256) ((unsigned*)smem)[i] = i * 2654435761u;
    __syncthreads();
    f32x16 acc[GROUPS];
    bf16x8 q[GROUPS];
#pragma unroll
    for (int g = 0; g < GROUPS; ++g) {
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[g][j] = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) q[g][j] = (short)(0x3c00 + lane * 7 + j + g);
        asm volatile("" : "+v"(q[g]));
    }
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem + 65536 + wave * 1024;
    const unsigned char* base = src + (size_t)blockIdx.x * 65536;
    const unsigned voff = (unsigned)(lane * 16 + wave * 1024);
    const unsigned char* gptr = base + voff;
    i32x4 rsrc;
    rsrc[0] = (int)(unsigned)(unsigned long long)base;
    rsrc[1] = (int)(((unsigned long long)base >> 32) & 0xFFFF);
    rsrc[2] = 0x7FFFFFFF;
    rsrc[3] = 0x00020000;
    bf16x8 af[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) af[s] = *(const bf16x8*)(smem + s * 4096 + lane * 16);
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    unsigned piece = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 24; ++s) {
#pragma unroll
            for (int g = 0; g < GROUPS; ++g) mfma(acc[g], af[s & 3], q[g]);
            af[s & 3] = *(const bf16x8*)(smem + ((s + 4) % 8) * 4096 + lane * 16);
            if (FORM != 0 && s % PER == 1) {
                const unsigned off = (piece & 15) * 4096;  // stays inside this workgroup's 64 KiB (L2 resident)
                if (FORM == 1) dma_global(gptr + off, lds0 + (piece & 7) * 4096);
                else dma_buffer(voff, rsrc, off, lds0 + (piece & 7) * 4096);
                ++piece;
            }
        }
        if (FORM != 0) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float sum = 0.f;
#pragma unroll
    for (int g = 0; g < GROUPS; ++g) sum += acc[g][0] + acc[g][7];
    if (sum == 12345.678f) out[1] = 1;  // keep the accumulators alive
    if (threadIdx.x == 0) out[2 + blockIdx.x] = t1 - t0;
}

template <int FORM, int PER, int GROUPS>
static double run(const unsigned char* src, size_t span, unsigned long long* out, int grid, int iters) {
    auto k = loop_kernel<FORM, PER, GROUPS>;
    CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + 32768));
    std::vector<unsigned long long> h(2 + grid);
    double best = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
        k<<<grid, 256, 65536 + 32768>>>(src, span, iters, out);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost));
        double tot = 0;
        for (int i = 0; i < grid; ++i) tot += (double)h[2 + i];
        best = std::min(best, tot / grid / iters / 24.0);
    }
    return best;  // cycles per k-step
}

int main() {
    const int grid = 256, iters = 2000;
    const size_t span = (size_t)grid * 65536 + 65536;
    unsigned char* src;
    unsigned long long* out;
    CHECK(hipMalloc(&src, span));
    CHECK(hipMemset(src, 1, span));
    CHECK(hipMalloc(&out, (2 + grid) * 8));
    CHECK(hipMemset(out, 0, (2 + grid) * 8));
    printf("cycles per k-step (s_memtime), %d workgroups x 4 waves, 24 k-steps x %d iterations\n", grid, iters);
    printf("groups=2 (2 MFMA per k-step):  none %.1f | global/4 %.1f  buffer/4 %.1f | global/2 %.1f  buffer/2 %.1f\n",
           run<0, 4, 2>(src, span, out, grid, iters), run<1, 4, 2>(src, span, out, grid, iters), run<2, 4, 2>(src, span, out, grid, iters),
           run<1, 2, 2>(src, span, out, grid, iters), run<2, 2, 2>(src, span, out, grid, iters));
    printf("groups=1 (1 MFMA per k-step):  none %.1f | global/4 %.1f  buffer/4 %.1f | global/2 %.1f  buffer/2 %.1f\n",
           run<0, 4, 1>(src, span, out, grid, iters), run<1, 4, 1>(src, span, out, grid, iters), run<2, 4, 1>(src, span, out, grid, iters),
           run<1, 2, 1>(src, span, out, grid, iters), run<2, 2, 1>(src, span, out, grid, iters));
    return 0;
}
