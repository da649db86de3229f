// Micro-benchmark (tools only, not part of the product): sustained L2 -> LDS (LDS-DMA) and L2 -> VGPR streaming rate per CU
// as a function of waves per CU and bytes in flight — the operand path that bounds emage_gemm at M = 4096.
// Every block sweeps the same `span` bytes (L2-resident after the first touch, like a GEMM's shared W / A panels).
//   usage: l2_stream [span_kb=2048] [iters=200]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }

// mode 0: buffer_load_dwordx4 ... lds, G wave-instructions (1 KiB each) per stage, DEPTH stages in flight per wave
// mode 1: buffer_load_dwordx4 -> VGPR, same shape
template <int MODE, int G, int DEPTH>
__global__ __launch_bounds__(1024) void stream_kernel(const unsigned char* __restrict__ src, unsigned span, int iters, unsigned* sink) {
    extern __shared__ __attribute__((aligned(128))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = blockDim.x >> 6;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, span, 0x00020000);
    unsigned char* my = smem + wave * (G * DEPTH * 1024);
    const unsigned voff = lane * 16;
    // this wave's stream: stage s covers bytes ((blockIdx*nw + wave)*977 + s*nw... ) — stride the waves apart, wrap in span
    unsigned pos = ((blockIdx.x * nw + wave) * 7919u * 1024u) % span;
    u32x4 acc = {0, 0, 0, 0};
    u32x4 regs[MODE == 1 ? G * DEPTH : 1];
    auto issue = [&](int slot) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            if constexpr (MODE == 0) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(my + (slot * G + g) * 1024), 16, (int)voff, (int)pos, 0, 0);
            } else {
                regs[slot * G + g] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, (int)pos, 0);
            }
            pos += 1024u * nw * 3u;                 // neighbouring waves interleave lines; stay inside the span
            if (pos >= span) pos -= span;
        }
    };
#pragma unroll
    for (int s = 0; s < DEPTH - 1; ++s) issue(s);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < DEPTH; ++s) {
            issue((s + DEPTH - 1) % DEPTH);
            wait_vmcnt<G*(DEPTH - 1)>();
            if constexpr (MODE == 1) {
#pragma unroll
                for (int g = 0; g < G; ++g) acc ^= regs[s * G + g];
            }
        }
    }
    wait_vmcnt<0>();
    if (acc[0] == 0x12345678u && acc[1] == 1u) sink[0] = acc[2] ^ acc[3];
    if constexpr (MODE == 0) { if (((unsigned*)smem)[threadIdx.x] == 0x9abcdef1u) sink[1] = 1; }
}

template <int MODE, int G, int DEPTH>
void run(const unsigned char* src, unsigned span, int iters, int blocks, int waves, unsigned* sink, const char* tag) {
    const size_t lds = MODE == 0 ? (size_t)waves * G * DEPTH * 1024 : 1024;
    if (lds > 160 * 1024) return;
    hipFuncSetAttribute((const void*)stream_kernel<MODE, G, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((stream_kernel<MODE, G, DEPTH>), dim3(blocks), dim3(waves * 64), lds, 0, src, span, 4, sink);   // warm L2
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((stream_kernel<MODE, G, DEPTH>), dim3(blocks), dim3(waves * 64), lds, 0, src, span, iters, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)blocks * waves * ((double)iters * DEPTH + DEPTH - 1) * G * 1024.0;
    printf("%-8s blocks %4d waves/blk %2d G %d depth %d inflight/CU %6.1f KiB : %8.3f ms  %7.2f TB/s  %6.1f GB/s per block\n", tag, blocks, waves, G, DEPTH,
           (double)(blocks > 256 ? blocks / 256 : 1) * waves * G * (DEPTH - 1), ms, bytes / ms / 1e9, bytes / ms / 1e6 / blocks);
}

int main(int argc, char** argv) {
    const unsigned span = (argc > 1 ? atoi(argv[1]) : 2048) * 1024u;
    const int iters = argc > 2 ? atoi(argv[2]) : 200;
    unsigned char* src; unsigned* sink;
    hipMalloc(&src, span); hipMalloc(&sink, 64);
    hipMemset(src, 1, span); hipMemset(sink, 0, 64);
    printf("span %u KiB (shared by every block), iters %d\n", span / 1024, iters);
    for (int blocks : {256, 512}) {
        run<0, 4, 2>(src, span, iters, blocks, 4, sink, "lds-dma");
        run<0, 4, 4>(src, span, iters, blocks, 4, sink, "lds-dma");
        run<0, 4, 8>(src, span, iters, blocks, 4, sink, "lds-dma");
        run<0, 4, 2>(src, span, iters, blocks, 8, sink, "lds-dma");
        run<0, 4, 4>(src, span, iters, blocks, 8, sink, "lds-dma");
        run<0, 2, 4>(src, span, iters, blocks, 16, sink, "lds-dma");
        run<0, 8, 2>(src, span, iters, blocks, 8, sink, "lds-dma");
        run<1, 4, 2>(src, span, iters, blocks, 4, sink, "vgpr");
        run<1, 4, 4>(src, span, iters, blocks, 4, sink, "vgpr");
        run<1, 4, 2>(src, span, iters, blocks, 8, sink, "vgpr");
        run<1, 4, 4>(src, span, iters, blocks, 8, sink, "vgpr");
        run<1, 4, 2>(src, span, iters, blocks, 16, sink, "vgpr");
        run<1, 8, 2>(src, span, iters, blocks, 8, sink, "vgpr");
    }
    return 0;
}
