// Timing skeleton of a backward that keeps the output gradients on chip: what one net's backward at the bench size
// (12.58 M samples, four 128-wide layers) would cost if the dgrad kernel contracted D_j with the layer inputs itself.  SYNTHETIC data,
// real instruction mix and real memory traffic; the numbers price the structure, nothing here is a product kernel.
//
// Workgroup = four waves of up to 512 registers (one per SIMD), one round = 256 samples (64 per wave, two 32-sample column tiles).
// Per round and layer j:
//   W   the layer's bf16 weight image (32 KB, L2-resident) HBM/L2 -> LDS ring by LDS-DMA, one layer ahead
//   H   the layer input's e4m3 tile [256 samples][128 B] HBM -> LDS by LDS-DMA (every round reads fresh bytes: 32 KB)
//   C   the dgrad chain: 4 row tiles x 8 k-steps x 2 column tiles = 64 v_mfma_f32_32x32x16_bf16, A fragments by ds_read_b128
//   E   the epilogue per (row tile, column tile): mask, bf16 pack (the next layer's B operand), e5m2 conversion, and ONE ds_write_b128
//       of the 16 bytes a lane holds into the D tile [sample][chunk = (row tile, lane half)][16 features]  (16-byte chunks XOR-swizzled
//       by the sample row so that the transposed reads below are conflict-free)
//   G   the weight gradient: wave w owns output rows 32w .. 32w+31 of dW_j: per 64-sample k-step its A operand (D^T) by 4
//       ds_read_b64_tr_b8 and, for each of the four 32-column blocks, the B operand (H^T) by 4 more; 16 v_mfma_scale_f32_32x32x64_f8f6f4
//       into 4 x 16 accumulators per layer (256 for the four layers)
// with two workgroup barriers per layer (tiles complete -> G; G done -> tiles free).  PARTS selects what is compiled in.
//   hipcc --offload-arch=gfx950 -O3 tools/onchip_bwd_probe.hip -o /tmp/obp && /tmp/obp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <type_traits>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

constexpr int NL = 4, TS = 256, WBYTES = 32768, TILE = TS * 128;
enum { P_W = 1, P_H = 2, P_C = 4, P_E = 8, P_G = 16 };

__device__ __forceinline__ int swz(int row, int chunk) { return chunk ^ (((row >> 1) & 3) << 1); }

template <int PARTS>
__global__ __launch_bounds__(256, 1) void probe(const char* __restrict__ W, const char* __restrict__ H, long long hbytes, int rounds_total, float* out,
                                                unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* const ring = lds;
    char* const dt = lds + 2 * WBYTES;
    char* const ht = dt + TILE;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lr = lane & 31, lh = lane >> 5;
    for (int i = threadIdx.x; i < (2 * WBYTES + 2 * TILE) / 4; i += 256) reinterpret_cast<unsigned*>(lds)[i] = 0x38383838u;
    __syncthreads();

    u32x4 B[2][8];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int s = 0; s < 8; ++s) B[c][s] = u32x4{0x3c003c00u + lane, 0x3c003c00u, 0x3c003c00u + s, 0x3c003c00u + c};
    f32x16 accW[NL][4];
#pragma unroll
    for (int j = 0; j < NL; ++j)
#pragma unroll
        for (int n = 0; n < 4; ++n) accW[j][n] = (f32x16)(0.f);
    const unsigned maskbits = 0x5a5a5a5au ^ (unsigned)lane;
    const int scale127 = 127;

    auto dma = [&](const char* src, char* dst, int bytes) __attribute__((always_inline)) {     // whole workgroup: 1 KB per wave instruction
        for (int c = wave; c < bytes / 1024; c += 4)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)c * 1024 + lane * 16),
                                             (__attribute__((address_space(3))) void*)(dst + c * 1024), 16, 0, 0);
    };
    // addresses of the transposed reads: lane i of a 16-lane group supplies row (i >> 1), bytes 8 (i & 1) .. +7 of a 16-byte chunk
    const int tq = (lane & 15) >> 1, tp = lane & 1, tg = (lane >> 4) & 1;
    const int tsw = (tq >> 1) & 3;
    const char* const abase = dt + (lh * 32 + tq) * 128 + (2 * (wave ^ tsw) + tg) * 16 + tp * 8;
    const char* hbase[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) hbase[nb] = ht + (lh * 32 + tq) * 128 + (2 * (nb ^ tsw) + tg) * 16 + tp * 8;

    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (PARTS & P_W) dma(W, ring, WBYTES);
    for (int round = blockIdx.x; round < rounds_total; round += gridDim.x) {
#pragma unroll 1
        for (int j = 0; j < NL; ++j) {
            const char* img = ring + (j & 1) * WBYTES + lane * 16;
            if (PARTS & P_W) dma(W + ((j + 1) % NL) * WBYTES, ring + ((j + 1) & 1) * WBYTES, WBYTES);          // next layer's image
            if (PARTS & P_H) {
                const long long off = (((long long)round * NL + j) * TILE) % hbytes;
                dma(H + off, ht, TILE);
            }
            // ---- dgrad chain + epilogue -------------------------------------------------------------------------------------------
            u32x4 Bn[2][8];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                f32x16 acc0 = (f32x16)(0.f), acc1 = (f32x16)(0.f);
                if (PARTS & P_C) {
#pragma unroll
                    for (int ks = 0; ks < 8; ++ks) {
                        const u32x4 a = *reinterpret_cast<const u32x4*>(img + (m * 8 + ks) * 1024);
                        // (inline assembly with VGPR accumulators: left to itself the register allocator puts the chain's accumulators into
                        // AGPRs as well and shuffles the 256 weight-gradient accumulators around them -- 1 478 v_accvgpr moves, 160 spills)
                        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc0) : "v"(a), "v"(B[0][ks]));
                        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc1) : "v"(a), "v"(B[1][ks]));
                    }
                    asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");          // MFMA result -> vector ALU read: software wait states
                } else {
                    acc0[0] = __builtin_bit_cast(float, B[0][m][0]); acc1[1] = __builtin_bit_cast(float, B[1][m][1]);
                }
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const f32x16& acc = c == 0 ? acc0 : acc1;
                    unsigned w[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const bf16x2 pk = {(__bf16)acc[2 * u], (__bf16)acc[2 * u + 1]};
                        w[u] = __builtin_bit_cast(unsigned, pk);
                    }
                    if (PARTS & P_E) {
                        // ReLU mask of the layer (bits -> 0 / 0xffff per half word -> and)
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const unsigned bits = (maskbits >> (2 * u + 16 * c)) & 3u;
                            const unsigned mk = (bits & 1u ? 0xffffu : 0u) | (bits & 2u ? 0xffff0000u : 0u);
                            w[u] &= mk;
                        }
                        u32x4 q;
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            s16x2 v = {0, 0};
                            v = __builtin_amdgcn_cvt_scalef32_pk_bf8_bf16(v, __builtin_bit_cast(bf16x2, w[2 * u]), 1.0f, false);
                            v = __builtin_amdgcn_cvt_scalef32_pk_bf8_bf16(v, __builtin_bit_cast(bf16x2, w[2 * u + 1]), 1.0f, true);
                            q[u] = __builtin_bit_cast(unsigned, v);
                        }
                        const int row = wave * 64 + c * 32 + lr;
                        *reinterpret_cast<u32x4*>(dt + row * 128 + swz(row, 2 * m + lh) * 16) = q;
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) { Bn[c][2 * m][u] = w[u]; Bn[c][2 * m + 1][u] = w[4 + u]; }
                }
            }
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int s = 0; s < 8; ++s) B[c][s] = Bn[c][s];
            // ---- tiles complete -------------------------------------------------------------------------------------------------------
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            // ---- weight gradient of this layer: rows 32 * wave .. + 31 ---------------------------------------------------------------
            if (PARTS & P_G) {
                // (one copy per layer: accW[J] must be a static register index)
                auto wgrad = [&](auto jc) __attribute__((always_inline)) {
                    constexpr int J = decltype(jc)::value;
                    // 16 steps (k-step, column block); the operands of step t + 1 are read while step t multiplies (one fragment pair ahead)
                    // the swizzle of a read's row depends on the lane only ((row >> 1) & 3 = (tq >> 1) & 3: k-steps and 8-row sub-blocks move
                    // the row by multiples of 8), so a fragment is one lane address per 32-feature block + immediate offsets
                    auto rd = [&](const char* base, int ks) __attribute__((always_inline)) {
                        i32x8 f;
#pragma unroll
                        for (int sb = 0; sb < 4; ++sb) {
                            const i32x2 v = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) i32x2*)(base + (ks * 64 + sb * 8) * 128));
                            f[2 * sb] = v[0]; f[2 * sb + 1] = v[1];
                        }
                        return f;
                    };
                    i32x8 A = rd(abase, 0), Bc = rd(hbase[0], 0);
#pragma unroll
                    for (int t = 0; t < 16; ++t) {
                        const int ks = t >> 2, nb = t & 3;
                        i32x8 An = A, Bn2 = Bc;
                        if (t + 1 < 16) {
                            Bn2 = rd(hbase[(t + 1) & 3], (t + 1) >> 2);
                            if (((t + 1) & 3) == 0) An = rd(abase, (t + 1) >> 2);
                        }
                        asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0] cbsz:1" : "+a"(accW[J][nb]) : "v"(A), "v"(Bc), "v"(scale127));
                        __builtin_amdgcn_sched_barrier(0);
                        A = An; Bc = Bn2;
                        (void)ks;
                    }
                };
                switch (j) {
                    case 0: wgrad(std::integral_constant<int, 0>{}); break;
                    case 1: wgrad(std::integral_constant<int, 1>{}); break;
                    case 2: wgrad(std::integral_constant<int, 2>{}); break;
                    default: wgrad(std::integral_constant<int, 3>{}); break;
                }
            }
            // ---- tiles free ------------------------------------------------------------------------------------------------------------
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NL; ++j)
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int i = 0; i < 16; ++i) s += accW[j][n][i];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int k = 0; k < 8; ++k) s += __builtin_bit_cast(float, B[c][k][0]);
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int PARTS>
static void run(const char* name, const char* W, const char* H, long long hbytes, float* out, unsigned long long* dcyc) {
    const int grid = 256, rounds = 49152;                         // 12 582 912 samples / 256
    const size_t lds = 2 * WBYTES + 2 * TILE;
    (void)hipFuncSetAttribute((const void*)probe<PARTS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<PARTS>), dim3(grid), dim3(256), lds, 0, W, H, hbytes, rounds / 8, out, dcyc);       // warm
    float best = 1e9f;
    unsigned long long med = 0;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((probe<PARTS>), dim3(grid), dim3(256), lds, 0, W, H, hbytes, rounds, out, dcyc);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        best = std::min(best, ms);
        std::vector<unsigned long long> h(grid);
        (void)hipMemcpy(h.data(), dcyc, grid * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        med = h[grid / 2];
    }
    const hipError_t e = hipGetLastError();
    // s_memtime ticks at 100 MHz: ticks * 10 ns = the workgroup's wall time
    printf("%-58s %7.3f ms per net   (median workgroup %7.3f ms; %s)\n", name, best, med * 1e-5, e == hipSuccess ? "ok" : hipGetErrorString(e));
    fflush(stdout);
}

int main() {
    char *W, *H; float* out; unsigned long long* dcyc;
    const long long hbytes = 6LL << 30;
    (void)hipMalloc(&W, NL * WBYTES); (void)hipMalloc(&H, hbytes + TILE); (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&dcyc, 256 * 8);
    (void)hipMemset(W, 0x3c, NL * WBYTES); (void)hipMemset(H, 0x38, hbytes + TILE);
    printf("one net's backward at the bench size (49 152 rounds of 256 samples x 4 layers of 128), 256 workgroups of 4 waves:\n");
    run<P_W | P_C>("chain MFMAs + weight ring", W, H, hbytes, out, dcyc);
    run<P_W | P_C | P_E>("  + epilogue (mask, bf16 pack, e5m2, D tile to LDS)", W, H, hbytes, out, dcyc);
    run<P_W | P_C | P_E | P_H>("  + H tiles HBM -> LDS (6.4 GB)", W, H, hbytes, out, dcyc);
    run<P_W | P_C | P_E | P_H | P_G>("  + weight gradient (transposed reads, MX MFMAs) = all", W, H, hbytes, out, dcyc);
    run<P_W | P_H | P_G>("weight gradient alone (tiles + transposed reads + MX MFMAs)", W, H, hbytes, out, dcyc);
    run<P_W | P_C | P_E | P_G>("all but the H traffic", W, H, hbytes, out, dcyc);
    return 0;
}
