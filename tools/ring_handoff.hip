// ring_handoff.hip -- can blocks be handed from producer workgroups to consumer workgroups of the SAME kernel through the
// caches (L2 / the 256 MB memory-side cache) instead of HBM?  Question behind DESIGN.md section 7(b): the staged layer inputs /
// output gradients of the training step (9 KB per sample) would stay on chip if the kernel that produces them and the kernel
// that contracts them over samples ran side by side.
//
// 2 * npairs workgroups of 512 threads, all co-resident.  Producer p writes blocks of `bs` bytes into its own ring of `nb`
// blocks and publishes a counter (release, agent scope); consumer p waits for the counter (acquire), reads the block, checks
// every word, and publishes its own counter so that the producer may reuse the slot.  Every spin is bounded and watches a
// global abort flag: a lost partner ends the kernel, it cannot hang.
//   build: hipcc --offload-arch=gfx950 -O3 tools/ring_handoff.hip -o gpurun_out/ring_handoff     run: ./gpurun_out/ring_handoff
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define NT 512      // threads per workgroup
#define UN 8        // 16-byte accesses a thread keeps in flight
#define STRIDE (NT * 16)
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Ctl {                        // one per pair; the two counters on separate 128 B lines
    unsigned long long prod; char pad0[120];
    unsigned long long cons; char pad1[120];
};

template <bool ACQ>
__device__ __forceinline__ bool wait_ge(const unsigned long long* p, unsigned long long want, int* abort_flag) {
    for (long it = 0; it < 200000000L; ++it) {
        if (__hip_atomic_load(p, ACQ ? __ATOMIC_ACQUIRE : __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) return true;
        if ((it & 1023) == 1023 && __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
        __builtin_amdgcn_s_sleep(2);
    }
    return false;
}

// agent-scope accesses without fences: the sc1 bit makes a store write through to, and a load read from, the level at which
// the XCDs' L2s agree (what a relaxed agent-scope atomic gets); ordering by s_waitcnt alone
__device__ __forceinline__ void store_sc1(char* p, u32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }
// UN loads in flight, then the wait INSIDE the statement: the compiler does not know that a load written in assembly
// completes later, and would otherwise reuse or copy the destination registers before the data has arrived
#define LOAD8(NAME, POL)                                                                                                                   \
__device__ __forceinline__ void NAME(const char* p, u32x4 (&v)[8]) {                                                                      \
    const char *p1 = p + STRIDE, *p2 = p + 2 * STRIDE, *p3 = p + 3 * STRIDE, *p4 = p + 4 * STRIDE, *p5 = p + 5 * STRIDE, *p6 = p + 6 * STRIDE, *p7 = p + 7 * STRIDE; \
    asm volatile("global_load_dwordx4 %0, %8, off " POL "\n\tglobal_load_dwordx4 %1, %9, off " POL "\n\tglobal_load_dwordx4 %2, %10, off " POL "\n\t"   \
                 "global_load_dwordx4 %3, %11, off " POL "\n\tglobal_load_dwordx4 %4, %12, off " POL "\n\tglobal_load_dwordx4 %5, %13, off " POL "\n\t" \
                 "global_load_dwordx4 %6, %14, off " POL "\n\tglobal_load_dwordx4 %7, %15, off " POL "\n\ts_waitcnt vmcnt(0)"                        \
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])                  \
                 : "v"(p), "v"(p1), "v"(p2), "v"(p3), "v"(p4), "v"(p5), "v"(p6), "v"(p7) : "memory");                                    \
}
LOAD8(load8_sc1, "sc1")
LOAD8(load8_sc0, "sc0")

// mode 0: plain stores / loads + release / acquire fences;  mode 1: non-temporal stores and loads + the same fences;
// mode 2: sc1 stores / loads, no fences (s_waitcnt vmcnt(0) before the counter moves);
// mode 3: plain stores (the per-CU cache writes through, the data then sit in the XCD's L2) and sc0 loads (past the reader's
//         per-CU cache): coherent ONLY between workgroups that share an L2, i.e. run on the same XCD -- mismatches show if they do not
template <int MODE>
__global__ __launch_bounds__(NT) void k_handoff(char* ring, Ctl* ctl, int npairs, int nb, int bs, int T, int shift,
                                                 unsigned long long* bad, int* abort_flag) {
    __shared__ int ok;
    const int tid = threadIdx.x;
    const bool producer = (int)blockIdx.x < npairs;
    const int p = producer ? (int)blockIdx.x : (((int)blockIdx.x - npairs) - shift + npairs) % npairs;   // consumer j serves pair j - shift
    char* base = ring + (size_t)p * nb * bs;
    const int nvec = bs / STRIDE;                     // 16-byte accesses per thread per block (a multiple of UN)
    unsigned long long mism = 0;
    for (int t = 0; t < T; ++t) {
        if (tid == 0) {
            bool good = true;
            if (producer) { if (t >= nb) good = wait_ge<(MODE < 2)>(&ctl[p].cons, (unsigned long long)(t - nb + 1), abort_flag); }
            else good = wait_ge<(MODE < 2)>(&ctl[p].prod, (unsigned long long)(t + 1), abort_flag);
            if (!good) __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = good ? 1 : 0;
        }
        __syncthreads();
        if (!ok) break;
        char* blk = base + (size_t)(t % nb) * bs + tid * 16;
        const unsigned tag = (unsigned)t * 2654435761u + (unsigned)p;
        if (producer) {
            const u32x4 v = {tag, tag ^ (unsigned)tid, tag + 1u, tag + 2u};
            for (int i = 0; i < nvec; ++i) {
                if (MODE == 2) store_sc1(blk + (size_t)i * STRIDE, v);
                else if (MODE == 3) *reinterpret_cast<u32x4*>(blk + (size_t)i * STRIDE) = v;
                else if (MODE == 1) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(blk + (size_t)i * STRIDE));
                else *reinterpret_cast<u32x4*>(blk + (size_t)i * STRIDE) = v;
            }
            if (MODE >= 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");   // agent scope: my stores are visible before the counter moves
            __syncthreads();
            if (tid == 0) __hip_atomic_store(&ctl[p].prod, (unsigned long long)(t + 1), MODE >= 2 ? __ATOMIC_RELAXED : __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (MODE < 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // every thread: nothing stale from my caches
            for (int i = 0; i < nvec; i += UN) {
                u32x4 v[UN];
                if (MODE == 2) load8_sc1(blk + (size_t)i * STRIDE, v);
                else if (MODE == 3) load8_sc0(blk + (size_t)i * STRIDE, v);
                else {
#pragma unroll
                    for (int j = 0; j < UN; ++j)
                        v[j] = MODE == 1 ? __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(blk + (size_t)(i + j) * STRIDE))
                                         : *reinterpret_cast<const u32x4*>(blk + (size_t)(i + j) * STRIDE);
                }
#pragma unroll
                for (int j = 0; j < UN; ++j)
                    mism += (v[j][0] != tag) + (v[j][1] != (tag ^ (unsigned)tid)) + (v[j][2] != tag + 1u) + (v[j][3] != tag + 2u);
            }
            __syncthreads();
            if (tid == 0) __hip_atomic_store(&ctl[p].cons, (unsigned long long)(t + 1), MODE >= 2 ? __ATOMIC_RELAXED : __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (mism) atomicAdd(bad, mism);
}

// the same bytes without a partner: producers only / consumers only, each over its own ring (what HBM gives for this shape)
template <int MODE>
__global__ __launch_bounds__(NT) void k_solo(char* ring, int nb, int bs, int T, int write, unsigned long long* sink) {
    const int tid = threadIdx.x, p = blockIdx.x;
    char* base = ring + (size_t)p * nb * bs;
    const int nvec = bs / STRIDE;
    unsigned acc = 0;
    for (int t = 0; t < T; ++t) {
        char* blk = base + (size_t)(t % nb) * bs + tid * 16;
        const u32x4 v = {(unsigned)t, (unsigned)tid, 1u, 2u};
        for (int i = 0; i < nvec; i += UN) {
            if (write) {
#pragma unroll
                for (int j = 0; j < UN; ++j) {
                    if (MODE == 1) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(blk + (size_t)(i + j) * STRIDE));
                    else *reinterpret_cast<u32x4*>(blk + (size_t)(i + j) * STRIDE) = v;
                }
            } else {
                u32x4 x[UN];
#pragma unroll
                for (int j = 0; j < UN; ++j)
                    x[j] = MODE == 1 ? __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(blk + (size_t)(i + j) * STRIDE))
                                     : *reinterpret_cast<const u32x4*>(blk + (size_t)(i + j) * STRIDE);
#pragma unroll
                for (int j = 0; j < UN; ++j) acc += x[j][0] + x[j][3];
            }
        }
    }
    if (acc == 0x12345678u) *sink = acc;
}

int main() {
    const int npairs = 128;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    Ctl* ctl; unsigned long long* bad; int* abort_flag;
    CK(hipMalloc(&ctl, npairs * sizeof(Ctl))); CK(hipMalloc(&bad, 8)); CK(hipMalloc(&abort_flag, 4));
    const size_t max_ring = (size_t)8 << 30;
    char* ring; CK(hipMalloc(&ring, max_ring));
    CK(hipMemset(ring, 0, max_ring));
    printf("%d producer + %d consumer workgroups of 512 threads; TB/s figures count the bytes once per direction (written + read = 2x)\n", npairs, npairs);
    const int bss[] = {64 << 10, 256 << 10, 1 << 20};      // (nb = 1 would serialise producer and consumer)
    for (int bs : bss) {
        for (int nb : {2, 4, 8, 32, 128}) {
            if (bs == (64 << 10) && nb == 2) printf("(16 MiB of rings = 2 MiB per XCD: inside the 4 MiB L2s)\n");
            const size_t ring_bytes = (size_t)npairs * nb * bs;
            if (ring_bytes > max_ring) continue;
            const int T = (int)(((size_t)12 << 30) / ((size_t)npairs * bs));        // 12 GiB per direction
            for (int mode = 0; mode < 4; ++mode) {
                for (int shift : {0, 1}) {
                    CK(hipMemset(ctl, 0, npairs * sizeof(Ctl))); CK(hipMemset(bad, 0, 8)); CK(hipMemset(abort_flag, 0, 4));
                    CK(hipEventRecord(e0));
                    if (mode == 0) k_handoff<0><<<2 * npairs, NT>>>(ring, ctl, npairs, nb, bs, T, shift, bad, abort_flag);
                    else if (mode == 1) k_handoff<1><<<2 * npairs, NT>>>(ring, ctl, npairs, nb, bs, T, shift, bad, abort_flag);
                    else if (mode == 2) k_handoff<2><<<2 * npairs, NT>>>(ring, ctl, npairs, nb, bs, T, shift, bad, abort_flag);
                    else k_handoff<3><<<2 * npairs, NT>>>(ring, ctl, npairs, nb, bs, T, shift, bad, abort_flag);
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    unsigned long long hb; int ab;
                    CK(hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&ab, abort_flag, 4, hipMemcpyDeviceToHost));
                    const double gb = (double)npairs * T * bs / 1e9;
                    printf("block %5d KiB  ring %6.0f MiB (%3d blocks/pair)  %s  partner %s  %8.2f ms  %7.2f TB/s handed over (x2 moved)  mismatches %llu%s\n",
                           bs >> 10, ring_bytes / 1048576.0, nb, mode == 3 ? "L2 (sc0 load)" : mode == 2 ? "sc1, no fence" : (mode ? "nt + fences  " : "plain+fences "), shift ? "other XCD" : "same XCD ",
                           ms, gb / ms, hb, ab ? "  ABORTED (a wait ran out)" : "");
                    fflush(stdout);
                    if (ab) { printf("stopping after an aborted run\n"); return 2; }
                }
            }
        }
        // reference points for this block size: the same streams without a partner, over rings far larger than any cache
        size_t nbs = max_ring / ((size_t)npairs * bs);
        const int nb = nbs > 128 ? 128 : (int)nbs;
        const int T = (int)(((size_t)12 << 30) / ((size_t)npairs * bs));
        for (int mode = 0; mode < 2; ++mode)
            for (int write = 1; write >= 0; --write) {
                CK(hipEventRecord(e0));
                if (mode == 0) k_solo<0><<<npairs, NT>>>(ring, nb, bs, T, write, bad); else k_solo<1><<<npairs, NT>>>(ring, nb, bs, T, write, bad);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                printf("block %5d KiB  solo %s %s over %d MiB rings, %d workgroups: %7.2f TB/s\n", bs >> 10, write ? "write" : "read ", mode ? "non-temporal" : "plain       ",
                       (int)((size_t)nb * bs >> 20), npairs, (double)npairs * T * bs / 1e9 / ms);
            }
    }
    return 0;
}
