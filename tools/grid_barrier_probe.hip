// tools/grid_barrier_probe.hip -- round 6, step 0 of the persistent utterance-minor kernel (VERDICT r5 "Next round" item 1):
// what does ONE frame boundary cost INSIDE a launch, with the traffic the real frame has, before any kernel is built on it?
//
// A persistent grid of G workgroups x 256 threads walks T frames.  Per frame a workgroup
//   (1) reads R float4 per lane from its COMBO's vector (buffer f & 1): 128-byte segments at hashed positions, 8 loads in flight
//       per lane (the frame kernel's gathers: one segment = the UL = 32 utterances of one state);
//   (2) writes its W float4 per lane of the combo's next vector (buffer (f + 1) & 1), tagged with the frame number;
//   (3) crosses the grid barrier.
// Every value read is CHECKED against the tag the previous frame must have written (stale reads are counted, not assumed away).
// The workgroups of a combo sit on the XCDs x with x % C == combo (block b runs on XCD b % 8), as crf_batch_frame_kernel's do.
//
// data modes   0  no data at all (the empty barrier)
//              1  sc1 (write-through) stores, sc1 loads (L1 bypassed, L2-served), no fence anywhere
//              2  sc1 stores, ONE lane per workgroup runs an agent-scope acquire behind the barrier, plain loads
//              3  plain stores, one lane's agent-scope release before arriving, one lane's acquire behind the barrier, plain loads
//              4  like 3 but EVERY wave fences (round 3's persistent experiment, for the record)
// barriers     0  one counter, every workgroup's lane 0 arrives and polls it (relaxed sc1 loads, s_sleep)
//              1  XCD-hierarchical: per-XCD arrival counter, the XCD's last arriver goes to the top counter and then publishes the
//                 XCD's generation word, which the others poll
// baseline     --launches: the same frame body as ONE LAUNCH PER FRAME (what the product does today), data mode 3 without fences
//
//   hipcc --offload-arch=gfx950 -O3 tools/grid_barrier_probe.hip -o tools/grid_barrier_probe.bin
//   tools/grid_barrier_probe.bin            (prints one line per configuration)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

struct Params {
    unsigned *ctr;       // [0] flat counter | [64 + 16 x] arrivals of XCD x | [256 + 16 x] generation of XCD x | [512] top | [520 + x] census
    u32x4 *vec;          // [2][C][n4]
    unsigned *err;       // [0] stale / wrong values seen, [1] time-outs
    int T, G, C, n4;     // frames, workgroups, combos, float4 per combo vector
    int R, W;            // float4 read / written per lane and frame
    int mode, bar, f0;   // f0: first frame of this launch (launch-per-frame baseline: T = 1)
};

__device__ __forceinline__ unsigned hash32(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

__device__ __forceinline__ bool poll_ge(unsigned *w, unsigned target, unsigned *err) {
    for (unsigned spins = 0;; ++spins) {
        if (__hip_atomic_load(w, RLX_AGENT) >= target) return true;
        __builtin_amdgcn_s_sleep(1);
        if (spins > (1u << 22)) { atomicAdd(err + 1, 1u); return false; }
    }
}

template <int MODE>
__device__ __forceinline__ void frame_body(const Params &p, int f, int combo, int idx, int nc, int tid) {
    if (MODE == 0) return;
    const int lane = tid & 63, wave = tid >> 6;
    const u32x4 *src = p.vec + ((size_t)(f & 1) * p.C + combo) * p.n4;
    u32x4 *dst = p.vec + ((size_t)((f + 1) & 1) * p.C + combo) * p.n4;
    const unsigned nseg = (unsigned)p.n4 / 8u;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, p.n4 * 16, 0x27000);
    __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void *)dst, 0, p.n4 * 16, 0x27000);
    unsigned bad = 0;
    for (int r = 0; r < p.R; r += 8) {
        u32x4 v[8];
        unsigned at[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const unsigned seg = hash32((unsigned)(((idx * 4 + wave) * 4096 + r + k) * 8 + (lane >> 3)) ^ (unsigned)(f * 0x9e3779b9u)) % nseg;
            at[k] = seg * 8u + (unsigned)(lane & 7);
            if (MODE == 1) v[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, at[k] * 16u, 0, 16);
            else v[k] = src[at[k]];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) bad += (v[k].x != (unsigned)f) | (v[k].y != at[k]);
    }
    if (bad) atomicAdd(p.err, bad);
    for (int k = 0; k < p.W; ++k) {
        const unsigned i = (unsigned)((k * nc + idx) * 256 + tid);
        const u32x4 x = {(unsigned)(f + 1), i, hash32(i), ~i};
        if (MODE == 1 || MODE == 2) __builtin_amdgcn_raw_buffer_store_b128(x, rd, i * 16u, 0, 16);
        else dst[i] = x;
    }
}

template <int MODE, int BAR>
__global__ __launch_bounds__(256) void persistent_kernel(Params p) {
    const int tid = threadIdx.x, b = blockIdx.x;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 7u;
    // the combo and this workgroup's index among the combo's workgroups follow the BLOCK id (placement is a matter of speed only)
    const int per = 8 / p.C, combo = (b & 7) % p.C, nc = p.G / p.C, idx = (b >> 3) * per + (b & 7) / p.C;
    __shared__ unsigned nx_s;
    if (BAR == 1) {   // census: workgroups per XCD, then one flat barrier so that every count is final
        if (tid == 0) {
            atomicAdd(p.ctr + 520 + xcc, 1u);
            atomicAdd(p.ctr + 0, 1u);
            poll_ge(p.ctr + 0, (unsigned)p.G, p.err);
            nx_s = __hip_atomic_load(p.ctr + 520 + xcc, RLX_AGENT);
        }
        __syncthreads();
    }
    const unsigned nx = BAR == 1 ? nx_s : 0u;
    unsigned nxcd = 0;
    if (BAR == 1) { for (int x = 0; x < 8; ++x) nxcd += __hip_atomic_load(p.ctr + 520 + x, RLX_AGENT) != 0u; }
    for (int f = 0; f < p.T; ++f) {
        frame_body<MODE>(p, f, combo, idx, nc, tid);
        // ---- the frame boundary ----
        if (MODE != 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // every writing wave drains its own stores
        if (MODE == 4) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        __syncthreads();
        if (tid == 0) {
            if (MODE == 3) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
            if (BAR == 0) {
                __hip_atomic_fetch_add(p.ctr + 0, 1u, RLX_AGENT);
                poll_ge(p.ctr + 0, (unsigned)p.G * (unsigned)(f + 1 + (BAR == 1)), p.err);
            } else {
                const unsigned old = __hip_atomic_fetch_add(p.ctr + 64 + 16 * xcc, 1u, RLX_AGENT);
                if (old + 1u == nx * (unsigned)(f + 1)) {                       // the XCD's last arriver
                    __hip_atomic_fetch_add(p.ctr + 512, 1u, RLX_AGENT);
                    poll_ge(p.ctr + 512, nxcd * (unsigned)(f + 1), p.err);
                    __hip_atomic_store(p.ctr + 256 + 16 * xcc, (unsigned)(f + 1), RLX_AGENT);
                } else {
                    poll_ge(p.ctr + 256 + 16 * xcc, (unsigned)(f + 1), p.err);
                }
            }
            if (MODE == 2 || MODE == 3) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        if (MODE == 4) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
}

// ---- which LOAD instruction?  (round 6, after the first persistent kernel: its sc1 gathers as raw BUFFER loads ran 40 % slower than the
// per-frame kernel's plain global loads.)  The frame's gathers alone, one launch per frame so that every flavour is correct, 16 loads in
// flight per lane: LD 0 plain global_load_dwordx4, 1 buffer_load_dwordx4 (aux 0), 2 buffer_load_dwordx4 sc1, 3 global_load_dwordx4 sc1
// (inline asm, waited for by hand), 4 global_load_dwordx4 nt
template <int LD>
__global__ __launch_bounds__(256) void gather_kernel(Params p) {
    const int tid = threadIdx.x, b = blockIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = 8 / p.C, combo = (b & 7) % p.C, idx = (b >> 3) * per + (b & 7) / p.C;
    const u32x4 *src = p.vec + (size_t)combo * p.n4;
    const unsigned nseg = (unsigned)p.n4 / 8u;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, p.n4 * 16, 0x27000);
    unsigned acc = 0;
    for (int r = 0; r < p.R; r += 16) {
        u32x4 v[16];
        unsigned at[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const unsigned seg = (hash32((unsigned)(((idx * 4 + wave) * 4096 + r + k) * 8 + (lane >> 3)) ^ (unsigned)(p.f0 * 0x9e3779b9u)) & 0xffffffu) * nseg >> 24;
            at[k] = seg * 8u + (unsigned)(lane & 7);
            if (LD == 0) v[k] = src[at[k]];
            else if (LD == 1) v[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, at[k] * 16u, 0, 0);
            else if (LD == 2) v[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, at[k] * 16u, 0, 16);
            else if (LD == 3) asm volatile("global_load_dwordx4 %0, %1, %2 sc1" : "=v"(v[k]) : "v"(at[k] * 16u), "s"(src) : "memory");
            else v[k] = __builtin_nontemporal_load(src + at[k]);
        }
        if (LD == 3) {
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) : : "memory");
            asm volatile("" : "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]) : : "memory");
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) acc += (v[k].y != at[k]);
    }
    if (acc) atomicAdd(p.err, acc);
}

// the same frame as its own launch (the kernel boundary is the barrier): plain stores and loads
__global__ __launch_bounds__(256) void frame_kernel(Params p) {
    const int tid = threadIdx.x, b = blockIdx.x;
    const int per = 8 / p.C, combo = (b & 7) % p.C, nc = p.G / p.C, idx = (b >> 3) * per + (b & 7) / p.C;
    frame_body<3>(p, p.f0, combo, idx, nc, tid);
}

template <int MODE>
static void launch_persistent(const Params &p, hipStream_t st) {
    if (p.bar == 0) hipLaunchKernelGGL((persistent_kernel<MODE, 0>), dim3(p.G), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((persistent_kernel<MODE, 1>), dim3(p.G), dim3(256), 0, st, p);
}

int main(int argc, char **argv) {
    int T = 1500;
    bool quick = argc > 1 && !strcmp(argv[1], "quick");
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("# device %s, %d CUs\n", prop.name, prop.multiProcessorCount);
    unsigned *ctr, *err;
    CK(hipMalloc(&ctr, 4096));
    CK(hipMalloc(&err, 64));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct Cfg { const char *name; int G, C, W, R; };
    // "large": S = 16 385, B = 64 in two groups of 32: 4 combos, ~2.6 MB vectors, 150 MB of gathers per frame
    // "c5":    config #5 at B = 8 (one group of 8: 32-byte segments there; modelled with 128-byte ones): 2 combos
    const Cfg cfgs[] = {
        {"exchange 1 MB, light reads", 256, 4, 1, 8},
        {"large-graph frame (10 MB written, 150 MB gathered)", 256, 4, 10, 144},
        {"large-graph frame, 2 workgroups per CU", 512, 4, 5, 72},
        {"gathers only quarter (10 MB written, 38 MB gathered)", 256, 4, 10, 40},
    };
    for (const Cfg &c : cfgs) {
        const int nc = c.G / c.C, n4 = nc * 256 * c.W;
        u32x4 *vec;
        CK(hipMalloc(&vec, (size_t)2 * c.C * n4 * 16));
        std::vector<u32x4> h((size_t)2 * c.C * n4);
        for (int bf = 0; bf < 2; ++bf) for (int cc = 0; cc < c.C; ++cc) for (int i = 0; i < n4; ++i)
            h[((size_t)bf * c.C + cc) * n4 + i] = u32x4{0u, (unsigned)i, 0u, ~(unsigned)i};
        printf("## %s: G = %d, %d combos, vector %.2f MB per combo, written %.1f MB, gathered %.1f MB per frame\n", c.name, c.G, c.C,
               n4 * 16 / 1048576.0, (double)c.C * n4 * 16 / 1048576.0, (double)c.G * 256 * c.R * 16 / 1048576.0);
        for (int bar = 0; bar < 2; ++bar)
            for (int mode = 0; mode <= 4; ++mode) {
                if (mode == 0 && &c != &cfgs[0] && &c != &cfgs[2]) continue;
                if (quick && mode == 4) continue;
                Params p{ctr, vec, err, T, c.G, c.C, n4, c.R, c.W, mode, bar, 0};
                float best = 1e30f;
                unsigned herr[2] = {0, 0};
                for (int rep = 0; rep < 3; ++rep) {
                    CK(hipMemcpy(vec, h.data(), h.size() * 16, hipMemcpyHostToDevice));
                    CK(hipMemsetAsync(ctr, 0, 4096, st));
                    CK(hipMemsetAsync(err, 0, 64, st));
                    CK(hipEventRecord(e0, st));
                    switch (mode) {
                        case 0: launch_persistent<0>(p, st); break;
                        case 1: launch_persistent<1>(p, st); break;
                        case 2: launch_persistent<2>(p, st); break;
                        case 3: launch_persistent<3>(p, st); break;
                        default: launch_persistent<4>(p, st); break;
                    }
                    CK(hipEventRecord(e1, st));
                    CK(hipStreamSynchronize(st));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    if (ms < best) best = ms;
                    unsigned e2[2]; CK(hipMemcpy(e2, err, 8, hipMemcpyDeviceToHost));
                    herr[0] += e2[0]; herr[1] += e2[1];
                }
                printf("persistent  barrier %s  data mode %d: %8.3f us per frame   wrong values %u, time-outs %u\n",
                       bar ? "xcd " : "flat", mode, best * 1000.f / T, herr[0], herr[1]);
                fflush(stdout);
            }
        {   // one launch per frame
            Params p{ctr, vec, err, 1, c.G, c.C, n4, c.R, c.W, 3, 0, 0};
            float best = 1e30f;
            unsigned herr = 0;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipMemcpy(vec, h.data(), h.size() * 16, hipMemcpyHostToDevice));
                CK(hipMemsetAsync(err, 0, 64, st));
                CK(hipEventRecord(e0, st));
                for (int f = 0; f < T; ++f) { p.f0 = f; hipLaunchKernelGGL(frame_kernel, dim3(c.G), dim3(256), 0, st, p); }
                CK(hipEventRecord(e1, st));
                CK(hipStreamSynchronize(st));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
                unsigned e2; CK(hipMemcpy(&e2, err, 4, hipMemcpyDeviceToHost));
                herr += e2;
            }
            printf("one launch per frame                 : %8.3f us per frame   wrong values %u\n", best * 1000.f / T, herr);
            fflush(stdout);
        }
        if (&c == &cfgs[1] || &c == &cfgs[2]) {   // load flavours
            Params p{ctr, vec, err, 1, c.G, c.C, n4, c.R / 16 * 16, c.W, 3, 0, 0};
            const char *names[] = {"global_load plain", "buffer_load aux 0", "buffer_load sc1", "global_load sc1 (asm)", "global_load nt"};
            for (int ld = 0; ld < 5; ++ld) {
                float best = 1e30f;
                unsigned herr = 0;
                for (int rep = 0; rep < 3; ++rep) {
                    CK(hipMemcpy(vec, h.data(), h.size() * 16, hipMemcpyHostToDevice));
                    CK(hipMemsetAsync(err, 0, 64, st));
                    CK(hipEventRecord(e0, st));
                    for (int f = 0; f < 300; ++f) {
                        p.f0 = f;
                        switch (ld) {
                            case 0: hipLaunchKernelGGL(gather_kernel<0>, dim3(c.G), dim3(256), 0, st, p); break;
                            case 1: hipLaunchKernelGGL(gather_kernel<1>, dim3(c.G), dim3(256), 0, st, p); break;
                            case 2: hipLaunchKernelGGL(gather_kernel<2>, dim3(c.G), dim3(256), 0, st, p); break;
                            case 3: hipLaunchKernelGGL(gather_kernel<3>, dim3(c.G), dim3(256), 0, st, p); break;
                            default: hipLaunchKernelGGL(gather_kernel<4>, dim3(c.G), dim3(256), 0, st, p); break;
                        }
                    }
                    CK(hipEventRecord(e1, st));
                    CK(hipStreamSynchronize(st));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    if (ms < best) best = ms;
                    unsigned e2; CK(hipMemcpy(&e2, err, 4, hipMemcpyDeviceToHost));
                    herr += e2;
                }
                printf("gathers only, one launch per frame, %-24s: %8.3f us per frame   wrong values %u\n", names[ld], best * 1000.f / 300, herr);
                fflush(stdout);
            }
        }
        CK(hipFree(vec));
    }
    return 0;
}
