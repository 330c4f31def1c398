// cat_amd/csrc/k_res.hip -- register-resident denominator recursions, generic layout over K compute units
// (one translation unit per kernel family, built in parallel by cat_amd/build.py; the explicit instantiations at the end are the
//  ones the host side in crf_host.hip launches -- a missing one is a link error, -Wl,-z,defs)
#include "crf_device.h"
#include "crf_kernels_decl.h"
#include "k_res_common.h"

namespace crf {

// =============================================================================================
// Register-resident denominator recursions (crf_internal.h: ResDev, res_layout.cpp).
// A recursion of one utterance runs on K compute units; each thread holds its arcs in VGPRs, so a
// frame is: max-reduce -> kResNCH x (4 LDS gathers + 4 FMA), row epilogue at every slice end ->
// barrier -> (K > 1) all-gather of the new state vector through tagged 8-byte granules in L2.
// =============================================================================================

// (block reductions, the tagged-granule exchange and the gather / accumulate macros: k_res_common.h, shared with the factored family)

template <int DIR>
__device__ __forceinline__ void res_chain_body(const ResParams &p, float *lds, const int bx, const int gx) {
    const ResDirDev &L = p.L;
    const int K = p.K;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Peers of one recursion are placed 8 block ids apart: the dispatcher is observed to put block x on
    // XCD x % 8, so the K CUs that exchange a state vector every frame share one L2 (speed only; the
    // protocol is placement-independent).
    int b, k;
    {
        const int x = bx, total = gx, full = total / (8 * K) * (8 * K);
        if (x < full) { const int grp = x / (8 * K), within = x % (8 * K); k = within / 8; b = p.b0 + grp * 8 + within % 8; }
        else { const int y = x - full; k = y % K; b = p.b0 + full / K + y / K; }
    }
    const int V = p.V, lx = p.lx[b], G = L.G;
    const int Vp = rup64(V + 1);                             // emissions + a zero at [V] for rows that produce nothing
    const int64_t bt0 = (int64_t)b * p.T;
    const int rows_cu_max = p.rows_cu_max;
    float *X = lds;                                          // [2][kResGmax]: ping-pong state vectors
    int *RL = (int *)((char *)lds + 2 * kResXB);             // [rows_cu_max] emission index (label) of this CU's rows
    float *EP = (float *)(RL + rows_cu_max);                 // [2][Vp]
    float *wm = EP + 2 * Vp;                                 // [2][kResWaves] per-wave maxima of the next vector
    double *red = (double *)(wm + 2 * kResWaves);            // [kResWaves]
    int *PL = (int *)(red + kResWaves);                      // [2 * (kResMaxK - 1)] entry ranges of the peers

    // ---- one-time: arcs -> registers, row metadata -> LDS
    unsigned A[kResWords];
    {
        const unsigned *src = L.arcs + (size_t)k * kResWords * kResThreads + tid;
#pragma unroll
        for (int i = 0; i < kResWords; ++i) A[i] = src[(size_t)i * kResThreads];
    }
    const uint4 wi = L.wave_info[k * kResWaves + wave];
    const unsigned ends = __builtin_amdgcn_readfirstlane(wi.x);
    const int nch = __builtin_amdgcn_readfirstlane(wi.y);
    const int row0 = __builtin_amdgcn_readfirstlane(wi.z);
    const int cu_row0 = L.cu_row_off[k], cu_rows = L.cu_row_off[k + 1] - cu_row0;
    const int own0 = __builtin_amdgcn_readfirstlane(L.own_off[k]);  // first gather entry produced by this CU
    for (int r = tid; r < cu_rows; r += kResThreads) { const int l = L.row_lab[cu_row0 + r]; RL[r] = l < 0 ? V : l; }
    if (tid < 2) EP[tid * Vp + V] = 0.f;
    // forward slots first ([B][2][Gf]), backward slots ([B][2][Gb]) after them
    gu64 *xch = (gu64 *)(p.xch + (DIR == 0 ? 0 : (size_t)p.B * 2 * (size_t)p.Gf) + (size_t)b * 2 * (size_t)G);
    int E = kScaleExp;
    float zpart = 0.f;

    // ---- initial vector (complete on every CU, no exchange needed).  Entries nobody produces stay 0 in
    // both buffers for ever; every other entry is rewritten by its one producing row in every frame, so
    // the buffers never need clearing.
    for (int s = tid; s < 2 * kResGmax; s += kResThreads) X[s] = 0.f;
    if (lx > 0)
        for (int v = tid; v < V; v += kResThreads) {
            EP[v] = p.ep[(bt0 + (DIR == 0 ? 0 : lx - 1)) * V + v];
            if (DIR == 1 && lx > 1) EP[Vp + v] = p.ep[(bt0 + lx - 2) * V + v];
        }
    __syncthreads();
    {
        float m0 = 0.f;
        if (DIR == 0) {
            for (int s = tid; s < G; s += kResThreads) { const float v = p.x_start[s] * pow2f(kScaleExp); X[s] = v; m0 = fmaxf(m0, v); }
        } else if (lx > 0) {
            for (int z = tid; z < G; z += kResThreads) { const float v = EP[p.z_lab[z]] * (p.z_end[z] * pow2f(kScaleExp)); X[z] = v; m0 = fmaxf(m0, v); }
            float *BProw = p.Out + (bt0 + lx - 1) * p.Rout;
            for (int r = tid; r < cu_rows; r += kResThreads) BProw[cu_row0 + r] = p.brow_end[cu_row0 + r] * pow2f(kScaleExp);
            if (tid == 0 && k == 0) p.Eout[bt0 + lx - 1] = E;
        } else {
            for (int r = tid; r < cu_rows; r += kResThreads) zpart += p.brow_start[cu_row0 + r] * p.brow_end[cu_row0 + r] * pow2f(kScaleExp);
        }
        m0 = wave_max(m0);
        if (lane == 0) wm[wave] = m0;
    }
    __syncthreads();

    // ---- where are my peers?  Every CU publishes the id of its XCD (write-through, placement-independent)
    // and reads its peers'; all peers on one XCD => the cheap same-L2 hand-off is used.  Both sides take the
    // decision from the same K ids, so they always agree.
    bool same_l2 = false;
    if (K > 1) {
        gu64 *hs = (gu64 *)(p.xch + (size_t)p.B * 2 * ((size_t)p.Gf + (size_t)p.Gb)) + ((size_t)DIR * p.B + b) * kResMaxK;
        const unsigned my_xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf;  // HW_REG_XCC_ID[3:0]
        if (tid == 0) __hip_atomic_store(hs + k, (1ull << 32) | my_xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int same = 1;
        if (tid < K && tid != k) {
            unsigned long long g = 0;
            for (unsigned spins = 0; (g >> 32) != 1ull; ++spins) {
                g = __hip_atomic_load(hs + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (spins > (1u << 22)) { __hip_atomic_store(p.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                if ((spins & 255u) == 255u && __hip_atomic_load(p.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                __builtin_amdgcn_s_sleep(2);
            }
            same = ((unsigned)g & 0xf) == my_xcc && (g >> 32) == 1ull;
        }
        same_l2 = __syncthreads_and(same) != 0;
        static_assert(kResMaxK <= kWave, "peer ids are read by the first K threads");
    }

    // Ranges of the gather vector produced by each peer: loop-invariant, but read through a pointer, so
    // inside the frame loop (whose barriers and publishes clobber memory) the compiler re-loaded them from
    // global memory in every frame and waited vmcnt(0) twice.  They live in LDS (a few words; keeping them
    // in registers unrolls the peer loop three times, and this kernel's code has to stay small: the 64 KiB
    // instruction cache is shared by two CUs that usually run the forward and the backward kernel).
    if (tid < kResMaxK - 1) {
        int lo = 0, n = 0;
        if (tid + 1 < K) { const int pj = (k + tid + 1) % K; lo = L.own_off[pj]; n = L.ex_cnt[pj]; }
        PL[2 * tid] = lo;
        PL[2 * tid + 1] = lo + n;
    }
    __syncthreads();
    // Everything loaded so far (the arc registers above all) has landed: tell the compiler, whose
    // wait-count bookkeeping otherwise carries "arc registers may still be in flight" into the loop and
    // answers it with vmcnt(0) right after the emission prefetch is issued (vmcnt(0), expcnt/lgkmcnt free).
    __builtin_amdgcn_s_waitcnt(0x0F70);

    // one frame; par = parity of i = which buffer is the gather source
    // MODE (compile-time, one copy of the loop per value; a workgroup runs exactly one): how a row result
    // reaches the peers -- 0: not at all (K = 1), 1: plain store into the shared L2, 2: write-through store.
    auto frame = [&](auto MODE, const int par, int i) __attribute__((always_inline)) {
        constexpr int mode = decltype(MODE)::value;
        const int t = DIR == 0 ? i : lx - 1 - i;                     // frame whose emissions are consumed
        const bool produce = DIR == 0 || t > 0;                      // a next vector exists
        [[maybe_unused]] const bool tm_on = b == p.b0 + 3 && i >= 100 && i < 228 && wave == 0;
        [[maybe_unused]] const int tm_i = (DIR * 4 + k) * 1024 + (i - 100) * 8;
        CRF_TM(tm_on, tm_i + 0);
#ifdef CRF_TIMING
        if (b == p.b0 + 3 && i >= 150 && i < 158 && wave == 0) CRF_TM(true, 12288 + 1024 + (DIR * 4 + k) * 8 + (i - 150));  // frame start
#endif
        float *Xn = X + (1 - par) * kResGmax;
        const float *EPu = EP + (DIR == 0 ? par : 1 - par) * Vp;     // e'_t (fwd) / e'_{t-1} (bwd)
        const int tpre = DIR == 0 ? t + 1 : t - 2;                   // emission row to prefetch
        const bool pre = DIR == 0 ? (t + 1 < lx) : (t >= 2);
        float epn[kEpRegsR];
        if (pre) {
            const float *er = p.ep + (bt0 + tpre) * V;
#pragma unroll
            for (int q = 0; q < kEpRegsR; ++q) { const int v = tid + q * kResThreads; epn[q] = v < V ? er[v] : 0.f; }
        }
        const int ksc = rescale_exp(res_frame_max(wm + par * kResWaves));
        const float sc = pow2f(ksc);
        float *Orow;
        if (DIR == 0) {
            E += ksc;                         // exponent of q_t
            if (tid == 0 && k == 0) p.Eout[bt0 + t] = E;
            E += kEpExp;                      // a_{t+1} = sum e'_t q_t carries the 2^kEpExp of e'_t
            Orow = p.Out + (bt0 + t) * p.Rout;
        } else {
            E += ksc + kEpExp;                // z_t = e'_t b_{t+1} carries the 2^kEpExp of e'_t
            if (t > 0 && tid == 0 && k == 0) p.Eout[bt0 + t - 1] = E;
            // b_t rows feed the grad pass as BP[t-1]; the last one (t = 0) only feeds logZ and goes to a spare row
            Orow = t > 0 ? p.Out + (bt0 + t - 1) * p.Rout : p.Row0 + (int64_t)b * p.Rout;
        }
        gu64 *slot = xch + (size_t)(1 - par) * G;
        const unsigned tag = (unsigned)(i + 1);
        const bool xchg = mode != 0 && produce;
        const char *xb = (const char *)lds + par * kResXB;
        // keep the slice-end mask opaque per frame: otherwise hipcc hoists all 30 loop-invariant
        // "bit c set?" conditions out of the time loop as 64-bit lane masks (60 SGPRs), spills them
        // to VGPR lanes and pays two v_readlane per chunk to get them back
        unsigned ends_f = ends;
        int nch_f = nch;
        asm volatile("" : "+s"(ends_f), "+s"(nch_f));
        f32x2 acc = {0.f, 0.f};
        float mymax = 0.f;
        // Row `rid` of the wave's current slice, kept as the byte offset r4 = 4*rid: the row's slot in the
        // per-frame HBM row, its label (RL), its LDS word and (times two) its exchange granule are all "uniform
        // base + r4" -- SGPR-base addressing, no per-epilogue 64-bit address arithmetic.  The entry produced by
        // row rid is implicit: rid + eoff (res_layout.cpp), so an epilogue needs ONE table value, the emission.
        unsigned r4 = (unsigned)(row0 + lane) * 4u;
        const int eoff = own0 - cu_row0;
        const char *RLb = (const char *)RL - (size_t)cu_row0 * 4;
        char *Xeb = (char *)(Xn + eoff);
        char *Ob = (char *)Orow;
        char *Sb = (char *)(slot + eoff);
#pragma unroll
        for (int c0 = 0; c0 < kResNCH; c0 += kResBatch) {
#if CRF_X_PRIO
            // issue priority by progress through the frame's chunks, as in fac_chain_body (two waves per SIMD here: the older one used to run ahead)
            if (c0 == 0) __builtin_amdgcn_s_setprio(3);
            else if (c0 == 2 * kResBatch) __builtin_amdgcn_s_setprio(2);
            else if (c0 == 3 * kResBatch) __builtin_amdgcn_s_setprio(1);
            else if (c0 == 4 * kResBatch) __builtin_amdgcn_s_setprio(0);
#endif
            if (c0 < nch_f) {
                // Row epilogues: two dependent LDS reads (label, then e'[label]).  (Tried and measured
                // slower, both of them: prefetching label and e' for ALL row ends of a batch ahead of the
                // gathers -- the second code path per batch cost more, in moves, branches and instruction-cache
                // misses, than the waves with many short slices gained; and a rolling prefetch, label when
                // the previous row ends and e' at every batch top: +4% on both kernels.)
                f32x2 g01[kResBatch], g23[kResBatch];
                CRF_RES_GATHER(g01, g23, A, xb, c0);
#pragma unroll
                for (int ci = 0; ci < kResBatch; ++ci) {
                    CRF_RES_CHUNK_ACC(acc, g01, g23, A, c0 + ci, ci);
                    if (__builtin_expect_with_probability((ends_f >> (c0 + ci) & 1u) != 0u, 0, 0.8)) {   // (the common path falls through: fac_chain_body)
                        const float rv = (acc.x + acc.y) * sc;       // q_t[row] (fwd) / b_t[state copy] (bwd)
                        *(float *)(Ob + r4) = rv;
                        const float av = EPu[*(const int *)(RLb + r4)] * rv;  // a_{t+1}[dst] (fwd) / z_{t-1}[pair] (bwd): final, one producer per entry
                        *(float *)(Xeb + r4) = av;
                        mymax = fmaxf(mymax, av);
                        if (mode != 0 && (DIR == 0 || produce)) {
                            const unsigned long long g = ((unsigned long long)tag << 32) | __float_as_uint(av);
                            gu64 *dst = (gu64 *)(Sb + 2u * r4);
                            if (mode == 1) asm volatile("global_store_dwordx2 %0, %1, off" : : "v"(dst), "v"(g) : "memory");
                            else __hip_atomic_store(dst, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                        acc = f32x2{0.f, 0.f};
                        r4 += kWave * 4u;
                    }
                }
            }
        }
        CRF_TM(tm_on, tm_i + 1);
#ifdef CRF_TIMING
        if (b == p.b0 + 3 && i >= 150 && i < 158) {  // per-wave compute end (8 frames), chunks and slices of the wave
            const int o = 12288 + ((DIR * 4 + k) * 8 + wave) * 16;
            CRF_TM(true, o + (i - 150));
            if (lane == 0 && i == 150) { g_tm[o + 8] = (unsigned long long)nch; g_tm[o + 9] = (unsigned long long)__builtin_popcount(ends); }
        }
#endif
        if (xchg) {  // the peers' entries (published from their row epilogues)
#pragma clang loop unroll(disable)
            for (int j = 0; j < K - 1; ++j) {
                const int lo = __builtin_amdgcn_readfirstlane(PL[2 * j]), hi = __builtin_amdgcn_readfirstlane(PL[2 * j + 1]);
                mymax = fmaxf(mymax, res_fetch(slot, Xn, lo, hi, tag, p.err, tid));
            }
        }
        CRF_TM(tm_on, tm_i + 2);
        mymax = wave_max(mymax);
        if (lane == 0) wm[(1 - par) * kResWaves + wave] = mymax;
        if (pre) {
            float *EPw = EP + (DIR == 0 ? 1 - par : par) * Vp;
#pragma unroll
            for (int q = 0; q < kEpRegsR; ++q) { const int v = tid + q * kResThreads; if (v < V) EPw[v] = epn[q]; }
        }
        CRF_TM(tm_on, tm_i + 3);
        sync_lds();
        CRF_TM(tm_on, tm_i + 4);
    };
    // NOT unrolled by two for compile-time buffer offsets: the gathers add an SGPR base either way, and the
    // doubled loop body (2 x 30 KiB) did not fit the instruction cache next to the other direction's kernel
    // (after frame 0: entries no row produces -- the start state -- still hold a_0 in buffer 0 and nothing rewrites
    // them; cleared before the buffer is the source again, see fac_chain_body)
    // (between two runs of the one frame loop -- frame 0 alone, then the rest: inside the loop body it cost 2.6 % of the step)
    auto run = [&](auto xmode) __attribute__((always_inline)) {
        int i = 0;
#pragma clang loop unroll(disable)
        for (int seg = 0; seg < 2; ++seg) {
            const int iend = (DIR == 0 && seg == 0) ? min(lx, 1) : lx;
#pragma clang loop unroll(disable)
            for (; i < iend; ++i) frame(xmode, i & 1, i);
            if (DIR == 0 && seg == 0 && lx > 0) {
                for (int s = tid; s < G; s += kResThreads) if (p.x_start[s] != 0.f) X[s] = 0.f;
                sync_lds();
            }
        }
    };
    if (K == 1) run(std::integral_constant<int, 0>{});
    else if (same_l2) run(std::integral_constant<int, 1>{});
    else run(std::integral_constant<int, 2>{});

    if (DIR == 0) {
        if (k == 0) {
            const float *Xf = X + (lx & 1) * kResGmax;
            float part = 0.f;
            for (int s = tid; s < G; s += kResThreads) part += Xf[s] * p.x_end[s];
            const float zs = res_block_sum(part, (float *)red, tid);
            const double mxs = res_mx_total(p, b, lx, red, tid);
            if (tid == 0) { p.den_zs[b] = zs; p.den_ez[b] = E; p.cost_alpha[b] = to_log(zs, E, mxs); if (!(zs > 0.f && zs < INFINITY)) p.redo[b] = 1; }
        }
    } else {
        if (lx > 0) {  // logZ from the backward side: sum_s start(s) b_0(s) over this CU's rows (written above)
            __syncthreads();  // drains vmcnt: this workgroup's own stores to the spare row are visible to it
            const float *r0 = p.Row0 + (int64_t)b * p.Rout;
            for (int r = tid; r < cu_rows; r += kResThreads) zpart += p.brow_start[cu_row0 + r] * r0[cu_row0 + r];
        }
        const float zb = res_block_sum(zpart, (float *)red, tid);
        if (tid == 0) p.cb_part[(size_t)b * kResMaxK + k] = zb;
        if (k == 0) {
            const double mxs = res_mx_total(p, b, lx, red, tid);
            if (tid == 0) { p.cb_F[b] = E; p.cb_mxs[b] = mxs; }
        }
    }
}

// Forward and backward recursion of a group of utterances as ONE grid (the first half of the blocks runs the forward
// recursion): one launch, one stream -- the loss no longer needs a hardware queue per recursion.
__global__ __launch_bounds__(kResThreads) void crf_res_pair_kernel(ResParams pf, ResParams pb) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int half = (int)gridDim.x >> 1;
    if ((int)blockIdx.x < half) res_chain_body<0>(pf, lds, (int)blockIdx.x, half);
    else res_chain_body<1>(pb, lds, (int)blockIdx.x - half, half);
}


// ---- explicit instantiations ----
// (crf_res_pair_kernel is not a template)

}  // namespace crf
