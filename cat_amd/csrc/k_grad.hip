// cat_amd/csrc/k_grad.hip -- the grad pass: generic, streaming denominator half, numerator half
// (one translation unit per kernel family, built in parallel by cat_amd/build.py; the explicit instantiations at the end are the
//  ones the host side in crf_host.hip launches -- a missing one is a link error, -Wl,-z,defs)
#include "crf_device.h"
#include "crf_kernels_decl.h"

namespace crf {

// ---------------------------------------------------------------------------------------------
// grad: one workgroup per (utterance, kGradFrames consecutive frames)
// LDS: prod[Pr] | csum[NC] | gd[Vp] | gc[Vp]
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kGradThreads) void crf_grad_kernel(LossParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const GraphDev &g = p.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int b = blockIdx.y, V = p.V, Vp = rup64(V);
    const int lx = p.lx[b];
    const bool do_den = p.c_den != 0.f && p.grad_phase != 2, do_ctc = p.c_ctc != 0.f && p.grad_phase != 1;
    const bool accumulate = p.grad_phase == 2;
    const int Rq = do_den ? p.Rq : 0, Rb = do_den ? p.Rb : 0, NC = do_den ? p.gNC : 0;
    const bool stage = p.grad_stage != 0;
    float *Qs = lds;                              // [Rq] staged q_t row       (when the rows fit in LDS)
    float *Bs = Qs + (stage ? rup64(Rq) : 0);     // [Rb] staged b_{t+1} row
    float *csum = Bs + (stage ? rup64(Rb) : 0);
    float *gd = csum + rup64(NC);
    float *gc = gd + Vp;
    const int64_t bt0 = (int64_t)b * p.T;
    float zs = 0.f;
    double zc = 0.0;
    int ez = 0, ezc = 0, Sx = 0;
    const int *ul = nullptr;
    if (do_den) { zs = p.den_zs[b]; ez = p.den_ez[b]; }
    if (do_ctc) { zc = ctc_zc_for_grad(p, b); ezc = p.ctc_ez[b]; Sx = 2 * p.ly[b] + 1; ul = p.labels + p.lab_off[b]; }
    const float inv = zs > 0.f ? 1.f / zs : 0.f;
    const double invc = zc > 0.0 ? 1.0 / zc : 0.0;

    const int t0 = blockIdx.x * kGradFrames;
    for (int t = t0; t < t0 + kGradFrames && t < p.T; ++t) {
        float *row = p.grad + (bt0 + t) * V;
        if (t >= lx) {
            if (!accumulate)
                for (int v = tid; v < V; v += kGradThreads) row[v] = 0.f;
            continue;
        }
        if (do_den) {
            const float *Qr = p.Q + (bt0 + t) * Rq, *Br = p.BP + (bt0 + t) * Rb;
            if (stage) {
                for (int r = tid; r < Rq; r += kGradThreads) Qs[r] = Qr[r];
                for (int r = tid; r < Rb; r += kGradThreads) Bs[r] = Br[r];
                __syncthreads();
            }
            const float *Qg = stage ? Qs : Qr, *Bg = stage ? Bs : Br;
            for (int c = tid; c < NC; c += kGradThreads) {
                float s = 0.f;
                for (int j = p.gchunk[c]; j < p.gchunk[c + 1]; ++j) s += Qg[p.gq[j]] * Bg[p.gb[j]];
                csum[c] = s;
            }
            __syncthreads();
            const int e = ez - p.EQ[bt0 + t] - p.EB[bt0 + t] - kEpExp;  // er[] carries 2^kEpExp
            const float *er = p.ep + (bt0 + t) * V;
            for (int v = tid; v < V; v += kGradThreads) {
                float s = 0.f;
                if (v <= g.max_label)
                    for (int c = p.glab[v]; c < p.glab[v + 1]; ++c) s += csum[c];
                gd[v] = er[v] * (ldexpf(s, e) * inv);
            }
        }
        if (do_ctc) {
            for (int v = tid; v < V; v += kGradThreads) gc[v] = 0.f;
            __syncthreads();
            if (zc > 0.0) {
                const double *Ar = p.CA + (bt0 + t) * p.Sc, *Br = p.CB + (bt0 + t) * p.Sc;
                const double fc = ctc_frame_factor(p, b, bt0 + t, invc, ezc);
                float blank = 0.f;
                for (int s = tid; s < Sx; s += kGradThreads) {
                    const float pr = (float)(Ar[s] * Br[s] * fc);  // a posterior, in [0,1]
                    if (s & 1) atomicAdd(&gc[ul[s >> 1]], pr);
                    else blank += pr;
                }
                blank = wave_sum(blank);
                if (lane == 0) atomicAdd(&gc[0], blank);
            }
        }
        __syncthreads();
        for (int v = tid; v < V; v += kGradThreads) {
            float o = accumulate ? row[v] : 0.f;
            if (do_den) o = p.c_den * gd[v];
            if (do_ctc) o -= p.c_ctc * gc[v];
            // fused log_softmax: d/dx = d/dlogp - softmax(x) * sum_v d/dlogp[v]; the posteriors of a frame sum to 1
            if (do_ctc && p.fused) o -= (p.c_den - (zc > 0.0 ? p.c_ctc : 0.f)) * (p.ep[(bt0 + t) * V + v] * pow2f(-kEpExp)) * p.inv_s[bt0 + t];
            row[v] = o;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// grad, denominator half, streaming form: gamma_den[t][v] = e'_t[v] * sum_{j in label v} Q_t[gq_j] * BP_t[gb_j] / Z.
// One workgroup walks kGDFrames consecutive frames of one utterance.  The (gq, gb) index pairs of a
// thread's chunk(s) are loaded ONCE into registers (packed 16+16 bit) -- in the generic kernel they were
// re-read from L2 for every frame and doubled its traffic; the rows of frame t+1 are prefetched into
// registers while frame t is reduced out of LDS.  HBM-bound: two coalesced rows per frame.
// ---------------------------------------------------------------------------------------------
// Nothing in the frame loop may wait on global memory except for data requested a whole frame earlier:
// the rows AND the emission row of frame t+1 are requested while frame t is reduced, the per-frame
// exponents are read once per workgroup, and the barriers are LDS-only (sync_lds) -- __syncthreads()
// would drain vmcnt, i.e. wait for the prefetch it has just issued (that alone was ~2/3 of this kernel).
// NT threads: 256, or 512 for graphs whose rows do not fit 256 threads' prefetch registers (5 float4 each per row)
// CH: entries per chunk the index registers hold (kChunk; 8 for graphs with few pairs per label -- V = 500: ~8 -- whose chunk lists the
// graph compiler cuts at 8: a chunk of 32 slots with 8 pairs spends three quarters of its gathers on padding)
// RR: float4 registers per thread and row (rows of up to 4 * RR * NT floats); WPE: waves per SIMD the register budget is held to (4 = 128 VGPRs:
// two 512-thread workgroups per CU)
// The mass checks of frame `tt` of the grad den pass (see the comment at the normaliser): on the SCALAR unit, on the floats' bits (non-negative
// floats order like integers; times 2^-82 = 82 off the exponent field), collected in one uniform word
#define CRF_GD_CHECK(tt)                                                                                                     \
    do {                                                                                                                     \
        const unsigned nvb = (unsigned)__builtin_amdgcn_readfirstlane(__float_as_int(nrm[(tt) & 3]));                         \
        const unsigned emb = (unsigned)__builtin_amdgcn_readfirstlane(__float_as_int(nrm[8 + ((tt) & 3)]));                   \
        const unsigned thr = emb > (82u << 23) ? emb - (82u << 23) : 0u;                                                     \
        frame_bad |= (unsigned)(!(nvb >= 0x03800000u && nvb < 0x7f800000u && nvb >= thr));                                   \
    } while (0)
template <int NCPT, int EPR, int NT, int CH, int RR, int WPE>
// (Registers, a property of the SCHEDULE: they are granted in eights, a SIMD has 512 per lane.  The instantiation of the metric graph, <1, 1>, runs
// three workgroups to a CU (one wave each per SIMD) and, for batches whose grad pass is not staged (B >= 128, two CUs per recursion), BESIDE the
// numerator chains, whose workgroup puts two waves on every SIMD: 2 x 160 + 2 x 96 = 512.  Round 5 found it the hard way: this kernel went from 152
// to 153 registers (-> 160) while the chains were at 97 (-> 104): a chain workgroup then fitted beside ONE grad workgroup only, the grad launch's
// queue never lets a CU fall that low, and the chains ran AFTER the grad pass: B = 128 4.8 -> 5.6 ms per step, found by bisecting the round's own
// commits (profiles/round5_ab_one_register.txt).  Now: chains <= 96 (crf_ctc_pair_kernel<1>, held by its launch bounds), this kernel <= 160;
// tests/test_isa_checks.py holds both numbers on the compiler's own metadata.  amdgpu_num_vgpr is ignored by this compiler.)
__global__ __launch_bounds__(NT, WPE) void crf_grad_den_kernel(LossParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const GraphDev &g = p.g;
    const int tid = threadIdx.x;
    int b = blockIdx.y;
    int blk = blockIdx.x;
    int stg = p.gd_stage, nfc = p.gd_nf, fpb = kGDFrames, sub = 0;
    if (p.gd_persist) {
        // ONE launch for all remaining stages (round 5).  Per-stage launches behind stream-level waits cost the pass ~5 us per wait even when the
        // counter had long passed, and every launch ends on a partly empty device (1 024 workgroups on 384 slots: 2.67 rounds) before the next may
        // begin: 232 us per 128-iteration stage against the 219 us the recursions take to release one -- the pass fell 14 us further behind
        // with every stage and reached the recursions' end 75 us late with the whole last stage still to do (profiles/round5_ab_grad_one_launch.txt).
        // Here the grid is 1-D and stage-major (the dispatcher hands out workgroups in that order), a workgroup finds its stage, leaves at once if
        // its block is not of that stage, sets up, and only then waits (one lane, s_sleep between polls) for the counter the den kernels bump: the
        // last stage's workgroups sit ready when the recursions end.  No deadlock: the launch is enqueued behind stage 1's stream-level wait, i.e.
        // when every den workgroup has run half of its frames -- all of them hold their CUs and need nothing from this kernel.
        stg = p.gd_stage;
        while (stg + 1 < p.gd_nb && blk >= p.gd_poff[stg + 1]) ++stg;
        nfc = (p.gd_bound[stg] - p.gd_bound[stg - 1] + kGDFrames - 1) / kGDFrames + 3;
        fpb = p.gd_fpb[stg];
        // candidate-major, the utterances side by side: workgroup i runs on XCD i % 8, and utterance-major with 2 * nfc = 16 candidates
        // (pieces of 80 iterations) put every utterance's candidate j on the same XCD -- the five with work on five XCDs, three XCDs idle:
        // 2.73 -> 2.92 ms per step (profiles/round5_ab_grad_one_launch.txt)
        const int nsub = kGDFrames / fpb;
        const int r = blk - p.gd_poff[stg];
        const int r2 = r / p.B;
        b = r - r2 * p.B;
        blk = r2 / nsub;
        sub = r2 - blk * nsub;
        // (uniform, but integer division runs on the vector unit: back into scalar registers)
        b = __builtin_amdgcn_readfirstlane(b); blk = __builtin_amdgcn_readfirstlane(blk); sub = __builtin_amdgcn_readfirstlane(sub);
        stg = __builtin_amdgcn_readfirstlane(stg); nfc = __builtin_amdgcn_readfirstlane(nfc); fpb = __builtin_amdgcn_readfirstlane(fpb);
    }
    const int V = p.V;
    const int lx = p.lx[b], Rq = p.Rq, Rb = p.Rb, NC = p.gNC;
    if (nfc > 0) {
        // Compact launch of stage k > 1: the blocks a stage completes are two short runs -- those whose last frame
        // the FORWARD recursion reached in this stage, from block bound[k-1]/16 on, and those whose first frame the
        // BACKWARD recursion reached, from block (lx-1-bound[k])/16 on -- so the launch holds gd_nf candidates of
        // each run instead of every block of the utterance (94 per utterance, of which 6 had work: each of the
        // others held a workgroup slot and 37 KB of LDS for the ~2 us it takes to find that out).  The runs are
        // taken one block wider than needed on both sides; the exact test below decides, and a block of both runs
        // is taken by the forward one.
        const int k = stg, nf = nfc;
        const int flo = p.gd_bound[k - 1] / kGDFrames - 1;
        const int blo = (lx - 1 - p.gd_bound[k]) / kGDFrames - 1;
        if (blk < nf) blk = flo + blk;
        else {
            blk = blo + (blk - nf);
            if (blk >= flo && blk < flo + nf) return;
        }
        if (blk < 0 || blk * kGDFrames >= p.T) return;
    }
    const int Vp = rup64(V);
    float *Qs = lds, *Bs = Qs + rup64(Rq + 1), *gd = Bs + rup64(Rb + 1);    // Qs[Rq] = 0: target of padding index pairs
    float *nrm = gd + 4 * Vp;                                               // [4] per-frame normalisers; gd: [4][Vp] label sums, both in rotation
    int *clab_s = (int *)(nrm + kGDFrames);                                 // [NC] label of each chunk (prologue only)
    const int64_t bt0 = (int64_t)b * p.T;
    int t0 = blk * kGDFrames, t1 = min(t0 + kGDFrames, p.T), tl = min(t1, lx);
    if (p.gd_stage > 0) {
        // Staged mode: the den recursions run in segments (iteration bounds gd_bound[]), and after segment k an event
        // releases the launch with gd_stage = k.  A block belongs to the FIRST stage at which both its Q rows
        // (forward has passed frame tl) and its BP rows (backward has come down to frame t0) exist; the other
        // stages' launches skip it.  (No waiting inside kernels: see crf_loss_fwd_bwd.)
        int sf = 1, sb = 1;                                        // first segment that has run `tl` / `lx-1-t0` iterations
        for (int k = 1; k < p.gd_nb; ++k) {                        // (BP[t0] is stored by iteration lx-2-t0; BP[lx-1] by the set-up)
            if (p.gd_bound[k] < tl) sf = k + 1;
            if (p.gd_bound[k] < lx - 1 - t0) sb = k + 1;
        }
        const int mine = t0 < tl ? max(sf, sb) : 1;                // blocks past the utterance: first stage
        if (mine != stg) return;
        if (fpb < kGDFrames) {                                     // this workgroup's share of the block
            t0 += sub * fpb;
            t1 = min(t0 + fpb, t1);
            tl = min(t1, lx);
            if (t0 >= tl) return;                                  // (stages > 1 add to rows the numerator half has written: nothing to zero)
        }
    }
    t0 = __builtin_amdgcn_readfirstlane(t0); t1 = __builtin_amdgcn_readfirstlane(t1); tl = __builtin_amdgcn_readfirstlane(tl);

    unsigned idx[NCPT][CH];
    constexpr int HS = CH < 16 ? CH : 16;
    {
        // unconditional (clamped) loads, selected afterwards: predicated loads were issued one at a time,
        // 32 L2 round trips in a row before the first frame
        const int nlist = p.gchunk[NC];
#pragma unroll
        for (int i = 0; i < NCPT; ++i) {
            const int c = tid + i * NT;
            const int j0 = c < NC ? p.gchunk[c] : 0;
            const int clen = c < NC ? p.gchunk[c + 1] - j0 : 0;
#pragma unroll
            for (int h = 0; h < CH; h += HS) {   // in halves of 16: 64 loads in flight were the register peak of the kernel
                unsigned short gqv[HS], gbv[HS];
#pragma unroll
                for (int j = 0; j < HS; ++j) {
                    const int jj = min(j0 + h + j, nlist - 1);
                    gqv[j] = (unsigned short)p.gq[jj];
                    gbv[j] = (unsigned short)p.gb[jj];
                }
#pragma unroll
                for (int j = 0; j < HS; ++j) idx[i][h + j] = h + j < clen ? ((unsigned)gqv[j] | (unsigned)gbv[j] << 16) : (unsigned)Rq;
            }
        }
    }
    // label of each chunk: the per-label chunk ranges, inverted once per workgroup.  The chunk sums of a
    // label are combined with LDS float adds -- a per-label loop over its chunks made one thread (the blank
    // label owns a third of all pairs) walk 64 chunks in every frame, half the time of this kernel.
    for (int v = tid; v <= g.max_label && v < V; v += NT)
        for (int c = p.glab[v]; c < p.glab[v + 1]; ++c) clab_s[c] = v;
    for (int v = tid; v < 4 * Vp; v += NT) gd[v] = 0.f;
    if (tid == 0) Qs[Rq] = 0.f;
    __syncthreads();
    // Chunks are label-sorted, so the lanes of a wave that share a label are neighbours: a segmented
    // shuffle reduction (the "same label d lanes up" tests are static, bit j of segm) leaves one LDS add per
    // (wave, label) -- 64 lanes adding to ONE address (the blank label) took ~50 cycles per lane.
    int clab[NCPT];
    unsigned segm[NCPT];
    const int lane = tid & 63;
#pragma unroll
    for (int i = 0; i < NCPT; ++i) {
        const int c = tid + i * NT;
        clab[i] = c < NC ? clab_s[c] : -1;
        segm[i] = 0;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int d = 1 << j;
            const int other = (lane + d < 64 && c + d < NC) ? clab_s[c + d] : -2;
            if (other == clab[i]) segm[i] |= 1u << j;
        }
        const bool head = lane == 0 || c >= NC || clab_s[c - 1] != clab[i];
        if (!head || c >= NC) segm[i] |= 1u << 31;   // bit 31: not the lane that adds the segment's sum
    }
    if (tid < 12) nrm[tid] = 0.f;
    // rows are multiples of 64 floats and 256-byte aligned: 16-byte loads, prefetched one frame ahead
    // (one chunk and one emission register per thread only -- the kernel of graphs over <= 256 classes with <= 256 label chunks: V = 72 step 2.89 -> 2.865 ms;
    // in the other instantiations the same reordering was 4 % slower at V = 217 / 500 while the copy-behind-the-loads wait described below was still in it, and
    // makes no difference without it (levels 2 / 3 of the switch): profiles/round4_ab_grad_pass_variants.txt, round4_ab_waits_found_in_the_isa.txt)
    constexpr bool GDE = CRF_X_GDEARLY != 0 && WPE == 1 && ((NCPT == 1 && (EPR == 1 || CRF_X_GDEARLY >= 2)) || CRF_X_GDEARLY >= 3);
    constexpr bool GDM = GDE && CRF_X_GDMOVE != 0;
    f32x4 qr[RR], br[RR];
    unsigned frame_bad = 0;          // (uniform: a frame of this block failed the mass checks below)
    // where a frame's checks are made: behind its normaliser, or -- the metric graph's instantiation, which has no register to spare there (see the
    // kernel's head) -- at the top of the next frame (the others grew by 20 registers or spilled when theirs were moved: measured on the compiler's report)
    constexpr bool kCheckTop = NCPT == 1 && EPR == 1;
    float ern[EPR], rwn[EPR] = {};   // next frame's emissions and (accumulate mode) grad row
#define CRF_GD_FETCH(t)                                                                                  \
    {                                                                                                    \
        const f32x4 *Qr = (const f32x4 *)(p.Q + (bt0 + (t)) * Rq), *Br = (const f32x4 *)(p.BP + (bt0 + (t)) * Rb); \
        _Pragma("unroll") for (int i = 0; i < RR; ++i) {                                         \
            const int r = tid + i * NT;                                                          \
            qr[i] = 4 * r < Rq ? Qr[r] : f32x4{0.f, 0.f, 0.f, 0.f};                                      \
            br[i] = 4 * r < Rb ? Br[r] : f32x4{0.f, 0.f, 0.f, 0.f};                                      \
        }                                                                                                \
        const float *er_ = p.ep + (bt0 + (t)) * V;                                                       \
        const float *gr_ = p.grad + (bt0 + (t)) * V;                                                     \
        _Pragma("unroll") for (int q = 0; q < EPR; ++q) {                                                \
            if constexpr (GDM) {   /* clamped, not predicated (every use is behind v < V): the loads write the loop registers themselves */ \
                const int v = min(tid + q * NT, V - 1);                                                  \
                ern[q] = er_[v];                                                                         \
                if (p.grad_den_acc == 1) rwn[q] = gr_[v];   /* (else: stays 0) */                        \
            } else {                                                                                     \
                const int v = tid + q * NT;                                                              \
                ern[q] = v < V ? er_[v] : 0.f;                                                           \
                rwn[q] = (p.grad_den_acc == 1 && v < V) ? gr_[v] : 0.f;                                  \
            }                                                                                            \
        }                                                                                                \
    }
#define CRF_GD_STAGE()                                                                                  \
    _Pragma("unroll") for (int i = 0; i < RR; ++i) {                                             \
        const int r = tid + i * NT;                                                              \
        if (4 * r < Rq) ((f32x4 *)Qs)[r] = qr[i];                                                        \
        if (4 * r < Rb) ((f32x4 *)Bs)[r] = br[i];                                                        \
    }
    if (p.gd_persist) {
        // set up; now the stage's rows.  The counter is bumped behind a drain of every wave's row stores and a barrier (publish_stage), the
        // words are uncached in L2; an agent-scope acquire before the first row load (rows of the call before may sit in this XCD's L2).
        if (tid == 0) {
            const int *c = p.gd_cnt + stg;
            // bounded in WALL-CLOCK time (s_memrealtime: 100 MHz, whatever the shader clock does): 2 s, far beyond any recursion; a time-out sets
            // the error word (the call's loss becomes NaN) and a pinned host word (the context goes back to per-stage launches)
            const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
            for (unsigned spins = 0; __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < p.gd_target; ++spins) {
                if ((spins & 63u) == 63u) {
                    if (__hip_atomic_load(p.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                    if (__builtin_amdgcn_s_memrealtime() - t_start > 200000000ull) {
                        __hip_atomic_store(p.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (p.gd_timeout_host) __hip_atomic_store(p.gd_timeout_host, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        break;
                    }
                }
                __builtin_amdgcn_s_sleep(127);
            }
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        // a workgroup that leaves on the error word must not read rows that are not there and add garbage to the gradient: the rows keep what
        // the numerator half wrote, the loss is NaN (crf_finalize_kernel)
        if (__hip_atomic_load(p.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    }
    float erc[EPR], rwc[EPR];
    if constexpr (GDE) {
        // Round 4: the rows of frame t+2 are requested as soon as frame t+1's have left the registers for the LDS (behind the first barrier of
        // frame t) instead of at the top of frame t+1 -- a normalise-and-store phase and a barrier earlier.  The timing build had shown the
        // pass waiting ~1 800 of a frame's 8 600 cycles for rows requested only ~4 000 cycles before (issue 2 000 -- the requests queue --,
        // gathers 2 000): memory latency under load is ~2.4 us.  The emission / grad-row registers get a third set (cur, next, in flight).
        float erx[EPR], rwx[EPR];   // frame t+1's emissions and grad row, landed (ern / rwn: in flight for t+2)
        if (t0 < tl) {
            CRF_GD_FETCH(t0);
            CRF_GD_STAGE();
#pragma unroll
            for (int q = 0; q < EPR; ++q) { erc[q] = ern[q]; rwc[q] = rwn[q]; }
            if (t0 + 1 < tl) CRF_GD_FETCH(t0 + 1);
        }
        sync_lds();
        [[maybe_unused]] const bool tm_on = blk == 46 && b == 3 && tid < 64;   // (T = 1500: a block of the first stage, the middle of utterance 3)
        for (int t = t0; t < tl; ++t) {
            CRF_TM(tm_on, 8192 + (t - t0) * 8 + 0);
            float *gsum = gd + (t & 3) * Vp, *gzero = gd + ((t + 2) & 3) * Vp;
            if (kCheckTop && t > t0) CRF_GD_CHECK(t - 1);   // (frame t-1's, here, where few registers live)
            CRF_TM(tm_on, 8192 + (t - t0) * 8 + 1);
#pragma unroll
            for (int i = 0; i < NCPT; ++i) {
                float s0 = 0.f, s1 = 0.f;
#pragma unroll
                for (int j = 0; j < CH; j += 2) {  // padding pairs read Qs[Rq] = 0 (times Bs[0])
                    s0 = fmaf(Qs[idx[i][j] & 0xffffu], Bs[idx[i][j] >> 16], s0);
                    s1 = fmaf(Qs[idx[i][j + 1] & 0xffffu], Bs[idx[i][j + 1] >> 16], s1);
                }
                float sv = s0 + s1;
#pragma unroll
                for (int j = 0; j < 6; ++j) {   // suffix sums within the label segment: lane gets sum over [lane, segment end]
                    const float o = __shfl_down(sv, 1 << j, 64);
                    if (segm[i] >> j & 1u) sv += o;
                }
                if (!(segm[i] >> 31)) {
                    if constexpr (WPE > 1) {   // the label re-read from the LDS: as a register it was spilled, and a scratch reload waits for the row prefetch
                        int lab;
                        asm volatile("v_lshl_add_u32 %0, %1, 2, %2\n\tds_read_b32 %0, %0\n\ts_waitcnt lgkmcnt(0)"
                                     : "=&v"(lab) : "v"(tid), "s"((unsigned)(uintptr_t)(clab_s + i * NT)) : "memory");   // (the address formed here: as a value it was spilled too)
                        atomicAdd(&gsum[lab], sv);
                    } else atomicAdd(&gsum[clab[i]], sv);
                }
            }
#pragma unroll
            for (int q = 0; q < EPR; ++q) {  // cleared two frames ahead: its last readers are behind frame t-1's barrier
                const int v = tid + q * NT;
                if (v < V) gzero[v] = 0.f;
            }
            if (tid == 0) { nrm[(t + 2) & 3] = 0.f; nrm[8 + ((t + 2) & 3)] = 0.f; }
            CRF_TM(tm_on, 8192 + (t - t0) * 8 + 2);
            sync_lds();                             // every gather of frame t is done: the row buffers are free
            CRF_TM(tm_on, 8192 + (t - t0) * 8 + 3);
            if (t + 1 < tl) {
                CRF_GD_STAGE();                     // (the compiler's vmcnt wait in front of these LDS writes: the rows of frame t+1 -- and frame t-1's row store)
#pragma unroll
                for (int q = 0; q < EPR; ++q) {
                    // real moves, here: left to the compiler, the copy became "new ern -> its loop register" BEHIND the loads below, with a
                    // vmcnt(0) in front of it -- the frame waited for the rows of t+2 the moment it had asked for them
                    if constexpr (GDM) asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=&v"(erx[q]), "=&v"(rwx[q]) : "v"(ern[q]), "v"(rwn[q]));
                    else { erx[q] = ern[q]; rwx[q] = rwn[q]; }
                }
                if (t + 2 < tl) CRF_GD_FETCH(t + 2);
            }
            CRF_TM(tm_on, 8192 + (t - t0) * 8 + 4);
            float u[EPR], part = 0.f, emx = 0.f;
#pragma unroll
            for (int q = 0; q < EPR; ++q) {
                const int v = tid + q * NT;
                u[q] = v < V ? (erc[q] * pow2f(-kGradDescale)) * gsum[v] : 0.f;
                part += u[q];
                if (CRF_X_GCHK) emx = fmaxf(emx, v <= g.max_label && v < V ? erc[q] : 0.f);
            }
            part = wave_sum(part);
            if (CRF_X_GCHK) emx = wave_max(emx);
            if ((tid & 63) == 0 && part != 0.f) atomicAdd(&nrm[t & 3], part);
            if (CRF_X_GCHK && (tid & 63) == 0 && emx > 0.f) atomicMax((int *)&nrm[8 + (t & 3)], __float_as_int(emx));   // (non-negative floats order like their bits)
            CRF_TM(tm_on, 8192 + (t - t0) * 8 + 5);
            sync_lds();                             // rows of frame t+1 visible, normaliser complete
            CRF_TM(tm_on, 8192 + (t - t0) * 8 + 6);
            const float nv = nrm[t & 3];
            const float inv = nv > 0.f ? p.c_den / nv : 0.f;
            // The utterance goes to the log-domain fallback when a frame's mass is not a NORMAL positive float -- zero, denormal (c / nv = inf,
            // inf * 0 = NaN), inf, NaN -- or when products the rows can no longer hold could have mattered: a pair whose q (or b) lies below
            // 2^-126 is lost from rows scaled to 2^20, i.e. terms below 2^-105, at most 2^-92 of them together; a lost term weighs at most
            // e'max * 2^-4 * 2^-105 with e'max the largest emission of a label the graph has.  A frame mass below e'max * 2^-82 could be missing
            // more than 1e-4 of itself -- forward and backward mass ~100 nats apart, or two alignments, one through the frame's best label
            // and one 70 nats below it whose rows are the healthy ones (tests/test_gpu_fuzz.py, round 5: posteriors 0 / 1 instead of
            // 0.94 / 0.06 in single frames, costs exact).  CRF_GD_CHECK; the flag is stored once, behind the loop.
            if (!kCheckTop) CRF_GD_CHECK(t);
            float *row = p.grad + (bt0 + t) * V;
#pragma unroll
            for (int q = 0; q < EPR; ++q) {
                const int v = tid + q * NT;
                if (v < V) { if (p.grad_den_acc == 2) unsafeAtomicAdd(row + v, u[q] * inv); else row[v] = rwc[q] + u[q] * inv; }   // rwc = 0 unless accumulating onto the numerator half
                erc[q] = erx[q]; rwc[q] = rwx[q];
            }
        }
    } else {
        if (t0 < tl) {
            CRF_GD_FETCH(t0);
            CRF_GD_STAGE();
#pragma unroll
            for (int q = 0; q < EPR; ++q) { erc[q] = ern[q]; rwc[q] = rwn[q]; }
        }
        sync_lds();
        [[maybe_unused]] const bool tm_on = blk == 46 && b == 3 && tid < 64;   // (T = 1500: a block of the first stage, the middle of utterance 3)
        for (int t = t0; t < tl; ++t) {
            CRF_TM(tm_on, 8192 + (t - t0) * 8 + 0);
            float *gsum = gd + (t & 3) * Vp, *gzero = gd + ((t + 2) & 3) * Vp;
            if (kCheckTop && t > t0) CRF_GD_CHECK(t - 1);   // (frame t-1's, here, where few registers live)
            if (t + 1 < tl) CRF_GD_FETCH(t + 1);   // lands while frame t is reduced
            CRF_TM(tm_on, 8192 + (t - t0) * 8 + 1);
#pragma unroll
            for (int i = 0; i < NCPT; ++i) {
                float s0 = 0.f, s1 = 0.f;
#pragma unroll
                for (int j = 0; j < CH; j += 2) {  // padding pairs read Qs[Rq] = 0 (times Bs[0])
                    s0 = fmaf(Qs[idx[i][j] & 0xffffu], Bs[idx[i][j] >> 16], s0);
                    s1 = fmaf(Qs[idx[i][j + 1] & 0xffffu], Bs[idx[i][j + 1] >> 16], s1);
                }
                float sv = s0 + s1;
#pragma unroll
                for (int j = 0; j < 6; ++j) {   // suffix sums within the label segment: lane gets sum over [lane, segment end]
                    const float o = __shfl_down(sv, 1 << j, 64);
                    if (segm[i] >> j & 1u) sv += o;
                }
                if (!(segm[i] >> 31)) {
                    if constexpr (WPE > 1) {   // the label re-read from the LDS: as a register it was spilled, and a scratch reload waits for the row prefetch
                        int lab;
                        asm volatile("v_lshl_add_u32 %0, %1, 2, %2\n\tds_read_b32 %0, %0\n\ts_waitcnt lgkmcnt(0)"
                                     : "=&v"(lab) : "v"(tid), "s"((unsigned)(uintptr_t)(clab_s + i * NT)) : "memory");   // (the address formed here: as a value it was spilled too)
                        atomicAdd(&gsum[lab], sv);
                    } else atomicAdd(&gsum[clab[i]], sv);
                }
            }
#pragma unroll
            for (int q = 0; q < EPR; ++q) {  // cleared two frames ahead: its last readers are behind frame t-1's barrier
                const int v = tid + q * NT;
                if (v < V) gzero[v] = 0.f;
            }
            if (tid == 0) { nrm[(t + 2) & 3] = 0.f; nrm[8 + ((t + 2) & 3)] = 0.f; }
            CRF_TM(tm_on, 8192 + (t - t0) * 8 + 2);
            sync_lds();                             // every gather of frame t is done: the row buffers are free
            CRF_TM(tm_on, 8192 + (t - t0) * 8 + 3);
            // Stage frame t+1 BEFORE this frame's stores are issued: the vmcnt wait in front of the LDS writes
            // then covers loads only (vmcnt counts in order; behind the stores it would also wait for their
            // acknowledgement).
            if (t + 1 < tl) CRF_GD_STAGE();
            CRF_TM(tm_on, 8192 + (t - t0) * 8 + 4);
            // gamma[t][v] = u_v / sum_v u_v with u_v = e'_t[v] * (label sum): the posteriors of a frame sum to 1, so
            // the frame normalises itself -- no logZ, no per-frame exponents, hence no dependence on the END of the
            // recursions (the pass runs beside them).  e' is taken without its 2^kEpExp (range: label sums reach 2^50).
            float u[EPR], part = 0.f, emx = 0.f;
#pragma unroll
            for (int q = 0; q < EPR; ++q) {
                const int v = tid + q * NT;
                u[q] = v < V ? (erc[q] * pow2f(-kGradDescale)) * gsum[v] : 0.f;
                part += u[q];
                if (CRF_X_GCHK) emx = fmaxf(emx, v <= g.max_label && v < V ? erc[q] : 0.f);
                erc[q] = ern[q];
            }
            float rw[EPR];
#pragma unroll
            for (int q = 0; q < EPR; ++q) { rw[q] = rwc[q]; rwc[q] = rwn[q]; }
            part = wave_sum(part);
            if (CRF_X_GCHK) emx = wave_max(emx);
            if ((tid & 63) == 0 && part != 0.f) atomicAdd(&nrm[t & 3], part);
            if (CRF_X_GCHK && (tid & 63) == 0 && emx > 0.f) atomicMax((int *)&nrm[8 + (t & 3)], __float_as_int(emx));
            CRF_TM(tm_on, 8192 + (t - t0) * 8 + 5);
            sync_lds();                             // rows of frame t+1 visible, normaliser complete
            CRF_TM(tm_on, 8192 + (t - t0) * 8 + 6);
            const float nv = nrm[t & 3];
            const float inv = nv > 0.f ? p.c_den / nv : 0.f;
            // The utterance goes to the log-domain fallback when a frame's mass is not a NORMAL positive float -- zero, denormal (c / nv = inf,
            // inf * 0 = NaN), inf, NaN -- or when products the rows can no longer hold could have mattered: a pair whose q (or b) lies below
            // 2^-126 is lost from rows scaled to 2^20, i.e. terms below 2^-105, at most 2^-92 of them together; a lost term weighs at most
            // e'max * 2^-4 * 2^-105 with e'max the largest emission of a label the graph has.  A frame mass below e'max * 2^-82 could be missing
            // more than 1e-4 of itself -- forward and backward mass ~100 nats apart, or two alignments, one through the frame's best label
            // and one 70 nats below it whose rows are the healthy ones (tests/test_gpu_fuzz.py, round 5: posteriors 0 / 1 instead of
            // 0.94 / 0.06 in single frames, costs exact).  CRF_GD_CHECK; the flag is stored once, behind the loop.
            if (!kCheckTop) CRF_GD_CHECK(t);
            float *row = p.grad + (bt0 + t) * V;
#pragma unroll
            for (int q = 0; q < EPR; ++q) {
                const int v = tid + q * NT;
                if (v < V) { if (p.grad_den_acc == 2) unsafeAtomicAdd(row + v, u[q] * inv); else row[v] = rw[q] + u[q] * inv; }   // rw = 0 unless accumulating onto the numerator half
            }
        }
    }
#undef CRF_GD_STAGE
#undef CRF_GD_FETCH
    if (kCheckTop && tl > t0) CRF_GD_CHECK(tl - 1);   // (the last frame's: its normaliser was complete at the last barrier)
    if (frame_bad && tid == 0 && p.redo) p.redo[b] = 1;
#undef CRF_GD_CHECK
    if (!p.grad_den_acc)
        for (int t = max(t0, tl); t < t1; ++t) {  // frames past the utterance's length: zero rows
            float *row = p.grad + (bt0 + t) * V;
            for (int v = tid; v < V; v += NT) row[v] = 0.f;
        }
}

// ---------------------------------------------------------------------------------------------
// grad, numerator half, streaming form: grad[b][t][v] (-)= c_ctc * gamma_ctc[b][t][v],
// gamma_ctc[t][v] = sum_{s: l'_s = v} A_t[s] * Bx_t[s] / Z  (gpu_ctc_kernels.h:377-458 computes the same
// posterior per unique label after an in-kernel sort; here labels are scattered with LDS float adds).
// Same rules as crf_grad_den_kernel: everything a frame needs from global memory (the two fp64 rows,
// the grad row it accumulates into) is requested one frame ahead, the per-frame factors are read once
// per workgroup, barriers are LDS-only.  grad_phase 2 = subtract from the row the den half wrote,
// otherwise write -c_ctc * gamma (plain CTC).  blockDim is even, so a thread's s-values all have the
// parity of its tid: odd threads own label positions, even threads blanks.
// ---------------------------------------------------------------------------------------------
// REGS: label positions per thread (2L+1 <= REGS * 256).  The kernel is latency-bound (a workgroup walks its 16
// frames one after the other), so what counts is how many workgroups a CU holds: with 16 positions per thread (three
// fp64 arrays) that is 3, with 2 positions -- utterances of up to 255 labels -- 8.
template <int REGS>
__global__ __launch_bounds__(kGCThreads) void crf_grad_ctc_kernel(LossParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int b = blockIdx.y, V = p.V, Vp = rup64(V);
    const int lx = p.lx[b];
    const bool accumulate = p.grad_phase == 2;
    const bool atomic = p.grad_phase == 3;    // the row starts at zero (prep) and the den half adds its part from another stream: add, never read
    float *gc = lds;                          // [4][Vp] in rotation
    double *fcs = (double *)(gc + 4 * Vp);    // [kGCFrames]
    float *gt = (float *)(fcs + kGCFrames);   // [4] in rotation: a frame's posteriors must sum to one
    const int64_t bt0 = (int64_t)b * p.T;
    const double zc = ctc_zc_for_grad(p, b);
    const int ezc = p.ctc_ez[b], Sx = 2 * p.ly[b] + 1;
    const int *ul = p.labels + p.lab_off[b];
    const double invc = zc > 0.0 ? 1.0 / zc : 0.0;
    const float ksm = p.c_den - (zc > 0.0 ? p.c_ctc : 0.f);   // fused log_softmax: sum over v of d loss / d logp[t][v]
    const int t0 = blockIdx.x * kGCFrames, t1 = min(t0 + kGCFrames, p.T), tl = min(t1, lx);
    int mylab[REGS];
#pragma unroll
    for (int i = 0; i < REGS; ++i) {
        const int s = tid + i * kGCThreads;
        mylab[i] = (s < Sx && (s & 1)) ? ul[s >> 1] : 0;
    }
    if (tid < kGCFrames)
        fcs[tid] = (t0 + tid < tl && zc > 0.0) ? ctc_frame_factor(p, b, bt0 + t0 + tid, invc, ezc) : 0.0;
    double an[REGS], bn[REGS];
    // Loads of FETCH are clamped, not predicated, and nothing is computed from them before CONSUME: with `v < V ? row_[v] : 0.f` (and the fused
    // term subtracted on the spot) the compiler loaded into temporaries and copied them to the loop registers right behind the requests -- behind
    // a vmcnt(0), so that every frame waited for the rows it had just asked for (round 4, found in the ISA).
    float rown[kGCVRegs] = {}, esn[kGCVRegs] = {}, isn = 0.f;
#define CRF_GC_FETCH(t)                                                                          \
    {                                                                                            \
        const double *Ar = p.CA + (bt0 + (t)) * p.Sc, *Br = p.CB + (bt0 + (t)) * p.Sc;           \
        _Pragma("unroll") for (int i = 0; i < REGS; ++i) {                                    \
            const int s = tid + i * kGCThreads;                                                  \
            if (s < Sx) { an[i] = Ar[s]; bn[i] = Br[s]; }                                        \
        }                                                                                        \
        if (accumulate) {                                                                        \
            const float *row_ = p.grad + (bt0 + (t)) * V;                                        \
            _Pragma("unroll") for (int q = 0; q < kGCVRegs; ++q) rown[q] = row_[min(tid + q * kGCThreads, V - 1)]; \
        }                                                                                        \
        if (p.fused) {   /* softmax term of log_softmax's backward, folded (in CONSUME) into the row the frame starts from */ \
            const float *er_ = p.ep + (bt0 + (t)) * V;                                           \
            isn = p.inv_s[bt0 + (t)];                                                            \
            _Pragma("unroll") for (int q = 0; q < kGCVRegs; ++q) esn[q] = er_[min(tid + q * kGCThreads, V - 1)]; \
        }                                                                                        \
    }
#define CRF_GC_CONSUME()                                                                         \
    {                                                                                            \
        _Pragma("unroll") for (int i = 0; i < REGS; ++i) prod[i] = (tid + i * kGCThreads < Sx) ? an[i] * bn[i] : 0.0; \
        _Pragma("unroll") for (int q = 0; q < kGCVRegs; ++q) rowc[q] = rown[q];                  \
        if (p.fused) {                                                                           \
            const float ks_ = ksm * pow2f(-kEpExp) * isn;                                        \
            _Pragma("unroll") for (int q = 0; q < kGCVRegs; ++q) rowc[q] -= ks_ * esn[q];        \
        }                                                                                        \
    }
    double prod[REGS];
    float rowc[kGCVRegs];
    for (int v = tid; v < 4 * Vp; v += kGCThreads) gc[v] = 0.f;
    if (tid < 4) gt[tid] = 0.f;
    if (t0 < tl) {
        CRF_GC_FETCH(t0);
        CRF_GC_CONSUME();
    }
#pragma unroll
    for (int i = 0; i < REGS; ++i) asm volatile("" ::"v"(mylab[i]));   // (a use in front of the loop: the labels' loads are waited for HERE, not at their
                                                                        // first use inside a divergent block of every frame)
    sync_lds();
    // Four label-sum buffers in rotation: frame t adds into buffer t&3 and clears buffer (t+2)&3, whose last
    // readers (the stores of frame t-2) are behind the barrier of frame t-1 -- ONE barrier per frame.
    // (the last frame peeled off instead of `if (t + 1 < tl)` around FETCH and CONSUME: with the waits for a frame's loads inside a conditional
    // block the compiler assumed them still in flight at the top of the next frame and waited there -- for the requests of THAT frame too)
    auto frame = [&](const int t, auto more_c) {
        constexpr bool more = decltype(more_c)::value;
        float *g = gc + (t & 3) * Vp, *gz = gc + ((t + 2) & 3) * Vp;
        if constexpr (more) CRF_GC_FETCH(t + 1);
        const double fc = fcs[t - t0];
        if (zc > 0.0) {
            float blank = 0.f, tot = 0.f;
#pragma unroll
            for (int i = 0; i < REGS; ++i)
                if (tid + i * kGCThreads < Sx) {
                    const float pr = (float)(prod[i] * fc);  // a posterior, in [0,1]
                    if (tid & 1) { atomicAdd(&g[mylab[i]], pr); if (CRF_X_CTCSUM) tot += pr; }
                    else blank += pr;
                }
            blank = wave_sum(blank);
            if (CRF_X_CTCSUM) tot = wave_sum(tot);
            if (lane == 0) { atomicAdd(&g[0], blank); if (CRF_X_CTCSUM) atomicAdd(&gt[t & 3], tot + blank); }
        }
#pragma unroll
        for (int q = 0; q < kGCVRegs; ++q) {
            const int v = tid + q * kGCThreads;
            if (v < V) gz[v] = 0.f;
        }
        if (CRF_X_CTCSUM && tid == 0) gt[(t + 2) & 3] = 0.f;
        sync_lds();
        // The posteriors of a frame sum to ONE.  A frame whose scaled products do not -- the chains' rescaled fp64 rows lost the states that
        // carry it, or Z itself is off -- is left out here and MARKED like the frames beyond the factor's range: the log-domain chains redo
        // it (round 5: tests/test_gpu_fuzz.py found frames summing to 617, to inf and to 0 behind a finite, correct-looking cost).
        const bool fbad = CRF_X_CTCSUM != 0 && zc > 0.0 && fc != 0.0 && !(fabsf(gt[t & 3] - 1.f) <= 1e-3f);
        if (fbad && tid == 0) { p.ctc_bad[bt0 + t] = 1; atomicMax(&p.redo_ctc[b], 1); }
        float out[kGCVRegs];
#pragma unroll
        for (int q = 0; q < kGCVRegs; ++q) {
            const int v = tid + q * kGCThreads;
            out[q] = v < V ? (fbad ? rowc[q] : rowc[q] - p.c_ctc * g[v]) : 0.f;
        }
        // take the next frame's loads out of their registers BEFORE this frame's stores are issued (a
        // vmcnt wait behind the stores would also wait for their acknowledgement)
        if constexpr (more) CRF_GC_CONSUME();
        float *row = p.grad + (bt0 + t) * V;
#pragma unroll
        for (int q = 0; q < kGCVRegs; ++q) {
            const int v = tid + q * kGCThreads;
            if (v < V) {   // (atomic: global_atomic_add_f32 -- the gradient is ordinary device memory; most labels of a frame carry no numerator mass)
                if (atomic) { if (out[q] != 0.f) unsafeAtomicAdd(row + v, out[q]); }
                else row[v] = out[q];
            }
        }
    };
    for (int t = t0; t + 1 < tl; ++t) frame(t, std::true_type{});
    if (t0 < tl) frame(tl - 1, std::false_type{});
#undef CRF_GC_CONSUME
#undef CRF_GC_FETCH
    if (!accumulate && !atomic)
        for (int t = max(t0, tl); t < t1; ++t) {
            float *row = p.grad + (bt0 + t) * V;
            for (int v = tid; v < V; v += kGCThreads) row[v] = 0.f;
        }
}


// ---- explicit instantiations ----
template __global__ void crf_grad_den_kernel<1, 1, 256, 32, 5, 1>(LossParams);
template __global__ void crf_grad_den_kernel<1, 4, 256, 32, 5, 1>(LossParams);
template __global__ void crf_grad_den_kernel<2, 4, 256, 32, 5, 1>(LossParams);
template __global__ void crf_grad_den_kernel<1, 2, 512, 32, 5, 1>(LossParams);
template __global__ void crf_grad_den_kernel<1, 2, 512, 32, 4, 4>(LossParams);
template __global__ void crf_grad_den_kernel<1, 2, 512, 32, 6, 1>(LossParams);
template __global__ void crf_grad_den_kernel<2, 2, 512, 32, 5, 1>(LossParams);
template __global__ void crf_grad_den_kernel<2, 2, 512, 8, 5, 1>(LossParams);
template __global__ void crf_grad_ctc_kernel<2>(LossParams);
template __global__ void crf_grad_ctc_kernel<4>(LossParams);
template __global__ void crf_grad_ctc_kernel<16>(LossParams);

}  // namespace crf
