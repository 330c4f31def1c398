// cat_amd/csrc/crf_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the CTC-CRF loss and
// the C-ABI entry point that launches them.  No MFMA: this is a sparse sum-product recursion.
//
// What is computed (semantics of the reference):
//   denominator  den_calculate.cu:63-261   alpha/beta over the den graph, logZ, arc posteriors -> labels
//   numerator    gpu_ctc_kernels.h:87-458  CTC alpha/beta on log-probs (blank 0), label posteriors
//   combine      ctc_crf/__init__.py:78-87 grad = c_den*gamma_den - c_ctc*gamma_ctc, loss likewise
//
// How (the MI355X design):
//   * arithmetic is LINEAR domain with an exact per-frame power-of-two rescale (integer exponent
//     bookkeeping) instead of per-arc log1p(exp()) (den_calculate.cu:29-35): no transcendental in
//     any recursion, the only exp() is one per (b,t,v) in crf_prep_kernel.
//   * 3 launches instead of ~3T+7 (den_calculate.cu:443-476): prep -> chains -> grad (+ finalize).
//   * crf_chain_kernel<ROLE> runs the FOUR independent recursions of every utterance (den forward,
//     den backward, ctc forward, ctc backward) as concurrent persistent workgroups on forked HIP
//     streams, one workgroup (one CU) per utterance and recursion, the whole time loop in-kernel, state vectors in LDS, arcs streamed as coalesced 16-byte ELL elements.
//   * the den graph is factored through "pairs" p = (destination state, label):
//       forward   q_t[p]   = sum_{arcs k in p} a_t[src_k] * w_k          (one LDS gather + FMA per arc)
//                 a_{t+1}[dst_p] += e_t[lab_p] * q_t[p]                   (one LDS atomic per pair)
//       backward  b_t[s]   = sum_{arcs k out of s} w_k * z_t[pair_k],  z_t[p] = e_t[lab_p]*b_{t+1}[dst_p]
//     so the per-arc work has no label lookup, and the posterior needs no arc pass at all:
//       gamma_den[t][v] = e_t[v] * sum_{p: lab_p = v} q_t[p] * b_{t+1}[dst_p] / Z
//     (crf_grad_kernel: two coalesced streams q_t, b_{t+1}[dst_p] -- the "algorithmic bytes").
#include <hip/hip_runtime.h>
#include <math.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <type_traits>
#include <vector>

#include "../../include/ctc_crf_hip.h"
#include "crf_internal.h"

namespace crf {

constexpr int kEpRegs = 8;    // ep row prefetch registers per thread  -> V  <= 8 * 1024
constexpr int kCtcThreads = 512, kCtcWaves = kCtcThreads / 64;  // numerator chains: 8 waves (S' = 2L+1 is a few hundred)
constexpr int kCtcRegs = 8;   // ctc states per thread                  -> 2L+1 <= 8 * 512
constexpr int kCtcPF = 4;     // frames per emission prefetch batch (ctc_forward)
constexpr int kGradThreads = 256;
constexpr int kGradFrames = 4;  // frames per crf_grad_kernel workgroup
// The streaming grad kernels normalise every frame by its own sum, so the power of two they take out of e' * sum(q * b) is free -- and must
// leave the product in range at BOTH ends: the label sums sit at ~2^40 (two rows rescaled to 2^20; up to 2^55 for a label with thousands of
// pairs), e' = exp(logp - rowmax) * 2^64 reaches down to 2^-125.  Rounds 1 - 4 took out 2^-64 ("undo the emissions' factor"): a frame whose
// ALLOWED labels all lie 100 - 131 nats below the row maximum -- the recursions carry it, nothing is flagged -- then had e' * 2^-64 * 2^40
// below the smallest normal float and came out with zero or garbage posteriors, silently (found in round 5 by
// tests/test_gpu_parity.py::test_single_frame_shrink_window[110]: gradient 90 % off, loss right).  2^-4: the product stays normal wherever
// e' itself is, and 2^64 * 2^55 * 2^-4 < 2^127 at the other end.
constexpr int kGradDescale = 4;
constexpr int kFlagFallback = 40;   // words [40], [41] behind the error word: utterances of the last call redone by the denominator / numerator fallback (crf_finalize_kernel)

// ---- build-time A/B switches of the frame loops (the defaults are the measured best: DESIGN.md section 2, profiles/round4_ab_*) ----
#ifndef CRF_X_GDEARLY
#define CRF_X_GDEARLY 1     // crf_grad_den_kernel: the rows of frame t+2 are requested right behind the staging of frame t+1 (0: at the top of frame t+1;
                            // 1: in the one-chunk, one-emission-register instantiation; 2: in every one-chunk instantiation not held to 128 VGPRs; 3: two-chunk ones too)
#endif
#ifndef CRF_X_GDMOVE
#define CRF_X_GDMOVE 1      // ... with the cur <- next register moves spelled out in front of the requests (0: left to the compiler, which waited for the rows
                            // of t+2 right behind their requests)
#endif
#ifndef CRF_X_GDW2
#define CRF_X_GDW2 1        // crf_grad_den_kernel, rows of 5121 .. 8192 floats: 1 = four row registers per thread at 128 VGPRs (two workgroups per CU)
#endif
#ifndef CRF_X_PRIO
#define CRF_X_PRIO 2        // fac_chain_body: issue priority by progress through the frame's chunks: 0 off, 1 steps at 1/4, 1/2, 3/4 of the
                            // chunks, 2 at 1/2, 3/4, 7/8 (product), 3 at 1/8, 1/4, 1/2 -- profiles/round4_ab_setprio_by_progress.txt
#endif
#ifndef CRF_X_EARLY
#define CRF_X_EARLY 1       // fac_chain_body: the frame's scale / exponent bookkeeping behind the first batch of gathers (0: in front of it)
#endif
#ifndef CRF_X_CTCSUM
#define CRF_X_CTCSUM 0      // crf_grad_ctc_kernel: mark a frame whose posteriors do not sum to one (built with the round-5 fixes; +16 % on that kernel, and what it
                            // caught is decided in front of the grad pass by crf_ctc_check_kernel and the frame factor's range check: the fuzz is green without it)
#endif
#ifndef CRF_X_CTCWPE
#define CRF_X_CTCWPE 5
#endif
#ifndef CRF_X_GCHK
#define CRF_X_GCHK 1        // crf_grad_den_kernel: the emission-weighted lost-term bound per frame (0: only "the frame's mass is a normal float"; A/B of what the check costs)
#endif
#ifndef CRF_X_GFIRST
#define CRF_X_GFIRST 0      // fac_chain_body (one CU per recursion): a frame BEGINS with its first batch of gathers -- everything else a frame starts with
                            // (stage check, emission prefetch, scale, exponent bookkeeping, row pointers: ~35 scalar instructions, ~4 cycles of a
                            // wave's issue each) follows behind a scheduling barrier, while the gathers are on their way -- profiles/round5_ab_*.txt
#endif
#ifndef CRF_X_KCLATE
#define CRF_X_KCLATE 0      // fac_chain_body, table geometries: the first slice's row constants are requested behind the first batch of gathers (0: at the frame top)
#endif
#ifndef CRF_X_LAG
#define CRF_X_LAG 0         // fac_chain_body (one CU per recursion): the scale of frame t+1 is worked out in the TAIL of frame t from the maximum
                            // deposited in frame t-1 -- known before barrier t, so no frame starts with an LDS round trip for its scale (0: the scale of
                            // frame t from the maximum of its own source vector, read behind the barrier) -- profiles/round5_ab_lagged_scale.txt
#endif
// Lagged scale: the vector of frame t+1 is produced with a scale chosen before its size is known.  With u_t = (exponent of max X_t) + k_t
// the exponent of the SCALED source of frame t, the rule k_{t+1} = kLagTarget - u_t gives u_{t+1} = kLagTarget + kEpExp + g_t, where 2^g_t is
// what frame t's emissions and weights did to the maximum (g_t in [-8, 1] for ordinary network outputs): the scaled maximum of a frame depends
// on the growth of ONE earlier frame, nothing accumulates.  Any integer k is exact (a power of two; the exponent word E carries the sum), so
// only the range is at stake -- and not in the recursions first but in the GRAD pass, which multiplies a row of q ~ 2^u_t with a row of
// b ~ 2^u'_t and e' 2^-kEpExp ~ 2^g: with the unlagged rule both rows sit at 2^kScaleExp whatever the frame did, here they carry the frame's
// growth (found by tests/test_gpu_parity.py::test_lagged_scale_window: NaN gradients at 45 nats with the first thresholds).  Hence: a frame
// whose scaled maximum falls below 2^kLagLow -- it shrank the vector by more than 2^40, 28 nats below the row maximum in ONE frame, against
// 131 nats with the unlagged rule -- marks the utterance for the log-shifted fallback (crf_robust_den_kernel), as total underflow does.
constexpr int kLagTarget = -28;   // u_{t+1} = 36 + g_t (the next vector's maximum: 2^(100 + g_t + g_{t+1}) < 2^127)
constexpr int kLagLow = -4;       // u_t below this: q * b * e' could leave the fp32 range in the grad pass
struct LossParams {
    GraphDev g;
    const float *logp;
    const int *labels, *lab_off, *lx, *ly;
    int B, T, V;
    int res_lds_rows_f, res_lds_rows_b;  // max rows per CU (LDS carve of the resident kernels)
    int Sc;       // row stride of the ctc per-frame stores: 2*max_label_len+1 rounded up to 64
    float c_den, c_ctc;
    // workspace
    float *ep, *mx;               // [B*T*V] exp(logp - mx), [B*T] row max
    // fused log_softmax (crf_loss_fwd_bwd_logits): `logp` points to RAW logits of type in_dtype (0 f32, 1 bf16, 2 f16);
    // log_softmax(x)[v] - rowmax = x[v] - max x, so everything that works on differences to the row maximum is unchanged:
    // only the per-frame OFFSET that enters the log-likelihoods differs (moff = max x - lse x = -log sum exp(x - max x);
    // without fusion moff = mx), and the gradient w.r.t. x gets the softmax term of log_softmax's backward
    // (inv_s = 1 / sum exp(x - max x); applied by the numerator half of the grad pass, which runs once per call).
    int fused, in_dtype;
    float *moff, *inv_s;          // [B*T]
    float *Q, *BP;                // [B*T*Rq] q_t[row], [B*T*Rb] b_{t+1}[row]  (scaled)
    int Rq, Rb;                   // their row strides
    const int *gq, *gb;           // label-sorted pair list -> index into a Q row / a BP row
    const int *gchunk, *glab;     // its chunks (<= kChunk entries of one label) and per-label chunk ranges
    int gNC;
    int res;                      // 1: register-resident den kernels (g.res), 0: streaming kernels
    int gd_stage, gd_nb;          // crf_grad_den_kernel: process only the 16-frame blocks completed by den segment `gd_stage` (0 = all)
    int gd_nf;                    // > 0: the launch holds only 2 * gd_nf candidate blocks per utterance (see the kernel)
    int gd_bound[16];             //   segment k (1-based) runs the recursion iterations [gd_bound[k-1], gd_bound[k])
    int fin_fold;                 // crf_robust_grad_kernel: workgroup (0, 0) also does what crf_finalize_kernel does (which is then not launched)
    int gd_persist;               // crf_grad_den_kernel: 1 = ONE launch for the stages gd_stage .. gd_nb-1 -- 1-D grid, stage-major, the candidates of stage k
    int gd_poff[17];              //   from block gd_poff[k] on -- whose workgroups wait for their stage's counter themselves:
    const int *gd_cnt;            //   the den kernels' stage counters (FacParams::stage_cnt; fine-grained memory) ...
    int gd_target;                //   ... and the value that releases a stage (2 B: every recursion has published it)
    int gd_fpb[16];               //   frames per workgroup in stage k: kGDFrames, or a divisor of it -- the short last stages, where a workgroup's 16 frames one after the
                                  //   other (4 - 5 us each) would BE the tail: a block is still of the stage its 16 frames make it, and split among 16 / fpb workgroups there
    int grad_den_acc;             // crf_grad_den_kernel: 1 = add to the row (the numerator half has written it) instead of writing; 2 = atomic add into a
                                  // row the prep kernel has zeroed (the numerator half adds its part from ANOTHER stream at the same time)
    int zero_grad;                // crf_prep_kernel: zero the gradient rows (the two halves of the grad pass then ADD, in any order)
    int grad_phase;               // crf_grad_kernel: 0 = den and ctc in one pass, 1 = den part only (writes), 2 = ctc part only (subtracts)
                                  // (crf_grad_ctc_kernel: 0 writes, 2 subtracts from the row, 3 = atomic add into a zeroed row)
    int b0;                       // first utterance of this launch (resident kernels with K > 1 run in groups)
    unsigned long long *xch;      // [2][B][2][G] tagged granules for the K-way exchange of the state vector
    float *cb_part;               // [B][kResMaxK] partial backward partition sums
    double *cb_mxs;               // [B]
    int *cb_F;                    // [B]
    int *err;                     // [1] set if an exchange timed out
    int *clear; int nclear;       // words the prep kernel zeroes (error word, start and stage counters of the factored schedule)
    float *Row0;                  // [B][Rb] spare rows (b_0 of the resident backward recursion)
    float *dump; int dump_stride; // two utterances per workgroup (fac_chain_body2): [2][ceil(B / 2)][dump_stride] dump rows
    float *gvec;                  // streaming kernels, graphs too large for LDS: [B][3*Sp + 4*Pr] state vectors in global memory
    int grad_stage;               // crf_grad_kernel: 1 = stage the Q / BP rows in LDS, 0 = gather them from global memory
    int *EQ, *EB;                 // [B*T] their binary exponents
    double *CA, *CB;              // [B*T*Sc] ctc forward (incl. emission) / backward (excl.)  (scaled, fp64)
    int *ECA, *ECB;
    float *den_zs;                // [B] scaled partition sums
    double *ctc_zc;
    int *den_ez, *ctc_ez;         // [B] their exponents
    float *cost_alpha, *cost_beta, *cost_ctc;  // [B]
    int *invalid;                 // [B]
    int *redo;                    // [2][B] utterance whose scaled-fp32 denominator lost all its mass (forward / backward): redone by the robust kernels
    int force_redo;               // switch robust = 1: every utterance takes the robust path
    // Numerator fallback.  The fp64 chains rescale a frame's vector by its maximum; a frame whose posterior mass sits more than
    // ~650 nats below (max alpha) * (max beta) -- long utterances with many labels whose emissions do not follow the labels:
    // T = 3000, L = 500 on random inputs -- has the products that matter at the bottom of the fp64 range (and beyond: 0 * inf).
    // The grad kernels recognise such frames from the frame's exponents alone (ctc_frame_factor), contribute nothing for them
    // and mark them; crf_robust_ctc_kernel redoes the marked utterances' chains in the log domain and
    // crf_robust_ctc_fix_kernel adds the marked frames' posteriors.
    int *redo_ctc;                // [B] 0 = fine, 1 = some frames marked in ctc_bad, 2 = the whole utterance (the forward chain lost its mass, or forced)
    int *ctc_bad;                 // [B*T] marked frames (cleared by the prep kernel)
    int force_redo_ctc;           // switches robust = 1 / robust_ctc = 1: every utterance's numerator takes the log-domain path
    int *ctc_logdom;              // [B] 0, or the pass (1 = right behind the numerator's grad half, beside the denominator recursions of the
                                  // staged schedule; 2 = end of the call) in which the log-domain kernels redid the utterance: its CA / CB
                                  // rows and ctc_zc then hold LOGARITHMS
    int ctc_pass;                 // the pass of this launch of the log-domain kernels
    int ctc_tilt;                 // percent of the numerator chains' tilt (ctc_rho): 100 by default, 0 = none
    int *ctc_seen;                // host-visible word: the number of the last call in which an utterance took the log-domain chains
    int call_id;                  // this call's number (per context)
    int64_t gvec_stride;          // floats per utterance of `gvec`
    // outputs
    float *grad, *loss, *out_den, *out_beta, *out_ctc;
    int *out_invalid;
};

// Wave-wide max without touching the LDS crossbar (a __shfl_xor butterfly is six dependent
// ds_bpermute round trips, ~0.25 us on the per-frame critical path): rotate-and-max inside each row
// of 16 lanes with DPP (row_ror 8/4/2/1), then combine the four rows through readlane + scalar max.
__device__ __forceinline__ float wave_max(float v) {
#define CRF_DPP_MAX(ctrl) v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, false)))
    CRF_DPP_MAX(0x128);
    CRF_DPP_MAX(0x124);
    CRF_DPP_MAX(0x122);
    CRF_DPP_MAX(0x121);
#undef CRF_DPP_MAX
    const int iv = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}
// LDS float maximum without a returned value (ds_max_f32; the intrinsic, not atomicrmw: the compiler's atomic optimiser turns a
// same-address atomicrmw of several lanes into a scalar loop over the active lanes)
__device__ __forceinline__ void lds_fmax(float *lds_ptr, float v) {
    (void)__builtin_amdgcn_ds_fmaxf((__attribute__((address_space(3))) float *)lds_ptr, v, 0, 0, false);
}
// maximum over each row of 16 lanes of NON-NEGATIVE values, left in every lane of the row: on the float bits as integers (they
// order alike, and an integer maximum takes the DPP operand directly -- fmaxf costs two canonicalising v_max per step on top)
__device__ __forceinline__ float row_max16(float v) {
    int b = __builtin_bit_cast(int, v);
#define CRF_DPP_IMAXB(ctrl) b = max(b, __builtin_amdgcn_update_dpp(0, b, ctrl, 0xf, 0xf, false))
    CRF_DPP_IMAXB(0x128);
    CRF_DPP_IMAXB(0x124);
    CRF_DPP_IMAXB(0x122);
    CRF_DPP_IMAXB(0x121);
#undef CRF_DPP_IMAXB
    return __builtin_bit_cast(float, b);
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// Wave-wide max of NON-NEGATIVE doubles, exact in the top 32 bits (sign, exponent, 20 mantissa bits) --
// all the rescaling needs is the binary exponent.  Positive doubles order like their high words as
// integers, so this is an integer DPP max (no LDS round trips).
__device__ __forceinline__ double wave_max_d(double v) {
    int hi = (int)((unsigned long long)__double_as_longlong(v) >> 32);
#define CRF_DPP_IMAX(ctrl) hi = max(hi, __builtin_amdgcn_update_dpp(0, hi, ctrl, 0xf, 0xf, false))
    CRF_DPP_IMAX(0x128);
    CRF_DPP_IMAX(0x124);
    CRF_DPP_IMAX(0x122);
    CRF_DPP_IMAX(0x121);
#undef CRF_DPP_IMAX
    const int m = max(max(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(hi, 16)),
                      max(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(hi, 48)));
    return __longlong_as_double((long long)(unsigned)m << 32);
}
// ... the same maximum as its high word (an int): what the chains' frame maximum is kept as in LDS (one ds_max_i32 per wave)
__device__ __forceinline__ int wave_max_hi(double v) {
    int hi = (int)((unsigned long long)__double_as_longlong(v) >> 32);
#define CRF_DPP_IMAX(ctrl) hi = max(hi, __builtin_amdgcn_update_dpp(0, hi, ctrl, 0xf, 0xf, false))
    CRF_DPP_IMAX(0x128);
    CRF_DPP_IMAX(0x124);
    CRF_DPP_IMAX(0x122);
    CRF_DPP_IMAX(0x121);
#undef CRF_DPP_IMAX
    return max(max(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(hi, 16)), max(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(hi, 48)));
}
// rescale exponent from the HIGH WORD of a non-negative double maximum (0: nothing to scale by)
__device__ __forceinline__ int rescale_exp_hi(int hi) {
    if (hi <= 0) return 0;
    const int k = kScaleExpD - (((hi >> 20) & 0x7ff) - 1023);
    return k < -900 ? -900 : (k > 900 ? 900 : k);
}
// fp64 twin of rescale_exp / pow2f for the numerator chains
__device__ __forceinline__ int rescale_exp_d(double m) {
    if (!(m > 0.0)) return 0;
    int e = (int)(((unsigned long long)__double_as_longlong(m) >> 52) & 0x7ffull) - 1023;
    int k = kScaleExpD - e;
    return k < -900 ? -900 : (k > 900 ? 900 : k);
}
__device__ __forceinline__ double pow2d(int k) { return __longlong_as_double((long long)(k + 1023) << 52); }
// exp(d) * 2^add for d <= 0 without intermediate underflow: d = k ln2 + r, result = exp(r) * 2^(k+add)
__device__ __forceinline__ float exp_scaled(float d, int add) {
    const float k = rintf(d * 1.4426950408889634f);
    float r = fmaf(-k, 0.693145751953125f, d);
    r = fmaf(-k, 1.428606765330187e-06f, r);
    return ldexpf(expf(r), (int)k + add);
}
__device__ __forceinline__ double exp_scaled_d(float d) {
    const float k = rintf(d * 1.4426950408889634f);
    float r = fmaf(-k, 0.693145751953125f, d);
    r = fmaf(-k, 1.428606765330187e-06f, r);
    return ldexp((double)expf(r), (int)k);
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. it waits
// until every global store of the frame (the Q / BP rows) has been acknowledged by L2 -- about a
// microsecond per frame on a 1500-frame dependency chain.  Nothing inside the frame loops reads
// global data written by another wave of the same workgroup, so the LDS-only form is sufficient.
__device__ __forceinline__ void sync_lds() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// LDS reads whose ISSUE point is fixed in the source (the compiler sinks an ordinary read to its first use, behind whatever is computed in
// between).  The value may be used only behind lds_landed() of the same variable -- the compiler does not know these are LDS operations and
// inserts no wait of its own; its waits for its own LDS operations stay correct (lgkmcnt counts in order: at worst they wait for these too).
__device__ __forceinline__ double lds_issue_f64(const double *q) {
    double v;
    asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"((unsigned)(uintptr_t)q));
    return v;
}
__device__ __forceinline__ int lds_issue_i32(const int *q) {
    int v;
    asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"((unsigned)(uintptr_t)q));
    return v;
}
__device__ __forceinline__ void lds_landed(double &a, double &b, double &c, int &w) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(w));
}

// In-kernel phase timing for diagnosis (build with CRF_BUILD_DEFS=-DCRF_TIMING; tools/timing_probe.py):
// one chosen workgroup stamps s_memtime (shader cycles) at phase boundaries into g_tm, read back with
// crf_timing_read().  Compiled out of the product build.
#ifdef CRF_TIMING
__device__ unsigned long long g_tm[16384];
#define CRF_TM(on, idx) do { if (on) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if ((threadIdx.x & 63) == 0) g_tm[(idx)] = t_; } } while (0)
#else
#define CRF_TM(on, idx) do { } while (0)
#endif

// exact power-of-two rescale that brings m into [2^kScaleExp, 2^(kScaleExp+1))
__device__ __forceinline__ int rescale_exp(float m) {
    if (!(m > 0.f)) return 0;
    int e = (int)((__float_as_uint(m) >> 23) & 0xffu) - 127;
    int k = kScaleExp - e;
    return k < -100 ? -100 : (k > 100 ? 100 : k);
}
__device__ __forceinline__ float pow2f(int k) { return __uint_as_float((unsigned)(k + 127) << 23); }
// the same from the float BITS of a non-negative maximum (integer operations only: on a wave-uniform value they run on the
// scalar unit; a float compare would not)
__device__ __forceinline__ int rescale_exp_bits(unsigned bits) {
    if (bits == 0u) return 0;
    const int k = kScaleExp + 127 - (int)((bits >> 23) & 0xffu);
    return k < -100 ? -100 : (k > 100 ? 100 : k);
}
// ... for a value the caller KNOWS to be wave-uniform (an SGPR): the clamp as s_max / s_min.  The compiler selects v_med3_i32 for the
// C form above even on uniform operands, which costs the frame loops a VALU instruction and three VGPRs (the two bounds and the result).
__device__ __forceinline__ int rescale_exp_bits_uniform(unsigned bits) {
    int k = kScaleExp + 127 - (int)((bits >> 23) & 0xffu);
    asm("s_max_i32 %0, %0, %1\n\ts_min_i32 %0, %0, %2" : "+s"(k) : "s"(-100), "s"(100) : "scc");
    return bits == 0u ? 0 : k;
}

// ---------------------------------------------------------------------------------------------
// prep: e[b][t][v] = exp(logp[b][t][v] - max_v) * 2^kEpExp, mx[b][t] = max_v   (one wave per frame)
// ---------------------------------------------------------------------------------------------
// element `i` of the network output: fp32 log-probs (reference interface) or, fused, raw logits in fp32 / bf16 / fp16
__device__ __forceinline__ float ld_x(const LossParams &p, int64_t i) {
    if (p.in_dtype == 0) return p.logp[i];
    const unsigned short u = ((const unsigned short *)p.logp)[i];
    if (p.in_dtype == 1) return __uint_as_float((unsigned)u << 16);
    return (float)__builtin_bit_cast(_Float16, u);
}

// G lanes per frame (16 for small vocabularies: four frames per wave -- with a whole wave per 72-entry row the kernel ran
// at a quarter of the HBM rate; 64 otherwise).
template <int G>
__global__ __launch_bounds__(256) void crf_prep_kernel(LossParams p) {
    const int sub = threadIdx.x & (G - 1);
    // the counters of the staged schedule start at zero in every call (a memset in the stream cost two more
    // dispatches between this kernel and the recursions)
    if (blockIdx.x == 0) {
        for (int i = threadIdx.x; i < p.nclear; i += 256) __hip_atomic_store(p.clear + i, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (p.redo) for (int i = threadIdx.x; i < 2 * p.B; i += 256) p.redo[i] = p.force_redo;
        if (p.redo_ctc) for (int i = threadIdx.x; i < p.B; i += 256) { p.redo_ctc[i] = p.force_redo_ctc ? 2 : 0; p.ctc_logdom[i] = 0; }
    }
    const int64_t f = (int64_t)blockIdx.x * (256 / G) + (threadIdx.x / G);
    if (f >= (int64_t)p.B * p.T) return;
    const int b = (int)(f / p.T), t = (int)(f % p.T);
    if (p.zero_grad) {                            // (every frame, the ones past the utterance's length too)
        float *gr = p.grad + f * p.V;
        for (int v = sub; v < p.V; v += G) gr[v] = 0.f;
    }
    if (t >= p.lx[b]) return;                     // (whole groups of G lanes leave together)
    auto gmax = [](float v) {
#pragma unroll
        for (int o = G / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, G));
        return v;
    };
    auto gsum = [](float v) {
#pragma unroll
        for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, G);
        return v;
    };
    const int64_t r0 = f * p.V;
    float m = -INFINITY;
    float *er = p.ep + f * p.V;
    float ssum = 0.f;
    constexpr int NX = 16;                        // row entries a lane keeps (V <= 16 G: every vocabulary the factored kernels take)
    if (p.V <= NX * G) {
        // one pass over the row: its entries stay in registers between the maximum and the exp (round 5; the second pass re-read them through
        // the cache, a second memory round trip on the only kernel in front of the recursions: 23.7 -> see profiles/round5_ab_grad_one_launch.txt)
        float x[NX];
#pragma unroll
        for (int k = 0; k < NX; ++k) {
            if (k * G >= p.V) break;              // (uniform)
            const int v = sub + k * G;
            x[k] = v < p.V ? ld_x(p, r0 + v) : -INFINITY;
            m = fmaxf(m, x[k]);
        }
        m = gmax(m);
        if (m == -INFINITY) m = 0.f;
#pragma unroll
        for (int k = 0; k < NX; ++k) {
            if (k * G >= p.V) break;
            const int v = sub + k * G;
            if (v < p.V) {
                const float d = x[k] - m;
                er[v] = exp_scaled(d, kEpExp);
                if (p.fused) ssum += __expf(d);
            }
        }
    } else {
        for (int v = sub; v < p.V; v += G) m = fmaxf(m, ld_x(p, r0 + v));
        m = gmax(m);
        if (m == -INFINITY) m = 0.f;
        for (int v = sub; v < p.V; v += G) {
            const float d = ld_x(p, r0 + v) - m;
            er[v] = exp_scaled(d, kEpExp);
            if (p.fused) ssum += __expf(d);
        }
    }
    if (p.fused) {
        ssum = gsum(ssum);
        if (sub == 0) { p.moff[f] = -logf(ssum); p.inv_s[f] = 1.f / ssum; }   // (moff == mx without fusion: same array)
    }
    if (sub == 0) { p.mx[f] = m; if (p.ctc_bad) p.ctc_bad[f] = 0; }
}

// crf_stage_i32: the integer metadata of a call (labels, lengths, offsets) come from the host; a kernel reads
// them from PINNED host memory and writes the device copy.  A DMA copy in the caller's stream took ~0.15 ms to
// start between two calls (rocprofv3 timeline), and on its own stream it would need a fifth hardware queue.
__global__ __launch_bounds__(256) void crf_stage_i32_kernel(int *__restrict__ dst, const int *__restrict__ src, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = __builtin_nontemporal_load(src + i);
}

// block-wide helpers for the 1024-thread chain workgroups --------------------------------------
__device__ __forceinline__ float block_sum(float v, float *red, int tid) {
    v = wave_sum(v);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < kChainWaves; ++i) s += red[i];
    return s;
}
__device__ __forceinline__ double block_sum_d(double v, double *red, int tid) {
    v = wave_sum_d(v);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < kChainWaves; ++i) s += red[i];
    return s;
}
__device__ __forceinline__ double mx_total(const LossParams &p, int b, int lx, double *red, int tid) {
    double part = 0.0;
    for (int t = tid; t < lx; t += kChainThreads) part += (double)p.moff[(int64_t)b * p.T + t];
    return block_sum_d(part, red, tid);
}
__device__ __forceinline__ float frame_max(const float *wm) {
    float m = wm[0];
#pragma unroll
    for (int i = 1; i < kChainWaves; ++i) m = fmaxf(m, wm[i]);
    return m;
}
__device__ __forceinline__ float to_log(float zs, int e, double mxs) {
    return zs > 0.f ? (float)(log((double)zs) - (double)e * 0.6931471805599453 + mxs) : -INFINITY;
}

// Sum of one ELL row per lane: sum_k x[idx_k] * w_k over n 16-byte elements (2 arcs each) that are
// kWave elements apart.  n is wave-uniform; the host pads n to a multiple of 4 whenever n > 2, so
// the stream is consumed in groups of four elements with the NEXT group already in flight (4 KiB per
// wave, 64 KiB per CU): the L2 latency of the arc stream overlaps the LDS gathers of the group
// that has landed.  hipcc folds a source-level prefetch loop back into load->wait->use, so the
// group loads are issued from inline asm (invisible to its scheduler) and waited for explicitly; two
// register sets alternate.  Loads return in order, so compiler-issued loads/stores in between only
// make either side's waits more conservative (cdna_hip_programming.md 5.7).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void ell_consume(const u32x4 &c, const float *x, float &acc0, float &acc1) {
    acc0 = fmaf(x[c.x], __uint_as_float(c.y), acc0);
    acc1 = fmaf(x[c.z], __uint_as_float(c.w), acc1);
}
#define CRF_LOADG(R, P)                                                                                        \
    asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:1024\n\t"           \
                 "global_load_dwordx4 %2, %4, off offset:2048\n\tglobal_load_dwordx4 %3, %4, off offset:3072"   \
                 : "=&v"(R##0), "=&v"(R##1), "=&v"(R##2), "=&v"(R##3) : "v"(P) : "memory")
#define CRF_WAITG(N, R) \
    asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(R##0), "+v"(R##1), "+v"(R##2), "+v"(R##3) : : "memory")
#define CRF_USEG(R)                      \
    ell_consume(R##0, x, acc0, acc1);    \
    ell_consume(R##1, x, acc0, acc1);    \
    ell_consume(R##2, x, acc0, acc1);    \
    ell_consume(R##3, x, acc0, acc1)

__device__ __forceinline__ float ell_row_sum(const uint4 *a4, int n, const float *x) {
    const u32x4 *a = (const u32x4 *)a4;
    float acc0 = 0.f, acc1 = 0.f;
    if (n <= 2) {
        if (n > 0) {
            const u32x4 c0 = a[0];
            if (n > 1) {
                const u32x4 c1 = a[kWave];
                ell_consume(c1, x, acc0, acc1);
            }
            ell_consume(c0, x, acc0, acc1);
        }
        return acc0 + acc1;
    }
    // Double buffer: the loads of group g+1 are in flight while group g is gathered and summed.
    // Every group is waited for BEFORE the loop back-edge, so whatever register copies hipcc inserts
    // for the loop-carried set only ever touch data that has landed (cdna_hip_programming.md 5.7:
    // an asm load's destination counts as written at the end of the statement).
    const int ng = n >> 2;
    u32x4 A0, A1, A2, A3, B0, B1, B2, B3;
    const u32x4 *p = a;
    CRF_LOADG(A, p);
    CRF_WAITG(0, A);
    int g = 1;
#pragma unroll 1
    for (; g + 1 < ng; g += 2) {
        p += 4 * kWave;
        CRF_LOADG(B, p);
        CRF_USEG(A);
        p += 4 * kWave;
        CRF_WAITG(0, B);
        CRF_LOADG(A, p);
        CRF_USEG(B);
        CRF_WAITG(0, A);
    }
    if (g < ng) {
        p += 4 * kWave;
        CRF_LOADG(B, p);
        CRF_USEG(A);
        CRF_WAITG(0, B);
        CRF_USEG(B);
    } else {
        CRF_USEG(A);
    }
    return acc0 + acc1;
}

// LDS carve (floats) shared by host sizing and the kernels
__host__ __device__ inline int rup64(int x) { return (x + 63) & ~63; }

// ---------------------------------------------------------------------------------------------
// denominator forward.  LDS: X[3][Sp] | EP[2][Vp] | wmax[2][16] | red (16 doubles)
// ---------------------------------------------------------------------------------------------
// GV = true: the state vectors live in global memory (L2) instead of LDS -- the fallback for graphs whose
// vectors exceed the 160 KiB of a CU.  Same code; gathers become L2 hits, the frame barrier also drains vmcnt.
template <bool GV>
__device__ __forceinline__ void chain_sync() {
    if (GV) __syncthreads(); else sync_lds();
}
template <bool GV>
__device__ __forceinline__ void den_forward(const LossParams &p, int b, float *lds) {
    const GraphDev &g = p.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int S = g.S, V = p.V, Pr = g.Pr, lx = p.lx[b];
    const int Sp = rup64(S), Vp = rup64(V);
    float *X = GV ? p.gvec + (size_t)b * p.gvec_stride : lds;
    float *EP = GV ? lds : X + 3 * Sp;
    float *wm = EP + 2 * Vp;
    double *red = (double *)(wm + 2 * kChainWaves);
    const int64_t bt0 = (int64_t)b * p.T;

    for (int s = tid; s < 3 * Sp; s += kChainThreads) X[s] = (s < S) ? g.start_lin[s] * pow2f(kScaleExp) : 0.f;
    if (lx > 0)
        for (int v = tid; v < V; v += kChainThreads) EP[v] = p.ep[bt0 * V + v];
    int E = kScaleExp;
    __syncthreads();

    const int sl0 = g.fwd.wave_off[wave], sl1 = g.fwd.wave_off[wave + 1];
    for (int t = 0; t < lx; ++t) {
        float *Xc = X + (t % 3) * Sp, *Xn = X + ((t + 1) % 3) * Sp, *Xz = X + ((t + 2) % 3) * Sp;
        const float *EPc = EP + (t & 1) * Vp;
        // next frame's emission row -> registers now, LDS at the end of the frame
        float epn[kEpRegs];
        if (t + 1 < lx) {
            const float *er = p.ep + (bt0 + t + 1) * V;
#pragma unroll
            for (int i = 0; i < kEpRegs; ++i) {
                int v = tid + i * kChainThreads;
                epn[i] = v < V ? er[v] : 0.f;
            }
        }
        float m = 0.f;
        for (int s = tid; s < S; s += kChainThreads) m = fmaxf(m, Xc[s]);
        m = wave_max(m);
        if (lane == 0) wm[(t & 1) * kChainWaves + wave] = m;
        chain_sync<GV>();
        const int k = rescale_exp(frame_max(wm + (t & 1) * kChainWaves));
        const float sc = pow2f(k);
        E += k;                       // exponent of q_t
        if (tid == 0) p.EQ[bt0 + t] = E;
        E += kEpExp;                  // a_{t+1} = sum e'_t q_t carries the 2^kEpExp of e'_t
        for (int s = tid; s < Sp; s += kChainThreads) Xz[s] = 0.f;
        float *Qrow = p.Q + (bt0 + t) * Pr;
        for (int i = sl0; i < sl1; ++i) {
            const int j = __builtin_amdgcn_readfirstlane(g.fwd.wave_slices[i]);
            const int off = __builtin_amdgcn_readfirstlane(g.fwd.slice_off[j]);
            const int w2 = __builtin_amdgcn_readfirstlane(g.fwd.slice_w2[j]);
            const int r = j * kWave + lane;
            const int2 meta = g.pair_meta[r];  // {dst, label}; issued before the arc stream
            const float q = ell_row_sum(g.fwd.arcs + off + lane, w2, Xc) * sc;
            Qrow[r] = q;
            if (meta.x >= 0) {  // sole contributor to its destination state: plain store; LDS float atomics
                const float av = EPc[meta.y & 0xffff] * q;  // are lane-serial (~2.5 clk per lane)
                if (meta.y >> 16) Xn[meta.x] = av; else atomicAdd(&Xn[meta.x], av);
            }
        }
        if (t + 1 < lx) {
            float *EPn = EP + ((t + 1) & 1) * Vp;
#pragma unroll
            for (int i = 0; i < kEpRegs; ++i) {
                int v = tid + i * kChainThreads;
                if (v < V) EPn[v] = epn[i];
            }
        }
        chain_sync<GV>();
    }
    const float *Xf = X + (lx % 3) * Sp;
    float part = 0.f;
    for (int s = tid; s < S; s += kChainThreads) part += Xf[s] * g.end_lin[s];
    const float zs = block_sum(part, (float *)red, tid);
    const double mxs = mx_total(p, b, lx, red, tid);
    if (tid == 0) {
        p.den_zs[b] = zs;
        p.den_ez[b] = E;
        p.cost_alpha[b] = to_log(zs, E, mxs);
        if (!(zs > 0.f && zs < INFINITY)) p.redo[b] = 1;   // all mass lost in scaled fp32 (or overflow): the robust kernels redo it
    }
}

// ---------------------------------------------------------------------------------------------
// denominator backward.  LDS: Z[2][Pr] | BPst[2][Pr] | EP[2][Vp] | wmax[2][16] | red
// iteration i handles frame t = lx-1-i:  b_t[s] = sc * sum_k w_k * Z[cur][pair_k]
// ---------------------------------------------------------------------------------------------
template <bool GV>
__device__ __forceinline__ void den_backward(const LossParams &p, int b, float *lds) {
    const GraphDev &g = p.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int S = g.S, V = p.V, Pr = g.Pr, lx = p.lx[b];
    const int Vp = rup64(V);
    float *Z = GV ? p.gvec + (size_t)b * p.gvec_stride + 3 * (size_t)rup64(S) : lds;
    float *BPst = Z + 2 * Pr;
    float *EP = GV ? lds : BPst + 2 * Pr;
    float *wm = EP + 2 * Vp;
    double *red = (double *)(wm + 2 * kChainWaves);
    const int64_t bt0 = (int64_t)b * p.T;
    int F = kScaleExp;
    float zpart = 0.f;

    for (int r = tid; r < 4 * Pr; r += kChainThreads) Z[r] = 0.f;
    if (lx > 0) {
        for (int v = tid; v < V; v += kChainThreads) {
            EP[v] = p.ep[(bt0 + lx - 1) * V + v];
            if (lx > 1) EP[Vp + v] = p.ep[(bt0 + lx - 2) * V + v];
        }
        __syncthreads();
        float *BProw = p.BP + (bt0 + lx - 1) * Pr;
        for (int r = tid; r < Pr; r += kChainThreads) {
            const int d = g.pair_meta[r].x;
            const float bv = d >= 0 ? g.end_lin[d] * pow2f(kScaleExp) : 0.f;
            BProw[r] = bv;
            Z[r] = EP[g.pair_meta[r].y & 0xffff] * bv;
        }
        if (tid == 0) p.EB[bt0 + lx - 1] = F;
    } else {
        for (int s = tid; s < S; s += kChainThreads) zpart += g.start_lin[s] * g.end_lin[s] * pow2f(kScaleExp);
    }
    __syncthreads();

    const int sl0 = g.bwd.wave_off[wave], sl1 = g.bwd.wave_off[wave + 1];
    for (int i = 0; i < lx; ++i) {
        const int t = lx - 1 - i;
        const float *Zc = Z + (i & 1) * Pr;
        float *Zn = Z + ((i + 1) & 1) * Pr;
        float *BPc = BPst + (i & 1) * Pr;
        const float *EPn = EP + ((i + 1) & 1) * Vp;  // e_{t-1}
        float epn[kEpRegs];
        if (t >= 2) {
            const float *er = p.ep + (bt0 + t - 2) * V;
#pragma unroll
            for (int q = 0; q < kEpRegs; ++q) {
                int v = tid + q * kChainThreads;
                epn[q] = v < V ? er[v] : 0.f;
            }
        }
        float m = 0.f;
        for (int r = tid; r < Pr; r += kChainThreads) m = fmaxf(m, Zc[r]);
        m = wave_max(m);
        if (lane == 0) wm[(i & 1) * kChainWaves + wave] = m;
        if (i > 0) {  // b_{t+1}[dst_p], staged by the previous iteration -> BP[b][t]
            const float *BPp = BPst + ((i - 1) & 1) * Pr;
            float *BProw = p.BP + (bt0 + t) * Pr;
            for (int r = tid; r < Pr; r += kChainThreads) BProw[r] = BPp[r];
            if (tid == 0) p.EB[bt0 + t] = F;
        }
        chain_sync<GV>();
        const int k = rescale_exp(frame_max(wm + (i & 1) * kChainWaves));
        const float sc = pow2f(k);
        F += k + kEpExp;              // Z_t = e'_t b_{t+1} carries the 2^kEpExp of e'_t
        for (int ii = sl0; ii < sl1; ++ii) {
            const int j = __builtin_amdgcn_readfirstlane(g.bwd.wave_slices[ii]);
            const int off = __builtin_amdgcn_readfirstlane(g.bwd.slice_off[j]);
            const int w2 = __builtin_amdgcn_readfirstlane(g.bwd.slice_w2[j]);
            const int4 meta = g.bwd_row_meta[j * kWave + lane];  // {state, #pairs into it, first pair, its label}
            const float bv = ell_row_sum(g.bwd.arcs + off + lane, w2, Zc) * sc;
            const int s = meta.x;
            if (s >= 0) {
                if (t == 0) {
                    zpart += g.start_lin[s] * bv;
                } else if (meta.y == 1) {
                    BPc[meta.z] = bv;
                    Zn[meta.z] = EPn[meta.w] * bv;
                } else {
                    for (int pi = g.st_pair_off[s]; pi < g.st_pair_off[s + 1]; ++pi) {
                        const int r = g.st_pairs[pi];
                        BPc[r] = bv;
                        Zn[r] = EPn[g.pair_meta[r].y & 0xffff] * bv;
                    }
                }
            }
        }
        if (t >= 2) {
            float *EPw = EP + (i & 1) * Vp;
#pragma unroll
            for (int q = 0; q < kEpRegs; ++q) {
                int v = tid + q * kChainThreads;
                if (v < V) EPw[v] = epn[q];
            }
        }
        chain_sync<GV>();
    }
    const float zb = block_sum(zpart, (float *)red, tid);
    const double mxs = mx_total(p, b, lx, red, tid);
    if (tid == 0) {
        p.cost_beta[b] = to_log(zb, F, mxs);
        if (!(zb > 0.f && zb < INFINITY)) p.redo[p.B + b] = 1;
    }
}

__device__ __forceinline__ float ctc_block_sum(float v, float *red, int tid) {
    v = wave_sum(v);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < kCtcWaves; ++i) s += red[i];
    return s;
}
__device__ __forceinline__ double ctc_mx_total(const LossParams &p, int b, int lx, double *red, int tid) {
    double part = 0.0;
    for (int t = tid; t < lx; t += kCtcThreads) part += (double)p.moff[(int64_t)b * p.T + t];
    part = wave_sum_d(part);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = part;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < kCtcWaves; ++i) s += red[i];
    return s;
}

// ---------------------------------------------------------------------------------------------
// CTC numerator chains, in fp64: a forced alignment may have to pass through frames where the
// label is e^-100 below the row max, so the numerator gets the e^+-700 range of doubles (it is
// ~1% of the work).  Emissions are formed in-kernel as exp(logp - rowmax) in double.
// LDS: Abuf[2][Sxp] (double) | wmax[2][16] (double) | red[16] (double) | lab[Sxp] (int)
// validity rule L + repeats <= T_b: gpu_ctc.h:161-174
// ---------------------------------------------------------------------------------------------
struct CtcLds {
    double *A, *wm, *red;
    int *lab;
};
__device__ __forceinline__ CtcLds ctc_carve(float *lds, int Sxp) {
    CtcLds c;
    c.A = (double *)lds;
    c.wm = c.A + 2 * Sxp;
    c.red = c.wm + 2 * kCtcWaves;
    c.lab = (int *)(c.red + kCtcWaves);
    return c;
}
__device__ __forceinline__ bool ctc_setup(const LossParams &p, int b, const CtcLds &c, int L, int lx, int tid) {
    const int *ul = p.labels + p.lab_off[b];
    const int Sx = 2 * L + 1;
    float rep = 0.f;
    for (int s = tid; s < Sx; s += kCtcThreads) c.lab[s] = (s & 1) ? ul[s >> 1] : 0;
    for (int i = tid + 1; i < L; i += kCtcThreads) rep += (ul[i] == ul[i - 1]) ? 1.f : 0.f;
    const int repeats = (int)(ctc_block_sum(rep, (float *)c.red, tid) + 0.5f);  // also orders the lab[] writes
    __syncthreads();
    return lx > 0 && L + repeats <= lx;
}
// Tilt of the numerator chains.  The two chains are stored as A'_t[s] = A_t[s] rho^s and Bx'_t[s] = Bx_t[s] rho^(Sx-1-s): the
// recursions keep their form with the factors rho, rho^2 on the transitions that advance by one, two states (both directions), the
// products A' Bx' are the posteriors' numerators times the constant rho^(Sx-1), which the chain's own end sum Z' = A'[Sx-1] +
// rho A'[Sx-2] carries too -- EXACT for any rho > 0; log Z = log Z' - (Sx-1) log rho.  What rho buys is range: each chain is
// rescaled by its own maximum, and with diffuse emissions (an untrained network, a large output layer) the free forward mass runs
// ahead at ~0.8 states per frame whatever the labels need (Sx / lx), the backward mass likewise from the other end, so the states
// that carry a frame's posterior sit e^-0.8 per frame of the utterance's middle below both maxima: beyond fp64 from T ~ 1 600 on
// (frames marked for the log-domain chains; T = 3 000 / L = 500: every utterance).  rho_u solves "mean advance of the tilted free
// chain = Sx / lx" for uniform emissions ((rho + rho^2) / (1 + rho + rho^2 / 2): a state passes mass to itself, the next, and --
// half of the states -- the one after); a peaked network's chains follow the alignment by themselves and a tilt would only cost
// range where the alignment leaves the diagonal, so the tilt's strength is (1 - mean_t max_v p_t[v]).  Both chains' workgroups
// compute rho from the same inputs by the same instructions (it must be the same number, bit for bit).
__device__ __forceinline__ double ctc_rho(const LossParams &p, int b, int Sx, int lx, double *red, int tid) {
    if (p.ctc_tilt <= 0 || Sx < 2 || lx < 1) return 1.0;
    double part = 0.0;
    for (int t = tid; t < lx; t += kCtcThreads) part += exp((double)p.moff[(int64_t)b * p.T + t]);   // max_v p_t[v] (the offset IS its log)
    part = wave_sum_d(part);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = part;
    __syncthreads();
    double sum = 0.0;
#pragma unroll
    for (int i = 0; i < kCtcWaves; ++i) sum += red[i];
    __syncthreads();
    const double c = fmin(1.0, fmax(0.0, sum / (double)lx));
    const double r = fmin(0.79, (double)Sx / (double)lx);   // (0.8 = the untilted chain's own speed: rho_u = 1)
    const double a2 = 1.0 - 0.5 * r, a1 = 1.0 - r;
    const double rho_u = (-a1 + sqrt(a1 * a1 + 4.0 * r * a2)) / (2.0 * a2);
    const double theta = -log(rho_u) * (1.0 - c) * (double)p.ctc_tilt * 0.01;
    return fmin(1.0, fmax(0x1p-8, exp(-theta)));
}
__device__ __forceinline__ double frame_max_d(const double *wm) {
    double m = wm[0];
#pragma unroll
    for (int i = 1; i < kCtcWaves; ++i) m = fmax(m, wm[i]);
    return m;
}
__device__ __forceinline__ float to_log_d(double zs, int e, double mxs) {
    return zs > 0.0 ? (float)(log(zs) - (double)e * 0.6931471805599453 + mxs) : -INFINITY;
}

// The factor that turns the scaled products A~_t[s] * Bx~_t[s] of frame `bt` into posteriors, or 0 with the frame MARKED for the
// log-domain fallback: A~ and Bx~ are rescaled to a maximum in [2^40, 2^41), so a factor beyond 2^900 means the entries that carry
// the frame's mass are ~2^-900 below the maxima -- at the bottom of the fp64 range, where they are rounded away or flushed (and the
// factor itself overflows: 0 * inf).  A frame that passes has every entry that matters to 1e-9 of the posterior as a normal number,
// at this frame and -- mass never grows along a path -- at every frame before it.
constexpr int kCtcSafeExp = 900;
__device__ __forceinline__ double ctc_frame_factor(const LossParams &p, int b, int64_t bt, double invc, int ezc) {
    const int e = ezc - p.ECA[bt] - p.ECB[bt];
    // (atomicMax, not a store: blocks of ONE utterance run side by side, and a plain "= 1" here took back the 2 -- redo the whole utterance --
    // that the chains had set, after other blocks had already left their frames to the whole-utterance fix: frames with no numerator mass at
    // all, found by tests/test_gpu_fuzz.py in round 5)
    if (e + ilogb(invc) > kCtcSafeExp) { p.ctc_bad[bt] = 1; atomicMax(&p.redo_ctc[b], 1); return 0.0; }
    return ldexp(invc, e);
}
// scaled partition sum of the numerator as the grad kernels use it: 0 (= "contribute nothing") for an utterance that is redone whole
__device__ __forceinline__ double ctc_zc_for_grad(const LossParams &p, int b) { return p.redo_ctc[b] == 2 ? 0.0 : p.ctc_zc[b]; }

// NR = ctc states per thread actually needed (ceil((2L+1)/512) rounded up to 1, 2, 4, 8): a frame is ONE
// in-order instruction stream per wave (~5 cycles per instruction, dependent or not), so the predicated-off
// register iterations of a fixed NR = 8 were most of a frame's ~440 instructions for ordinary label lengths.
template <int NR>
__device__ __forceinline__ void ctc_forward(const LossParams &p, int b, float *lds) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int V = p.V, lx = p.lx[b], L = p.ly[b], Sx = 2 * L + 1, Sxp = rup64(Sx);
    const CtcLds c = ctc_carve(lds, Sxp);
    double *A = c.A;
    int *wmi = (int *)c.wm;                    // [3] frame maxima (high words) in rotation
    int sr = 1;                                // slot read by the next frame (frame 1 reads slot 1)
    const int *lab = c.lab;
    const int64_t bt0 = (int64_t)b * p.T;
    const bool valid = ctc_setup(p, b, c, L, lx, tid);
    if (!valid) {
        if (tid == 0) {
            const bool empty_ok = (lx <= 0 && L == 0);
            p.ctc_zc[b] = 0.0; p.ctc_ez[b] = 0; p.cost_ctc[b] = 0.f; p.invalid[b] = empty_ok ? 0 : 1;
        }
        return;
    }
    int mylab[NR];
    bool skip[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int s = tid + i * kCtcThreads;
        mylab[i] = s < Sx ? lab[s] : 0;
        skip[i] = s < Sx && s >= 2 && mylab[i] != 0 && mylab[i] != lab[s - 2];
    }
    // the label as an UNSIGNED byte offset: the emission loads are then `global_load v, v_off, s[row]` -- a uniform row base and one VGPR; with
    // a signed index the compiler kept a 64-bit pointer per lane (logp + label) and added the row to it with a v_lshl_add_u64 per load: the
    // kernel's 97th register, one more than lets a chain workgroup sit beside two grad workgroups of up to 160 (round 5, crf_grad_den_kernel's head)
    unsigned labo[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) labo[i] = (unsigned)mylab[i] * 4u;
    int E = kScaleExpD;
    const double rho = ctc_rho(p, b, Sx, lx, c.red, tid), rho2 = rho * rho;
    // The frame maximum used for the (exact, power-of-two) rescale is taken from the values as they are
    // WRITTEN: one barrier per frame instead of a separate reduction pass plus barrier.
    {   // t = 0 (gpu_ctc_kernels.h:146-152)
        const int64_t lr0 = bt0 * V;
        const float m0 = p.mx[bt0];
        double *CArow = p.CA + bt0 * p.Sc;
        double vmax = 0.0;
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int s = tid + i * kCtcThreads;
            if (s < Sxp) {
                const double v = (s < 2 && s < Sx) ? exp_scaled_d(ld_x(p, lr0 + mylab[i]) - m0) * pow2d(kScaleExpD) * (s == 1 ? rho : 1.0) : 0.0;
                A[s] = v;
                A[Sxp + s] = 0.0;
                if (s < Sx) CArow[s] = v;
                vmax = fmax(vmax, v);
            }
        }
        if (tid == 0) p.ECA[bt0] = E;
        // The frame maximum is ONE LDS word per frame (three in rotation: read / accumulated by ds_max_i32 / cleared), the high word of the
        // largest value: all the rescale needs is its exponent.  (Round 4; before, eight per-wave doubles that every thread read back and
        // reduced: 8 LDS reads and 7 fp64 maxima per thread and frame on a latency chain -- and the chains' end decides when the den half
        // of the grad pass may start.)
        if (tid < 3) wmi[tid] = 0;
        __syncthreads();
        const int hm = wave_max_hi(vmax);
        if (lane == 0) (void)__hip_atomic_fetch_max(wmi + 1, hm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // slot t % 3 is read by frame t: slot 1 for t = 1
    }
    __syncthreads();
    // Emissions are fetched in BATCHES of kCtcPF frames into two alternating register sets.  A gather from
    // L2/HBM takes ~1 us, a frame ~0.5 us, and vmcnt counts in order: with branches around the (Sx-
    // dependent) loads and stores the compiler cannot count what is younger than a prefetch and waits
    // vmcnt(0) -- i.e. for everything issued up to the previous frame, which ties the frame time to the
    // memory latency.  Batching leaves ONE such wait per kCtcPF frames, for loads issued kCtcPF frames ago.
    // The row maximum is read through a per-lane (VGPR) address: as a scalar load it would be counted by
    // lgkmcnt and the per-frame LDS barrier would wait for it.
    float lr[2][kCtcPF][NR], mr[2][kCtcPF];
    auto fetch1 = [&](auto SET, auto F, int t) __attribute__((always_inline)) {
        constexpr int st = decltype(SET)::value, f = decltype(F)::value;
        if (t < lx) {
            const int64_t row0 = (bt0 + t) * V;
            unsigned vz;   // (a fresh zero per load: hoisted out of the loop, `p.mx + vz` was a 64-bit pointer per lane -- and the register the kernel spilled)
            asm volatile("v_mov_b32 %0, 0" : "=v"(vz));
            mr[st][f] = *(const float *)((const char *)(p.mx + bt0 + t) + vz);
            if (p.in_dtype == 0) {   // (one uniform branch around the batch: the fp32 loads stay as they were)
                const float *row = p.logp + row0;
#pragma unroll
                for (int i = 0; i < NR; ++i) lr[st][f][i] = (tid + i * kCtcThreads < Sx) ? *(const float *)((const char *)row + labo[i]) : 0.f;
            } else {
#pragma unroll
                for (int i = 0; i < NR; ++i) lr[st][f][i] = (tid + i * kCtcThreads < Sx) ? ld_x(p, row0 + mylab[i]) : 0.f;
            }
        }
    };
    auto fetch4 = [&](auto SET, int t) __attribute__((always_inline)) {
        fetch1(SET, std::integral_constant<int, 0>{}, t);
        fetch1(SET, std::integral_constant<int, 1>{}, t + 1);
        fetch1(SET, std::integral_constant<int, 2>{}, t + 2);
        fetch1(SET, std::integral_constant<int, 3>{}, t + 3);
    };
    static_assert(kCtcPF == 4, "fetch4 / the frame loop are written for batches of four");
    auto step = [&](auto SET, auto F, int t) __attribute__((always_inline)) {
        constexpr int st = decltype(SET)::value, f = decltype(F)::value;
        const double *Ac = A + ((t - 1) & 1) * Sxp;
        double *An = A + (t & 1) * Sxp;
        double em[NR];
        [[maybe_unused]] const bool tm_on = b == 3 && t >= 100 && t < 228 && wave == 0;
        [[maybe_unused]] const int tm_i = 14336 + (t - 100) * 8;
        CRF_TM(tm_on, tm_i + 0);
        // The frame's LDS reads -- its maximum and the three neighbours of every state -- are issued TOGETHER, unconditionally (clamped index,
        // coefficient 0 where the transition does not exist: fma(0, x, a) = a exactly) and, with one or two states per thread, in FRONT of the
        // emissions' exp, which needs registers only.  As `if (s >= 1) ...; if (skip) ...` each neighbour sat in its own divergent block behind
        // its own lgkmcnt(0), and the maximum's read was cut off from them by thread 0's clearing store: four LDS round trips in a row on the
        // frame's dependency chain, all behind the exp (round 4, found in the ISA).  (NR > 2: that costs registers the kernel does not have.)
        constexpr bool RT = NR <= 2;
        int wv = 0;
        double a0[NR], a1[NR], a2[NR];
        if constexpr (RT) {
            wv = lds_issue_i32(wmi + sr);
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                const int s0 = min(tid + i * kCtcThreads, Sxp - 1);
                a0[i] = lds_issue_f64(Ac + s0); a1[i] = lds_issue_f64(Ac + max(s0 - 1, 0)); a2[i] = lds_issue_f64(Ac + max(s0 - 2, 0));
            }
        }
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            float x = lr[st][f][i] - mr[st][f];
            asm volatile("" : "+v"(x));  // keeps the fp64 exp of LATER frames of the batch from being hoisted up here (VGPRs)
            em[i] = (tid + i * kCtcThreads < Sx) ? exp_scaled_d(x) : 0.0;
        }
        CRF_TM(tm_on, tm_i + 1);
        if (f == 0) fetch4(std::integral_constant<int, 1 - st>{}, t + kCtcPF);  // after this batch has landed
        CRF_TM(tm_on, tm_i + 2);
        auto nbr = [&](int i) __attribute__((always_inline)) {
            const int s0 = min(tid + i * kCtcThreads, Sxp - 1);
            a0[i] = Ac[s0]; a1[i] = Ac[max(s0 - 1, 0)]; a2[i] = Ac[max(s0 - 2, 0)];
        };
        if constexpr (RT) {
#pragma unroll
            for (int i = 0; i < NR; ++i) lds_landed(a0[i], a1[i], a2[i], wv);
        } else wv = wmi[sr];
        const int k = rescale_exp_hi(wv);
        const int sw = sr == 2 ? 0 : sr + 1, sz = sw == 2 ? 0 : sw + 1;   // accumulated during this frame / cleared in it
        const double sc = pow2d(k);
        E += k;
        double *CArow = p.CA + (bt0 + t) * p.Sc;
        double vmax = 0.0;
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int s = tid + i * kCtcThreads;
            if constexpr (!RT) nbr(i);
            double a = fma((s >= 1 && s < Sx) ? rho : 0.0, a1[i], a0[i]);
            a = fma(skip[i] ? rho2 : 0.0, a2[i], a);
            const double v = sc * em[i] * a;
            if (s < Sx) {
                An[s] = v;
                CArow[s] = v;
            }
            vmax = fmax(vmax, s < Sx ? v : 0.0);
        }
        CRF_TM(tm_on, tm_i + 3);
        if (tid == 0) { wmi[sz] = 0; p.ECA[bt0 + t] = E; }
        const int hm = wave_max_hi(vmax);
        if (lane == 0) (void)__hip_atomic_fetch_max(wmi + sw, hm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        sr = sw;
        CRF_TM(tm_on, tm_i + 4);
        sync_lds();
        CRF_TM(tm_on, tm_i + 5);
    };
    {
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
        fetch4(I0{}, 1);
        for (int t = 1; t < lx; t += 2 * kCtcPF) {
            // NESTED, not eight independent `if (t + k < lx)`: behind independent conditions the compiler has to assume that the step which waited for
            // a batch may not have run, and every later step of the batch waited again -- vmcnt(0), i.e. for the previous frame's row store to be
            // acknowledged (three frames in eight; round 4, found in the ISA)
            step(I0{}, I0{}, t);
            if (t + 1 < lx) {
                step(I0{}, I1{}, t + 1);
                if (t + 2 < lx) {
                    step(I0{}, I2{}, t + 2);
                    if (t + 3 < lx) {
                        step(I0{}, I3{}, t + 3);
                        if (t + 4 < lx) {
                            step(I1{}, I0{}, t + 4);
                            if (t + 5 < lx) {
                                step(I1{}, I1{}, t + 5);
                                if (t + 6 < lx) {
                                    step(I1{}, I2{}, t + 6);
                                    if (t + 7 < lx) step(I1{}, I3{}, t + 7);
                                }
                            }
                        }
                    }
                }
            }
        }
    }
    const double *Af = A + ((lx - 1) & 1) * Sxp;
    const double mxs = ctc_mx_total(p, b, lx, c.red, tid);
    if (tid == 0) {
        const double zc = Af[Sx - 1] + (Sx > 1 ? rho * Af[Sx - 2] : 0.0);   // Z' (ctc_rho)
        const bool ok = zc > 0.0 && zc < INFINITY;
        p.ctc_zc[b] = ok ? zc : 0.0;
        p.ctc_ez[b] = E;
        p.cost_ctc[b] = ok ? (float)(log(zc) - (double)E * 0.6931471805599453 + mxs - (double)(Sx - 1) * log(rho)) : 0.f;
        p.invalid[b] = ok ? 0 : 1;
        if (!ok) p.redo_ctc[b] = 2;   // a VALID label sequence whose scaled chain lost all its mass: the log-domain kernels decide
    }
}

// backward, EXCLUDING the emission at t:  Bx_t[s] = sum_{s' in {s,s+1,s+2*}} e_{t+1}[l'_s'] Bx_{t+1}[s']
// LDS holds Y_t[s] = e_t[l'_s] * Bx_t[s]; Bx_t itself only goes to HBM (CB).
template <int NR>
__device__ __forceinline__ void ctc_backward(const LossParams &p, int b, float *lds) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int V = p.V, lx = p.lx[b], L = p.ly[b], Sx = 2 * L + 1, Sxp = rup64(Sx);
    const CtcLds c = ctc_carve(lds, Sxp);
    double *Y = c.A;
    int *wmi = (int *)c.wm;
    int sr = 1;
    const int *lab = c.lab;
    const int64_t bt0 = (int64_t)b * p.T;
    if (!ctc_setup(p, b, c, L, lx, tid)) return;
    int mylab[NR];
    bool skip[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int s = tid + i * kCtcThreads;
        mylab[i] = s < Sx ? lab[s] : 0;
        skip[i] = (s + 2 < Sx) && lab[s + 2] != 0 && lab[s + 2] != mylab[i];
    }
    unsigned labo[NR];   // (see ctc_forward)
#pragma unroll
    for (int i = 0; i < NR; ++i) labo[i] = (unsigned)mylab[i] * 4u;
    int F_ = kScaleExpD;
    const double rho = ctc_rho(p, b, Sx, lx, c.red, tid), rho2 = rho * rho;
    {   // t = lx-1
        const int64_t lr0 = (bt0 + lx - 1) * V;
        const float ml = p.mx[bt0 + lx - 1];
        double *CBrow = p.CB + (bt0 + lx - 1) * p.Sc;
        double vmax = 0.0;
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int s = tid + i * kCtcThreads;
            if (s < Sxp) {
                const double bx = (s < Sx && s >= Sx - 2) ? pow2d(kScaleExpD) * (s == Sx - 2 ? rho : 1.0) : 0.0;
                const double y = s < Sx ? exp_scaled_d(ld_x(p, lr0 + mylab[i]) - ml) * bx : 0.0;
                Y[s] = y;
                Y[Sxp + s] = 0.0;
                if (s < Sx) CBrow[s] = bx;
                vmax = fmax(vmax, y);
            }
        }
        if (tid == 0) p.ECB[bt0 + lx - 1] = F_;
        if (tid < 3) wmi[tid] = 0;                    // (frame maxima: ctc_forward)
        __syncthreads();
        const int hm = wave_max_hi(vmax);
        if (lane == 0) (void)__hip_atomic_fetch_max(wmi + 1, hm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // read by iteration i = 1
    }
    __syncthreads();
    // emissions in batches of kCtcPF frames, as in ctc_forward
    float lr[2][kCtcPF][NR], mr[2][kCtcPF];
    auto fetch1 = [&](auto SET, auto F, int t) __attribute__((always_inline)) {
        constexpr int st = decltype(SET)::value, f = decltype(F)::value;
        if (t >= 0) {
            const int64_t row0 = (bt0 + t) * V;
            unsigned vz;   // (a fresh zero per load: hoisted out of the loop, `p.mx + vz` was a 64-bit pointer per lane -- and the register the kernel spilled)
            asm volatile("v_mov_b32 %0, 0" : "=v"(vz));
            mr[st][f] = *(const float *)((const char *)(p.mx + bt0 + t) + vz);
            if (p.in_dtype == 0) {
                const float *row = p.logp + row0;
#pragma unroll
                for (int q = 0; q < NR; ++q) lr[st][f][q] = (tid + q * kCtcThreads < Sx) ? *(const float *)((const char *)row + labo[q]) : 0.f;
            } else {
#pragma unroll
                for (int q = 0; q < NR; ++q) lr[st][f][q] = (tid + q * kCtcThreads < Sx) ? ld_x(p, row0 + mylab[q]) : 0.f;
            }
        }
    };
    auto fetch4 = [&](auto SET, int t) __attribute__((always_inline)) {  // frames t, t-1, t-2, t-3
        fetch1(SET, std::integral_constant<int, 0>{}, t);
        fetch1(SET, std::integral_constant<int, 1>{}, t - 1);
        fetch1(SET, std::integral_constant<int, 2>{}, t - 2);
        fetch1(SET, std::integral_constant<int, 3>{}, t - 3);
    };
    // iteration i handles frame t = lx-1-i with the emissions of frame t
    auto step = [&](auto SET, auto F, int i) __attribute__((always_inline)) {
        constexpr int st = decltype(SET)::value, f = decltype(F)::value;
        const int t = lx - 1 - i;
        const double *Yc = Y + ((i - 1) & 1) * Sxp;
        double *Yn = Y + (i & 1) * Sxp;
        double em[NR];
        constexpr bool RT = NR <= 2;   // (the frame's LDS reads together, unconditional, in front of the exp: see ctc_forward)
        int wv = 0;
        double a0[NR], a1[NR], a2[NR];
        if constexpr (RT) {
            wv = lds_issue_i32(wmi + sr);
#pragma unroll
            for (int q = 0; q < NR; ++q) {
                const int s0 = min(tid + q * kCtcThreads, Sxp - 1);
                a0[q] = lds_issue_f64(Yc + s0); a1[q] = lds_issue_f64(Yc + min(s0 + 1, Sxp - 1)); a2[q] = lds_issue_f64(Yc + min(s0 + 2, Sxp - 1));
            }
        }
#pragma unroll
        for (int q = 0; q < NR; ++q) {
            float x = lr[st][f][q] - mr[st][f];
            asm volatile("" : "+v"(x));
            em[q] = (tid + q * kCtcThreads < Sx) ? exp_scaled_d(x) : 0.0;
        }
        if (f == 0) fetch4(std::integral_constant<int, 1 - st>{}, t - kCtcPF);
        auto nbr = [&](int q) __attribute__((always_inline)) {
            const int s0 = min(tid + q * kCtcThreads, Sxp - 1);
            a0[q] = Yc[s0]; a1[q] = Yc[min(s0 + 1, Sxp - 1)]; a2[q] = Yc[min(s0 + 2, Sxp - 1)];
        };
        if constexpr (RT) {
#pragma unroll
            for (int q = 0; q < NR; ++q) lds_landed(a0[q], a1[q], a2[q], wv);
        } else wv = wmi[sr];
        const int k = rescale_exp_hi(wv);
        const int sw = sr == 2 ? 0 : sr + 1, sz = sw == 2 ? 0 : sw + 1;
        const double sc = pow2d(k);
        F_ += k;
        double *CBrow = p.CB + (bt0 + t) * p.Sc;
        double vmax = 0.0;
#pragma unroll
        for (int q = 0; q < NR; ++q) {
            const int s = tid + q * kCtcThreads;
            if constexpr (!RT) nbr(q);
            double a = fma(s + 1 < Sx ? rho : 0.0, a1[q], a0[q]);
            a = fma(skip[q] ? rho2 : 0.0, a2[q], a);
            const double bx = sc * a;
            const double y = em[q] * bx;
            if (s < Sx) {
                CBrow[s] = bx;
                Yn[s] = y;
            }
            vmax = fmax(vmax, s < Sx ? y : 0.0);
        }
        if (tid == 0) { wmi[sz] = 0; p.ECB[bt0 + t] = F_; }
        const int hm = wave_max_hi(vmax);
        if (lane == 0) (void)__hip_atomic_fetch_max(wmi + sw, hm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        sr = sw;
        sync_lds();
    };
    {
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
        fetch4(I0{}, lx - 2);
        for (int i = 1; i < lx; i += 2 * kCtcPF) {
            // (nested: see ctc_forward)
            step(I0{}, I0{}, i);
            if (i + 1 < lx) {
                step(I0{}, I1{}, i + 1);
                if (i + 2 < lx) {
                    step(I0{}, I2{}, i + 2);
                    if (i + 3 < lx) {
                        step(I0{}, I3{}, i + 3);
                        if (i + 4 < lx) {
                            step(I1{}, I0{}, i + 4);
                            if (i + 5 < lx) {
                                step(I1{}, I1{}, i + 5);
                                if (i + 6 < lx) {
                                    step(I1{}, I2{}, i + 6);
                                    if (i + 7 < lx) step(I1{}, I3{}, i + 7);
                                }
                            }
                        }
                    }
                }
            }
        }
    }
}

// =============================================================================================
// Register-resident denominator recursions (crf_internal.h: ResDev, res_layout.cpp).
// A recursion of one utterance runs on K compute units; each thread holds its arcs in VGPRs, so a
// frame is: max-reduce -> kResNCH x (4 LDS gathers + 4 FMA), row epilogue at every slice end ->
// barrier -> (K > 1) all-gather of the new state vector through tagged 8-byte granules in L2.
// =============================================================================================
constexpr int kEpRegsR = 2;   // emission-row prefetch registers (V <= 2*512 for the resident kernels)
constexpr int kPoll = 4;      // granules polled concurrently per thread

template <int NW = kResWaves>
__device__ __forceinline__ float res_block_sum(float v, float *red, int tid) {
    v = wave_sum(v);
    sync_lds();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    sync_lds();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NW; ++i) s += red[i];
    return s;
}
template <int NW = kResWaves, typename P>
__device__ __forceinline__ double res_mx_total(const P &p, int b, int lx, double *red, int tid) {
    double part = 0.0;
    for (int t = tid; t < lx; t += NW * kWave) part += (double)p.mx[(int64_t)b * p.T + t];
    part = wave_sum_d(part);
    sync_lds();
    if ((tid & 63) == 0) red[tid >> 6] = part;
    sync_lds();
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < NW; ++i) s += red[i];
    return s;
}
template <int NW = kResWaves>
__device__ __forceinline__ float res_frame_max(const float *wm) {
    float m = wm[0];
#pragma unroll
    for (int i = 1; i < NW; ++i) m = fmaxf(m, wm[i]);
    return m;
}

// Exchange of the state vector between the K CUs of one recursion.  Granule = {tag << 32 | float
// bits}, ONE aligned 8-byte agent-scope store / load each: the data is the flag
// (cdna_hip_programming.md G16 R2).  Slots alternate per frame; a peer can be at most one exchange
// ahead, so two slots suffice.  Values are published from the row epilogues as soon as they are
// final, so most of the hand-off latency overlaps the rest of the frame's gathers.
typedef __attribute__((address_space(1))) unsigned long long gu64;
// `same_l2`: every peer of this recursion was found (at run time, from HW_REG_XCC_ID) on this CU's XCD.
// Then a PLAIN 8-byte store is enough: it is written through to the shared L2 and stays there, and the
// peers' loads bypass their L1 (sc1), so the hand-off is an L2 round trip.  Otherwise the store is
// write-through to memory (sc1): correct under any placement, ~1 us slower per frame (an sc1 store drops
// the line from L2, MI355X_MICROARCH.md "stores of each flavour").
__device__ __forceinline__ void res_publish(gu64 *slot, int i, unsigned tag, float v, bool same_l2) {
    const unsigned long long g = ((unsigned long long)tag << 32) | __float_as_uint(v);
    if (same_l2) {
        asm volatile("global_store_dwordx2 %0, %1, off" : : "v"(slot + i), "v"(g) : "memory");
    } else {
        __hip_atomic_store(slot + i, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// Fetch entries [lo,hi) published by a peer into LDS `v`; returns the largest value seen.
template <int NT = kResThreads>
__device__ __forceinline__ float res_fetch(gu64 *slot, float *v, int lo, int hi, unsigned tag, int *err, int tid) {
    float mx = 0.f;
    for (int base = lo; base < hi; base += kPoll * NT) {
        unsigned pending = 0;
#pragma unroll
        for (int q = 0; q < kPoll; ++q)
            if (base + q * NT + tid < hi) pending |= 1u << q;
        for (unsigned spins = 0; pending; ++spins) {
            unsigned long long gv[kPoll];
#pragma unroll
            for (int q = 0; q < kPoll; ++q)
                if (pending >> q & 1) gv[q] = __hip_atomic_load(slot + base + q * NT + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int q = 0; q < kPoll; ++q)
                if ((pending >> q & 1) && (unsigned)(gv[q] >> 32) == tag) {
                    const float x = __uint_as_float((unsigned)gv[q]);
                    v[base + q * NT + tid] = x;
                    mx = fmaxf(mx, x);
                    pending &= ~(1u << q);
                }
            if (!pending) break;
            // a peer died or was never scheduled: give up loudly (error word -> NaN loss), never hang;
            // once the word is set nobody waits again
            if (spins > (1u << 24)) { __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }  // ~tens of seconds
            if ((spins & 255u) == 255u && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
            if (spins > 64) __builtin_amdgcn_s_sleep(8);  // back off when the peer is clearly not there yet
            else if (spins > 2) __builtin_amdgcn_s_sleep(1);
        }
    }
    return mx;
}

// ... the same for a LIST of entries (one per thread and round)
template <int NT>
__device__ __forceinline__ float res_fetch_list(gu64 *slot, float *v, const int *__restrict__ list, int l0, int l1, unsigned tag, int *err, int tid) {
    float mx = 0.f;
    for (int j = l0 + tid; j < l1; j += NT) {
        const int e = list[j];
        for (unsigned spins = 0;; ++spins) {
            const unsigned long long g = __hip_atomic_load(slot + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((unsigned)(g >> 32) == tag) { const float x = __uint_as_float((unsigned)g); v[e] = x; mx = fmaxf(mx, x); break; }
            if (spins > (1u << 24)) { __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            if ((spins & 255u) == 255u && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
            if (spins > 64) __builtin_amdgcn_s_sleep(8);
            else if (spins > 2) __builtin_amdgcn_s_sleep(1);
        }
    }
    return mx;
}

// ... a range AND a list in ONE polling loop (round 4: the two-CU forward frame fetched the peer's U range and then, behind it, its list of
// L / A entries -- two dependent L2 round trips per frame where one will do): every thread polls up to kPoll granules of the range and one
// entry of the list together; a list longer than the workgroup takes res_fetch_list for the rest.
template <int NT>
__device__ __forceinline__ float res_fetch_both(gu64 *slot, float *v, int lo, int hi, const int *__restrict__ list, int l0, int l1, unsigned tag, int *err, int tid) {
    float mx = 0.f;
    const int le = l0 + tid < l1 ? list[l0 + tid] : -1;   // this thread's list entry (first round of the list)
    bool first = true;
    for (int base = lo; base < hi || first; base += kPoll * NT) {
        unsigned pending = 0;
#pragma unroll
        for (int q = 0; q < kPoll; ++q)
            if (base + q * NT + tid < hi) pending |= 1u << q;
        if (first && le >= 0) pending |= 1u << kPoll;
        first = false;
        for (unsigned spins = 0; pending; ++spins) {
            unsigned long long gv[kPoll + 1];
#pragma unroll
            for (int q = 0; q < kPoll; ++q)
                if (pending >> q & 1) gv[q] = __hip_atomic_load(slot + base + q * NT + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (pending >> kPoll & 1) gv[kPoll] = __hip_atomic_load(slot + le, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int q = 0; q < kPoll; ++q)
                if ((pending >> q & 1) && (unsigned)(gv[q] >> 32) == tag) {
                    const float x = __uint_as_float((unsigned)gv[q]);
                    v[base + q * NT + tid] = x;
                    mx = fmaxf(mx, x);
                    pending &= ~(1u << q);
                }
            if ((pending >> kPoll & 1) && (unsigned)(gv[kPoll] >> 32) == tag) {
                const float x = __uint_as_float((unsigned)gv[kPoll]);
                v[le] = x;
                mx = fmaxf(mx, x);
                pending &= ~(1u << kPoll);
            }
            if (!pending) break;
            if (spins > (1u << 24)) { __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            if ((spins & 255u) == 255u && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
            if (spins > 64) __builtin_amdgcn_s_sleep(8);
            else if (spins > 2) __builtin_amdgcn_s_sleep(1);
        }
    }
    if (l0 + NT < l1) mx = fmaxf(mx, res_fetch_list<NT>(slot, v, list, l0 + NT, l1, tag, err, tid));
    return mx;
}

// Chunk sums in batches of kResBatch chunks: the 4*kResBatch gathers of a batch are one straight-line
// block (no branch between them), so every wave keeps 24 independent ds_read_b32 in flight -- with
// only 2 waves per SIMD that, not occupancy, is what hides the LDS latency.  The (uniform) slice-end
// branches sit between batches.  Unused chunks hold zero weights.
constexpr int kResBatch = 6;   // measured: 5 -> 6 = -2% (fewer, longer straight-line blocks); 10 spills
static_assert(kResNCH % kResBatch == 0, "kResNCH must be a multiple of kResBatch");
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// Packed math: the four products of a chunk are two v_pk_fma_f32 lanes (PMC showed the frame loop is
// as much VALU-issue-bound as LDS-bound: ~530 VALU instructions per wave and frame before packing).
// The 4*kResBatch gathers of a batch; the products are chained FMAs into the row accumulator (measured:
// the frame loop is bound by instruction issue as much as by LDS -- one packed instruction less per
// chunk than "multiply, FMA, add" was worth 3%).
#define CRF_RES_GATHER(g01, g23, A, xb, c0) CRF_RES_GATHER_N(g01, g23, A, xb, c0, kResBatch)
#define CRF_RES_GATHER_N(g01, g23, A, xb, c0, NB_)                                                        \
    _Pragma("unroll") for (int ci = 0; ci < (NB_); ++ci) {                                            \
        const int c = (c0) + ci;                                                                          \
        const unsigned i01 = A[6 * c], i23 = A[6 * c + 1];                                                \
        g01[ci].x = *(const float *)(xb + (i01 & 0xffffu)); g01[ci].y = *(const float *)(xb + (i01 >> 16)); \
        g23[ci].x = *(const float *)(xb + (i23 & 0xffffu)); g23[ci].y = *(const float *)(xb + (i23 >> 16)); \
    }
#define CRF_RES_CHUNK_ACC(accv, g01, g23, A, c, ci)                                                       \
    {                                                                                                     \
        f32x2 w01, w23;                                                                                   \
        w01.x = __uint_as_float(A[6 * (c) + 2]); w01.y = __uint_as_float(A[6 * (c) + 3]);                 \
        w23.x = __uint_as_float(A[6 * (c) + 4]); w23.y = __uint_as_float(A[6 * (c) + 5]);                 \
        accv = __builtin_elementwise_fma(g23[ci], w23, __builtin_elementwise_fma(g01[ci], w01, accv));    \
    }

// Kernel arguments of the resident kernels: only what ONE direction needs (the full LossParams is ~90
// SGPRs of pointers, most of which the compiler would keep live or spill around the unrolled frame body).
struct ResParams {
    ResDirDev L;
    int K, B, T, V, b0, rows_cu_max, Rout, Gf, Gb;
    const int *lx;
    const float *ep, *mx;
    float *Out;                 // Q (fwd) or BP (bwd) rows
    float *Row0;                // [B][Rout] spare rows: b_0 of the backward recursion
    int *Eout;                  // EQ (fwd) or EB (bwd)
    unsigned long long *xch;
    int *err;
    // forward side tables / results
    const float *x_start, *x_end;
    float *den_zs, *cost_alpha;
    int *den_ez;
    // backward side tables / results
    const int *z_lab;
    const float *z_end, *brow_start, *brow_end;
    const int2 *bcsr;
    float *cb_part;
    double *cb_mxs;
    int *cb_F;
    int *redo;                  // [2][B], see LossParams
};

// LDS map of the resident kernels: the two state-vector buffers sit at byte offsets 0 and kResXB; a gather
// is `ds_read_b32 v, (buffer base SGPR + 16-bit offset from the packed arc word)` -- one VALU (an SDWA add)
// per arc besides the FMA.
constexpr int kResXB = 65536;                 // bytes per state-vector buffer  -> gather vector <= 16384 entries (16-bit byte offsets);
                                              // 32 KiB until round 2: graphs of 8 k - 16 k states fell to the utterance-minor kernels (27 ms
                                              // instead of ~14 at S = 8193 / A = 208 k)
constexpr int kResGmax = kResXB / 4;

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

// =============================================================================================
// FACTORED resident recursions (crf_internal.h: FacDev, res_layout.cpp: build_factored): one compute unit
// per recursion and utterance, no exchange.  Same arithmetic and scaling as crf_res_chain_kernel, same arc
// format (4 gathers of one float + 4 weights per chunk); what differs is the row epilogue, which has TWO outputs:
//   forward : a row belongs to the main state s2 of a pair (s1, s2).  L' = e'[l2] * rowsum is a[s2];
//             A' = e'[l1] * w * U is a[s1] (its own row is exactly w * (a[s1] + a[s2]) = w * U); U' = A' + L'
//             is the entry every other row gathers for the pair.  Q row: [rowsum of each row | w*U of each row].
//   backward: a row is the common out-arc sum of one or two states; the epilogue adds each state's one
//             extra arc (BP positions / z entries 2*rid, 2*rid + 1).
// LDS: V0 | V1 (two state vectors of Gp floats) | row metadata int4[R] | EP[2][Vp] | wm | red
// =============================================================================================
struct FacParams {
    FacDirDev L;
    int B, T, V, Rout, NT, Rf;
    const int *lx;
    const float *ep, *mx;
    float *Out;                 // Q (fwd) or BP (bwd) rows
    float *Row0;                // [B][Rout] spare rows: b_0 of the backward recursion
    int *Eout;                  // EQ (fwd) or EB (bwd)
    int *started;               // workgroups of the den kernels that have started (gate for the numerator chains)
    int i0, i1;                 // iterations of this launch (segment)
    int nb;                     // stage bounds (iterations), bound[0] = 0 < ... < bound[nb-1] >= T; nb <= 1: no stage flags
    int bound[16];
    int *stage_cnt;             // [16] fine-grained counters: += 1 per utterance when the rows of all iterations < bound[k] are in memory
    float *state;               // [B][rup64(G) + 64] parked state vector and exponent between segments
    const int4 *frow_meta;
    const float *x_start, *x_end;
    float *den_zs, *cost_alpha;
    int *den_ez;
    const int4 *brow_meta;
    const int *z_lab;
    const float *z_end, *brow_start, *brow_end;
    const int *bx_idx; const float *bx_w; int nbx; float bx_se;   // rowless states of the backward recursion (FacDev)
    float *cb_part;
    double *cb_mxs;
    int *cb_F;
    int *redo;                  // [2][B], see LossParams
    // two CUs per recursion (FacDev::K = 2): the utterances [b0, b0 + nbu) of this launch, the exchange granules, the error word
    int K, b0, nbu, Gf, Gb;
    const int *xlist;           // forward: the L / A entries this CU fetches every frame (FacDev::xlist), [xl0, xl1)
    int xlist_off[3];
    unsigned long long *xch;
    int *err;
    // two utterances per workgroup (fac_chain_body2): pairs of this launch, rows that take the stores of an utterance that has ended
    int npair, dump_stride;
    float *dump;                // [2 directions][npair][dump_stride]
};

// FLAG: publish stage flags and store rows write-through (one instantiation per use: the frame loop has no
// run-time switch for it)
// NTH threads with NCH chunks of arcs each, gathered in batches of NB chunks: 512 x 30 (2 waves per SIMD) or
// 768 x 21 (3 waves per SIMD at <= 168 VGPRs -- the frame is latency-bound, a third wave fills the gaps).
// ML: some rows are cut into pieces on adjacent lanes (graphs with long rows; a separate instantiation, the check costs the
// row epilogue of the others 2 %)
// K2: TWO CUs per recursion (graphs of 120 k - 240 k arcs).  Each CU holds half of the rows (its share of the arc registers)
// and the WHOLE state vector: what its row epilogues produce -- U', L', A' of a forward row, the two z of a backward row -- is
// also published as {frame tag, value} granules (res_chain_body's protocol: the data is the flag), and after its last chunk the
// CU fetches the peer's entries into its own vector before the frame barrier.  Table geometry (RL) only, one copy of the
// gathered entries, no stages (2 B x 2 workgroups are every CU of the device: nothing runs beside the recursions).
template <int DIR, bool FLAG, int NTH, int NCH, int NB, bool ML, bool RL, bool K2 = false>
__device__ __forceinline__ void fac_chain_body(const FacParams &p, float *lds, const int b, const int k = 0) {
    static_assert(!K2 || (RL && !FLAG), "two CUs per recursion: table geometry, no stage flags");
    constexpr int NW = NTH / kWave;
    // 768-thread geometry: the last chunk slot of a thread holds ROW CONSTANTS instead of arcs -- two words for each of
    // the (at most three) rows the lane finishes per frame -- and the entries of a row sit where its row id says
    // (res_layout.cpp, "implicit"): a row epilogue asks for everything it needs from LDS in ONE round trip.  With a
    // table of row constants in LDS it was a chain of three (constants -> the values they point to -> emissions),
    // ~320 cycles of a wave's time per slice, 70 % of the frame loop (timing build).
    // RL (768 threads): the row constants are a TABLE in LDS after all -- 8 (forward) / 16 (backward) bytes per row, read one
    // slice AHEAD (the next slice's constants are requested in the epilogue of this one and arrive behind its gathers), so an
    // epilogue is still one round trip, a wave may finish any number of slices per frame, and no select between registers
    // is needed (with three slices that select is ~10 VALU instructions per epilogue).  For graphs with many short rows.
    static_assert(!RL || NTH != kResThreads, "the row-constant table goes with the 768- and 1024-thread geometries");
    constexpr bool IMP = NTH != kResThreads;                 // entries of a row lie where its row id says
    constexpr bool RC = IMP && !RL;
    constexpr int NCHA = RC ? kFac3ArcCh : NCH;              // chunk slots that hold arcs (RC: the last slot holds the row constants)
    constexpr int RCW = NCHA * 6;                            // first row-constant word
    static_assert(!RC || NCH == kFac3NCH, "row constants in registers: 20 chunks of arcs + the constants' slot");

    const FacDirDev &L = p.L;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int V = p.V, lx = p.lx[b], G = L.G, R = L.R;
    const unsigned dup = (unsigned)L.dup;                   // second copy of the gathered entries, other banks (res_layout.cpp pack_arcs)
    const int Vp = rup64(V + 1), Gp = rup64(G);
    const int XB = Gp * 4;                                   // bytes per state vector
    const int64_t bt0 = (int64_t)b * p.T;
    float *X = lds;                                          // [2][Gp]
    // row table.  K2: only the rows of this CU, [tr0, tr1) -- RMc is then the table's VIRTUAL base, so that "RMc + row bytes * rid"
    // addresses it by the global row id like everywhere else (the state vectors in front are longer than any such shift)
    constexpr int kRowB = RL ? (DIR == 0 ? 8 : 16) : 16;
    const int tr0 = K2 ? L.cu_row[k] : 0, tr1 = K2 ? L.cu_row[k + 1] : R;
    const int trn = K2 ? max(L.cu_row[1] - L.cu_row[0], L.cu_row[2] - L.cu_row[1]) : R;   // rows the table has room for (both CUs the same: fac_lds_bytes)
    char *RMc0 = (char *)(X + 2 * Gp);                       // int4[R]
    char *RMc = RMc0 - (size_t)tr0 * kRowB;
    float *EP = (float *)(RMc0 + (RL ? (size_t)(trn + 64) * kRowB : (size_t)R * 16));   // [2][Vp]  (RL: 64 rows of slack for the read-ahead)
    float *wm = EP + 2 * Vp;                                 // [2][NW]
    double *red = (double *)(wm + 2 * NW);            // [NW]
    if (tid == 0 && p.started && p.i0 == 0) atomicAdd(p.started, 1);   // this workgroup holds its CU: see crf_gate_kernel
    [[maybe_unused]] const bool lead = !K2 || k == 0;        // the CU that writes what exists once per recursion (exponents, logZ)
    [[maybe_unused]] gu64 *xchd = nullptr;                   // K2: this recursion's granules, [2 slots][G]
    [[maybe_unused]] gu64 *hs = nullptr;                     // K2: two handshake words of this recursion
    [[maybe_unused]] bool same_l2 = false;
    if constexpr (K2) {
        xchd = (gu64 *)p.xch + (DIR == 0 ? 0 : (size_t)p.B * 2 * (size_t)p.Gf) + (size_t)b * 2 * (size_t)G;
        hs = (gu64 *)p.xch + (size_t)p.B * 2 * ((size_t)p.Gf + (size_t)p.Gb) + ((size_t)DIR * p.B + b) * kResMaxK;
    }

    unsigned A[(NCH * 6)];
    unsigned rc00 = 0, rc01 = 0, rc10 = 0, rc11 = 0, rc20 = 0, rc21 = 0;   // row constants of the lane's (up to) three rows: scalars,
    {                                                                      // not array elements (a select between array elements
        const unsigned *src = L.arcs + (K2 ? (size_t)k * (NCH * 6) * NTH : (size_t)0) + tid;   // becomes a variable index and the array leaves the registers)
#pragma unroll
        for (int i = 0; i < (RC ? RCW : NCH * 6); ++i) A[i] = src[(size_t)i * NTH];
        if (RC) {
            rc00 = src[(size_t)(RCW + 0) * NTH]; rc01 = src[(size_t)(RCW + 1) * NTH];
            rc10 = src[(size_t)(RCW + 2) * NTH]; rc11 = src[(size_t)(RCW + 3) * NTH];
            rc20 = src[(size_t)(RCW + 4) * NTH]; rc21 = src[(size_t)(RCW + 5) * NTH];
        }
    }
    const uint4 wi = L.wave_info[(K2 ? k * NW : 0) + wave];
    const unsigned ends = __builtin_amdgcn_readfirstlane(wi.x);
    const int nch = __builtin_amdgcn_readfirstlane(wi.y);
    const int row0 = __builtin_amdgcn_readfirstlane(wi.z);
    const unsigned lgbits = __builtin_amdgcn_readfirstlane(wi.w);   // 3 bits per slice: its rows are cut into 2^lg pieces on adjacent lanes
    if (RL) {
        if (DIR == 0) {
            uint2 *RT = (uint2 *)RMc;                        // {emission byte offsets main | tail << 16, tail weight}
            for (int r = tr0 + tid; r < tr1; r += NTH) {
                const int4 m = p.frow_meta[r];
                RT[r] = uint2{(((unsigned)m.x >> 16) * 4u) | (((unsigned)m.w * 4u) << 16), (unsigned)m.z};
            }
        } else {
            uint4 *RT = (uint4 *)RMc;                        // {z byte offsets of the two extra arcs, emission byte offsets, their weights}
            for (int r = tr0 + tid; r < tr1; r += NTH) {
                const int4 m = p.brow_meta[r];
                const unsigned l0 = (unsigned)m.w & 0xffffu, l1 = (unsigned)m.w >> 16;   // 0xffff = no label: emission 0 at EP[V]
                RT[r] = uint4{(unsigned)m.x, ((l0 == 0xffffu ? (unsigned)V : l0) * 4u) | (((l1 == 0xffffu ? (unsigned)V : l1) * 4u) << 16), (unsigned)m.y, (unsigned)m.z};
            }
        }
    } else if (!RC) {
        int4 *RM = (int4 *)RMc;
        for (int r = tid; r < R; r += NTH) {
            int4 m = DIR == 0 ? p.frow_meta[r] : p.brow_meta[r];
            if (DIR == 1) {
                const int l0 = (short)(m.w & 0xffff), l1 = m.w >> 16;       // -1 = no label: emission 0 at EP[V]
                m.w = ((l0 < 0 ? V : l0) & 0xffff) | ((l1 < 0 ? V : l1) << 16);
            }
            RM[r] = m;
        }
    } else if (DIR == 1) {
        f32x2 *RW = (f32x2 *)RMc;                            // the weights of the two extra arcs of a row
        for (int r = tid; r < R; r += NTH) { const int4 m = p.brow_meta[r]; RW[r] = f32x2{__int_as_float(m.y), __int_as_float(m.z)}; }
        auto fix = [&](unsigned w) {                         // emission byte offsets: 0xffff = no label -> emission 0 at EP[V]
            const unsigned l0 = w & 0xffffu, l1 = w >> 16;
            return (l0 == 0xffffu ? (unsigned)V * 4u : l0) | ((l1 == 0xffffu ? (unsigned)V * 4u : l1) << 16);
        };
        rc01 = fix(rc01); rc11 = fix(rc11); rc21 = fix(rc21);
    }
    // One launch runs the iterations [i0, i1) of the recursion ("segment"): the host cuts a long recursion into
    // a few launches so that the grad pass can be released stage by stage with stream events (a kernel that
    // waits for another kernel's progress is not safe, see crf_loss_fwd_bwd).  Between launches the state --
    // the current vector and its exponent -- rests in HBM (p.state: [B][Gp + 64] floats per direction).
    const int i0 = p.i0, i1 = min(p.i1, lx);
    if (i0 > 0 && i0 >= lx) return;                          // this utterance was finished by an earlier segment
    float *state = p.state + (size_t)b * (Gp + 64);
    const int par0 = i0 & 1;
    int E = kScaleExp;
    float zpart = 0.f;
    for (int s = tid; s < 2 * Gp; s += NTH) X[s] = 0.f;
    if (tid < 2) EP[tid * Vp + V] = 0.f;
    // The frame maximum is FOUR LDS words per frame (three sets in rotation: read / accumulated by ds_max_f32 / cleared), one per
    // row of 16 lanes: a wave's frame top reads them with one ds_read_b128 and works the scale out on the scalar unit; with twelve wave maxima per frame every wave spent ~17 VALU instructions there and ~8 more in the tail --
    // the frame is bound by instruction issue (timing build: a wave with NO rows still took 520 cycles per frame).
    if (tid < 12) wm[tid] = 0.f;                             // [3 frames][4 rows of 16 lanes]: distinct addresses per lane (a same-address
                                                             // LDS atomic of several lanes is turned into a scalar loop by the compiler)
    const bool rowlead = (lane & 15) == 0;
    int sr = i0 % 3;                                         // word read by the next frame
    if (lx > 0)
        for (int v = tid; v < V; v += NTH) {
            if (DIR == 0) EP[par0 * Vp + v] = p.ep[(bt0 + i0) * V + v];                 // e'_t of the first frame
            else {
                const int t = lx - 1 - i0;                                             // first frame of this segment
                if (i0 == 0) EP[v] = p.ep[(bt0 + lx - 1) * V + v];                     // for the initial z vector
                if (t >= 1) EP[(1 - par0) * Vp + v] = p.ep[(bt0 + t - 1) * V + v];     // e'_{t-1}
            }
        }
    __syncthreads();
    {
        float m0 = 0.f;
        if (i0 > 0) {                                        // resume
            float *Xc = X + par0 * Gp;
            for (int s = tid; s < G; s += NTH) { const float v = state[s]; Xc[s] = v; m0 = fmaxf(m0, v); }
            E = __builtin_amdgcn_readfirstlane(__float_as_int(state[Gp]));   // (uniform: E lives on the scalar unit)
        } else if (DIR == 0) {
            for (int s = tid; s < G; s += NTH) { const float v = p.x_start[s] * pow2f(kScaleExp); X[s] = v; m0 = fmaxf(m0, v); }
        } else if (lx > 0) {
            for (int z = tid; z < G; z += NTH) {
                const int l = p.z_lab[z];
                const float v = EP[l < 0 ? V : l] * (p.z_end[z] * pow2f(kScaleExp));
                X[z] = v; m0 = fmaxf(m0, v);
            }
            float *BProw = p.Out + (bt0 + lx - 1) * p.Rout;
            if (lead) for (int r = tid; r < 2 * R; r += NTH) BProw[r] = p.brow_end[r] * pow2f(kScaleExp);
            if (tid == 0 && lead) p.Eout[bt0 + lx - 1] = E;
        } else {
            if (lead) for (int r = tid; r < 2 * R; r += NTH) zpart += p.brow_start[r] * p.brow_end[r] * pow2f(kScaleExp);
        }
        m0 = row_max16(m0);
        if (rowlead) lds_fmax(wm + sr * 4 + (lane >> 4), m0);
    }
    __syncthreads();
    if constexpr (K2) {
        // where is my peer?  (res_chain_body: both publish the id of their XCD, write-through; both on one XCD => plain stores
        // into the shared L2 are enough for the hand-off, otherwise write-through stores.  Both take the decision from the same
        // two ids.)
        const unsigned my_xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf;  // HW_REG_XCC_ID[3:0]
        if (tid == 0) __hip_atomic_store(hs + k, (1ull << 32) | my_xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int same = 1;
        if (tid == 0) {
            unsigned long long g = 0;
            for (unsigned spins = 0; (g >> 32) != 1ull; ++spins) {
                g = __hip_atomic_load(hs + (1 - k), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (spins > (1u << 22)) { __hip_atomic_store(p.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                if ((spins & 255u) == 255u && __hip_atomic_load(p.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                __builtin_amdgcn_s_sleep(2);
            }
            same = ((unsigned)g & 0xf) == my_xcc && (g >> 32) == 1ull;
        }
        same_l2 = __syncthreads_and(same) != 0;
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);   // arcs and tables have landed (see crf_res_chain_kernel)

    // Stage flags (p.nb > 1): when the recursion reaches iteration bound[k], the rows of all earlier iterations
    // are made visible device-wide (they are stored write-through; every wave drains its stores, then a barrier)
    // and a counter in fine-grained memory is bumped.  The host has queued the grad launch of stage k behind a
    // STREAM-level wait on that counter (hipStreamWaitValue32: the command processor polls, no wave spins, the
    // launch is not even dispatched before) -- so the grad pass follows the recursions without a relaunch.
    constexpr bool flagged = FLAG;
    int next_stage = 1;
    int next_bound = (FLAG && p.nb > 1) ? p.bound[1] : 0x7fffffff;   // iteration at which the next stage is published
    auto publish_stage = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(p.stage_cnt + next_stage, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        ++next_stage;
        next_bound = next_stage < p.nb ? p.bound[next_stage] : 0x7fffffff;
    };
    const bool pre_w = wave * kWave < V;                     // this wave holds emissions
    float last_sc = 1.f;                                     // scale of the last frame (rowless states, after the loop)
    // lagged scale (CRF_X_LAG): exponent of the NEXT frame's scale, worked out in the tail of the frame before; the first frame of a
    // launch takes the unlagged rule on the vector it starts from (any integer is exact).  lag_lo: the smallest scaled maximum seen (below kLagLow: fallback).
    constexpr bool LAG = CRF_X_LAG != 0 && !K2;
    // The maximum read in the tail of frame t travels to frame t+1 in a VECTOR register (lag_mx, the same bits in every lane) and is turned
    // into the scale there, BEHIND the frame's first gathers: lag_k is the exponent of the scale of the frame before.  First frame of a
    // launch: lag_mx = the maximum of the vector it starts from and lag_k = kLagTarget - kScaleExp give exactly the unlagged rule.
    [[maybe_unused]] int lag_k = kLagTarget - kScaleExp, lag_lo = 0x7fffffff, lag_mx = 0;
    if constexpr (LAG) {
        typedef int i32x4_t __attribute__((ext_vector_type(4)));
        const i32x4_t m4 = *(const i32x4_t *)(wm + sr * 4);
        lag_mx = max(max(m4.x, m4.y), max(m4.z, m4.w));
    }
    constexpr bool GFIRST = CRF_X_GFIRST != 0 && CRF_X_EARLY != 0 && !K2;
    constexpr int EPR = NTH >= 2 * kResThreads ? 1 : kEpRegsR;   // (V <= 2 * 512 everywhere: use_factored)
    float epn[EPR] = {};                                    // next emission row, in flight across the frame (waves that hold emissions only)
    // this utterance's emissions, rows and exponents (the frame loop adds 32-bit offsets: one s_mul instead of a 64-bit product per address)
    const float *ep_b = p.ep + bt0 * V;
    float *Out_b = p.Out + bt0 * p.Rout;
    int *Eo_b = p.Eout + bt0;
    auto frame = [&](const int par, int i) __attribute__((always_inline)) {
        [[maybe_unused]] const bool tm_on = b == 3 && i >= 100 && i < 228 && wave == 0;
        [[maybe_unused]] const int tm_i = (DIR * 4) * 1024 + (i - 100) * 8;
        CRF_TM(tm_on, tm_i + 0);
#ifdef CRF_TIMING
        if (b == 3 && i >= 150 && i < 158 && wave == 0) CRF_TM(true, 12288 + 1024 + (DIR * 4) * 8 + (i - 150));  // frame start
#endif
        const char *xb = (const char *)lds + par * XB;
        // GFIRST: the frame's first batch of gathers needs the source vector and nothing else -- it is requested before anything else is
        // worked out (every wave: idle waves hold padding arcs, offset 0 and weight 0).  A wave issues one instruction per ~4 cycles whatever
        // it is (tools/ubench_issue.py), and the ~35 scalar instructions a frame used to begin with kept the LDS idle for ~200 cycles right
        // behind every barrier (timing build, round 5: "frame top" 240 cycles with nothing to wait for).
        constexpr int NB0 = NB < NCHA ? NB : NCHA;
        [[maybe_unused]] f32x2 g01_0[NB0], g23_0[NB0];
        if constexpr (GFIRST) {
#if CRF_X_PRIO
            __builtin_amdgcn_s_setprio(3);
#endif
            CRF_RES_GATHER_N(g01_0, g23_0, A, xb, 0, NB0);
            __builtin_amdgcn_sched_barrier(0);
        }
        const int t = DIR == 0 ? i : lx - 1 - i;
        if (FLAG && i == next_bound) publish_stage();
        char *xnb = (char *)lds + (1 - par) * XB;
        const float *EPu = EP + (DIR == 0 ? par : 1 - par) * Vp;     // e'_t (fwd) / e'_{t-1} (bwd)
        const int tpre = DIR == 0 ? t + 1 : t - 2;
        // (only the waves that hold emissions take part in the prefetch: the compiler waits for vmcnt(0) around these
        // loads -- i.e. for the acknowledgement of the previous frame's row stores -- and the other waves need not)
        // The frame's side jobs belong to a few waves -- the emission prefetch to the waves that hold emissions, the exponent store
        // and the clearing of the maximum words to wave 0 -- and are tested as UNIFORM, UNLIKELY conditions: the other waves fall
        // through.  (As lane conditions they were exec-masked blocks that every other wave JUMPED over, ~32 cycles of instruction
        // refetch per taken branch and wave, six of them before the frame's first gather: tools/ubench_issue.py.)
        // (NOT the prefetch: as a cold block its register -- live across the whole frame -- is what the allocator spills first, and a
        // reload from scratch waits for vmcnt(0), i.e. for the frame's row stores: the S = 513 graph went from 1.84 to 2.35 ms)
        const bool pre = pre_w && (DIR == 0 ? (t + 1 < lx) : (t >= 2));
        if (pre) {
            const float *er = ep_b + (unsigned)tpre * (unsigned)V;   // (32-bit products: B * T * max(V, Rout) floats per utterance < 2^32, checked by the host)
#pragma unroll
            for (int q = 0; q < EPR; ++q) { const int v = tid + q * NTH; if (v < V) epn[q] = er[v]; }
        }
        const int sw = sr == 2 ? 0 : sr + 1, sz = sw == 2 ? 0 : sw + 1;   // accumulated during this frame / cleared in it
        typedef int i32x4_t __attribute__((ext_vector_type(4)));
        [[maybe_unused]] i32x4_t m4{};
        if constexpr (!LAG) m4 = *(const i32x4_t *)(wm + sr * 4);    // (non-negative floats: their bits order like integers)
        // The frame's scale and exponent are worked out BEHIND the first batch of gathers (EARLY: the batch loop calls `bookkeeping`
        // once its first gathers are requested -- they need nothing but the vector; the scale enters in the row epilogues only):
        // with the scale first, every wave of the workgroup sat out one LDS round trip right after the frame barrier, the LDS idle.
        float sc = 1.f;
        float *Orow = nullptr;
        auto bookkeeping = [&]() __attribute__((always_inline)) {
            // (the maxima's use ends up in FRONT of the first gathers all the same -- the scheduler gives their four registers to the gathers' addresses -- so
            // the frame still starts with one LDS round trip.  Reading them by inline asm BEHIND the gathers instead: metric step 2.833 vs 2.848 ms, but the
            // S = 513 graph 2.012 vs 1.978 -- dropped, profiles/round4_ab_waits_found_in_the_isa.txt)
            int ksc;
            if constexpr (LAG) {
                // lag_mx: maximum of the source vector of the frame BEFORE (read in that frame's tail), lag_k: that frame's scale
                const unsigned bits = (unsigned)__builtin_amdgcn_readfirstlane(lag_mx);
                int u = (int)(bits >> 23) - 127 + lag_k;             // exponent of the scaled maximum of that frame's source
                ksc = kLagTarget - u;
                // (all on the scalar unit; lag_lo: the smallest u of the recursion -- an all-zero vector lands far below kLagLow too, it is flagged at the end anyway)
                const int uu = i == i0 ? 0x7fffffff : u;           // (the first frame's u is the start-up convention above, not a measurement)
                asm("s_min_i32 %0, %0, %1" : "+s"(lag_lo) : "s"(uu) : "scc");
                asm("s_max_i32 %0, %0, %1\n\ts_min_i32 %0, %0, %2" : "+s"(ksc) : "s"(-100), "s"(100) : "scc");
                lag_k = ksc;
            } else ksc = rescale_exp_bits_uniform((unsigned)__builtin_amdgcn_readfirstlane(max(max(m4.x, m4.y), max(m4.z, m4.w))));   // (uniform)
            sc = pow2f(ksc);
            if (DIR == 1) last_sc = sc;
            if (DIR == 0) {
                E += ksc;
                if (__builtin_expect(wave == 0, 0)) {
                    if (lane < 4) wm[sz * 4 + lane] = 0.f;
                    if (tid == 0 && lead) Eo_b[t] = E;
                }
                E += kEpExp;
                Orow = Out_b + (unsigned)t * (unsigned)p.Rout;
            } else {
                E += ksc + kEpExp;
                if (__builtin_expect(wave == 0, 0)) {
                    if (lane < 4) wm[sz * 4 + lane] = 0.f;
                    if (t > 0 && tid == 0 && lead) Eo_b[t - 1] = E;
                }
                Orow = t > 0 ? Out_b + (unsigned)(t - 1) * (unsigned)p.Rout : p.Row0 + (int64_t)b * p.Rout;
            }
        };
        constexpr bool EARLY = CRF_X_EARLY != 0;   // (also with two CUs per recursion: H = 3 072 recursions 4.22 -> 4.04 ms)
        if constexpr (!EARLY) bookkeeping();
        unsigned ends_f = ends;
        int nch_f = nch;
        asm volatile("" : "+s"(ends_f), "+s"(nch_f));
        f32x2 acc = {0.f, 0.f};
        float mymax = 0.f;
        CRF_TM(tm_on, tm_i + 1);
        unsigned r4 = (unsigned)(row0 + lane) * 4u;   // 4 * row id
        constexpr bool EARLY_ = CRF_X_EARLY != 0;
        [[maybe_unused]] gu64 *slot = nullptr;        // K2: the granules of the vector this frame produces
        [[maybe_unused]] const unsigned tag = (unsigned)(i + 1);
        if constexpr (K2) slot = xchd + (size_t)(1 - par) * G;
        typedef std::conditional_t<DIR == 0, uint2, uint4> rct_t;
        [[maybe_unused]] rct_t kc{};                  // RL: constants of the slice that ends next
        // (KCLATE: the first slice's constants are requested BEHIND the first batch of gathers -- sixteen waves' 8- / 16-byte-per-lane reads
        // in front of them keep the LDS busy for 64 / 128 cycles right behind the barrier before the first gather is served)
        constexpr bool KCLATE = CRF_X_KCLATE != 0 && RL && EARLY_ && !K2;
        if constexpr (RL && !KCLATE) kc = *(const rct_t *)(RMc + (DIR == 0 ? 2u : 4u) * r4);
        // the end of a slice (row): everything between the row's sum and its entries of the next vector
        auto row_end = [&](const unsigned ks) __attribute__((always_inline)) {   // ks: slice number (uniform)
            // rows longer than a lane's registers lie on 2^lg adjacent lanes (res_layout.cpp place_rows): a butterfly
            // leaves the row's sum in every lane of the group, the first one owns the outputs
            float tot = acc.x + acc.y;
            if constexpr (ML) {
                const unsigned lg = ks < 10u ? (lgbits >> (3u * ks)) & 7u : 0u;   // (ten 3-bit fields; later slices have whole rows)
                if (lg) {
                    // DPP for groups of up to 16 lanes (pair swap, quad half swap, mirror of 8, mirror of 16: after
                    // each step every lane of the growing group holds the group's sum, so ANY lane of the other half
                    // will do); a __shfl_xor is a ds_bpermute round trip (~100+ cycles each, dependent) and cost the
                    // graphs with long rows -- every den_lm estimated from text -- a quarter of the frame
#define CRF_DPP_ADD(ctrl) tot += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, tot), ctrl, 0xf, 0xf, false))
                    CRF_DPP_ADD(0xB1);                            // quad_perm [1,0,3,2]
                    if (lg >= 2) CRF_DPP_ADD(0x4E);               // quad_perm [2,3,0,1]
                    if (lg >= 3) CRF_DPP_ADD(0x141);              // row_half_mirror
                    if (lg >= 4) CRF_DPP_ADD(0x140);              // row_mirror
#undef CRF_DPP_ADD
                    if (lg >= 5) tot += __shfl_xor(tot, 16, 64);
                    if (lg >= 6) tot += __shfl_xor(tot, 32, 64);
                }
            }
            if constexpr (IMP) {
                unsigned k0, k1;
                [[maybe_unused]] f32x2 wrl{};
                if constexpr (RC) {
                    // (masks, not ?: -- the compiler turns a three-way select of registers by a uniform index
                    // into an indexed array, which it then cannot keep in registers)
                    const unsigned s0 = 0u - (unsigned)(ks == 0), s1 = 0u - (unsigned)(ks == 1), s2 = 0u - (unsigned)(ks >= 2);
                    k0 = (rc00 & s0) | (rc10 & s1) | (rc20 & s2);
                    k1 = (rc01 & s0) | (rc11 & s1) | (rc21 & s2);
                } else {
                    k0 = kc.x; k1 = kc.y;
                    if constexpr (DIR == 1) wrl = f32x2{__uint_as_float(kc.z), __uint_as_float(kc.w)};
                    kc = *(const rct_t *)(RMc + (DIR == 0 ? 2u : 4u) * (r4 + kWave * 4u));   // the next slice's (64 rows of slack behind the table)
                }

                if (DIR == 0) {   // k0 = main label | tail label << 16, k1 = tail weight; U, L, A at rid, R + rid, 2R + rid
                    const float uold = *(const float *)(xb + r4);                   // U_t of the row's pair
                    const float em = *(const float *)((const char *)EPu + (k0 & 0xffffu)), et = *(const float *)((const char *)EPu + (k0 >> 16));   // (byte offsets)
                    const float rv = tot * sc;                                      // q_t[pair of the main state]
                    const float qt = __uint_as_float(k1) * uold * sc;               // q_t[pair of the tail state]
                    if (flagged) {
                        __hip_atomic_store((unsigned *)((char *)Orow + r4), __float_as_uint(rv), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store((unsigned *)((char *)Orow + r4 + 4u * (unsigned)R), __float_as_uint(qt), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } else {
                        *(float *)((char *)Orow + r4) = rv;
                        *(float *)((char *)Orow + r4 + 4u * (unsigned)R) = qt;
                    }
                    const float Lp = em * rv, Ap = et * qt, Up = Ap + Lp;           // a_{t+1}[main], [tail], their sum
                    *(float *)(xnb + r4) = Up;
                    *(float *)(xnb + r4 + dup) = Up;
                    *(float *)(xnb + r4 + 4u * (unsigned)R) = Lp;
                    *(float *)(xnb + r4 + 8u * (unsigned)R) = Ap;
                    if constexpr (K2) {           // entries rid, R + rid, 2 R + rid of the peer's vector
                        gu64 *gs = (gu64 *)((char *)slot + 2u * r4);
                        res_publish(gs, 0, tag, Up, same_l2); res_publish(gs, R, tag, Lp, same_l2); res_publish(gs, 2 * R, tag, Ap, same_l2);
                    }
                    mymax = __int_as_float(max(__float_as_int(mymax), __float_as_int(Up)));   // (non-negative: bits order like integers; fmaxf canonicalises first)
                } else {          // k0 = z offsets of the two extra arcs, k1 = label 0 | label 1 << 16
                    const float z0 = *(const float *)(xb + (k0 & 0xffffu)), z1 = *(const float *)(xb + (k0 >> 16));
                    const float e0 = *(const float *)((const char *)EPu + (k1 & 0xffffu)), e1 = *(const float *)((const char *)EPu + (k1 >> 16));
                    const f32x2 w01 = RL ? wrl : *(const f32x2 *)(RMc + 2u * r4);
                    const float craw = tot;                                         // common out-arcs of the row's states
                    f32x2 bv;                                                        // b_t of the two states
                    bv.x = fmaf(w01.x, z0, craw) * sc;
                    bv.y = fmaf(w01.y, z1, craw) * sc;
                    if (flagged)
                        __hip_atomic_store((unsigned long long *)((char *)Orow + 2u * r4),
                                           (unsigned long long)__float_as_uint(bv.x) | ((unsigned long long)__float_as_uint(bv.y) << 32),
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else
                        *(f32x2 *)((char *)Orow + 2u * r4) = bv;
                    f32x2 zv;                                                        // z_{t-1} of the pairs entering them
                    zv.x = e0 * bv.x;
                    zv.y = e1 * bv.y;
                    *(f32x2 *)(xnb + 2u * r4) = zv;
                    typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));
                    *(f32x2u *)(xnb + 2u * r4 + dup) = zv;
                    if constexpr (K2) {           // entries 2 rid, 2 rid + 1
                        gu64 *gs = (gu64 *)((char *)slot + 4u * r4);
                        res_publish(gs, 0, tag, zv.x, same_l2); res_publish(gs, 1, tag, zv.y, same_l2);
                    }
                    mymax = __int_as_float(max(__float_as_int(mymax), max(__float_as_int(zv.x), __float_as_int(zv.y))));
                }
            } else {
            const int4 m = *(const int4 *)(RMc + 4u * r4);
            if (DIR == 0) {
                const float uold = *(const float *)(xb + (m.x & 0xffff));   // U_t of the row's pair
                const float em = EPu[(unsigned)m.x >> 16], et = EPu[m.w];   // (requested together: one LDS round trip)
                const float rv = tot * sc;                                  // q_t[pair of the main state]
                const float qt = __int_as_float(m.z) * uold * sc;           // q_t[pair of the tail state]
                if (flagged) {   // write-through: the grad pass reads the rows from other XCDs while this kernel runs
                    __hip_atomic_store((unsigned *)((char *)Orow + r4), __float_as_uint(rv), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store((unsigned *)((char *)Orow + r4 + 4u * (unsigned)R), __float_as_uint(qt), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    *(float *)((char *)Orow + r4) = rv;
                    *(float *)((char *)Orow + r4 + 4u * (unsigned)R) = qt;
                }
                const float Lp = em * rv;                                   // a_{t+1}[main]
                const float Ap = et * qt;                                   // a_{t+1}[tail]
                const float Up = Ap + Lp;
                *(float *)(xnb + (m.x & 0xffff)) = Up;
                if (dup) *(float *)(xnb + (m.x & 0xffff) + dup) = Up;
                *(float *)(xnb + (m.y & 0xffff)) = Lp;
                *(float *)(xnb + ((unsigned)m.y >> 16)) = Ap;
                mymax = fmaxf(mymax, Up);
            } else {
                const float craw = tot;                                     // common out-arcs of the row's states
                const float z0 = *(const float *)(xb + (m.x & 0xffff)), z1 = *(const float *)(xb + ((unsigned)m.x >> 16));
                const float e0 = EPu[m.w & 0xffff], e1 = EPu[(unsigned)m.w >> 16];
                f32x2 bv;                                                    // b_t of the two states
                bv.x = fmaf(__int_as_float(m.y), z0, craw) * sc;
                bv.y = fmaf(__int_as_float(m.z), z1, craw) * sc;
                if (flagged)
                    __hip_atomic_store((unsigned long long *)((char *)Orow + 2u * r4),
                                       (unsigned long long)__float_as_uint(bv.x) | ((unsigned long long)__float_as_uint(bv.y) << 32),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else
                    *(f32x2 *)((char *)Orow + 2u * r4) = bv;
                f32x2 zv;                                                    // z_{t-1} of the pairs entering them
                zv.x = e0 * bv.x;
                zv.y = e1 * bv.y;
                *(f32x2 *)(xnb + 2u * r4) = zv;
                typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));
                if (dup) *(f32x2u *)(xnb + 2u * r4 + dup) = zv;   // the copy's distance is an odd number of floats: ds_write2_b32
                mymax = fmaxf(mymax, fmaxf(zv.x, zv.y));
            }
            }
            acc = f32x2{0.f, 0.f};
            r4 += kWave * 4u;
        };
#pragma unroll
        for (int c0 = 0; c0 < NCHA; c0 += NB) {
            const int nb = NCHA - c0 < NB ? NCHA - c0 : NB;   // (the last batch may be short: 21 chunks in batches of 4)
#if CRF_X_PRIO
            // least progress first: a wave's issue priority (s_setprio) falls as it advances through its chunks, so that the four
            // waves of a SIMD reach the frame barrier together.  Without it the SIMD issues its OLDEST wave first: in the timing
            // build the oldest waves were through their chunks at 2 560 cycles and the youngest -- alone on their SIMDs at the end,
            // one wave's latency hiding -- at 4 140 of a 4 700-cycle frame; with it 3 050 ... 3 790 of 4 350.  (Round 2 had tried
            // STATIC priorities for the younger waves: slower.)
            {
                constexpr int f1 = CRF_X_PRIO == 2 ? 4 : CRF_X_PRIO == 3 ? 1 : 2, f2 = CRF_X_PRIO == 2 ? 6 : CRF_X_PRIO == 3 ? 2 : 4, f3 = CRF_X_PRIO == 2 ? 7 : CRF_X_PRIO == 3 ? 4 : 6;   // eighths of the chunks
                constexpr int qlast = (NCHA - 1) / NB * NB;   // start of the last batch: a threshold rounded up beyond it would never be reached (768 x 20 in batches of 4: 7/8 -> 20)
                constexpr int q1 = (f1 * NCHA / 8 + NB - 1) / NB * NB, q2 = (f2 * NCHA / 8 + NB - 1) / NB * NB,
                              q3 = (f3 * NCHA / 8 + NB - 1) / NB * NB < qlast ? (f3 * NCHA / 8 + NB - 1) / NB * NB : qlast;
                // (not with two CUs per recursion: a wave that polls for the peer's entries at the lowest priority delays BOTH CUs -- H = 3 072 recursions 4.17 without, 4.24 ms with)
                if constexpr (!K2) {
                    if (c0 == 0) { if constexpr (!GFIRST) __builtin_amdgcn_s_setprio(3); }
                    else if (c0 == q1) __builtin_amdgcn_s_setprio(2);
                    else if (c0 == q2) __builtin_amdgcn_s_setprio(1);
                    else if (c0 == q3) __builtin_amdgcn_s_setprio(0);
                }
            }
#endif
            // (EARLY: the first batch is gathered by every wave -- the slots of a wave without arcs hold padding, offset 0 and weight 0)
            if ((EARLY && c0 == 0) || c0 < nch_f) {
                f32x2 g01[NB], g23[NB];
                if (GFIRST && c0 == 0) {
#pragma unroll
                    for (int ci = 0; ci < NB0; ++ci) { g01[ci] = g01_0[ci]; g23[ci] = g23_0[ci]; }
                } else {
                    CRF_RES_GATHER_N(g01, g23, A, xb, c0, nb);
                }
                if constexpr (KCLATE) {
                    if (c0 == 0) {
                        asm volatile("" ::: "memory");   // (keeps the read behind the gathers in the instruction stream)
                        kc = *(const rct_t *)(RMc + (DIR == 0 ? 2u : 4u) * r4);
                    }
                }
                if constexpr (EARLY) { if (c0 == 0) bookkeeping(); }
#pragma unroll
                for (int ci = 0; ci < nb; ++ci) {
                    CRF_RES_CHUNK_ACC(acc, g01, g23, A, c0 + ci, ci);
                    // (a row ends after 2 - 3 of a wave's 15 chunks: the test is laid out so that the common path falls through --
                    // a taken branch costs a wave ~32 cycles of instruction refetch, tools/ubench_issue.py.  With a probability, not
                    // "never": blocks the allocator believes cold are where it spills, and a reload from scratch waits for
                    // vmcnt(0), i.e. for the acknowledgement of the frame's write-through row stores)
                    if (__builtin_expect_with_probability((ends_f >> (c0 + ci) & 1u) != 0u, 0, 0.8)) row_end((unsigned)__builtin_popcount(ends_f & ((1u << (c0 + ci)) - 1u)));
                }
            }
        }
        CRF_TM(tm_on, tm_i + 2);
#ifdef CRF_TIMING
        if (b == 3 && i >= 150 && i < 158) {
            const int o = 12288 + ((DIR * 4) * 8 + wave) * 16;
            CRF_TM(true, o + (i - 150));
            if (lane == 0 && i == 150) { g_tm[o + 8] = (unsigned long long)nch; g_tm[o + 9] = (unsigned long long)__builtin_popcount(ends); }
        }
#endif
        if constexpr (K2) {   // the peer's entries (published from its row epilogues) into this CU's vector
            const int p0 = L.cu_row[1 - k], p1 = L.cu_row[2 - k];
            float *Xn = (float *)xnb;
            float fm;
            if (DIR == 0) {
                // the U entries, and of the L / A entries those this CU's rows gather (a list) -- all of them after the last frame
                // (logZ; only the CU that computes it)
                if (i == lx - 1) {
                    fm = res_fetch<NTH>(slot, Xn, p0, p1, tag, p.err, tid);
                    if (lead) {
                        fm = fmaxf(fm, res_fetch<NTH>(slot, Xn, R + p0, R + p1, tag, p.err, tid));
                        fm = fmaxf(fm, res_fetch<NTH>(slot, Xn, 2 * R + p0, 2 * R + p1, tag, p.err, tid));
                    }
                } else {
                    fm = res_fetch_both<NTH>(slot, Xn, p0, p1, p.xlist, p.xlist_off[k], p.xlist_off[k + 1], tag, p.err, tid);   // (one polling loop for both)
                }
            } else {
                fm = res_fetch<NTH>(slot, Xn, 2 * p0, 2 * p1, tag, p.err, tid);
            }
            mymax = fmaxf(mymax, fm);
        }
        // lagged scale: the maximum of THIS frame's source vector (deposited during the frame before, complete since the last barrier) gives
        // the scale of the NEXT frame -- requested here, where the gathers' registers are free, ahead of the wave maximum's DPP chain; the
        // scalar arithmetic runs while the deposit below is on its way
        [[maybe_unused]] i32x4_t m4n{};
        if constexpr (LAG) m4n = *(const i32x4_t *)(wm + sr * 4);
        mymax = row_max16(mymax);   // (sending the maximum from every row end instead -- no reduction in the tail -- was measured 3 % slower, round 4)
        if (rowlead) lds_fmax(wm + sw * 4 + (lane >> 4), mymax);
        sr = sw;
        if (pre) {
            float *EPw = EP + (DIR == 0 ? 1 - par : par) * Vp;
#pragma unroll
            for (int q = 0; q < EPR; ++q) { const int v = tid + q * NTH; if (v < V) EPw[v] = epn[q]; }
        }
        if constexpr (LAG) lag_mx = max(max(m4n.x, m4n.y), max(m4n.z, m4n.w));   // (behind the emission staging: ONE wait for the LDS in the tail, the one the barrier needs anyway)
        CRF_TM(tm_on, tm_i + 3);
#ifdef CRF_TIMING
        if (b == 3 && i >= 150 && i < 158) CRF_TM(true, 15360 + (DIR * 16 + wave) * 8 + (i - 150));   // this wave's arrival at the frame barrier
#endif
        sync_lds();
        CRF_TM(tm_on, tm_i + 4);
    };
    // Entries no row produces (states nobody enters: the start state) still hold a_0 in the buffer frame 0 read from, and
    // nothing rewrites them: they are cleared before that buffer becomes the source again (frame 2).  In an ordinary frame
    // the stale start mass is ~2^-60 of the vector; once the vector underflows it would be ALL of it -- a finite, wrong
    // logZ instead of the zero that sends the utterance to the robust kernels.  The clearing sits BETWEEN two runs of the
    // one frame loop (frame 0 alone, then the rest): inside the loop body it cost 2.6 % of the step (measured).
    int i = i0;
#pragma clang loop unroll(disable)
    for (int seg = 0; seg < 2; ++seg) {
        const int iend = (DIR == 0 && seg == 0) ? min(i1, 1) : i1;
#pragma clang loop unroll(disable)
        for (; i < iend; ++i) frame(i & 1, i);
        if (DIR == 0 && seg == 0 && i0 == 0 && i1 > 0) {
            for (int s = tid; s < G; s += NTH) if (p.x_start[s] != 0.f) X[s] = 0.f;
            sync_lds();
        }
    }
    if (i1 < lx) {                                           // not the last segment of this utterance: park the state
        const float *Xc = X + (i1 & 1) * Gp;
        for (int s = tid; s < G; s += NTH) state[s] = Xc[s];
        if (tid == 0) state[Gp] = __int_as_float(E);
        if (tid == 0 && LAG && lag_lo < kLagLow) p.redo[DIR * p.B + b] = 1;  // (the flag does not travel with the parked state)
        return;
    }
    if (flagged)
        while (next_stage < p.nb) publish_stage();           // the rest (at least the last stage: bound = T)
    if (DIR == 0) {
        if (!lead) return;                                   // (the vector is complete on both CUs)
        const float *Xf = X + (lx & 1) * Gp;
        float part = 0.f;
        for (int s = tid; s < G; s += NTH) part += Xf[s] * p.x_end[s];
        const float zs = res_block_sum<NW>(part, (float *)red, tid);
        const double mxs = res_mx_total<NW>(p, b, lx, red, tid);
        if (tid == 0) { p.den_zs[b] = zs; p.den_ez[b] = E; p.cost_alpha[b] = to_log(zs, E, mxs); if (!(zs > 0.f && zs < INFINITY) || (LAG && lag_lo < kLagLow)) p.redo[b] = 1; }
    } else {
        if (lx > 0) {
            __syncthreads();  // drains vmcnt: this workgroup's own stores to the spare row are visible to it
            const float *r0 = p.Row0 + (int64_t)b * p.Rout;
            const int ra = K2 ? 2 * L.cu_row[k] : 0, rb = K2 ? 2 * L.cu_row[k + 1] : 2 * R;   // the rows this CU wrote
            for (int r = ra + tid; r < rb; r += NTH) zpart += p.brow_start[r] * r0[r];
            // states without a row: b_0 = (sum over their arcs of w * z_0) * (scale of the last frame); z_0 is the vector the
            // last frame read
            const float *Xl = X + ((lx - 1) & 1) * Gp;
            if (lead) for (int a = tid; a < p.nbx; a += NTH) zpart += p.bx_w[a] * Xl[p.bx_idx[a]] * last_sc;
        } else if (tid == 0 && lead) zpart += p.bx_se * pow2f(kScaleExp);
        float zb = res_block_sum<NW>(zpart, (float *)red, tid);
        if constexpr (K2) {   // the peer's part of the sum: one more granule through the handshake words (tag 2)
            if (!lead) {
                if (tid == 0) __hip_atomic_store(hs + k, (2ull << 32) | __float_as_uint(zb), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
            if (tid == 0) {
                unsigned long long g = 0;
                for (unsigned spins = 0; (g >> 32) != 2ull; ++spins) {
                    g = __hip_atomic_load(hs + (1 - k), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (spins > (1u << 22)) { __hip_atomic_store(p.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                    if ((spins & 255u) == 255u && __hip_atomic_load(p.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                    __builtin_amdgcn_s_sleep(2);
                }
                zb += (g >> 32) == 2ull ? __uint_as_float((unsigned)g) : 0.f;
            }
        }
        const double mxs = res_mx_total<NW>(p, b, lx, red, tid);
        if (tid == 0) { p.cb_part[(size_t)b * kResMaxK] = zb; p.cb_F[b] = E; p.cb_mxs[b] = mxs; if (!(zb > 0.f && zb < INFINITY) || (LAG && lag_lo < kLagLow)) p.redo[p.B + b] = 1; }
    }
}

// =============================================================================================
// TWO UTTERANCES per workgroup (throughput mode: batches above CUs / 4 utterances per GPU).  Same layout tables, same arithmetic,
// same rows in memory as fac_chain_body -- bit for bit -- but the state vector is [entry][2 utterances] (float2): ONE address
// computation and ONE ds_read_b64 (the LDS cycles of a ds_read_b32) gather an entry for both, the weight is shared in its register,
// the product is one v_pk_fma_f32 whose two lanes are the two utterances.  Per arc and utterance that is half an address
// instruction, half an LDS instruction and half a packed FMA, against 1 + 1 + 1/2 with one utterance per workgroup; the price is
// two row epilogues per slice and twice the frame's bookkeeping (two scales, two exponents, two row pointers).  A pair runs
// max(lx0, lx1) frames: the shorter utterance's sums are taken when it ends (its half of the vector then holds garbage nobody
// reads, its row stores go to a dump row).  768-thread geometries only (row constants in registers or in the LDS table).
// LDS: X2[2][Gp] float2 | row table (RL: as fac_chain_body; RC backward: the two extra-arc weights per row) | EP2[2][Vp] float2 |
//      wm[3][2][4] | red
// =============================================================================================
typedef __attribute__((address_space(3))) char lds_char;
__device__ __forceinline__ f32x2 lds_ld2(unsigned a) { return *(const __attribute__((address_space(3))) f32x2 *)(uintptr_t)a; }
__device__ __forceinline__ float lds_ld1(unsigned a) { return *(const __attribute__((address_space(3))) float *)(uintptr_t)a; }
__device__ __forceinline__ void lds_st2(unsigned a, f32x2 v) { *(__attribute__((address_space(3))) f32x2 *)(uintptr_t)a = v; }
// address of entry (16-bit byte offset of the one-utterance layout, low / high half of `w`) in a float2 vector at LDS address `base`
__device__ __forceinline__ unsigned addr2_lo(unsigned w, unsigned base) { unsigned r; asm("v_mad_u32_u16 %0, %1, 2, %2" : "=v"(r) : "v"(w), "s"(base)); return r; }
__device__ __forceinline__ unsigned addr2_hi(unsigned w, unsigned base) { unsigned r; asm("v_mad_u32_u16 %0, %1, 2, %2 op_sel:[1,0,0,0]" : "=v"(r) : "v"(w), "s"(base)); return r; }

template <int DIR, bool FLAG, int NTH, int NCH, int NB, bool ML, bool RL>
__device__ __forceinline__ void fac_chain_body2(const FacParams &p, float *lds, const int pair) {
    // NTH: 768 (the one-utterance kernel's 768-thread layouts as they are; 168 registers per wave -- this kernel spills there) or
    // 512 x 30 chunks (round 5: a layout of its own, HostGraph::facp; two waves per SIMD with 256 registers each)
    constexpr int NW = NTH / kWave;
    constexpr bool RC = !RL;
    static_assert(NTH == kFac3Threads || (NTH == kResThreads && RL), "two utterances per workgroup: 768 threads, or 512 threads with the row table");
    constexpr int NCHA = RC ? kFac3ArcCh : NCH;
    constexpr int RCW = NCHA * 6;
    static_assert(!RC || NCH == kFac3NCH, "row constants in registers: 20 chunks of arcs + the constants' slot");
    const FacDirDev &L = p.L;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int V = p.V, G = L.G, R = L.R;
    const int bu[2] = {2 * pair, 2 * pair + 1};
    const bool real1 = bu[1] < p.B;                          // (an odd batch: the last pair's second utterance does not exist)
    const int lxu[2] = {p.lx[bu[0]], real1 ? p.lx[bu[1]] : 0};
    const unsigned dup2 = 2u * (unsigned)L.dup;
    const int Vp = rup64(V + 1), Gp = rup64(G);
    const unsigned XB2 = (unsigned)Gp * 8u;                  // bytes per state vector
    constexpr int kRowB = RL ? (DIR == 0 ? 8 : 16) : 8;
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_char *)lds;   // LDS address of the carve
    f32x2 *X2 = (f32x2 *)lds;                                // [2][Gp]
    char *RMc = (char *)(X2 + 2 * Gp);
    const size_t tbytes = RL ? (size_t)(R + 64) * kRowB : (DIR == 1 ? (size_t)R * 8 : (size_t)0);
    f32x2 *EP2 = (f32x2 *)(RMc + tbytes);                    // [2][Vp]
    float *wm = (float *)(EP2 + 2 * Vp);                     // [3][2][4]
    double *red = (double *)(wm + 24);
    const unsigned rm0 = lds0 + 2u * XB2, ep0 = rm0 + (unsigned)tbytes, EB2 = (unsigned)Vp * 8u;
    if (tid == 0 && p.started) atomicAdd(p.started, 1);

    unsigned A[(NCH * 6)];
    unsigned rc00 = 0, rc01 = 0, rc10 = 0, rc11 = 0, rc20 = 0, rc21 = 0;
    {
        const unsigned *src = L.arcs + tid;
#pragma unroll
        for (int i = 0; i < (RC ? RCW : NCH * 6); ++i) A[i] = src[(size_t)i * NTH];
        if (RC) {
            rc00 = src[(size_t)(RCW + 0) * NTH]; rc01 = src[(size_t)(RCW + 1) * NTH];
            rc10 = src[(size_t)(RCW + 2) * NTH]; rc11 = src[(size_t)(RCW + 3) * NTH];
            rc20 = src[(size_t)(RCW + 4) * NTH]; rc21 = src[(size_t)(RCW + 5) * NTH];
        }
    }
    const uint4 wi = L.wave_info[wave];
    const unsigned ends = __builtin_amdgcn_readfirstlane(wi.x);
    const int nch = __builtin_amdgcn_readfirstlane(wi.y);
    const int row0 = __builtin_amdgcn_readfirstlane(wi.z);
    const unsigned lgbits = __builtin_amdgcn_readfirstlane(wi.w);
    if (RL) {
        if (DIR == 0) {
            uint2 *RT = (uint2 *)RMc;
            for (int r = tid; r < R; r += NTH) {
                const int4 m = p.frow_meta[r];
                RT[r] = uint2{(((unsigned)m.x >> 16) * 4u) | (((unsigned)m.w * 4u) << 16), (unsigned)m.z};
            }
        } else {
            uint4 *RT = (uint4 *)RMc;
            for (int r = tid; r < R; r += NTH) {
                const int4 m = p.brow_meta[r];
                const unsigned l0 = (unsigned)m.w & 0xffffu, l1 = (unsigned)m.w >> 16;
                RT[r] = uint4{(unsigned)m.x, ((l0 == 0xffffu ? (unsigned)V : l0) * 4u) | (((l1 == 0xffffu ? (unsigned)V : l1) * 4u) << 16), (unsigned)m.y, (unsigned)m.z};
            }
        }
    } else if (DIR == 1) {
        f32x2 *RW = (f32x2 *)RMc;
        for (int r = tid; r < R; r += NTH) { const int4 m = p.brow_meta[r]; RW[r] = f32x2{__int_as_float(m.y), __int_as_float(m.z)}; }
        auto fix = [&](unsigned w) {
            const unsigned l0 = w & 0xffffu, l1 = w >> 16;
            return (l0 == 0xffffu ? (unsigned)V * 4u : l0) | ((l1 == 0xffffu ? (unsigned)V * 4u : l1) << 16);
        };
        rc01 = fix(rc01); rc11 = fix(rc11); rc21 = fix(rc21);
    }
    const int64_t bt0u[2] = {(int64_t)bu[0] * p.T, (int64_t)bu[1] * p.T};
    const float *ep_b[2] = {p.ep + bt0u[0] * V, p.ep + (real1 ? bt0u[1] : bt0u[0]) * V};
    float *Out_b[2] = {p.Out + bt0u[0] * p.Rout, p.Out + (real1 ? bt0u[1] : bt0u[0]) * p.Rout};
    int *Eo_b[2] = {p.Eout + bt0u[0], p.Eout + (real1 ? bt0u[1] : bt0u[0])};
    float *dump = p.dump + ((size_t)DIR * p.npair + pair) * (size_t)p.dump_stride;   // rows of an utterance that has ended
    int E[2] = {kScaleExp, kScaleExp};
    float zpart[2] = {0.f, 0.f};
    for (int s = tid; s < 2 * Gp; s += NTH) X2[s] = f32x2{0.f, 0.f};
    if (tid < 2) EP2[tid * Vp + V] = f32x2{0.f, 0.f};
    if (tid < 24) wm[tid] = 0.f;                             // [3 frames][2 utterances][4 rows of 16 lanes]
    const bool rowlead = (lane & 15) == 0;
    int sr = 0;
    for (int v = tid; v < V; v += NTH) {
        f32x2 e0{0.f, 0.f}, e1{0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int lx = lxu[u];
            if (lx <= 0) continue;
            if (DIR == 0) e0[u] = ep_b[u][v];                                             // e'_0
            else { e0[u] = ep_b[u][(unsigned)(lx - 1) * (unsigned)V + v]; if (lx >= 2) e1[u] = ep_b[u][(unsigned)(lx - 2) * (unsigned)V + v]; }
        }
        EP2[v] = e0;
        if (DIR == 1) EP2[Vp + v] = e1;
    }
    __syncthreads();
    {
        f32x2 m0{0.f, 0.f};
        if (DIR == 0) {
            for (int s = tid; s < G; s += NTH) { const float v = p.x_start[s] * pow2f(kScaleExp); X2[s] = f32x2{v, v}; m0.x = fmaxf(m0.x, v); }
            m0.y = m0.x;
        } else {
            for (int z = tid; z < G; z += NTH) {
                const int l = p.z_lab[z];
                const f32x2 e = EP2[l < 0 ? V : l];
                const float ze = p.z_end[z] * pow2f(kScaleExp);
                const f32x2 v{lxu[0] > 0 ? e.x * ze : 0.f, lxu[1] > 0 ? e.y * ze : 0.f};
                X2[z] = v; m0.x = fmaxf(m0.x, v.x); m0.y = fmaxf(m0.y, v.y);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (u == 1 && !real1) continue;
                if (lxu[u] > 0) {
                    float *BProw = Out_b[u] + (unsigned)(lxu[u] - 1) * (unsigned)p.Rout;
                    for (int r = tid; r < 2 * R; r += NTH) BProw[r] = p.brow_end[r] * pow2f(kScaleExp);
                    if (tid == 0) Eo_b[u][lxu[u] - 1] = E[u];
                } else {
                    for (int r = tid; r < 2 * R; r += NTH) zpart[u] += p.brow_start[r] * p.brow_end[r] * pow2f(kScaleExp);
                }
            }
        }
        m0.x = row_max16(m0.x); m0.y = row_max16(m0.y);
        if (rowlead) { lds_fmax(wm + (lane >> 4), m0.x); lds_fmax(wm + 4 + (lane >> 4), m0.y); }
    }
    __syncthreads();
    __builtin_amdgcn_s_waitcnt(0x0F70);

    int next_stage = 1;
    int next_bound = (FLAG && p.nb > 1) ? p.bound[1] : 0x7fffffff;
    const int nreal = real1 ? 2 : 1;
    auto publish_stage = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(p.stage_cnt + next_stage, nreal, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        ++next_stage;
        next_bound = next_stage < p.nb ? p.bound[next_stage] : 0x7fffffff;
    };
    const bool pre_w = wave * kWave < V;
    f32x2 last_sc{1.f, 1.f};
    f32x2 epn[kEpRegsR] = {};
    // one frame; `act`: bit u = utterance u has not ended (its rows, exponents and emissions are real)
    auto frame = [&](const int par, const int i, const int act) __attribute__((always_inline)) {
        if (FLAG && i == next_bound) publish_stage();
        const unsigned xb = lds0 + (unsigned)par * XB2, xnb = lds0 + (unsigned)(1 - par) * XB2;
        const unsigned epu = ep0 + (unsigned)(DIR == 0 ? par : 1 - par) * EB2;
        int tu[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) tu[u] = DIR == 0 ? i : lxu[u] - 1 - i;
        bool pre[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            pre[u] = pre_w && (act >> u & 1) && (DIR == 0 ? (tu[u] + 1 < lxu[u]) : (tu[u] >= 2));
            if (pre[u]) {
                const float *er = ep_b[u] + (unsigned)(DIR == 0 ? tu[u] + 1 : tu[u] - 2) * (unsigned)V;
#pragma unroll
                for (int q = 0; q < kEpRegsR; ++q) { const int v = tid + q * NTH; if (v < V) epn[q][u] = er[v]; }
            }
        }
        const int sw = sr == 2 ? 0 : sr + 1, sz = sw == 2 ? 0 : sw + 1;
        const int4 ma = *(const int4 *)(wm + sr * 8), mb = *(const int4 *)(wm + sr * 8 + 4);
        const int ksc0 = rescale_exp_bits((unsigned)__builtin_amdgcn_readfirstlane(max(max(ma.x, ma.y), max(ma.z, ma.w))));
        const int ksc1 = rescale_exp_bits((unsigned)__builtin_amdgcn_readfirstlane(max(max(mb.x, mb.y), max(mb.z, mb.w))));
        if (wave == 0 && lane < 8) wm[sz * 8 + lane] = 0.f;
        const f32x2 sc{pow2f(ksc0), pow2f(ksc1)};
        if (DIR == 1) last_sc = sc;
        float *Orow[2];
        const int ksc[2] = {ksc0, ksc1};
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const bool on = act >> u & 1;
            if (DIR == 0) {
                E[u] += ksc[u];
                if (tid == 0 && on) Eo_b[u][tu[u]] = E[u];
                E[u] += kEpExp;
                Orow[u] = on ? Out_b[u] + (unsigned)tu[u] * (unsigned)p.Rout : dump;
            } else {
                E[u] += ksc[u] + kEpExp;
                if (tid == 0 && on && tu[u] > 0) Eo_b[u][tu[u] - 1] = E[u];
                Orow[u] = !on ? dump : tu[u] > 0 ? Out_b[u] + (unsigned)(tu[u] - 1) * (unsigned)p.Rout : p.Row0 + (int64_t)bu[u] * p.Rout;
            }
        }
        unsigned ends_f = ends;
        int nch_f = nch;
        asm volatile("" : "+s"(ends_f), "+s"(nch_f));
        f32x2 acc{0.f, 0.f}, accb{0.f, 0.f};
        f32x2 mymax{0.f, 0.f};
        unsigned r4 = (unsigned)(row0 + lane) * 4u;   // 4 * row id
        typedef std::conditional_t<DIR == 0, uint2, uint4> rct_t;
        [[maybe_unused]] rct_t kc{};
        if constexpr (RL) kc = *(const rct_t *)(RMc + (DIR == 0 ? 2u : 4u) * r4);
        auto row_end = [&](const unsigned ks) __attribute__((always_inline)) {
            f32x2 tot = acc + accb;
            if constexpr (ML) {
                const unsigned lg = ks < 10u ? (lgbits >> (3u * ks)) & 7u : 0u;
                if (lg) {
                    // (two scalars, not the halves of the float2: with the DPP source a sub-register of a 64-bit tuple the compiler's
                    // DPP combiner took the OTHER half as source -- v_add_f32_dpp v7, v6, v7 -- found by the parity tests)
                    float tx = tot.x, ty = tot.y;
                    asm volatile("" : "+v"(tx), "+v"(ty));
#define CRF_DPP_ADD2(ctrl) { tx += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, tx), ctrl, 0xf, 0xf, false)); \
                             ty += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, ty), ctrl, 0xf, 0xf, false)); }
                    CRF_DPP_ADD2(0xB1);
                    if (lg >= 2) CRF_DPP_ADD2(0x4E);
                    if (lg >= 3) CRF_DPP_ADD2(0x141);
                    if (lg >= 4) CRF_DPP_ADD2(0x140);
#undef CRF_DPP_ADD2
                    if (lg >= 5) { tx += __shfl_xor(tx, 16, 64); ty += __shfl_xor(ty, 16, 64); }
                    if (lg >= 6) { tx += __shfl_xor(tx, 32, 64); ty += __shfl_xor(ty, 32, 64); }
                    tot = f32x2{tx, ty};
                }
            }
            unsigned k0, k1;
            [[maybe_unused]] f32x2 wrl{};
            if constexpr (RC) {
                const unsigned s0 = 0u - (unsigned)(ks == 0), s1 = 0u - (unsigned)(ks == 1), s2 = 0u - (unsigned)(ks >= 2);
                k0 = (rc00 & s0) | (rc10 & s1) | (rc20 & s2);
                k1 = (rc01 & s0) | (rc11 & s1) | (rc21 & s2);
            } else {
                k0 = kc.x; k1 = kc.y;
                if constexpr (DIR == 1) wrl = f32x2{__uint_as_float(kc.z), __uint_as_float(kc.w)};
                kc = *(const rct_t *)(RMc + (DIR == 0 ? 2u : 4u) * (r4 + kWave * 4u));
            }
            if (DIR == 0) {   // k0 = main label | tail label << 16 (byte offsets of a float row), k1 = tail weight; U, L, A at rid, R + rid, 2R + rid
                const f32x2 uold = lds_ld2(xb + 2u * r4);
                const f32x2 em = lds_ld2(epu + 2u * (k0 & 0xffffu)), et = lds_ld2(epu + 2u * (k0 >> 16));
                const f32x2 rv = tot * sc;
                const f32x2 qt = __uint_as_float(k1) * uold * sc;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    if (FLAG) {
                        __hip_atomic_store((unsigned *)((char *)Orow[u] + r4), __float_as_uint(rv[u]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store((unsigned *)((char *)Orow[u] + r4 + 4u * (unsigned)R), __float_as_uint(qt[u]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } else {
                        *(float *)((char *)Orow[u] + r4) = rv[u];
                        *(float *)((char *)Orow[u] + r4 + 4u * (unsigned)R) = qt[u];
                    }
                }
                const f32x2 Lp = em * rv, Ap = et * qt, Up = Ap + Lp;
                lds_st2(xnb + 2u * r4, Up);
                lds_st2(xnb + 2u * r4 + dup2, Up);
                lds_st2(xnb + 2u * r4 + 8u * (unsigned)R, Lp);
                lds_st2(xnb + 2u * r4 + 16u * (unsigned)R, Ap);
                mymax.x = __int_as_float(max(__float_as_int(mymax.x), __float_as_int(Up.x)));
                mymax.y = __int_as_float(max(__float_as_int(mymax.y), __float_as_int(Up.y)));
            } else {          // k0 = z offsets of the two extra arcs, k1 = label 0 | label 1 << 16 (byte offsets of a float row)
                const f32x2 z0 = lds_ld2(xb + 2u * (k0 & 0xffffu)), z1 = lds_ld2(xb + 2u * (k0 >> 16));
                const f32x2 e0 = lds_ld2(epu + 2u * (k1 & 0xffffu)), e1 = lds_ld2(epu + 2u * (k1 >> 16));
                const f32x2 w01 = RL ? wrl : *(const f32x2 *)(RMc + 2u * r4);
                f32x2 bx, by;                                     // b_t of the row's two states, per utterance
                bx = __builtin_elementwise_fma(f32x2{w01.x, w01.x}, z0, tot) * sc;
                by = __builtin_elementwise_fma(f32x2{w01.y, w01.y}, z1, tot) * sc;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    if (FLAG)
                        __hip_atomic_store((unsigned long long *)((char *)Orow[u] + 2u * r4),
                                           (unsigned long long)__float_as_uint(bx[u]) | ((unsigned long long)__float_as_uint(by[u]) << 32),
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else
                        *(f32x2 *)((char *)Orow[u] + 2u * r4) = f32x2{bx[u], by[u]};
                }
                const f32x2 zx = e0 * bx, zy = e1 * by;           // z_{t-1} of the pairs entering them
                lds_st2(xnb + 4u * r4, zx);
                lds_st2(xnb + 4u * r4 + 8u, zy);
                lds_st2(xnb + 4u * r4 + dup2, zx);
                lds_st2(xnb + 4u * r4 + dup2 + 8u, zy);
                mymax.x = __int_as_float(max(__float_as_int(mymax.x), max(__float_as_int(zx.x), __float_as_int(zy.x))));
                mymax.y = __int_as_float(max(__float_as_int(mymax.y), max(__float_as_int(zx.y), __float_as_int(zy.y))));
            }
            acc = f32x2{0.f, 0.f}; accb = f32x2{0.f, 0.f};
            r4 += kWave * 4u;
        };
#pragma unroll
        for (int c0 = 0; c0 < NCHA; c0 += NB) {
            const int nb = NCHA - c0 < NB ? NCHA - c0 : NB;
            if (c0 < nch_f) {
                f32x2 g[NB][4];
#pragma unroll
                for (int ci = 0; ci < nb; ++ci) {
                    const unsigned i01 = A[6 * (c0 + ci)], i23 = A[6 * (c0 + ci) + 1];
                    g[ci][0] = lds_ld2(addr2_lo(i01, xb)); g[ci][1] = lds_ld2(addr2_hi(i01, xb));
                    g[ci][2] = lds_ld2(addr2_lo(i23, xb)); g[ci][3] = lds_ld2(addr2_hi(i23, xb));
                }
#pragma unroll
                for (int ci = 0; ci < nb; ++ci) {
                    const int c = c0 + ci;
                    // the weights stay the register PAIRS the one-utterance kernel multiplies with; a packed FMA takes one half of a
                    // pair for both of its lanes (op_sel) -- a splat built in C makes the compiler keep a second register per weight
                    // alive across the whole kernel (400 spilled registers).  The one-utterance kernel sums (a0 w0 + a2 w2) and
                    // (a1 w1 + a3 w3) in the two lanes of its packed FMA and adds the two at the row's end: the same two chains here,
                    // per utterance -- the results are bit-identical
                    f32x2 w01, w23;
                    w01.x = __uint_as_float(A[6 * c + 2]); w01.y = __uint_as_float(A[6 * c + 3]);
                    w23.x = __uint_as_float(A[6 * c + 4]); w23.y = __uint_as_float(A[6 * c + 5]);
                    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(g[ci][0]), "v"(w01));
                    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(accb) : "v"(g[ci][1]), "v"(w01));
                    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(g[ci][2]), "v"(w23));
                    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(accb) : "v"(g[ci][3]), "v"(w23));
                    if (ends_f >> c & 1u) row_end((unsigned)__builtin_popcount(ends_f & ((1u << c) - 1u)));
                }
            }
        }
        mymax.x = row_max16(mymax.x); mymax.y = row_max16(mymax.y);
        if (rowlead) { lds_fmax(wm + sw * 8 + (lane >> 4), mymax.x); lds_fmax(wm + sw * 8 + 4 + (lane >> 4), mymax.y); }
        sr = sw;
        if (pre[0] | pre[1]) {
            f32x2 *EPw = EP2 + (DIR == 0 ? 1 - par : par) * Vp;
#pragma unroll
            for (int q = 0; q < kEpRegsR; ++q) {
                const int v = tid + q * NTH;
                if (v < V) {   // (an utterance that is not prefetched keeps its old emissions: never read again, or read by garbage only)
                    if (pre[0] && pre[1]) EPw[v] = epn[q];
                    else if (pre[0]) ((float *)(EPw + v))[0] = epn[q].x;
                    else ((float *)(EPw + v))[1] = epn[q].y;
                }
            }
        }
        sync_lds();
    };
    // sums of an utterance that has just ended (forward: its logZ; backward: its part of the backward logZ)
    auto finish = [&](const int u, const int nfr) __attribute__((always_inline)) {
        if (u == 1 && !real1) return;
        const int b = bu[u], lx = lxu[u];
        if (DIR == 0) {
            const f32x2 *Xf = X2 + (nfr & 1) * Gp;
            float part = 0.f;
            for (int s = tid; s < G; s += NTH) part += Xf[s][u] * p.x_end[s];
            const float zs = res_block_sum<NW>(part, (float *)red, tid);
            const double mxs = res_mx_total<NW>(p, b, lx, red, tid);
            if (tid == 0) { p.den_zs[b] = zs; p.den_ez[b] = E[u]; p.cost_alpha[b] = to_log(zs, E[u], mxs); if (!(zs > 0.f && zs < INFINITY)) p.redo[b] = 1; }
        } else {
            float zp = zpart[u];
            if (lx > 0) {
                __syncthreads();  // drains vmcnt: this workgroup's own stores to the spare row are visible to it
                const float *r0 = p.Row0 + (int64_t)b * p.Rout;
                for (int r = tid; r < 2 * R; r += NTH) zp += p.brow_start[r] * r0[r];
                const f32x2 *Xl = X2 + ((lx - 1) & 1) * Gp;     // the vector the utterance's last frame read
                for (int a = tid; a < p.nbx; a += NTH) zp += p.bx_w[a] * Xl[p.bx_idx[a]][u] * last_sc[u];
            } else if (tid == 0) zp += p.bx_se * pow2f(kScaleExp);
            const float zb = res_block_sum<NW>(zp, (float *)red, tid);
            const double mxs = res_mx_total<NW>(p, b, lx, red, tid);
            if (tid == 0) { p.cb_part[(size_t)b * kResMaxK] = zb; p.cb_F[b] = E[u]; p.cb_mxs[b] = mxs; if (!(zb > 0.f && zb < INFINITY)) p.redo[p.B + b] = 1; }
        }
    };
    const int us = lxu[1] < lxu[0] ? 1 : 0, ul = 1 - us;      // the utterance that ends first / last
    const int lmin = lxu[us], lmax = lxu[ul];
    if (lmin == 0) finish(us, 0);                             // an empty utterance: its sums come from the untouched start vector
    // ONE call site of the frame (the loop body is ~10 KB of code; the 64 KiB instruction cache is shared by two CUs): three runs of
    // the same loop -- frame 0 of the forward recursion alone (the entries no row produces, the start state, are cleared before their
    // buffer becomes the source again: see fac_chain_body), the frames with both utterances, the frames of the longer one alone
    int i = 0;
#pragma clang loop unroll(disable)
    for (int seg = 0; seg < 3; ++seg) {
        const int iend = seg == 0 ? (DIR == 0 ? min(lmax, 1) : 0) : seg == 1 ? lmin : lmax;
#pragma clang loop unroll(disable)
        for (; i < iend; ++i) frame(i & 1, i, i < lmin ? 3 : (1 << ul));
        if (seg == 0 && DIR == 0 && lmax > 0) {
            for (int s = tid; s < G; s += NTH) if (p.x_start[s] != 0.f) X2[s] = f32x2{0.f, 0.f};
            sync_lds();
        }
        if (seg == 1 && lmin > 0 && lmin < lmax) finish(us, lmin);
    }
    if (FLAG)
        while (next_stage < p.nb) publish_stage();
    if (lmin > 0 && lmin == lmax) finish(us, lmin);
    finish(ul, lmax);
}

// Both recursions of every utterance as ONE grid of 2B workgroups (block x < B: forward recursion of utterance x,
// else the backward recursion of utterance x - B).  One launch on one stream: the two directions used to be two
// kernels on two streams, which ran side by side only while those streams sat on different hardware queues -- not
// guaranteed inside a training process (HIP maps all streams of a process onto GPU_MAX_HW_QUEUES = 4 queues; with
// RCCL's and torch's streams around, the recursions were observed to run one after the other: 5.3 instead of 3.25 ms).
// NBF / NBB: chunks gathered per batch, forward / backward.
template <bool FLAG, int NTH, int NCH, int NBF, int NBB, bool ML, bool RL = false>
__global__ __launch_bounds__(NTH) void crf_fac_pair_kernel(FacParams pf, FacParams pb) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int B = pf.B;
    if ((int)blockIdx.x < B) fac_chain_body<0, FLAG, NTH, NCH, NBF, ML, RL>(pf, lds, (int)blockIdx.x);
    else fac_chain_body<1, FLAG, NTH, NCH, NBB, ML, RL>(pb, lds, (int)blockIdx.x - B);
}
// ... with TWO UTTERANCES per workgroup: 2 * ceil(B / 2) workgroups, forward recursions first
template <bool FLAG, int NTH, int NCH, int NBF, int NBB, bool ML, bool RL>
__global__ __launch_bounds__(NTH) void crf_fac_pair2_kernel(FacParams pf, FacParams pb) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int np = pf.npair;
    if ((int)blockIdx.x < np) fac_chain_body2<0, FLAG, NTH, NCH, NBF, ML, RL>(pf, lds, (int)blockIdx.x);
    else fac_chain_body2<1, FLAG, NTH, NCH, NBB, ML, RL>(pb, lds, (int)blockIdx.x - np);
}
// ... with TWO CUs per recursion: 2 * nbu * 2 workgroups for the utterances [b0, b0 + nbu), forward recursions first; the
// two CUs of a recursion 8 block ids apart (block x is observed on XCD x % 8: one L2 for the hand-off; a matter of speed only)
template <int NTH, int NCH, int NBF, int NBB>
__global__ __launch_bounds__(NTH) void crf_fac2_pair_kernel(FacParams pf, FacParams pb) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int K = 2;
    const int total = pf.nbu * K;
    const bool fwd = (int)blockIdx.x < total;
    const int x = fwd ? (int)blockIdx.x : (int)blockIdx.x - total;
    const int full = total / (8 * K) * (8 * K);
    int b, k;
    if (x < full) { const int grp = x / (8 * K), within = x % (8 * K); k = within / 8; b = pf.b0 + grp * 8 + within % 8; }
    else { const int y = x - full; k = y % K; b = pf.b0 + full / K + y / K; }
    if (fwd) fac_chain_body<0, false, NTH, NCH, NBF, true, true, true>(pf, lds, b, k);
    else fac_chain_body<1, false, NTH, NCH, NBB, true, true, true>(pb, lds, b, k);
}

// Holds a (side) stream until `target` workgroups of the den kernels have started, i.e. own their compute
// units: the numerator chains launched behind it then land on the remaining CUs instead of scattering over
// all of them and keeping den workgroups (which need a whole CU's registers) waiting.  Bounded: after ~0.2 ms
// it lets go regardless (a speed matter only; the den kernels do not depend on this kernel).
__global__ void crf_gate_kernel(const int *started, int target) {
    for (int spins = 0; spins < 120; ++spins) {   // ~0.2 ms at most
        if (__hip_atomic_load(started, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) return;
        __builtin_amdgcn_s_sleep(64);
    }
    // timed out: nothing depends on this for correctness
}

// Streaming denominator recursions and the numerator chains: forward and backward of every utterance in ONE grid of 2B
// workgroups each (block x < B: forward).  The two grids (denominator pair, numerator pair) are independent and run side
// by side on the caller's stream and one side stream (crf_loss_fwd_bwd).
template <bool GV>
__global__ __launch_bounds__(kChainThreads) void crf_den_pair_kernel(LossParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if ((int)blockIdx.x < p.B) den_forward<GV>(p, (int)blockIdx.x, lds);
    else den_backward<GV>(p, (int)blockIdx.x - p.B, lds);
}
// NR: ctc states per thread, chosen by the host from the batch's longest label sequence.
template <int NR>
__global__ __launch_bounds__(kCtcThreads, NR == 1 ? CRF_X_CTCWPE : 1) void crf_ctc_pair_kernel(LossParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if ((int)blockIdx.x < p.B) ctc_forward<NR>(p, (int)blockIdx.x, lds);
    else ctc_backward<NR>(p, (int)blockIdx.x - p.B, lds);
}

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
#ifndef CRF_GD_FRAMES
#define CRF_GD_FRAMES 16
#endif
constexpr int kGDThreads = 256, kGDFrames = CRF_GD_FRAMES, kGDRowRegs = 5;   // rows of up to 5*256 float4 = 5120 floats
constexpr int kGDEpRegs = 4;                                      // V <= 4*256
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
template <int NCPT, int EPR, int NT = kGDThreads, int CH = kChunk, int RR = kGDRowRegs, int WPE = 1>
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
            for (unsigned spins = 0; __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < p.gd_target; ++spins) {
                if (spins > (1u << 21)) { __hip_atomic_store(p.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }   // (~7 s)
                if ((spins & 63u) == 63u && __hip_atomic_load(p.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                __builtin_amdgcn_s_sleep(127);
            }
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
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
constexpr int kGCThreads = 256, kGCFrames = 16, kGCRegs = 16, kGCVRegs = 4;  // 2L+1 <= 4096, V <= 1024
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

// =============================================================================================
// UTTERANCE-MINOR ("batch") denominator for graphs that do not fit the register-resident layouts (crf_internal.h:
// BatchDev).  The reference runs ANY graph with one launch per frame and one block per utterance, re-reading every arc
// for every utterance (den_calculate.cu:75-103, 189-227, 443-476); the streaming kernels above do the same from one
// persistent workgroup per utterance.  Here the batch is the minor dimension of everything:
//     a_t  [group][state][ul]     z_t [group][pair][ul]     Q_t, BP_t [group][pair][ul]     e'_t [group][label][ul]
// (utterance u = group * UL + ul; UL utterances = one 32..256-byte segment per entry, chosen so that ONE group's state
// vector stays in an XCD's 4 MiB L2: ws_layout).  A wave takes one row (a destination state forward, a source state backward) with the utterances in its lanes (UL
// utterances x 64/UL arcs of the row side by side), so an arc is fetched ONCE per frame for the whole batch and every
// gather of a state-vector entry is one contiguous UL*4-byte segment.  One launch per frame -- the kernel boundary is
// the grid barrier and makes the vectors visible across XCDs -- with the forward step of frame j and the backward step
// of frame T-j in the same launch.  XCD placement (speed only): the per-XCD L2s do not share, and with every XCD
// gathering from every group's vectors of both directions (8.4 MB at S = 16 k, B = 64) nearly every gather missed L2
// (measured 93 us per launch = 3 TB/s of fabric reads).  A (group, direction) "combo" therefore belongs to 8 / #combos
// XCDs (block b runs on XCD b % 8): an XCD gathers from ONE vector that fits its L2 and streams its share of the arcs.  Scaling: per utterance and frame an exact power of two from the maximum of the
// vector (atomic max per utterance, three slots in rotation), integer exponents carried per utterance.
// Backward frames are aligned at the END of the padded batch (iteration i handles frame T-1-i of every utterance); an
// utterance joins when the iteration reaches its last frame.
// =============================================================================================
struct BatchParams {
    BatchDev g;
    StreamDev st;                  // arc streams for AL = 64 / UL lane groups
    const float *start_lin, *end_lin;
    const float *x_start;          // [SX] a_0 of the forward vector's entries: the S states, then the U entries of factored streams
    int S, P, B, Bp, T, V, max_label, ngrp;
    int SX;                        // entries of the forward vector (S + StreamDev::NU)
    const int *lx;
    const float *ep, *moff;        // [B][T][V] e' (prep kernel), [B][T] log-likelihood offset per frame
    float *ept;                    // [T][grp][V][UL] e' transposed
    float *Af, *Zb;                // [2][grp][SX][UL], [2][grp][P][UL]
    float *Q, *BP;                 // [T][grp][P][UL]
    unsigned *mxf, *mxb;           // [3][Bp] maxima of the vectors (float bits; the values are non-negative)
    int *Ef, *Fb;                  // [Bp] running exponents
    float *zs, *zb;                // [Bp] scaled partition sums
    float *den_zs, *cost_alpha, *cost_beta;
    int *den_ez, *redo;
    float *grad;                   // [B][T][V]
    float c_den;
    int j;                         // launch number: forward frame j, backward frame T - j
};
constexpr int kBatThreads = 256, kBatWaves = kBatThreads / kWave;

// ep [B][T][V] -> ept [T][grp][V][UL] through a 64 x UL tile in LDS.  grid (ceil(V / 64), T, Bp / UL)
template <int UL>
__global__ __launch_bounds__(kBatThreads) void crf_batch_transpose_kernel(BatchParams p) {
    __shared__ float tile[UL][65];
    const int v0 = blockIdx.x * 64, t = blockIdx.y, u0 = blockIdx.z * UL, tid = threadIdx.x;
    for (int i = tid; i < UL * 64; i += kBatThreads) {
        const int u = i >> 6, v = i & 63;
        const bool ok = u0 + u < p.B && v0 + v < p.V && t < p.lx[u0 + u];
        tile[u][v] = ok ? p.ep[((int64_t)(u0 + u) * p.T + t) * p.V + v0 + v] : 0.f;
    }
    __syncthreads();
    for (int i = tid; i < UL * 64; i += kBatThreads) {
        const int v = i / UL, u = i % UL;
        if (v0 + v < p.V) p.ept[(((int64_t)t * p.ngrp + blockIdx.z) * p.V + v0 + v) * UL + u] = tile[u][v];
    }
}

// a_0, the slots, the exponents.  grid: enough blocks for S * Bp elements
__global__ __launch_bounds__(kBatThreads) void crf_batch_init_kernel(BatchParams p) {
    const int64_t i = (int64_t)blockIdx.x * kBatThreads + threadIdx.x;
    const int UL = p.Bp / p.ngrp;
    if (i < (int64_t)p.SX * p.Bp) p.Af[i] = p.x_start[(i / UL) % p.SX] * pow2f(kScaleExp);
    if (blockIdx.x == 0) {
        float m = 0.f;
        for (int s = threadIdx.x; s < p.SX; s += kBatThreads) m = fmaxf(m, p.x_start[s]);
        __shared__ float red[kBatWaves];
        m = wave_max(m);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
        __syncthreads();
        m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) * pow2f(kScaleExp);
        for (int u = threadIdx.x; u < p.Bp; u += kBatThreads) {
            p.mxf[u] = __float_as_uint(m); p.mxf[p.Bp + u] = 0u; p.mxf[2 * p.Bp + u] = 0u;
            p.mxb[u] = 0u; p.mxb[p.Bp + u] = 0u; p.mxb[2 * p.Bp + u] = 0u;
            p.Ef[u] = kScaleExp; p.Fb[u] = kScaleExp; p.zs[u] = 0.f; p.zb[u] = 0.f;
        }
    }
}

// sum over the AL = 64/UL arc lanes that share an utterance (lanes u, u + UL, u + 2 UL, ...)
template <int UL>
__device__ __forceinline__ float arc_lane_sum(float v) {
#pragma unroll
    for (int o = 32; o >= UL; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
template <int UL>
__device__ __forceinline__ float arc_lane_max(float v) {
#pragma unroll
    for (int o = 32; o >= UL; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// sum over the arcs [a0, a1) of w * X[idx][u].  The wave fetches 64 arcs at a time (one coalesced 512-byte load), the arc
// lanes of an utterance then walk them through lane broadcasts: the gathers of a chunk are independent loads, all in flight
// together -- with the arc fetched inside the loop every gather waited for its own arc first (two dependent trips to L2 per
// arc: 77 us per frame on the S = 16 k graph).
template <int UL>
__device__ __forceinline__ float bat_row_sum(const int2 *__restrict__ arcs, int a0, int a1, const float *__restrict__ X, int ul, int lane, int aj) {
    constexpr int AL = 64 / UL;
    float acc0 = 0.f, acc1 = 0.f;
    for (int c = a0; c < a1; c += 64) {
        const int n = min(64, a1 - c);
        int2 arc = int2{0, 0};                                    // (lanes past the end: state 0, weight 0)
        if (lane < n) arc = arcs[c + lane];
        const int steps = (n + AL - 1) / AL;
#pragma unroll 8
        for (int i = 0; i < steps; i += 2) {
            int s0, w0, s1, w1;
            if (AL == 1) {
                s0 = __builtin_amdgcn_readlane(arc.x, i); w0 = __builtin_amdgcn_readlane(arc.y, i);
                s1 = __builtin_amdgcn_readlane(arc.x, min(i + 1, 63)); w1 = i + 1 < steps ? __builtin_amdgcn_readlane(arc.y, min(i + 1, 63)) : 0;
            } else {
                s0 = __shfl(arc.x, i * AL + aj); w0 = __shfl(arc.y, i * AL + aj);
                const int l1 = min((i + 1) * AL + aj, 63);
                s1 = __shfl(arc.x, l1); w1 = i + 1 < steps ? __shfl(arc.y, l1) : 0;
            }
            acc0 = fmaf(X[(size_t)s0 * UL + ul], __int_as_float(w0), acc0);
            acc1 = fmaf(X[(size_t)s1 * UL + ul], __int_as_float(w1), acc1);
        }
    }
    return arc_lane_sum<UL>(acc0 + acc1);
}

// A wave's TASK of an arc stream (crf_internal.h: StreamDirDev).  A lane takes FOUR utterances (uq: which four of the
// group's UL) -- one 16-byte gather per arc and lane, LG = UL / 4 lanes a row -- and lane group aj walks row aj of every
// bundle: measured with one utterance per lane, the frame kernel was bound by the NUMBER of memory instructions a CU can
// take (~9 cycles each; with every gather hitting the same cached lines it was no faster), not by what they fetched.
// X = the gathered vector of this utterance group ([entry][UL]).  Everything but the gathers reaches the wave through its
// slice of LDS: the records (two 4 KB halves, kStreamChunk / AL batches each, staged one chunk ahead: the loads of the chunk
// after the next are the OLDEST entries of the memory queue), the row descriptors of the task's (at most kStreamBundles)
// bundles, and a ring of two emission rows filled from a register set loaded one bundle earlier.  So the memory queue holds
// the gathers -- D batches of kStreamBatch in flight, consumed in order behind partial vmcnt waits -- and a row's stores;
// a wait for something just requested happens nowhere.  epi(acc, m, e) is called at every bundle end with the row's
// descriptor {state, pair, label} and the four utterances' emissions et[label].
constexpr int kStreamBatch = 4, kStreamChunk = 128;   // steps per batch; batches * AL per 4 KB chunk (batches in flight: template parameter D)
constexpr int kStreamBundles = 8;                     // bundles per task at most (plain streams; factored: crf_internal.h stream_max_bundles)
constexpr int kStreamRecB = kStreamChunk * 32;        // bytes of a chunk of records
// Factored streams (NM = 3 descriptor words per row, crf_internal.h StreamDev): the ring holds NR values per row instead of the
// one emission row -- forward {e[label 0], e[label 1], U of the row's couple}, backward {e[label 0], e[label 1], z of the two
// extra arcs} -- all known from the descriptor, so all requested a bundle ahead like the emissions; epi gets them as e[NR].
constexpr int kStreamLds = 2 * kStreamRecB + 2 * 64 * 16 + kStreamBundles * 32 * 16;   // per wave: records (2 halves) | emission ring | descriptors
// (factored streams of small utterance groups -- many rows side by side -- have tasks of fewer bundles, so that a workgroup's
// four slices stay below half of the LDS: crf_internal.h stream_max_bundles, shared with the host's task cutter)
template <int UL, bool FAC>
constexpr int stream_lds() { return FAC ? 2 * kStreamRecB + 2 * 64 * 16 * 4 + stream_max_bundles(UL, true) * (256 / UL) * 48 : kStreamLds; }
template <int UL, int D, int NM, int NR, typename Epi>
__device__ __forceinline__ void bat_stream(const StreamDirDev &sd, const int4 tk, const float *__restrict__ X, const float *__restrict__ et,
                                           int uq, int aj, int lane, char *ldsw, int tmi, Epi &&epi) {
    constexpr int LG = UL / 4, AL = 64 / LG, CB = kStreamChunk / AL;
    static_assert(kStreamBatch == 4 && CB % D == 0 && CB >= D, "a chunk holds a whole number of pipeline rounds");
    static_assert((NM == 1 && NR == 1) || (NM == 3 && (NR == 3 || NR == 4)), "plain or factored rows");
    f32x4 *elds = (f32x4 *)(ldsw + 2 * kStreamRecB);               // [2][NR][lane]
    int4 *mlds = (int4 *)(ldsw + 2 * kStreamRecB + 2 * 64 * 16 * (NM == 1 ? 1 : 4));   // [bundle][AL][NM]
    const int b0 = __builtin_amdgcn_readfirstlane(tk.x), nb = __builtin_amdgcn_readfirstlane(tk.y);
    const int bund0 = __builtin_amdgcn_readfirstlane(tk.z), nbund = __builtin_amdgcn_readfirstlane(tk.w);
    CRF_TM(tmi >= 0, tmi + 2);
    const int4 *gsrc = (const int4 *)sd.recs + (size_t)b0 * AL * 2 + lane;   // a chunk = 256 int4: four per lane (the stream is padded)
    {   // chunk 0 -> LDS half 0 (the only wait for something just requested: once per task)
        const int4 s0 = gsrc[0], s1 = gsrc[64], s2 = gsrc[128], s3 = gsrc[192];
        const int4 *mp = sd.meta + (size_t)bund0 * AL * NM;
        constexpr int MAXB = stream_max_bundles(UL, NM == 3);
        constexpr int NMW = (MAXB * AL * NM + 63) / 64;
        int4 mm[NMW];
#pragma unroll
        for (int q = 0; q < NMW; ++q) mm[q] = (q * 64 + lane < nbund * AL * NM) ? mp[q * 64 + lane] : int4{-1, 0, 0, 0};
        *(int4 *)(ldsw + lane * 16) = s0; *(int4 *)(ldsw + (64 + lane) * 16) = s1;
        *(int4 *)(ldsw + (128 + lane) * 16) = s2; *(int4 *)(ldsw + (192 + lane) * 16) = s3;
#pragma unroll
        for (int q = 0; q < NMW; ++q)
            if ((NMW * 64 == MAXB * AL * NM) || q * 64 + lane < MAXB * AL * NM) mlds[q * 64 + lane] = mm[q];   // (the slice ends there)
    }
    CRF_TM(tmi >= 0, tmi + 3);
    int4 st0 = gsrc[256], st1 = gsrc[320], st2 = gsrc[384], st3 = gsrc[448];   // chunk 1 (padding if there is none)
    gsrc += 512;
    const f32x4 *et4 = (const f32x4 *)et + uq;                     // et[label * UL + 4 uq ..]
    const f32x4 *X4 = (const f32x4 *)X + uq;                       // X[entry * UL + 4 uq ..]
    // value k of the row (bundle bd, lane group aj): emissions of its label(s), entries its epilogue reads
    auto ringsrc = [&](const int bd, const int k) __attribute__((always_inline)) -> f32x4 {
        const int4 *m = mlds + ((size_t)bd * AL + aj) * NM;
        if (k == 0) return et4[(size_t)m[0].z * LG];
        if (k == 1) return et4[(size_t)m[NM > 1 ? 1 : 0].z * LG];
        return X4[(size_t)(k == 2 ? m[NM > 1 ? 2 : 0].x : m[NM > 1 ? 2 : 0].z) * LG];
    };
    // values of bundles 0 and 1 -> ring (written once the first gathers are out), bundle 2 -> eA
    f32x4 ei0[NR], ei1[NR], eA[NR];
#pragma unroll
    for (int k = 0; k < NR; ++k) { ei0[k] = ringsrc(0, k); ei1[k] = ringsrc(nbund > 1 ? 1 : 0, k); eA[k] = ringsrc(nbund > 2 ? 2 : 0, k); }
    const unsigned uq16 = (unsigned)uq * 16u;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    f32x4 x[D][4];
    float w[D][4];
    unsigned fl[D];
    int bund = 0;
    auto issue = [&](const int j, const int b) __attribute__((always_inline)) {   // gathers of batch b of the task into slot j
        const int4 *lp = (const int4 *)(ldsw + (((b % (2 * CB)) * AL + aj) * 32));
        int4 r0 = lp[0], r1 = lp[1];
        // (a record's index is entry * UL with the flag in bit 31: the shift to bytes drops the flag -- one VALU per gather --
        // and the address is a 32-bit offset to a uniform base)
        fl[j] = (unsigned)__builtin_amdgcn_readfirstlane(r0.x) >> 31;
        x[j][0] = *(const f32x4 *)((const char *)X + (((unsigned)r0.x << 2) + uq16));
        x[j][1] = *(const f32x4 *)((const char *)X + (((unsigned)r0.z << 2) + uq16));
        x[j][2] = *(const f32x4 *)((const char *)X + (((unsigned)r1.x << 2) + uq16));
        x[j][3] = *(const f32x4 *)((const char *)X + (((unsigned)r1.z << 2) + uq16));
        w[j][0] = __int_as_float(r0.y); w[j][1] = __int_as_float(r0.w); w[j][2] = __int_as_float(r1.y); w[j][3] = __int_as_float(r1.w);
    };
    auto consume = [&](const int j) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < 4; ++k) acc = __builtin_elementwise_fma(x[j][k], (f32x4){w[j][k], w[j][k], w[j][k], w[j][k]}, acc);
        if (fl[j] & 1u) {                                          // (uniform) the rows of the bundle end with this batch
            f32x4 e[NR];
#pragma unroll
            for (int k = 0; k < NR; ++k) e[k] = elds[((bund & 1) * NR + k) * 64 + lane];
            epi(acc, mlds + ((size_t)bund * AL + aj) * NM, e);
            acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < NR; ++k) elds[((bund & 1) * NR + k) * 64 + lane] = eA[k];   // values of bundle + 2 (asked for one bundle ago)
            if (bund + 3 < nbund) {
#pragma unroll
                for (int k = 0; k < NR; ++k) eA[k] = ringsrc(bund + 3, k);
            }
            ++bund;
        }
    };
    auto stage = [&](const int c) __attribute__((always_inline)) {   // entering chunk c: chunk c + 1 -> the other half, ask for chunk c + 2
        if ((c + 1) * CB < nb) {
            char *h = ldsw + ((c + 1) & 1) * kStreamRecB;
            *(int4 *)(h + lane * 16) = st0; *(int4 *)(h + (64 + lane) * 16) = st1;
            *(int4 *)(h + (128 + lane) * 16) = st2; *(int4 *)(h + (192 + lane) * 16) = st3;
            if ((c + 2) * CB < nb) { st0 = gsrc[0]; st1 = gsrc[64]; st2 = gsrc[128]; st3 = gsrc[192]; gsrc += 256; }
        }
    };
    stage(0);
#pragma unroll
    for (int j = 0; j < D - 1; ++j)
        if (j < nb) issue(j, j);
#pragma unroll
    for (int k = 0; k < NR; ++k) { elds[k * 64 + lane] = ei0[k]; elds[(NR + k) * 64 + lane] = ei1[k]; }   // (older than the gathers just issued)
    CRF_TM(tmi >= 0, tmi + 4);
#ifdef CRF_TIMING
    if (tmi >= 0 && lane == 0) { g_tm[tmi + 8] = (unsigned long long)nb; g_tm[tmi + 9] = (unsigned long long)nbund; }
#endif
    int b = 0;
    for (; b + 2 * D - 1 <= nb; b += D) {                          // steady state: no conditions around the loads
        if (b > 0 && b % CB == 0) stage(b / CB);
#pragma unroll
        for (int j = 0; j < D; ++j) { issue((j + D - 1) % D, b + j + D - 1); consume(j); }
    }
    for (; b < nb; b += D) {
        if (b > 0 && b % CB == 0) stage(b / CB);
#pragma unroll
        for (int j = 0; j < D; ++j) {
            if (b + j + D - 1 < nb) issue((j + D - 1) % D, b + j + D - 1);
            if (b + j < nb) consume(j);
        }
    }
    CRF_TM(tmi >= 0, tmi + 5);
}

// One frame of both recursions.  1-D grid of 8 * nslot workgroups; block b sits on XCD b % 8 (observed; a matter of speed
// only) and works for ONE combo = (utterance group, direction):
//   #combos <  8: XCD x serves combo x % #combos together with the other XCDs of that residue, the combo's workgroups
//                 ("chunks") dealt round-robin among them;
//   #combos >= 8: XCD x serves the combos x, x + 8, ..., its slots dealt round-robin among them.
// The waves of a combo take the tasks of its arc stream (rows with one entering pair: all of a T o LM graph) and then the
// remaining rows one at a time (bat_row_sum: one utterance per lane, 64 / UL arcs of the row side by side).
template <int UL, int D, bool FAC = false>
__global__ __launch_bounds__(kBatThreads) void crf_batch_frame_kernel(BatchParams p) {
    constexpr int ALR = 64 / UL;                                   // rest rows: arc lanes per utterance
    constexpr int LG = UL / 4;                                     // stream: lanes per row (64 / LG rows side by side)
    constexpr int NM = FAC ? 3 : 1;                                // descriptor words per stream row
    __shared__ unsigned umax[UL];                                  // maximum of the vector this workgroup wrote, per utterance (float bits)
    __shared__ __attribute__((aligned(16))) char stage[kBatWaves][stream_lds<UL, FAC>()];   // bat_stream: records, value ring, descriptors of a wave's task
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // (timing build: launch 700, four workgroups of the first XCD x their four waves, 16 stamps each from g_tm[14000])
    const int tmi = (p.j == 700 && (blockIdx.x & 7) == 0 && (blockIdx.x >> 3) % 24 == 0 && (blockIdx.x >> 3) < 96) ? 14000 + (((blockIdx.x >> 3) / 24) * 4 + wave) * 16 : -1;
    CRF_TM(tmi >= 0, tmi + 0);
    char *ldsw = stage[wave];
    const int ul = lane % UL, aj = lane / UL;                      // rest rows and the per-utterance scalars: utterance ul of the group
    const int uq = lane % LG, sj = lane / LG;                      // stream: utterances 4 uq .. 4 uq + 3, row sj of the bundle
    const int T = p.T, P = p.P;
    int combo, chunk, nchunk;
    bat_decode((int)blockIdx.x, (int)gridDim.x, 2 * p.ngrp, &combo, &chunk, &nchunk);   // (crf_internal.h)
    const int dir = combo & 1, grp = combo >> 1;
    const int u = grp * UL + ul;
    const int lx = u < p.B ? p.lx[u] : 0;
    const int u4 = grp * UL + 4 * uq;                              // first of the lane's four utterances (stream)
    int lx4[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) lx4[c] = u4 + c < p.B ? p.lx[u4 + c] : 0;
    const int w0 = chunk * kBatWaves + wave, NW = nchunk * kBatWaves;   // this wave among the waves of its combo
    // the wave's first task descriptor: asked for before anything else (it heads the chain descriptor -> records and row
    // descriptors -> emissions -> first gathers, three dependent trips to a cold L2 at the start of every launch)
    const StreamDirDev &sdd = dir == 0 ? p.st.f : p.st.b;
    int4 tk0 = int4{0, 0, 0, 0};
    if (w0 < sdd.ntasks) tk0 = sdd.tasks[w0];
    const bool lead = chunk == 0 && wave == 0 && aj == 0;          // one writer per utterance for the scalars
    const BatchDev &g = p.g;
    const size_t gS = (size_t)grp * p.SX * UL, gP = (size_t)grp * P * UL, gV = (size_t)grp * p.V * UL;
    const size_t Sall = (size_t)p.SX * p.Bp, Pall = (size_t)P * p.Bp, Vall = (size_t)p.V * p.Bp;
    if (tid < UL) umax[tid] = 0u;
    __syncthreads();
    CRF_TM(tmi >= 0 && lx4[0] + lx4[1] + lx4[2] + lx4[3] + lx >= 0, tmi + 1);   // (the scalar loads have landed)
    float mymax = 0.f;                                             // rest rows: utterance ul
    f32x4 mymax4 = {0.f, 0.f, 0.f, 0.f};                           // stream: the lane's four utterances
    if (dir == 0) {
        const int t = p.j;
        if (t >= T) return;
        const bool active = t < lx;
        const int k = rescale_exp(__uint_as_float(p.mxf[(t % 3) * p.Bp + u]));
        const float sc = pow2f(k);
        f32x4 sc4;
        bool act4[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) { sc4[c] = pow2f(rescale_exp(__uint_as_float(p.mxf[(t % 3) * p.Bp + u4 + c]))); act4[c] = t < lx4[c]; }
        const bool all4 = act4[0] && act4[1] && act4[2] && act4[3];
        const float *Ac = p.Af + (size_t)(t & 1) * Sall + gS;
        float *An = p.Af + (size_t)((t + 1) & 1) * Sall + gS;
        const float *et = p.ept + (size_t)t * Vall + gV;
        float *Qt = p.Q + (size_t)t * Pall + gP;
        for (int task = w0; task < p.st.f.ntasks; task += NW)
            bat_stream<UL, D, NM, FAC ? 3 : 1>(p.st.f, task == w0 ? tk0 : p.st.f.tasks[task], Ac, et, uq, sj, lane, ldsw, tmi, [&](const f32x4 &acc, const int4 *m, const f32x4 *e) __attribute__((always_inline)) {
                if (m[0].x < 0) return;                            // padding row of the last bundle
                // an utterance that has ended keeps a_lx where it is: nobody writes that buffer for it again
                // (crf_batch_zsum_kernel reads it there)
                const f32x4 q = acc * sc4, an = e[0] * q;
                float *qp = Qt + (size_t)m[0].y * UL + 4 * uq, *ap = An + (size_t)m[0].x * UL + 4 * uq;
                if (all4) { *(f32x4 *)qp = q; *(f32x4 *)ap = an; }
                else {
#pragma unroll
                    for (int c = 0; c < 4; ++c) if (act4[c]) { qp[c] = q[c]; ap[c] = an[c]; }
                }
                f32x4 top = an;                                    // the largest entry this row writes
                if constexpr (FAC) {
                    if (m[1].x >= 0) {                             // the couple's tail row, folded in: q = w * U_t, and U_{t+1} = both states' a
                        const float tw = __int_as_float(m[2].y);
                        const f32x4 qt = e[2] * (f32x4){tw, tw, tw, tw} * sc4, at = e[1] * qt, un = an + at;
                        float *qp1 = Qt + (size_t)m[1].y * UL + 4 * uq, *ap1 = An + (size_t)m[1].x * UL + 4 * uq, *up = An + (size_t)m[2].x * UL + 4 * uq;
                        if (all4) { *(f32x4 *)qp1 = qt; *(f32x4 *)ap1 = at; *(f32x4 *)up = un; }
                        else {
#pragma unroll
                            for (int c = 0; c < 4; ++c) if (act4[c]) { qp1[c] = qt[c]; ap1[c] = at[c]; up[c] = un[c]; }
                        }
                        top = un;
                    }
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) if (act4[c]) mymax4[c] = fmaxf(mymax4[c], top[c]);
            });
        for (int i = w0; i < p.st.f.nrest; i += NW) {
            const int r = __builtin_amdgcn_readfirstlane(p.st.f.rest[i]);
            const int d = __builtin_amdgcn_readfirstlane(g.frow_d[r]);   // (the row is the wave's: everything about it is uniform)
            int4 ds = g.frow[r];
            ds.x = __builtin_amdgcn_readfirstlane(ds.x); ds.y = __builtin_amdgcn_readfirstlane(ds.y);
            ds.z = __builtin_amdgcn_readfirstlane(ds.z); ds.w = __builtin_amdgcn_readfirstlane(ds.w);
            float acc = 0.f;
            if (ds.w & 0x40000000) {                               // one pair enters the state
                const float e = et[(size_t)(ds.w & 0xffff) * UL + ul];   // (requested before the arcs)
                const float q = bat_row_sum<UL>(g.farcs, ds.x, ds.y, Ac, ul, lane, aj) * sc;
                if (aj == 0 && active) Qt[(size_t)ds.z * UL + ul] = q;
                acc = e * q;
            } else {
                for (int kk = ds.z; kk < ds.w; ++kk) {
                    int4 pl = g.stp[kk];
                    pl.z = __builtin_amdgcn_readfirstlane(pl.z); pl.w = __builtin_amdgcn_readfirstlane(pl.w);
                    const float q = bat_row_sum<UL>(g.farcs, pl.z, pl.w, Ac, ul, lane, aj) * sc;
                    if (aj == 0 && active) Qt[(size_t)pl.x * UL + ul] = q;
                    acc = fmaf(et[(size_t)pl.y * UL + ul], q, acc);
                }
            }
            if (aj == 0 && active) An[(size_t)d * UL + ul] = acc;
            if (active) mymax = fmaxf(mymax, acc);
        }
        if (lead) {
            if (active) p.Ef[u] += k + kEpExp;                    // exponent of a_{t+1}
            p.mxf[((t + 2) % 3) * p.Bp + u] = 0u;                  // the slot the launch after next adds to
        }
    } else {
        const int t = T - p.j;                                     // t = T (nothing active yet) ... 0
        const bool active = t < lx;                                // b_t of this utterance is computed
        const bool starts = t - 1 == lx - 1 && lx > 0;             // frame t-1 is its last frame: z_{lx-1} is set up
        const int k = rescale_exp(__uint_as_float(p.mxb[(p.j % 3) * p.Bp + u]));
        const float sc = pow2f(k);
        f32x4 sc4;
        bool act4[4], st4[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            sc4[c] = pow2f(rescale_exp(__uint_as_float(p.mxb[(p.j % 3) * p.Bp + u4 + c])));
            act4[c] = t < lx4[c]; st4[c] = t == lx4[c] && lx4[c] > 0;
        }
        const bool all4 = act4[0] && act4[1] && act4[2] && act4[3];
        const float *Zc = p.Zb + (size_t)(p.j & 1) * Pall + gP;
        float *Zn = p.Zb + (size_t)((p.j + 1) & 1) * Pall + gP;
        const float *ep1 = p.ept + (size_t)(t >= 1 ? t - 1 : 0) * Vall + gV;   // (t = 0: read, not used)
        float *BPt = t >= 1 ? p.BP + (size_t)(t - 1) * Pall + gP : nullptr;
        const bool any_active = __ballot(active) != 0ull;
        if (any_active || __ballot(starts) != 0ull)
            for (int task = w0; task < p.st.b.ntasks; task += NW)
                bat_stream<UL, D, NM, FAC ? 4 : 1>(p.st.b, task == w0 ? tk0 : p.st.b.tasks[task], Zc, ep1, uq, sj, lane, ldsw, tmi, [&](const f32x4 &acc, const int4 *m, const f32x4 *e) __attribute__((always_inline)) {
                    if (m[0].x < 0) return;
                    // one output (plain rows), or the two states of a couple: the common out-arcs' sum + each state's extra arc
                    auto output = [&](const f32x4 &bv, const int st_, const int pr_, const f32x4 &em) __attribute__((always_inline)) {
                        if (t == 0) {
                            const float st = p.start_lin[st_];
                            if (st != 0.f) {
#pragma unroll
                                for (int c = 0; c < 4; ++c) if (act4[c]) atomicAdd(&p.zb[u4 + c], st * bv[c]);
                            }
                            return;
                        }
                        float *bp = BPt + (size_t)pr_ * UL + 4 * uq, *zp = Zn + (size_t)pr_ * UL + 4 * uq;
                        if (all4) {
                            const f32x4 z = em * bv;
                            *(f32x4 *)bp = bv; *(f32x4 *)zp = z;
                            mymax4 = __builtin_elementwise_max(mymax4, z);
                        } else {
                            const float eend = p.end_lin[st_] * pow2f(kScaleExp);
#pragma unroll
                            for (int c = 0; c < 4; ++c)
                                if (act4[c] || st4[c]) {
                                    const float out = act4[c] ? bv[c] : eend, z = em[c] * out;
                                    bp[c] = out; zp[c] = z;
                                    mymax4[c] = fmaxf(mymax4[c], z);
                                }
                        }
                    };
                    if constexpr (FAC) {
                        const float w0 = __int_as_float(m[2].y), w1 = __int_as_float(m[2].w);
                        output(__builtin_elementwise_fma(e[2], (f32x4){w0, w0, w0, w0}, acc) * sc4, m[0].x, m[0].y, e[0]);
                        if (m[1].x >= 0) output(__builtin_elementwise_fma(e[3], (f32x4){w1, w1, w1, w1}, acc) * sc4, m[1].x, m[1].y, e[1]);
                    } else {
                        output(acc * sc4, m[0].x, m[0].y, e[0]);
                    }
                });
        for (int i = w0; i < p.st.b.nrest; i += NW) {
            const int r = __builtin_amdgcn_readfirstlane(p.st.b.rest[i]);
            const int s = __builtin_amdgcn_readfirstlane(g.brow_s[r]);
            int4 ds = g.brow[r];
            ds.x = __builtin_amdgcn_readfirstlane(ds.x); ds.y = __builtin_amdgcn_readfirstlane(ds.y);
            ds.z = __builtin_amdgcn_readfirstlane(ds.z); ds.w = __builtin_amdgcn_readfirstlane(ds.w);
            const bool one = (ds.w & 0x40000000) != 0;
            float e1 = 0.f;
            if (one && t >= 1) e1 = ep1[(size_t)(ds.w & 0xffff) * UL + ul];      // (requested before the arcs)
            const float bv = any_active ? bat_row_sum<UL>(g.barcs, ds.x, ds.y, Zc, ul, lane, aj) * sc : 0.f;
            if (t == 0) {
                const float st = p.start_lin[s];
                if (st != 0.f && active && aj == 0) atomicAdd(&p.zb[u], st * bv);
            } else if (active || starts) {
                const float out = active ? bv : p.end_lin[s] * pow2f(kScaleExp);
                if (one) {
                    const float z = e1 * out;
                    if (aj == 0) { BPt[(size_t)ds.z * UL + ul] = out; Zn[(size_t)ds.z * UL + ul] = z; }
                    mymax = fmaxf(mymax, z);
                } else {
                    for (int kk = ds.z + aj; kk < ds.w; kk += ALR) {
                        const int4 pl = g.stp[kk];
                        BPt[(size_t)pl.x * UL + ul] = out;
                        const float z = ep1[(size_t)pl.y * UL + ul] * out;
                        Zn[(size_t)pl.x * UL + ul] = z;
                        mymax = fmaxf(mymax, z);
                    }
                }
            }
        }
        if (lead) {
            if (starts) p.Fb[u] = kScaleExp;
            else if (active) p.Fb[u] += k + kEpExp;
            p.mxb[((p.j + 2) % 3) * p.Bp + u] = 0u;
        }
    }
    // maximum of the vector this launch wrote, per utterance: lanes -> LDS (the values are non-negative: their bits order
    // like unsigned integers) -> one atomic per utterance and workgroup
    if (mymax > 0.f) atomicMax(&umax[ul], __float_as_uint(mymax));
#pragma unroll
    for (int c = 0; c < 4; ++c) if (mymax4[c] > 0.f) atomicMax(&umax[4 * uq + c], __float_as_uint(mymax4[c]));
    __syncthreads();
    if (tid < UL) {
        const unsigned m = umax[tid];
        unsigned *slot = (dir == 0 ? p.mxf : p.mxb) + ((p.j + 1) % 3) * p.Bp + grp * UL + tid;
        if (m != 0u) atomicMax(slot, m);
    }
    CRF_TM(tmi >= 0, tmi + 6);
}

// zs[u] = sum_s a_{lx}[s][u] * end[s].  grid (ceil(S / (4 * 64)), 1, Bp / UL): a wave sums 64 states
template <int UL>
__global__ __launch_bounds__(kBatThreads) void crf_batch_zsum_kernel(BatchParams p) {
    constexpr int AL = 64 / UL;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ul = lane % UL, aj = lane / UL, u = blockIdx.z * UL + ul;
    const int lxu = u < p.B ? p.lx[u] : 0;                         // a_lx sits in the buffer frame lx - 1 wrote: parity lx & 1
    const float *Af = p.Af + (size_t)(lxu & 1) * p.SX * p.Bp + (size_t)blockIdx.z * p.SX * UL;
    const int s0 = (blockIdx.x * kBatWaves + wave) * 64;
    float acc = 0.f;
    for (int s = s0 + aj; s < min(s0 + 64, p.S); s += AL) acc = fmaf(Af[(size_t)s * UL + ul], p.end_lin[s], acc);
    acc = arc_lane_sum<UL>(acc);
    if (aj == 0 && acc != 0.f) atomicAdd(&p.zs[u], acc);
}

// per utterance: costs from the scaled sums, the exponents and the per-frame offsets; flags for the robust fallback
__global__ __launch_bounds__(kBatThreads) void crf_batch_cost_kernel(BatchParams p) {
    __shared__ double red[kBatWaves];
    const int b = blockIdx.x, tid = threadIdx.x, lx = p.lx[b];
    double part = 0.0;
    for (int t = tid; t < lx; t += kBatThreads) part += (double)p.moff[(int64_t)b * p.T + t];
    part = wave_sum_d(part);
    if ((tid & 63) == 0) red[tid >> 6] = part;
    __syncthreads();
    if (tid == 0) {
        const double mxs = red[0] + red[1] + red[2] + red[3];
        float zs = p.zs[b], zb = p.zb[b];
        int ef = p.Ef[b], fb = p.Fb[b];
        if (lx <= 0) {                                             // empty utterance: logZ = LSE(start + end)
            float z0 = 0.f;
            for (int s = 0; s < p.S; ++s) z0 += p.start_lin[s] * p.end_lin[s];
            zs = zb = z0 * pow2f(kScaleExp); ef = fb = kScaleExp;
        }
        p.den_zs[b] = zs; p.den_ez[b] = ef;
        p.cost_alpha[b] = to_log(zs, ef, mxs);
        p.cost_beta[b] = to_log(zb, fb, mxs);
        if (!(zs > 0.f && zs < INFINITY)) p.redo[b] = 1;
        if (!(zb > 0.f && zb < INFINITY)) p.redo[p.B + b] = 1;
    }
}

// gamma_den[u][t][v] = u_v / sum_v u_v,  u_v = e'_t[v][u] * sum_{p: lab_p = v} Q_t[p][u] * BP_t[p][u]
// (each frame normalises itself).  grid (T, 1, Bp / UL); a wave takes the labels wave, wave + 4, ...; the un-normalised
// values go to the grad row first and are scaled by the same lanes afterwards.
template <int UL>
__global__ __launch_bounds__(kBatThreads) void crf_batch_grad_kernel(BatchParams p) {
    constexpr int AL = 64 / UL;
    __shared__ float wsum[kBatWaves][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ul = lane % UL, aj = lane / UL, u = blockIdx.z * UL + ul;
    const int t = blockIdx.x, V = p.V, Bp = p.Bp;
    const bool real = u < p.B;
    const bool active = real && t < p.lx[u];
    const float *Qt = p.Q + (size_t)t * p.P * Bp + (size_t)blockIdx.z * p.P * UL, *Bt = p.BP + (size_t)t * p.P * Bp + (size_t)blockIdx.z * p.P * UL;
    const float *et = p.ept + (size_t)t * V * Bp + (size_t)blockIdx.z * V * UL;
    float *row = real ? p.grad + ((size_t)u * p.T + t) * V : nullptr;
    float part = 0.f;
    for (int v = wave; v < V; v += kBatWaves) {
        float acc = 0.f;
        if (v <= p.max_label && active) {
            const int p0 = p.g.lab_off[v], p1 = p.g.lab_off[v + 1];
#pragma unroll 4
            for (int q = p0 + aj; q < p1; q += AL) acc = fmaf(Qt[(size_t)q * UL + ul], Bt[(size_t)q * UL + ul], acc);
            acc = arc_lane_sum<UL>(acc);                          // (every arc lane of the utterance holds the sum)
        }
        const float uv = active ? (et[(size_t)v * UL + ul] * pow2f(-kGradDescale)) * acc : 0.f;
        part += uv;
        if (aj == 0 && real) row[v] = uv;                          // un-normalised; 0 past the utterance's length
    }
    wsum[wave][lane] = part;
    __syncthreads();
    const float nrm = wsum[0][lane] + wsum[1][lane] + wsum[2][lane] + wsum[3][lane];
    const float inv = nrm > 0.f ? p.c_den / nrm : 0.f;
    if (aj != 0 || !active) return;
    if (wave == 0 && !(nrm >= 0x1p-120f && nrm < INFINITY) && p.redo) p.redo[u] = 1;   // a frame without (normal, finite) mass: log-domain fallback
    for (int v = wave; v < V; v += kBatWaves) row[v] *= inv;       // the lane's own stores: program order
}

// =============================================================================================
// ROBUST denominator (fallback, rare).  The fast recursions run in fp32 scaled per frame by a power of two and take
// the emissions as e' = exp(logp - rowmax) * 2^64: an utterance in which, at some frame, EVERY live (state, label)
// lies more than ~131 nats below the row maximum loses all its mass (logZ = -inf) where the reference's log-domain
// arithmetic (den_calculate.cu:29-35, 75-103, 189-227) stays finite -- e.g. a peaked network output whose arg-max
// label the un-smoothed n-gram den_lm forbids.  Such utterances (flagged by the fast kernels, p.redo) are redone here
// with the emission scale taken from the largest REACHED product instead of the row maximum:
//     D_t = max over live pairs p of ( d_t[lab_p] + ln q_t[p] ),   d = logp - rowmax   (fp64)
//     a_{t+1}[dst_p] = exp(d_t[lab_p] - D_t) * 2^20 * q_t[p]
// so the largest new entry is 2^20 whatever the emissions are; ln(total scale) is carried in fp64.  The rows Q / BP
// are stored as in the streaming kernels (pair order, first Pr entries of the workspace rows); the grad pass of a
// frame is a softmax over labels of d_t[v] + ln(sum of its pairs' Q * BP) -- no per-frame exponents needed.
// One workgroup per flagged utterance and direction; unflagged utterances leave at once.
// LDS fwd: X[3][Sp] | Ql[Pr] | Dv (double)[Vp] | wm[32] | red[16] (double)      (GV: X and Ql in global memory)
// LDS bwd: Z[2][Pr] | BPst[2][Pr] | Dv (double)[Vp] | wm[32] | red[16] (double)
// =============================================================================================
__device__ __forceinline__ double block_max_d(double v, double *red, int tid) {   // any sign; -inf = nothing
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    double m = red[0];
#pragma unroll
    for (int i = 1; i < kChainWaves; ++i) m = fmax(m, red[i]);
    return m;
}
__device__ __forceinline__ void load_drow(const LossParams &p, int64_t frame, double *Dv, int tid) {
    const double mx = (double)p.mx[frame];
    for (int v = tid; v < p.V; v += kChainThreads) Dv[v] = (double)ld_x(p, frame * p.V + v) - mx;
}

// log(e^m1 * s1 + e^v) kept as (m, s): running maximum in fp64, the sum of exp(differences) -- all <= 1 -- in fp32 (relative 1e-7 per term:
// an absolute 1e-7 on a logarithm; over 3 000 frames a random walk of ~5e-6, cf. lse3)
__device__ __forceinline__ void lse_add(double &m, float &s, double v) {
    if (!(v > -INFINITY)) return;
    if (v > m) { s = s * __expf((float)(m - v)) + 1.f; m = v; }   // (m = -inf: s = 0 * 0 + 1)
    else s += __expf((float)(v - m));
}
__device__ __forceinline__ double lse_value(double m, float s) { return m > -INFINITY ? m + (double)logf(s) : -INFINITY; }
// log of one ELL row's sum over its arcs of w * exp(x[index]): x holds LOGARITHMS (fp64), the weights are the tables' linear ones
__device__ __forceinline__ double ell_row_lse(const uint4 *a4, int n, const double *x) {
    double m = -INFINITY;
    float sm = 0.f;
    for (int k = 0; k < n; ++k) {
        const uint4 c = a4[(size_t)k * kWave];
        const float w0 = __uint_as_float(c.y), w1 = __uint_as_float(c.w);
        if (w0 > 0.f) lse_add(m, sm, x[c.x] + (double)logf(w0));
        if (w1 > 0.f) lse_add(m, sm, x[c.z] + (double)logf(w1));
    }
    return lse_value(m, sm);
}
// log-sum-exp over the workgroup of one (m, s) pair per thread
__device__ __forceinline__ double block_lse(double m, float sm, double *red, int tid) {
    const double M = block_max_d(m, red, tid);
    const double part = (M > -INFINITY && m > -INFINITY) ? (double)sm * exp(m - M) : 0.0;
    const double tot = block_sum_d(part, red, tid);
    return (M > -INFINITY && tot > 0.0) ? M + log(tot) : -INFINITY;
}

// The recursions of the reference in ITS domain -- logarithms (den_calculate.cu:29-35 log_plus, :75-103 alpha_next, :189-227 beta) -- at
// double width: alpha_{t+1}[s] = logsumexp over the pairs p entering s of ( logp_t[lab_p] + lq_t[p] ), lq_t[p] = logsumexp over the arcs
// of p of ( alpha_t[src] + ln w ).  No scale, no range: a state a thousand nats below the frame's best keeps its value, which the scaled
// fp32 vectors of the fast kernels (and of this fallback's first form, rounds 2 - 4: linear fp32 with a per-frame shift) cannot -- their
// entries end 2^-146 below the frame maximum, and a path that far behind at ONE frame was lost for good even if later frames made it the
// dominant one (tests/test_gpu_fuzz.py, round 5: network outputs hundreds of nats apart over den_lm with one or two arcs per state).
// Rows for crf_robust_grad_kernel: lq_t[p] and lb_t[p] as fp32 relative to their frame's maximum (the constants cancel in the frame's
// softmax over labels), pair order, first Pr entries of the workspace rows.
// LDS fwd: A[2][Sp] | Ql[Pr] | Dv[Vp] | red[16]   (doubles; GV: A and Ql in global memory)
// LDS bwd: Z[2][Pr] | BPst[2][Pr] | Dv[Vp] | red[16]
template <bool GV>
__device__ __forceinline__ void den_forward_robust(const LossParams &p, int b, float *lds) {
    const GraphDev &g = p.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int S = g.S, V = p.V, Pr = g.Pr, lx = p.lx[b];
    const int Sp = rup64(S), Vp = rup64(V);
    double *X = GV ? (double *)(p.gvec + (size_t)b * p.gvec_stride) : (double *)lds;   // [2][Sp]
    double *Ql = X + 2 * (size_t)Sp;                                                    // [Pr]
    double *Dv = GV ? (double *)lds : Ql + Pr;
    double *red = Dv + Vp;
    const int64_t bt0 = (int64_t)b * p.T;
    if (lx <= 0) return;                                    // nothing to redo: no emission is involved
    for (int s = tid; s < Sp; s += kChainThreads) X[s] = (s < S && g.start_lin[s] > 0.f) ? log((double)g.start_lin[s]) : -INFINITY;
    double off = 0.0;                                       // sum of the frames' log-likelihood offsets (logp = d + moff)
    __syncthreads();
    const int sl0 = g.fwd.wave_off[wave], sl1 = g.fwd.wave_off[wave + 1];
    for (int t = 0; t < lx; ++t) {
        const double *Xc = X + (size_t)(t & 1) * Sp;
        double *Xn = X + (size_t)((t + 1) & 1) * Sp;
        load_drow(p, bt0 + t, Dv, tid);
        float *Qrow = p.Q + (bt0 + t) * p.Rq;
        double smax = -INFINITY;
        for (int i = sl0; i < sl1; ++i) {
            const int j = __builtin_amdgcn_readfirstlane(g.fwd.wave_slices[i]);
            const int o = __builtin_amdgcn_readfirstlane(g.fwd.slice_off[j]);
            const int w2 = __builtin_amdgcn_readfirstlane(g.fwd.slice_w2[j]);
            const int r = j * kWave + lane;
            const double lq = g.pair_meta[r].x >= 0 ? ell_row_lse(g.fwd.arcs + o + lane, w2, Xc) : -INFINITY;
            Ql[r] = lq;
            smax = fmax(smax, lq);
        }
        const double M = block_max_d(smax, red, tid);       // (its barriers also publish Ql and Dv)
        for (int r = tid; r < Pr; r += kChainThreads) Qrow[r] = Ql[r] > -INFINITY ? (float)(Ql[r] - M) : -INFINITY;
        for (int s2 = tid; s2 < S; s2 += kChainThreads) {   // every state from the pairs that enter it (one, in T o LM)
            double m = -INFINITY;
            float sm = 0.f;
            for (int pi = g.st_pair_off[s2]; pi < g.st_pair_off[s2 + 1]; ++pi) {
                const int r = g.st_pairs[pi];
                lse_add(m, sm, Dv[g.pair_meta[r].y & 0xffff] + Ql[r]);
            }
            Xn[s2] = lse_value(m, sm);
        }
        off += (double)p.moff[bt0 + t];
        __syncthreads();
    }
    const double *Xf = X + (size_t)(lx & 1) * Sp;
    double m = -INFINITY;
    float sm = 0.f;
    for (int s2 = tid; s2 < S; s2 += kChainThreads) if (g.end_lin[s2] > 0.f) lse_add(m, sm, Xf[s2] + log((double)g.end_lin[s2]));
    const double lz = block_lse(m, sm, red, tid);
    if (tid == 0) p.cost_alpha[b] = (float)(lz + off);
}

template <bool GV>
__device__ __forceinline__ void den_backward_robust(const LossParams &p, int b, float *lds) {
    const GraphDev &g = p.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int S = g.S, V = p.V, Pr = g.Pr, lx = p.lx[b];
    const int Vp = rup64(V);
    double *Z = GV ? (double *)(p.gvec + (size_t)b * p.gvec_stride) + 2 * (size_t)rup64(S) + Pr : (double *)lds;   // [2][Pr] log z_t of every pair
    double *BPst = Z + 2 * (size_t)Pr;                      // [2][Pr] log b_{t+1}[dst_p], written one iteration late
    double *Dv = GV ? (double *)lds : BPst + 2 * (size_t)Pr;
    double *red = Dv + Vp;
    const int64_t bt0 = (int64_t)b * p.T;
    if (lx <= 0) return;
    load_drow(p, bt0 + lx - 1, Dv, tid);
    double off = 0.0;
    for (int t = 0; t < lx; ++t) off += (double)p.moff[bt0 + t];   // (every thread: lx <= T adds, once per redone utterance)
    __syncthreads();
    {   // z_{lx-1}[p] = logp_{lx-1}[lab_p] + ln end[dst_p];  BP[lx-1][p] = ln end[dst_p]
        float *BProw = p.BP + (bt0 + lx - 1) * p.Rb;
        double smax = -INFINITY;
        for (int r = tid; r < Pr; r += kChainThreads) {
            const int2 meta = g.pair_meta[r];
            const double lb = (meta.x >= 0 && g.end_lin[meta.x] > 0.f) ? log((double)g.end_lin[meta.x]) : -INFINITY;
            BPst[Pr + r] = lb;
            smax = fmax(smax, lb);
        }
        const double M = block_max_d(smax, red, tid);
        for (int r = tid; r < Pr; r += kChainThreads) {
            const double lb = BPst[Pr + r];
            BProw[r] = lb > -INFINITY ? (float)(lb - M) : -INFINITY;
            Z[r] = lb > -INFINITY ? Dv[g.pair_meta[r].y & 0xffff] + lb : -INFINITY;
        }
    }
    __syncthreads();
    double zm = -INFINITY;
    float zs = 0.f;
    double Mprev = 0.0;
    const int sl0 = g.bwd.wave_off[wave], sl1 = g.bwd.wave_off[wave + 1];
    for (int i = 0; i < lx; ++i) {
        const int t = lx - 1 - i;
        const double *Zc = Z + (size_t)(i & 1) * Pr;
        double *Zn = Z + (size_t)((i + 1) & 1) * Pr;
        double *BPc = BPst + (size_t)(i & 1) * Pr;
        if (t >= 1) load_drow(p, bt0 + t - 1, Dv, tid);      // (its last readers are behind the previous iteration's closing barrier)
        if (i > 0) {  // log b_{t+1}[dst_p], staged by the previous iteration -> BP[b][t]
            const double *BPp = BPst + (size_t)((i - 1) & 1) * Pr;
            float *BProw = p.BP + (bt0 + t) * p.Rb;
            for (int r = tid; r < Pr; r += kChainThreads) BProw[r] = BPp[r] > -INFINITY ? (float)(BPp[r] - Mprev) : -INFINITY;
        }
        for (int r = tid; r < Pr; r += kChainThreads) BPc[r] = -INFINITY;   // (pairs into states without a backward row)
        __syncthreads();
        double smax = -INFINITY;
        for (int ii = sl0; ii < sl1; ++ii) {
            const int j = __builtin_amdgcn_readfirstlane(g.bwd.wave_slices[ii]);
            const int o = __builtin_amdgcn_readfirstlane(g.bwd.slice_off[j]);
            const int w2 = __builtin_amdgcn_readfirstlane(g.bwd.slice_w2[j]);
            const int4 meta = g.bwd_row_meta[j * kWave + lane];  // {state, #pairs into it, first pair, its label}
            const int s2 = meta.x;
            if (s2 < 0) continue;
            const double lb = ell_row_lse(g.bwd.arcs + o + lane, w2, Zc);
            if (t == 0) { if (g.start_lin[s2] > 0.f) lse_add(zm, zs, log((double)g.start_lin[s2]) + lb); }
            else {
                smax = fmax(smax, lb);
                if (meta.y == 1) BPc[meta.z] = lb;
                else for (int pi = g.st_pair_off[s2]; pi < g.st_pair_off[s2 + 1]; ++pi) BPc[g.st_pairs[pi]] = lb;
            }
        }
        if (t >= 1) {
            Mprev = block_max_d(smax, red, tid);            // (barriers: BPc complete)
            for (int r = tid; r < Pr; r += kChainThreads) Zn[r] = BPc[r] > -INFINITY ? Dv[g.pair_meta[r].y & 0xffff] + BPc[r] : -INFINITY;
        }
        __syncthreads();
    }
    const double lz = block_lse(zm, zs, red, tid);
    if (tid == 0) p.cost_beta[b] = (float)(lz + off);
}

// Forward and backward recursion must arrive at the same log Z.  They are two independent computations over the same paths, and the
// one failure the scaled fp32 vectors cannot see by themselves -- a path that is 2^-146 below the frame's best at SOME frame and the
// dominant one in the end is lost for good (see den_forward_robust) -- shows up here, because the two directions lose different paths:
// an utterance whose two sums differ by more than kDenCheckTol is handed to the log-domain fallback (round 5; tests/test_gpu_fuzz.py found
// such utterances with costs off by 5 - 40 nats and NaN gradients, unflagged).  parts = 1: register-resident kernels, compared in fp64
// from the raw sums and exponents (the costs themselves are fp32: ulp 5e-4 at 4 000 nats); 0: the other families' fp32 costs.
constexpr double kDenCheckTol = 1e-3;
__global__ __launch_bounds__(256) void crf_den_check_kernel(LossParams p, int parts) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= p.B) return;
    bool bad;
    if (parts) {
        float zb = 0.f;
        const int nk = p.res == 2 ? 1 : p.g.res.K;
        for (int k = 0; k < nk; ++k) zb += p.cb_part[(size_t)b * kResMaxK + k];
        const float zs = p.den_zs[b];
        const double la = zs > 0.f ? log((double)zs) - (double)p.den_ez[b] * 0.6931471805599453 : -INFINITY;
        const double lb = zb > 0.f ? log((double)zb) - (double)p.cb_F[b] * 0.6931471805599453 : -INFINITY;
        bad = !(fabs(la - lb) <= kDenCheckTol);            // (-inf on both sides: NaN -> bad; the kernels have flagged those themselves)
    } else {
        const double a = (double)p.cost_alpha[b], c = (double)p.cost_beta[b];
        bad = !(fabs(a - c) <= kDenCheckTol + 3e-5 * fabs(a));
    }
    if (bad) p.redo[b] = 1;
}

template <bool GV>
__global__ __launch_bounds__(kChainThreads) void crf_robust_den_kernel(LossParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int b = (int)blockIdx.x < p.B ? (int)blockIdx.x : (int)blockIdx.x - p.B;
    if (!(p.redo[b] | p.redo[p.B + b])) return;
    if ((int)blockIdx.x < p.B) den_forward_robust<GV>(p, b, lds);
    else den_backward_robust<GV>(p, b, lds);
}

// grad rows of the redone utterances: gamma_den[t][v] = softmax_v( d_t[v] + ln sum_{p: lab_p = v} Q_t[p] * BP_t[p] ) in fp64,
// combined with the numerator half exactly as crf_grad_kernel does.  grid (frames-in-parallel, B); rows gathered from L2.
// LDS: csum[NC] | gl (double)[Vp] | gc[Vp] | red (double)[4]
__device__ __forceinline__ void finalize_body(const LossParams &p);
__device__ __forceinline__ void robust_grad_body(const LossParams &p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const GraphDev &g = p.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int b = blockIdx.y, V = p.V, Vp = rup64(V), NC = g.NC;
    if (!(p.redo[b] | p.redo[p.B + b])) return;
    const int lx = p.lx[b];
    float *csum = lds;
    double *gl = (double *)(csum + rup64(NC));
    float *gc = (float *)(gl + Vp);
    double *red = (double *)(gc + Vp);
    const int64_t bt0 = (int64_t)b * p.T;
    const bool do_ctc = p.c_ctc != 0.f;
    double zc = 0.0;
    int ezc = 0, Sx = 0;
    const int *ul = nullptr;
    if (do_ctc) { zc = ctc_zc_for_grad(p, b); ezc = p.ctc_ez[b]; Sx = 2 * p.ly[b] + 1; ul = p.labels + p.lab_off[b]; }
    // an utterance whose numerator the log-domain kernels have redone already (pass 1, beside the recursions): CA / CB hold logarithms
    const bool logdom = do_ctc && p.ctc_logdom[b] != 0;
    const double lzc = logdom ? p.ctc_zc[b] : 0.0;
    const bool logok = logdom && lzc > -INFINITY && lzc < INFINITY && !p.invalid[b];
    if (logdom) zc = logok ? 1.0 : 0.0;                      // (only its sign is used below)
    const double invc = zc > 0.0 ? 1.0 / zc : 0.0;
    auto bmax = [&](double v) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
        __syncthreads();
        if (lane == 0) red[tid >> 6] = v;
        __syncthreads();
        return fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    };
    auto bsum = [&](double v) {
        v = wave_sum_d(v);
        __syncthreads();
        if (lane == 0) red[tid >> 6] = v;
        __syncthreads();
        return red[0] + red[1] + red[2] + red[3];
    };
    for (int t = blockIdx.x; t < p.T; t += gridDim.x) {
        float *row = p.grad + (bt0 + t) * V;
        if (t >= lx) {
            for (int v = tid; v < V; v += kGradThreads) row[v] = 0.f;
            continue;
        }
        const float *Qr = p.Q + (bt0 + t) * p.Rq, *Br = p.BP + (bt0 + t) * p.Rb;   // ln q_t[p], ln b_{t+1}[dst_p] relative to the frame's maxima
        for (int c = tid; c < NC; c += kGradThreads) {
            double m = -INFINITY;
            float sm = 0.f;
            for (int j = g.chunk_off[c]; j < g.chunk_off[c + 1]; ++j) { const int r = g.perm[j]; lse_add(m, sm, (double)Qr[r] + (double)Br[r]); }
            csum[c] = (float)lse_value(m, sm);              // (<= 0 up to the rows' rounding: fp32 keeps 1e-7 absolute near 0, where it matters)
        }
        for (int v = tid; v < V; v += kGradThreads) gc[v] = 0.f;
        __syncthreads();
        const double mx = (double)p.mx[bt0 + t];
        double lmax = -INFINITY;
        for (int v = tid; v < V; v += kGradThreads) {
            double m = -INFINITY;
            float sm = 0.f;
            if (v <= g.max_label)
                for (int c = g.lab_chunk_off[v]; c < g.lab_chunk_off[v + 1]; ++c) lse_add(m, sm, (double)csum[c]);
            const double ls = lse_value(m, sm);
            const double l = ls > -INFINITY ? ((double)ld_x(p, (bt0 + t) * V + v) - mx) + ls : -INFINITY;
            gl[v] = l;
            lmax = fmax(lmax, l);
        }
        const double M = bmax(lmax);
        double part = 0.0;
        for (int v = tid; v < V; v += kGradThreads) {
            const double u = M > -INFINITY && gl[v] > -INFINITY ? exp(gl[v] - M) : 0.0;
            gl[v] = u;
            part += u;
        }
        const double nrm = bsum(part);
        if (do_ctc && zc > 0.0) {
            const double *Ar = p.CA + (bt0 + t) * p.Sc, *Bx = p.CB + (bt0 + t) * p.Sc;
            const double fc = logdom ? 0.0 : ctc_frame_factor(p, b, bt0 + t, invc, ezc);
            float blank = 0.f;
            for (int s = tid; s < Sx; s += kGradThreads) {
                const float pr = logdom ? (float)exp(Ar[s] + Bx[s] - lzc) : (float)(Ar[s] * Bx[s] * fc);
                if (s & 1) atomicAdd(&gc[ul[s >> 1]], pr);
                else blank += pr;
            }
            blank = wave_sum(blank);
            if (lane == 0) atomicAdd(&gc[0], blank);
        }
        __syncthreads();
        for (int v = tid; v < V; v += kGradThreads) {
            float o = p.c_den * (nrm > 0.0 ? (float)(gl[v] / nrm) : 0.f);
            if (do_ctc) o -= p.c_ctc * gc[v];
            if (do_ctc && p.fused)   // log_softmax's backward: - softmax(x)[v] * sum_v d loss / d logp[v]
                o -= (p.c_den - (zc > 0.0 ? p.c_ctc : 0.f)) * __expf(ld_x(p, (bt0 + t) * V + v) - (float)mx) * p.inv_s[bt0 + t];
            row[v] = o;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// Numerator fallback (LossParams::redo_ctc): the CTC recursions of the marked utterances in the LOG domain, fp64 -- the arithmetic
// of the reference's numerator (gpu_ctc_kernels.h:87-458 works on log-probabilities with log_plus) at double width.  Block
// x < B: forward chain of utterance x, log alpha_t[s] (emission included) into the CA rows and log p(labels | x) into the
// utterance's cost; else the backward chain, log beta_t[s] (emission excluded) into the CB rows.  One barrier per frame, the
// next frame's emissions requested a frame ahead.  Unmarked utterances leave at once.
// LDS: A[2][Sxp] (double) | red[16] (double) | lab[Sxp] (int)   (same carve as the scaled chains)
// ---------------------------------------------------------------------------------------------
// log(e^a + e^b + e^c): the maximum in fp64, the correction log(1 + ...) in [0, ln 3] with the hardware's fp32 exp / log (absolute
// error ~1e-7 per step; what all states of a frame have in common cancels in the posteriors, the rest is a random walk of ~5e-6
// over 3 000 frames) -- a software fp64 exp / log made the chain 1.8 us per frame, 2.7 ms for T = 1500
__device__ __forceinline__ double lse3(double a, double b, double c) {
    const double m = fmax(a, fmax(b, c));
    if (!(m > -INFINITY)) return -INFINITY;
    const float s = __expf((float)(a - m)) + __expf((float)(b - m)) + __expf((float)(c - m));
    return m + (double)__logf(s);
}
// NR = ctc states per thread actually needed (1, 2, 4, 8 <- the batch's longest label sequence), as for the scaled chains: the
// predicated-off iterations of a fixed NR = 8 are most of a frame's instructions for ordinary label lengths
template <int NR>
__global__ __launch_bounds__(kCtcThreads) void crf_robust_ctc_kernel(LossParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const bool fwd = (int)blockIdx.x < p.B;
    const int b = fwd ? (int)blockIdx.x : (int)blockIdx.x - p.B;
    const int redo = p.redo_ctc[b], done = p.ctc_logdom[b];
    if (!redo || (done != 0 && done != p.ctc_pass)) return;   // not marked, or redone by an earlier pass of this call
    const int tid = threadIdx.x;
    const int V = p.V, lx = p.lx[b], L = p.ly[b], Sx = 2 * L + 1, Sxp = rup64(Sx);
    const CtcLds c = ctc_carve(lds, Sxp);
    double *A = c.A;
    const int *lab = c.lab;
    const int64_t bt0 = (int64_t)b * p.T;
    if (!ctc_setup(p, b, c, L, lx, tid)) return;            // (not a valid label sequence: the scaled chain has said so)
    if (fwd && tid == 0 && p.ctc_seen) __hip_atomic_store(p.ctc_seen, p.call_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    int mylab[NR];
    bool skip[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int s = tid + i * kCtcThreads;
        mylab[i] = s < Sx ? lab[s] : 0;
        skip[i] = fwd ? (s < Sx && s >= 2 && mylab[i] != 0 && mylab[i] != lab[s - 2])
                      : ((s + 2 < Sx) && lab[s + 2] != 0 && lab[s + 2] != mylab[i]);
    }
    // log p_t[l'_s] = (x_t[l] - max_t) + offset_t (offset = the row maximum, or -log sum exp(x - max) with the fused log_softmax).
    // Emissions are fetched in BATCHES of kCtcPF frames into two alternating register sets, as in the scaled chains: one wait on
    // global memory per batch, for loads issued a batch ago (a wait per frame ties the frame to the memory latency: 1.2 us), and the
    // frame barrier orders LDS only.
    float lr[2][kCtcPF][NR];
    double of[2][kCtcPF];
    auto fetch = [&](auto SET, const int t0, const int dt) __attribute__((always_inline)) {   // frames t0, t0 + dt, ...
        constexpr int st = decltype(SET)::value;
#pragma unroll
        for (int f = 0; f < kCtcPF; ++f) {
            const int t = t0 + f * dt;
            if (t >= 0 && t < lx) {
                of[st][f] = (double)p.moff[bt0 + t] - (double)p.mx[bt0 + t];
#pragma unroll
                for (int i = 0; i < NR; ++i) lr[st][f][i] = (tid + i * kCtcThreads < Sx) ? ld_x(p, (bt0 + t) * V + mylab[i]) : 0.f;
            }
        }
    };
    auto lp0 = [&](int t, int i) -> double {   // (set-up frames only)
        return ((double)ld_x(p, (bt0 + t) * V + mylab[i]) - (double)p.mx[bt0 + t]) + (double)p.moff[bt0 + t];
    };
    if (fwd) {
        double *CArow = p.CA + bt0 * p.Sc;
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int s = tid + i * kCtcThreads;
            if (s < Sxp) {
                const double v = (s < 2 && s < Sx) ? lp0(0, i) : -INFINITY;
                A[s] = v;
                A[Sxp + s] = -INFINITY;
                if (s < Sx) CArow[s] = v;
            }
        }
        __syncthreads();
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
        auto batch = [&](auto SET, const int tb) __attribute__((always_inline)) {
            constexpr int st = decltype(SET)::value;
            fetch(std::integral_constant<int, 1 - st>{}, tb + kCtcPF, 1);   // the next batch, requested before this one is used
#pragma unroll
            for (int f = 0; f < kCtcPF; ++f) {
                const int t = tb + f;
                if (t < lx) {
                    const double *Ac = A + ((t - 1) & 1) * Sxp;
                    double *An = A + (t & 1) * Sxp;
                    double *row = p.CA + (bt0 + t) * p.Sc;
#pragma unroll
                    for (int i = 0; i < NR; ++i) {
                        const int s = tid + i * kCtcThreads;
                        if (s < Sx) {
                            const double v = lse3(Ac[s], s >= 1 ? Ac[s - 1] : -INFINITY, skip[i] ? Ac[s - 2] : -INFINITY) + ((double)lr[st][f][i] + of[st][f]);
                            An[s] = v;
                            row[s] = v;
                        }
                    }
                    sync_lds();
                }
            }
        };
        fetch(I0{}, 1, 1);
        for (int tb = 1; tb < lx; tb += 2 * kCtcPF) {
            batch(I0{}, tb);
            if (tb + kCtcPF < lx) batch(I1{}, tb + kCtcPF);
        }
        if (tid == 0) {
            const double *Af = A + ((lx - 1) & 1) * Sxp;
            const double lz = lse3(Af[Sx - 1], Sx > 1 ? Af[Sx - 2] : -INFINITY, -INFINITY);
            const bool ok = lz > -INFINITY && lz < INFINITY;
            p.ctc_zc[b] = lz;                    // (from here on the LOG of the partition sum)
            p.cost_ctc[b] = ok ? (float)lz : 0.f;
            p.invalid[b] = ok ? 0 : 1;
            p.ctc_logdom[b] = p.ctc_pass;        // (the backward workgroup of this pass may start later: it lets its own pass through)
        }
    } else {
        // Y_t[s] = log(e_t[l'_s] Bx_t[s]) in LDS; Bx_t itself goes to the CB rows
        double *CBrow = p.CB + (bt0 + lx - 1) * p.Sc;
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int s = tid + i * kCtcThreads;
            if (s < Sxp) {
                const double bx = (s < Sx && s >= Sx - 2) ? 0.0 : -INFINITY;
                A[s] = s < Sx ? bx + lp0(lx - 1, i) : -INFINITY;
                A[Sxp + s] = -INFINITY;
                if (s < Sx) CBrow[s] = bx;
            }
        }
        __syncthreads();
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
        auto batch = [&](auto SET, const int kb) __attribute__((always_inline)) {
            constexpr int st = decltype(SET)::value;
            fetch(std::integral_constant<int, 1 - st>{}, lx - 1 - (kb + kCtcPF), -1);
#pragma unroll
            for (int f = 0; f < kCtcPF; ++f) {
                const int k = kb + f;
                if (k < lx) {
                    const int t = lx - 1 - k;
                    const double *Yc = A + ((k - 1) & 1) * Sxp;
                    double *Yn = A + (k & 1) * Sxp;
                    double *row = p.CB + (bt0 + t) * p.Sc;
#pragma unroll
                    for (int i = 0; i < NR; ++i) {
                        const int s = tid + i * kCtcThreads;
                        if (s < Sx) {
                            const double bx = lse3(Yc[s], s + 1 < Sx ? Yc[s + 1] : -INFINITY, skip[i] ? Yc[s + 2] : -INFINITY);
                            row[s] = bx;
                            Yn[s] = bx + ((double)lr[st][f][i] + of[st][f]);
                        }
                    }
                    sync_lds();
                }
            }
        };
        fetch(I0{}, lx - 2, -1);
        for (int kb = 1; kb < lx; kb += 2 * kCtcPF) {
            batch(I0{}, kb);
            if (kb + kCtcPF < lx) batch(I1{}, kb + kCtcPF);
        }
    }
}
// ... and the posteriors of the marked frames (all frames of an utterance redone whole), subtracted from the rows the grad pass
// wrote without them: grad[b][t][v] -= c_ctc * sum_{s: l'_s = v} exp(log alpha_t[s] + log beta_t[s] - log Z).  grid (T / 16, B).
__global__ __launch_bounds__(kGradThreads) void crf_robust_ctc_fix_kernel(LossParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int b = blockIdx.y, V = p.V;
    const int redo = p.redo_ctc[b];
    if (!redo || p.ctc_logdom[b] != p.ctc_pass) return;      // not marked, or fixed by an earlier pass (or not a valid label sequence)
    const double lz = p.ctc_zc[b];
    if (!(lz > -INFINITY && lz < INFINITY) || p.invalid[b]) return;   // no alignment at all: the numerator contributes nothing
    const int lx = p.lx[b], Sx = 2 * p.ly[b] + 1;
    const int *ul = p.labels + p.lab_off[b];
    const int64_t bt0 = (int64_t)b * p.T;
    float *gc = lds;                                          // [Vp]
    const int t0 = blockIdx.x * kGCFrames, t1 = min(t0 + kGCFrames, lx);
    for (int t = t0; t < t1; ++t) {
        if (redo != 2 && !p.ctc_bad[bt0 + t]) continue;      // (uniform)
        for (int v = tid; v < V; v += kGradThreads) gc[v] = 0.f;
        __syncthreads();
        const double *Ar = p.CA + (bt0 + t) * p.Sc, *Bx = p.CB + (bt0 + t) * p.Sc;
        float blank = 0.f;
        for (int s = tid; s < Sx; s += kGradThreads) {
            const float pr = (float)exp(Ar[s] + Bx[s] - lz);
            if (s & 1) atomicAdd(&gc[ul[s >> 1]], pr);
            else blank += pr;
        }
        blank = wave_sum(blank);
        if (lane == 0) atomicAdd(&gc[0], blank);
        __syncthreads();
        float *row = p.grad + (bt0 + t) * V;
        // (an utterance redone whole went through the grad pass as "no numerator": with the fused log_softmax its softmax term
        // was taken with the factor c_den instead of c_den - c_ctc)
        const float ks = (redo == 2 && p.fused) ? p.c_ctc * pow2f(-kEpExp) * p.inv_s[bt0 + t] : 0.f;
        for (int v = tid; v < V; v += kGradThreads) row[v] += ks * p.ep[(bt0 + t) * V + v] - p.c_ctc * gc[v];
        __syncthreads();
    }
}

// loss = sum_b(c_den*logZ_b - c_ctc*logp_b); copies the per-utterance costs out (one workgroup of 256 threads)
__device__ __forceinline__ void finalize_body(const LossParams &p) {
    __shared__ double red[4];
    __shared__ int nfall[2];
    const int tid = threadIdx.x;
    double part = 0.0;
    if (tid < 2) nfall[tid] = 0;
    __syncthreads();
    for (int b = tid; b < p.B; b += 256) {
        double c = 0.0;
        // utterances that were redone by a fallback (denominator: log-shifted recursions; numerator: log-domain chains) -- crf_last_fallback_counts
        if (p.c_den != 0.f && (p.redo[b] | p.redo[p.B + b])) atomicAdd(&nfall[0], 1);
        if (p.c_ctc != 0.f && p.redo_ctc[b]) atomicAdd(&nfall[1], 1);
        if (p.c_den != 0.f) {
            if (p.res && !(p.redo[b] | p.redo[p.B + b])) {  // backward partition sum = sum of the K per-CU partials (redone utterances: written by the robust kernel)
                float zb = 0.f;
                const int nk = p.res == 2 ? 1 : p.g.res.K;
                for (int k = 0; k < nk; ++k) zb += p.cb_part[(size_t)b * kResMaxK + k];
                p.cost_beta[b] = to_log(zb, p.cb_F[b], p.cb_mxs[b]);
            }
            c += (double)p.c_den * (double)p.cost_alpha[b];
            if (p.out_den) p.out_den[b] = p.cost_alpha[b];
            if (p.out_beta) p.out_beta[b] = p.cost_beta[b];
        }
        if (p.c_ctc != 0.f) {
            c -= (double)p.c_ctc * (double)p.cost_ctc[b];
            if (p.out_ctc) p.out_ctc[b] = p.cost_ctc[b];
            if (p.out_invalid) p.out_invalid[b] = p.invalid[b];
        }
        part += c;
    }
    part = wave_sum_d(part);
    if ((tid & 63) == 0) red[tid >> 6] = part;
    __syncthreads();
    if (tid == 0) p.loss[0] = (p.res && *p.err) ? __builtin_nanf("") : (float)(red[0] + red[1] + red[2] + red[3]);
    if (tid < 2) p.err[kFlagFallback + tid] = nfall[tid];
}
__global__ __launch_bounds__(256) void crf_finalize_kernel(LossParams p) { finalize_body(p); }
// The last launch of a call with a denominator: the grad rows of the redone utterances (none, as a rule: every workgroup leaves at once) and, in
// workgroup (0, 0), the call's sums -- they need nothing of what the other workgroups write, and as a launch of their own they were one more
// dispatch (~5 us + the gap in front of it) behind the end of the grad pass (round 5; p.fin_fold, crf_loss_fwd_bwd)
static_assert(kGradThreads == 256, "finalize_body is written for 256 threads");
__global__ __launch_bounds__(kGradThreads) void crf_robust_grad_kernel(LossParams p) {
    robust_grad_body(p);
    if (p.fin_fold && blockIdx.x == 0 && blockIdx.y == 0) { __syncthreads(); finalize_body(p); }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
struct WsLayout {
    int64_t off_ep, off_mx, off_moff, off_invs, off_Q, off_BP, off_EQ, off_EB, off_CA, off_CB, off_ECA, off_ECB, off_pb, off_cbad, off_xch, off_row0, total;
    int64_t xch_bytes;
    int64_t Rq, Rb;
    bool res, gv, fac;
    int p2mode;                  // two utterances per workgroup: 0 no, 1 on the main factored layout, 2 on HostGraph::facp (pair2_mode)
    bool bat; int UL; int64_t Bp, off_ept, off_Af, off_Zb, off_bsm;   // utterance-minor layout (large graphs)
    bool gv_robust;              // the robust fallback kernels keep their vectors in global memory too
    int64_t off_gvec, off_state, state_stride, gvec_stride, off_dump, dump_stride;
};
static int64_t al(int64_t x) { return (x + 255) & ~(int64_t)255; }

// the register-resident kernels are used whenever the graph fits them (res.K > 0) and V fits their
// emission-row prefetch; CRF_NO_RESIDENT=1 at graph creation forces the streaming kernels
static size_t res_lds_bytes(const HostGraph *h, int V, int dir, int rows_cu_max);
static bool use_resident(const HostGraph *h, int64_t V) {
    // (the two fixed 64 KiB state-vector buffers leave 32 KiB for the CU's row table and the emission rows: a K = 1 layout
    // with ~6 k+ rows, or a call with far more classes than the den_lm has labels, takes the next kernel family instead)
    return h && h->dev.res.K > 0 && V <= (int64_t)kEpRegsR * kResThreads && h->dev.res.f.G <= kResGmax && h->dev.res.b.G <= kResGmax &&
           std::max(res_lds_bytes(h, (int)V, 0, h->res_rows_cu_f), res_lds_bytes(h, (int)V, 1, h->res_rows_cu_b)) <= (size_t)160 * 1024;
}

// the factored layout (one CU per recursion) is preferred whenever the graph has it; CRF_NO_FACTORED=1 at
// graph creation keeps the generic resident kernels
static size_t fac_lds_bytes(const HostGraph *h, int V, int dir);
static bool use_factored(const HostGraph *h, int64_t V) {
    // (the layout was budgeted for the graph's own label range, res_layout.cpp: a call with far more classes than the den_lm
    // has labels may not fit the LDS any more and takes the next kernel family)
    return h && h->dev.fac.ok && V <= (int64_t)kEpRegsR * kResThreads &&
           std::max(fac_lds_bytes(h, (int)V, 0), fac_lds_bytes(h, (int)V, 1)) <= (size_t)160 * 1024;
}

static int pair2_mode(const HostGraph *h, int64_t B, int64_t V);

// LDS of the robust fallback kernels (the larger of the two directions)
static size_t robust_lds_bytes(const HostGraph *h, int V, bool gv) {
    const size_t tail = (size_t)rup64(V) * 8 + 2 * kChainWaves * 4 + 16 * 8 + 64;
    if (gv) return tail;
    return std::max((size_t)2 * rup64(h->dev.S) + h->dev.Pr, (size_t)4 * h->dev.Pr) * 8 + tail;   // (fp64 logarithms: den_forward_robust / den_backward_robust)
}

static WsLayout ws_layout(const HostGraph *h, int64_t B, int64_t T, int64_t V, int64_t Sc) {
    WsLayout w{};
    int64_t o = 0;
    w.fac = use_factored(h, V);
    w.res = w.fac || use_resident(h, V);   // "res": register-resident kernels of either layout
    // graphs that fit neither register-resident layout take the utterance-minor kernels (CRF_NO_BATCH=1: the streaming
    // kernels instead; CRF_FORCE_BATCH=1: every graph, for tests)
    const bool force_bat = opt_on(kOpt_force_batch);
    w.bat = h && h->dev.bat.ok && (force_bat || (!w.res && !opt_on(kOpt_no_batch)));
    if (w.bat) w.res = w.fac = false;
    // utterances per group: as wide as the batch allows (an arc is fetched once per group); 32 instead of 64 when a group's state
    // vector [S][64] would not stay in an XCD's 4 MiB L2 next to the arc stream (> ~2.25 MB) -- but never narrower than 32 for
    // that reason: every group reads the whole arc stream again and narrower gathers are partial lines (measured: S = 16 385,
    // B = 64: UL 64 / 32 / 16 -> 31.6 / 29.7 / 34.5 ms; config #5 with B = 64, where not even [S][8] fits: UL 64 / 32 / 8 -> 379 /
    // 271 / 569 ms).  CRF_BAT_UL overrides, for sweeps.
    w.UL = B > 32 ? 64 : B > 16 ? 32 : B > 8 ? 16 : 8;
    if (h && w.UL == 64 && std::max<int64_t>(h->dev.S, h->dev.P) * w.UL * 4 > (int64_t)(2.25 * 1024 * 1024)) w.UL = 32;
    { const int v = opt(kOpt_bat_ul, 0); if (v == 8 || v == 16 || v == 32 || v == 64) w.UL = v; }
    w.Bp = (B + w.UL - 1) / w.UL * w.UL;
    // (rows are at least Pr wide: the robust fallback stores them in pair order, whatever layout the fast kernels use)
    w.p2mode = w.fac ? pair2_mode(h, B, V) : 0;
    const FacDev *FX = w.fac ? (w.p2mode == 2 ? &h->facp : &h->dev.fac) : nullptr;   // the factored layout this call works with
    w.Rq = h ? std::max<int64_t>(w.fac ? FX->Rq : w.res ? h->dev.res.f.R : h->dev.Pr, h->dev.Pr) : 0;
    w.Rb = h ? std::max<int64_t>(w.fac ? FX->Rbp : w.res ? h->dev.res.b.R : h->dev.Pr, h->dev.Pr) : 0;
    w.off_ep = o; o = al(o + B * T * V * 4);
    w.off_mx = o; o = al(o + B * T * 4);
    w.off_moff = o; o = al(o + B * T * 4);   // fused log_softmax only (crf_loss_fwd_bwd_logits)
    w.off_invs = o; o = al(o + B * T * 4);
    const int64_t qb = w.bat ? T * (int64_t)h->dev.P * w.Bp : 0;   // utterance-minor rows [T][P][Bp]
    w.off_Q = o; o = al(o + std::max(B * T * w.Rq, qb) * 4);
    w.off_BP = o; o = al(o + std::max(B * T * w.Rb, qb) * 4);
    w.off_EQ = o; o = al(o + B * T * 4);
    w.off_EB = o; o = al(o + B * T * 4);
    w.off_CA = o; o = al(o + B * T * Sc * 8);
    w.off_CB = o; o = al(o + B * T * Sc * 8);
    w.off_ECA = o; o = al(o + B * T * 4);
    w.off_ECB = o; o = al(o + B * T * 4);
    w.off_pb = o; o = al(o + 32 * B * 8);
    w.off_cbad = o; o = al(o + B * T * 4);   // frames of the numerator marked for the log-domain fallback
    // tagged granules [2 slots] of both directions, then one XCD-id word per CU of every recursion
    w.xch_bytes = (w.res && !w.fac && h->dev.res.K > 1) ? al((B * 2 * ((int64_t)h->dev.res.f.G + h->dev.res.b.G) + 2 * B * kResMaxK) * 8)
                : (w.fac && FX->K > 1) ? al((B * 2 * ((int64_t)FX->f.G + FX->b.G) + 2 * B * kResMaxK) * 8) : 0;
    w.off_xch = o; o = al(o + w.xch_bytes + 256 + 8 * B);   // granules | error word, start counter | per-utterance progress of the two den recursions
    w.off_row0 = o; o = al(o + (w.res ? B * w.Rb * 4 : 0));
    w.off_ept = o; o = al(o + (w.bat ? T * V * w.Bp * 4 : 0));
    w.off_Af = o; o = al(o + (w.bat ? 2 * ((int64_t)h->dev.S + (h->fb.ok ? h->fb.NU : 0)) * w.Bp * 4 : 0));   // (+ the U entries of factored streams)
    w.off_Zb = o; o = al(o + (w.bat ? 2 * (int64_t)h->dev.P * w.Bp * 4 : 0));
    w.off_bsm = o; o = al(o + (w.bat ? 12 * w.Bp * 4 : 0));        // mxf[3], mxb[3], Ef, Fb, zs, zb
    w.gv = h && !w.res && !w.bat && std::max((size_t)3 * rup64(h->dev.S), (size_t)4 * h->dev.Pr) * 4 + 2 * (size_t)rup64((int)V) * 4 + 1024 > 160 * 1024;
    w.gv_robust = h && robust_lds_bytes(h, (int)V, false) > 160 * 1024;
    // (floats per utterance: the streaming kernels' fp32 vectors, or the log-domain fallback's fp64 ones -- forward A[2][Sp] + Ql[Pr], backward Z[2][Pr] + BPst[2][Pr])
    w.gvec_stride = h ? std::max<int64_t>(3 * (int64_t)rup64(h->dev.S) + 5 * (int64_t)h->dev.Pr, 2 * (2 * (int64_t)rup64(h->dev.S) + 5 * (int64_t)h->dev.Pr)) : 0;
    w.off_gvec = o; o = al(o + ((w.gv || w.gv_robust) ? B * w.gvec_stride * 4 : 0));
    // factored recursions launched in segments park their state vector + exponent here: [2 dir][B][stride]
    w.state_stride = w.fac ? rup64(std::max(FX->f.G, FX->b.G)) + 64 : 0;
    w.off_state = o; o = al(o + 2 * B * w.state_stride * 4);
    // two utterances per workgroup: one dump row per (direction, pair) for the row stores of an utterance that has ended
    w.dump_stride = w.fac ? al(std::max(w.Rq, w.Rb)) : 0;
    w.off_dump = o; o = al(o + (w.fac ? 2 * ((B + 1) / 2) * w.dump_stride * 4 : 0));
    w.total = o;
    return w;
}

static size_t res_lds_bytes(const HostGraph *h, int V, int dir, int rows_cu_max) {
    const int G = dir == 0 ? h->dev.res.f.G : h->dev.res.b.G;
    (void)G;
    return (size_t)2 * kResXB + ((size_t)rows_cu_max + 2 * (size_t)rup64(V + 1) + 2 * kResWaves + 2 * kResWaves + 16) * sizeof(float);
}

static size_t chain_lds_bytes(const HostGraph *h, int V, int Sc, int role, bool gv = false) {
    const size_t tail = 2 * kChainWaves + 2 * kChainWaves + 16;  // wmax + 16 doubles + slack
    size_t fl;
    if (role == 0) fl = (gv ? 0 : (size_t)3 * rup64(h->dev.S)) + 2 * rup64(V) + tail;
    else if (role == 1) fl = (gv ? 0 : (size_t)4 * h->dev.Pr) + 2 * rup64(V) + tail;
    else fl = (size_t)2 * (2 * Sc + 3 * kChainWaves) + Sc + 16;  // doubles counted as 2 floats
    return fl * sizeof(float);
}

// ---------------------------------------------------------------------------------------------
// Per-(device, caller stream) context: ONE side stream, the fork/join events and a few flag words.
// A call needs two streams at most -- the caller's (denominator recursions: forward and backward are one grid) and
// one side stream (numerator recursions, then the grad pass that follows the recursions in stages).  Nothing is
// shared between two caller streams or two devices, so calls on different streams / from different host threads do
// not touch each other's events or counters; calls on ONE stream are ordered by the stream (the counters of call
// n+1 are cleared by its prep kernel, which runs after call n has joined everything back into that stream).
// ---------------------------------------------------------------------------------------------
constexpr int kMaxStages = 16;
constexpr int kFlagInts = 2048;
constexpr int kMaxDev = 64;
struct DevCtx {
    std::mutex mu;            // one call at a time is ENQUEUED through a context (host side only; nothing waits on the GPU)
    int dev = 0;
    hipStream_t owner{};
    hipStream_t side{};       // null: no side stream that runs beside the owner was found -> everything on the owner's stream
    hipStream_t aux{};        // a third stream that runs beside the owner's (numerator fallback chains beside the grad stages), or null
    hipEvent_t ev_a{}, ev_b{};
    int *seen = nullptr;      // pinned host word the log-domain numerator chains write the call number to (read without a sync: a hint)
    int call_id = 0;
    hipEvent_t fork{}, join{}, ev[kMaxStages]{}, evb[kMaxStages]{};
    int *flags = nullptr;     // fine-grained (uncached, cross-XCD coherent) words: [0] error word, [1] start counter, [16..32) stage counters
    bool warned = false;
    int reprobes = 0;         // probes for a side stream after the first one found none (loss_impl: at calls 256, 1 024, 4 096)
    int side_kind = 0, side_tries = 0;   // find_beside: what kind of stream the side stream is, how many candidates were probed
    char side_desc[96] = "none";
};
static std::mutex g_ctx_mu;
static std::vector<DevCtx *> g_ctxs;

// Do two streams run side by side?  HIP maps every stream of the process onto GPU_MAX_HW_QUEUES (default 4) hardware
// queues; two streams on one queue run their kernels one after the other.  Two single-wave kernels shake hands through
// fine-grained memory: each raises its flag and waits (bounded, ~5 ms) for the other's.  Both see the other only if
// they were resident at the same time.
__global__ void crf_probe_kernel(int *flags, int me, int other) {
    __hip_atomic_store(flags + me, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    int saw = 0;
    for (int spins = 0; spins < 2500 && !saw; ++spins) {   // ~2 us per spin: bounded at ~5 ms
        saw = __hip_atomic_load(flags + other, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (!saw) __builtin_amdgcn_s_sleep(64);
    }
    __hip_atomic_store(flags + 2 + me, saw ? 1 : 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// A stream that runs BESIDE `owner`, or null.  HIP (ROCclr) gives every stream one of GPU_MAX_HW_QUEUES (default 4) HSA queues PER
// PRIORITY -- a new stream takes the queue with the fewest users -- and a stream created with a CU mask gets a queue of its own.
// In a trainer process the pools are crowded before this library is loaded (torch's stream pool: 32 streams per priority as soon as
// c10d asks for one; RCCL's own), so the least-used queue may well be the owner's, and a candidate that is destroyed gives its
// slot back: the next one lands on the same queue again (rounds 2 - 3 probed six candidates that way and found all six behind the
// owner in every process that had initialised RCCL).  Hence: failed candidates stay alive until one passes -- each pushes the
// next one to another queue --, then the other priorities' pools (low first: the side stream's work is the filler, the owner's den
// grid is the critical path), then a CU-masked stream with every CU enabled.  Switch `side_kind` (1 plain, 2 high, 3 low, 4 masked)
// restricts the search to one kind.
enum { kSideNone = 0, kSidePlain, kSideHigh, kSideLow, kSideMask };
static const char *const kSideNames[] = {"none", "plain", "priority-high", "priority-low", "cu-mask"};
static hipStream_t make_candidate(int kind, int dev) {
    hipStream_t s{};
    hipError_t e = hipErrorInvalidValue;
    if (kind == kSidePlain) {
        e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    } else if (kind == kSideHigh || kind == kSideLow) {
        int least = 0, greatest = 0;   // (numerically lower = higher priority)
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest)
            e = hipStreamCreateWithPriority(&s, hipStreamNonBlocking, kind == kSideHigh ? greatest : least);
    } else if (kind == kSideMask) {
        int ncu = 0;
        if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && ncu > 0) {
            std::vector<uint32_t> mask((size_t)(ncu + 31) / 32, 0xffffffffu);
            if (ncu % 32) mask.back() = (1u << (ncu % 32)) - 1u;
            e = hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data());
        }
    }
    if (e != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return s;
}
static bool runs_beside(hipStream_t owner, hipStream_t cand, int *flags) {
    int res[4] = {0, 0, 0, 0};
    // (a first launch on a new stream may pay for its queue's set-up: get that out of the way, or the owner's probe kernel
    // gives up before the candidate's has started -- the kernel below shakes hands with itself)
    hipLaunchKernelGGL(crf_probe_kernel, dim3(1), dim3(1), 0, cand, flags + 8, 0, 0);
    bool ran = hipStreamSynchronize(cand) == hipSuccess && hipMemset(flags, 0, sizeof(res)) == hipSuccess;
    hipLaunchKernelGGL(crf_probe_kernel, dim3(1), dim3(1), 0, owner, flags, 0, 1);
    hipLaunchKernelGGL(crf_probe_kernel, dim3(1), dim3(1), 0, cand, flags, 1, 0);
    ran = ran && hipStreamSynchronize(cand) == hipSuccess && hipStreamSynchronize(owner) == hipSuccess &&
          hipMemcpy(res, flags, sizeof(res), hipMemcpyDeviceToHost) == hipSuccess;
    if (!ran) (void)hipGetLastError();
    return ran && res[2] == 1 && res[3] == 1;
}
static hipStream_t find_beside(hipStream_t owner, int *flags, int dev, int *kind_out, int *tries_out) {
    static const struct { int kind, count; } plan[] = {{kSidePlain, 8}, {kSideLow, 2}, {kSideHigh, 2}, {kSideMask, 1}};
    const int only = opt(kOpt_side_kind, 0);
    std::vector<hipStream_t> failed;
    hipStream_t found{};
    int tries = 0;
    for (const auto &ph : plan) {
        if (only > 0 && ph.kind != only) continue;
        // hipExtStreamCreateWithCUMask has no flags argument: the stream it makes is a BLOCKING one, i.e. it synchronises implicitly with the
        // legacy null stream -- beside the null stream (torch's default) the two probe kernels can never overlap, the candidate would only
        // cost its time-out.  (Beside any other owner it is a last resort with that caveat: null-stream work of the process orders with it.)
        if (ph.kind == kSideMask && owner == nullptr && only != kSideMask) continue;
        for (int i = 0; i < ph.count && !found; ++i) {
            hipStream_t cand = make_candidate(ph.kind, dev);
            if (!cand) break;
            ++tries;
            if (runs_beside(owner, cand, flags)) { found = cand; *kind_out = ph.kind; }
            else failed.push_back(cand);
        }
        if (found) break;
    }
    for (hipStream_t s : failed) (void)hipStreamDestroy(s);
    *tries_out = tries;
    if (!found) *kind_out = kSideNone;
    return found;
}

static int get_ctx(hipStream_t owner, DevCtx **out) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess || dev < 0 || dev >= kMaxDev) { set_error("hipGetDevice failed"); return CRF_ERR_HIP; }
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    for (DevCtx *c : g_ctxs)
        if (c->dev == dev && c->owner == owner) { *out = c; return CRF_OK; }
    DevCtx *c = new DevCtx();
    c->dev = dev; c->owner = owner;
    // Words that one kernel polls while another, on a different XCD, updates them must not be cached in the poller's
    // L2 (the per-XCD L2s are not coherent with each other): fine-grained device memory is uncached in L2.
    void *fl = nullptr;
    if (hipExtMallocWithFlags(&fl, kFlagInts * sizeof(int), hipDeviceMallocFinegrained) != hipSuccess) { (void)hipGetLastError(); fl = nullptr; }
    c->flags = (int *)fl;
    if ((e = hipEventCreateWithFlags(&c->fork, hipEventDisableTiming)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&c->join, hipEventDisableTiming)) != hipSuccess) {
        set_error(std::string("event create: ") + hipGetErrorString(e));
        delete c;
        return CRF_ERR_HIP;
    }
    for (int i = 0; i < kMaxStages; ++i) {
        (void)hipEventCreateWithFlags(&c->ev[i], hipEventDisableTiming);
        (void)hipEventCreateWithFlags(&c->evb[i], hipEventDisableTiming);
    }
    // The side stream: the first candidate that demonstrably runs beside the owner's stream (once per context; the owner's
    // stream is drained first so that both probe kernels start at once).  find_beside() keeps the candidates that failed
    // alive until one passes, and tries streams of other priorities (queue pools of their own) and a CU-masked stream (a
    // queue of its own) when no plain stream does.  Switch `no_side_stream` skips it (everything then runs on the caller's
    // stream, one kernel after the other).
    const bool want_side = !opt_on(kOpt_no_side_stream);
    const bool trust = opt_on(kOpt_trust_side);   // (counter passes of a profiler run one kernel at a time: the probe cannot succeed there)
    if (want_side && c->flags && !trust) {
        (void)hipStreamSynchronize(owner);
        c->side = find_beside(owner, c->flags, dev, &c->side_kind, &c->side_tries);
    } else if (want_side) {   // no fine-grained memory for the probe: take a stream on trust
        if (hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); c->side = nullptr; }
        c->side_kind = c->side ? kSidePlain : kSideNone;
    }
    if (want_side && !c->side && !c->warned) {
        fprintf(stderr, "[ctc_crf_hip] no stream of this process runs beside the caller's stream (%d candidates probed: plain, other "
                        "priorities, CU-masked): the loss runs its kernels one after the other on the caller's stream -- correct, but slower\n",
                c->side_tries);
        c->warned = true;
    }
    // A third stream for work that may take long beside the staged grad pass (the numerator's log-domain chains): it must not sit
    // behind the owner's stream (the den grid), which is all the probe asks; sharing a queue with the side stream only delays it.
    if (c->side && c->flags && !trust && !opt_on(kOpt_no_aux_stream) &&
        hipEventCreateWithFlags(&c->ev_a, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&c->ev_b, hipEventDisableTiming) == hipSuccess) {
        (void)hipStreamSynchronize(owner);
        int kind = 0, tries = 0;
        c->aux = find_beside(owner, c->flags, dev, &kind, &tries);
        void *hp = nullptr;
        if (c->aux && hipHostMalloc(&hp, 64, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess) { c->seen = (int *)hp; *c->seen = 0; }
        else (void)hipGetLastError();
    }
    snprintf(c->side_desc, sizeof(c->side_desc), "%s (candidate %d)%s", kSideNames[c->side_kind], c->side_tries, c->aux ? " + third stream" : "");
    g_ctxs.push_back(c);
    *out = c;
    return CRF_OK;
}

// Grids whose workgroups WAIT FOR EACH OTHER (layouts over K > 1 CUs per recursion: every workgroup of a launch spins on its
// peers) are sized to fill the device; two callers enqueueing such grids on different streams at once could each become
// partially resident and wait for peers that are not (the bounded spins would then time out into the error word).  They are
// chained per device: a caller's co-resident launch waits (stream-level, on an event) for the previous caller's.
struct CoresChain { std::mutex mu; hipEvent_t ev{}; bool have = false; hipStream_t last{}; };
static CoresChain g_cores[kMaxDev];
struct CoresGuard {
    CoresChain *c = nullptr; hipStream_t st{};
    CoresGuard(bool needed, int dev, hipStream_t s) : st(s) {
        if (!needed || dev < 0 || dev >= kMaxDev) return;
        c = &g_cores[dev];
        c->mu.lock();
        if (c->have && c->last != st) (void)hipStreamWaitEvent(st, c->ev, 0);
    }
    ~CoresGuard() {
        if (!c) return;
        if (!c->have) c->have = hipEventCreateWithFlags(&c->ev, hipEventDisableTiming) == hipSuccess;
        if (c->have) { (void)hipEventRecord(c->ev, st); c->last = st; }
        c->mu.unlock();
    }
};

// Dynamic LDS above 64 KiB must be opted into per kernel AND per device (hipFuncSetAttribute acts on the current
// device's copy of the function): high-water mark per device.
struct LdsMark { std::atomic<size_t> v[kMaxDev]; };
static int ensure_lds(const void *fn, size_t bytes, LdsMark &m, const char *what) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) { set_error("hipGetDevice failed"); return CRF_ERR_HIP; }
    if (bytes <= m.v[dev].load()) return CRF_OK;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) { set_error(std::string("hipFuncSetAttribute(") + what + "): " + hipGetErrorString(e)); return CRF_ERR_HIP; }
    m.v[dev] = bytes;
    return CRF_OK;
}

// optional per-kernel timing (crf_profile_enable / crf_profile_read)
struct Prof {
    bool on = false, have = false;
    hipEvent_t ev[16]{};  // start/stop per slot 0..6, [14],[15] whole call
    bool made = false, used[8]{};
};
static thread_local Prof g_prof;
static thread_local int g_call_streams = 1;          // crf_last_call_streams
static thread_local const char *g_side_desc = "none";   // crf_last_side_stream
static thread_local const int *g_last_err_word = nullptr;   // error word (+ kFlagFallback: fallback counts) of this thread's last call -- crf_last_fallback_counts
static thread_local const char *g_den_kernel = "";   // template instantiation of the denominator recursions' kernel in the last call (crf_last_den_kernel)
static void prof_mark(int slot, bool stop, hipStream_t st) {
    if (!g_prof.on) return;
    if (!g_prof.made) {
        for (auto &e : g_prof.ev) (void)hipEventCreate(&e);
        g_prof.made = true;
    }
    (void)hipEventRecord(g_prof.ev[2 * slot + (stop ? 1 : 0)], st);
    g_prof.used[slot] = true;
}

// streaming denominator recursions, forward + backward in one grid (profile slots 1 and 2 both time this launch)
template <bool GV>
static int launch_den_pair(const LossParams &p, size_t lds, hipStream_t st) {
    static LdsMark mark;
    int rc;
    if ((rc = ensure_lds((const void *)crf_den_pair_kernel<GV>, lds, mark, "den pair"))) return rc;
    g_den_kernel = GV ? "crf_den_pair_kernel<true>" : "crf_den_pair_kernel<false>";
    prof_mark(1, false, st); prof_mark(2, false, st);
    hipLaunchKernelGGL((crf_den_pair_kernel<GV>), dim3((unsigned)(2 * p.B)), dim3(kChainThreads), lds, st, p);
    prof_mark(1, true, st); prof_mark(2, true, st);
    hipError_t e;
    if ((e = hipGetLastError()) != hipSuccess) { set_error(std::string("crf_den_pair_kernel: ") + hipGetErrorString(e)); return CRF_ERR_HIP; }
    return CRF_OK;
}
// numerator chains, forward + backward in one grid (profile slots 3 and 4); states per thread from the longest label sequence
template <int NR>
static int launch_ctc_pair_nr(const LossParams &p, size_t lds, hipStream_t st) {
    static LdsMark mark;
    int rc;
    if ((rc = ensure_lds((const void *)crf_ctc_pair_kernel<NR>, lds, mark, "ctc pair"))) return rc;
    prof_mark(3, false, st); prof_mark(4, false, st);
    hipLaunchKernelGGL((crf_ctc_pair_kernel<NR>), dim3((unsigned)(2 * p.B)), dim3(kCtcThreads), lds, st, p);
    prof_mark(3, true, st); prof_mark(4, true, st);
    hipError_t e;
    if ((e = hipGetLastError()) != hipSuccess) { set_error(std::string("crf_ctc_pair_kernel: ") + hipGetErrorString(e)); return CRF_ERR_HIP; }
    return CRF_OK;
}
// Do the two numerator chains agree?  The posteriors of frame 0 -- A_0 is the chain's start, exact; Bx_0 the END of the backward chain; Z the
// end of the forward chain -- sum to one iff neither chain has lost the path mass on its way.  The rescaled fp64 rows end 2^-1074 below
// their frame's maximum: with network outputs a hundred nats apart per frame a chain can drop the states of the eventually dominant
// alignment, and every frame BEHIND the loss then looks consistent (its posteriors sum to one -- over the surviving alignments) while
// being wrong (tests/test_gpu_fuzz.py, round 5).  Such an utterance is redone WHOLE in the log domain (redo_ctc = 2), decided here, in
// front of the grad pass, so that every block of it sees the same verdict.  One workgroup per utterance.
__global__ __launch_bounds__(256) void crf_ctc_check_kernel(LossParams p) {
    __shared__ double red[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int lx = p.lx[b];
    if (lx <= 0 || p.invalid[b] || p.redo_ctc[b] == 2) return;
    const double zc = p.ctc_zc[b];
    if (!(zc > 0.0)) return;
    const int64_t bt0 = (int64_t)b * p.T;
    const int Sx = 2 * p.ly[b] + 1;
    const int e = p.ctc_ez[b] - p.ECA[bt0] - p.ECB[bt0];
    const double invc = 1.0 / zc;
    bool bad = e + ilogb(invc) > kCtcSafeExp;
    double part = 0.0;
    if (!bad) {
        const double fc = ldexp(invc, e);
        const double *Ar = p.CA + bt0 * p.Sc, *Br = p.CB + bt0 * p.Sc;
        for (int s = tid; s < Sx && s < 2; s += 256) part += Ar[s] * Br[s] * fc;   // (frame 0: only the first blank and the first label carry mass)
    }
    part = wave_sum_d(part);
    if ((tid & 63) == 0) red[tid >> 6] = part;
    __syncthreads();
    const double tot = red[0] + red[1] + red[2] + red[3];
    if (tid == 0 && (bad || !(fabs(tot - 1.0) <= 1e-3))) atomicMax(&p.redo_ctc[b], 2);
}

static int launch_ctc_pair(const LossParams &p, size_t lds, hipStream_t st, int64_t max_label_len) {
    const int64_t ni = (2 * max_label_len + 1 + kCtcThreads - 1) / kCtcThreads;
    int rc;
    if (ni <= 1) rc = launch_ctc_pair_nr<1>(p, lds, st);
    else if (ni <= 2) rc = launch_ctc_pair_nr<2>(p, lds, st);
    else if (ni <= 4) rc = launch_ctc_pair_nr<4>(p, lds, st);
    else rc = launch_ctc_pair_nr<kCtcRegs>(p, lds, st);
    if (rc) return rc;
    hipLaunchKernelGGL(crf_ctc_check_kernel, dim3((unsigned)p.B), dim3(256), 0, st, p);
    if (hipGetLastError() != hipSuccess) { set_error("crf_ctc_check_kernel"); return CRF_ERR_HIP; }
    return CRF_OK;
}

static ResParams res_params(const LossParams &lp, int dir, int b0) {
    const ResDev &R = lp.g.res;
    ResParams p{};
    p.L = dir == 0 ? R.f : R.b;
    p.K = R.K; p.B = lp.B; p.T = lp.T; p.V = lp.V; p.b0 = b0;
    p.rows_cu_max = dir == 0 ? lp.res_lds_rows_f : lp.res_lds_rows_b;
    p.Rout = dir == 0 ? lp.Rq : lp.Rb; p.Gf = R.f.G; p.Gb = R.b.G;
    p.lx = lp.lx; p.ep = lp.ep; p.mx = lp.moff;   // (the resident kernels use it for the log-likelihood offset only)
    p.Out = dir == 0 ? lp.Q : lp.BP; p.Eout = dir == 0 ? lp.EQ : lp.EB; p.Row0 = lp.Row0;
    p.xch = lp.xch; p.err = lp.err;
    p.x_start = R.x_start; p.x_end = R.x_end; p.den_zs = lp.den_zs; p.cost_alpha = lp.cost_alpha; p.den_ez = lp.den_ez;
    p.z_lab = R.z_lab; p.z_end = R.z_end; p.brow_start = R.brow_start; p.brow_end = R.brow_end; p.bcsr = R.bcsr;
    p.cb_part = lp.cb_part; p.cb_mxs = lp.cb_mxs; p.cb_F = lp.cb_F; p.redo = lp.redo;
    return p;
}
// generic register-resident recursions of the utterances [b0, b0 + nb): 2 * nb * K workgroups, forward first
static int launch_res_pair(const LossParams &lp, size_t lds, int b0, int nb, hipStream_t st) {
    static LdsMark mark;
    int rc;
    if ((rc = ensure_lds((const void *)crf_res_pair_kernel, lds, mark, "res pair"))) return rc;
    g_den_kernel = "crf_res_pair_kernel";
    const ResParams pf = res_params(lp, 0, b0), pb = res_params(lp, 1, b0);
    hipLaunchKernelGGL(crf_res_pair_kernel, dim3((unsigned)(2 * nb * lp.g.res.K)), dim3(kResThreads), lds, st, pf, pb);
    hipError_t e;
    if ((e = hipGetLastError()) != hipSuccess) { set_error(std::string("crf_res_pair_kernel: ") + hipGetErrorString(e)); return CRF_ERR_HIP; }
    return CRF_OK;
}

static size_t fac_lds_bytes(const HostGraph *h, int V, int dir) {
    const FacDev &F = h->dev.fac;
    const FacDirDev &L = dir == 0 ? F.f : F.b;
    const int nw = F.threads / kWave;
    const int trows = F.K > 1 ? std::max(L.cu_row[1] - L.cu_row[0], L.cu_row[2] - L.cu_row[1]) : L.R;   // two CUs: each holds its own rows' constants only
    const size_t table = F.rcl ? (size_t)(trows + 64) * (dir == 0 ? 8 : 16) : (size_t)L.R * 16;   // row constants (fac_chain_body)
    return (size_t)2 * rup64(L.G) * 4 + table +
           ((size_t)2 * rup64(V + 1) + 2 * nw + 2 * nw + 16) * sizeof(float);
}
// ... of the two-utterance kernels (fac_chain_body2): float2 state vectors and emission rows, the same row table
static size_t fac2u_lds_bytes(const FacDev &F, int V, int dir) {
    const FacDirDev &L = dir == 0 ? F.f : F.b;
    const size_t table = F.rcl ? (size_t)(L.R + 64) * (dir == 0 ? 8 : 16) : (dir == 1 ? (size_t)L.R * 8 : (size_t)0);
    return (size_t)2 * rup64(L.G) * 8 + table + (size_t)2 * rup64(V + 1) * 8 + 24 * 4 + (F.threads / kWave) * 8 + 64;
}
// Two utterances per workgroup?  -> 0: no; 1: on the main layout (its 768-thread geometries, as they are); 2: on the second layout
// (HostGraph::facp, 512 threads x 30 chunks: built beside a main layout of another geometry).  Switch fac_pair2 = 1 forces it for any
// batch, 0 forbids it.  One CU per recursion only, and both float2 vectors must fit the LDS.
//   main layout, 768 threads (round 3): the pair kernel's frame is 3.65 us against 1.98 us for one utterance -- 126 of a thread's 168
//   registers hold arcs, 26 - 39 dwords spill -- so it wins only where the one-utterance grid no longer fits the device at once
//   (2 B > CUs): B = 256: 10.1 against 10.5 ms per step; B = 128: 5.9 against 5.3; B = 96: 5.7 against 4.4.
//   second layout (round 5; 512 threads x 30 chunks, no spills inside the frame loop): the frame for two utterances is 3.25 us -- still
//   1.9 x the one-utterance frame: with eight waves the kernel is bound by what ONE wave can issue (30 chunks x 12 instructions + two
//   epilogues per slice, ~4.3 cycles each), not by the LDS whose gathers it halves.  Measured (metric graph, ms per step, this kernel /
//   the one-utterance kernel; profiles/round5_ab_two_utterances_512.txt): B = 64 5.08 / 2.81, B = 96 5.18 / 3.95, B = 128 5.39 / 4.92,
//   B = 192 7.83 / 8.77, B = 256 9.2 - 9.4 / 9.46 -- it pays where the one-utterance grid needs two rounds of the device: 2 B > CUs.
static int pair2_mode(const HostGraph *h, int64_t B, int64_t V) {
    const FacDev &F = h->dev.fac;
    const int sw = opt(kOpt_fac_pair2, -1);
    if (sw == 0 || !F.ok || F.K != 1) return 0;
    const int ncu = h->ncu > 0 ? h->ncu : 256;
    if (h->facp.ok && std::max(fac2u_lds_bytes(h->facp, (int)V, 0), fac2u_lds_bytes(h->facp, (int)V, 1)) <= (size_t)160 * 1024 &&
        V <= (int64_t)kEpRegsR * kResThreads) {
        if (sw == 1 || 2 * B > ncu) return 2;
        return 0;
    }
    if (F.threads != kFac3Threads) return 0;
    if (std::max(fac2u_lds_bytes(F, (int)V, 0), fac2u_lds_bytes(F, (int)V, 1)) > (size_t)160 * 1024) return 0;
    return (sw == 1 || 2 * B > ncu) ? 1 : 0;
}
#ifndef CRF_FAC4_NB
#define CRF_FAC4_NB 2       // chunks gathered per batch by the 1024-thread kernels (3: one weight pair spills INSIDE the frame loop, behind a vmcnt(0))
#endif
#ifndef CRF_FAC4_NB_ML
#define CRF_FAC4_NB_ML 2    // ... with multi-lane rows: the butterfly's registers make batches of 3 spill (V = 217: recursions 3.32 -> 3.02 ms)
#endif
#ifndef CRF_FAC5_NB2
#define CRF_FAC5_NB2 4      // ... by the two-utterance kernel on 512 threads x 30 chunks (256 registers per wave: 16 ds_read_b64 = 32 registers in flight)
#endif
#ifndef CRF_FAC3_NB2
#define CRF_FAC3_NB2 2      // chunks gathered per batch by the two-utterance kernels (8 ds_read_b64 = 16 registers in flight)
#endif
#define CRF_STR_(x) #x
#define CRF_STR(x) CRF_STR_(x)
#ifndef CRF_FAC3_NB_F
#define CRF_FAC3_NB_F 4
#endif
#ifndef CRF_FAC3_NB_B
#define CRF_FAC3_NB_B 4
#endif
static FacParams fac_params(const LossParams &lp, int dir, int *started, int i0, int i1, float *state, int nb, const int *bound, int *stage_cnt) {
    const FacDev &F = lp.g.fac;
    FacParams p{};
    p.L = dir == 0 ? F.f : F.b;
    p.bx_idx = F.bx_idx; p.bx_w = F.bx_w; p.nbx = F.nbx; p.bx_se = F.bx_se;
    p.B = lp.B; p.T = lp.T; p.V = lp.V; p.Rout = dir == 0 ? lp.Rq : lp.Rb; p.NT = F.NT; p.Rf = F.f.R;
    p.lx = lp.lx; p.ep = lp.ep; p.mx = lp.moff;   // (the resident kernels use it for the log-likelihood offset only)
    p.Out = dir == 0 ? lp.Q : lp.BP; p.Eout = dir == 0 ? lp.EQ : lp.EB; p.Row0 = lp.Row0;
    p.started = started; p.i0 = i0; p.i1 = i1; p.state = state;
    p.nb = nb; p.stage_cnt = stage_cnt;
    for (int k = 0; k < 16; ++k) p.bound[k] = (bound && k < nb) ? bound[k] : 0;
    p.frow_meta = F.frow_meta; p.x_start = F.x_start; p.x_end = F.x_end;
    p.den_zs = lp.den_zs; p.cost_alpha = lp.cost_alpha; p.den_ez = lp.den_ez;
    p.brow_meta = F.brow_meta; p.z_lab = F.z_lab; p.z_end = F.z_end; p.brow_start = F.brow_start; p.brow_end = F.brow_end;
    p.cb_part = lp.cb_part; p.cb_mxs = lp.cb_mxs; p.cb_F = lp.cb_F; p.redo = lp.redo;
    p.npair = (lp.B + 1) / 2; p.dump = lp.dump; p.dump_stride = lp.dump_stride;
    p.K = F.K; p.b0 = 0; p.nbu = lp.B; p.Gf = F.f.G; p.Gb = F.b.G; p.xch = lp.xch; p.err = lp.err; p.xlist = F.xlist; for (int k = 0; k < 3; ++k) p.xlist_off[k] = F.xlist_off[k];
    return p;
}
// factored recursions over TWO CUs each, utterances [b0, b0 + nbu): every workgroup of a launch must be resident at once
// (its peer spins on it), so the caller launches groups of at most CUs / 4 utterances
static int launch_fac2_pair(const LossParams &lp, size_t lds, hipStream_t st, int b0, int nbu) {
    static LdsMark mk;
    FacParams pf = fac_params(lp, 0, nullptr, 0, lp.T, nullptr, 0, nullptr, nullptr);
    FacParams pb = fac_params(lp, 1, nullptr, 0, lp.T, nullptr, 0, nullptr, nullptr);
    pf.b0 = pb.b0 = b0; pf.nbu = pb.nbu = nbu;
    int rc;
    if (lp.g.fac.threads == kFac4Threads) {   // 1024 threads x 15 chunks, four waves per SIMD (round 4)
        static LdsMark mk4;
        auto *k4 = crf_fac2_pair_kernel<kFac4Threads, kFac4NCH, CRF_FAC4_NB_ML, CRF_FAC4_NB_ML>;
        g_den_kernel = "crf_fac2_pair_kernel<1024,15," CRF_STR(CRF_FAC4_NB_ML) "," CRF_STR(CRF_FAC4_NB_ML) ">";
        if ((rc = ensure_lds((const void *)k4, lds, mk4, "fac2 pair"))) return rc;
        hipLaunchKernelGGL(k4, dim3((unsigned)(2 * nbu * 2)), dim3(kFac4Threads), lds, st, pf, pb);
    } else {
        auto *k = crf_fac2_pair_kernel<kFac3Threads, kFac3ArcCh, CRF_FAC3_NB_F, CRF_FAC3_NB_B>;
        g_den_kernel = "crf_fac2_pair_kernel<768,20,4,4>";
        if ((rc = ensure_lds((const void *)k, lds, mk, "fac2 pair"))) return rc;
        hipLaunchKernelGGL(k, dim3((unsigned)(2 * nbu * 2)), dim3(kFac3Threads), lds, st, pf, pb);
    }
    hipError_t e;
    if ((e = hipGetLastError()) != hipSuccess) { set_error(std::string("crf_fac2_pair_kernel: ") + hipGetErrorString(e)); return CRF_ERR_HIP; }
    return CRF_OK;
}
// factored recursions, iterations [i0, i1) of both directions as one grid of 2B workgroups; FLAG: publish stage counters
// (both directions bump the same counters: a stage is complete at 2B) and store the rows write-through
template <bool FLAG>
static int launch_fac_pair(const LossParams &lp, size_t lds, hipStream_t st, int *started, int i0, int i1, float *fstate, float *bstate,
                           int nb = 0, const int *bound = nullptr, int *stage_cnt = nullptr) {
    static LdsMark m3, m3m, m5;
    const FacDev &F = lp.g.fac;
    const bool g3 = F.threads == kFac3Threads, ml = F.multilane != 0;
    const FacParams pf = fac_params(lp, 0, started, i0, i1, fstate, nb, bound, stage_cnt);
    const FacParams pb = fac_params(lp, 1, started, i0, i1, bstate, nb, bound, stage_cnt);
    const dim3 grid((unsigned)(2 * lp.B));
    int rc;
#define CRF_LAUNCH_RL(NCH_, ML_, NAME_, MARK_)                                                                                   \
    {                                                                                                                           \
        static LdsMark MARK_;                                                                                                   \
        auto *k = crf_fac_pair_kernel<FLAG, kFac3Threads, NCH_, CRF_FAC3_NB_F, CRF_FAC3_NB_B, ML_, true>;                       \
        g_den_kernel = FLAG ? "crf_fac_pair_kernel<true,768," NAME_ ",true>" : "crf_fac_pair_kernel<false,768," NAME_ ",true>"; \
        if ((rc = ensure_lds((const void *)k, lds, MARK_, "fac pair"))) return rc;                                              \
        hipLaunchKernelGGL(k, grid, dim3(kFac3Threads), lds, st, pf, pb);                                                       \
    }
    // row constants in the LDS table: 20 chunks of arcs per thread (F.rcl == 1), or all 21 slots (== 2: graphs that need them)
    if (g3 && F.rcl == 1 && !ml) CRF_LAUNCH_RL(kFac3ArcCh, false, "20,4,4,false", mk20n)
    else if (g3 && F.rcl == 1) CRF_LAUNCH_RL(kFac3ArcCh, true, "20,4,4,true", mk20m)
    else if (g3 && F.rcl == 2 && !ml) CRF_LAUNCH_RL(kFac3LNCH, false, "21,4,4,false", mk21n)
    else if (g3 && F.rcl == 2) CRF_LAUNCH_RL(kFac3LNCH, true, "21,4,4,true", mk21m)
#undef CRF_LAUNCH_RL
    else if (F.threads == kFac4Threads) {   // 1024 threads: four waves per SIMD (the planner's first choice)
        static LdsMark m4n, m4m;
        if (ml) {
            auto *k = crf_fac_pair_kernel<FLAG, kFac4Threads, kFac4NCH, CRF_FAC4_NB_ML, CRF_FAC4_NB_ML, true, true>;
            g_den_kernel = FLAG ? "crf_fac_pair_kernel<true,1024," CRF_STR(CRF_FAC4_NCH) "," CRF_STR(CRF_FAC4_NB_ML) "," CRF_STR(CRF_FAC4_NB_ML) ",true,true>" : "crf_fac_pair_kernel<false,1024," CRF_STR(CRF_FAC4_NCH) "," CRF_STR(CRF_FAC4_NB_ML) "," CRF_STR(CRF_FAC4_NB_ML) ",true,true>";
            if ((rc = ensure_lds((const void *)k, lds, m4m, "fac pair"))) return rc;
            hipLaunchKernelGGL(k, grid, dim3(kFac4Threads), lds, st, pf, pb);
        } else {
            auto *k = crf_fac_pair_kernel<FLAG, kFac4Threads, kFac4NCH, CRF_FAC4_NB, CRF_FAC4_NB, false, true>;
            g_den_kernel = FLAG ? "crf_fac_pair_kernel<true,1024," CRF_STR(CRF_FAC4_NCH) "," CRF_STR(CRF_FAC4_NB) "," CRF_STR(CRF_FAC4_NB) ",false,true>" : "crf_fac_pair_kernel<false,1024," CRF_STR(CRF_FAC4_NCH) "," CRF_STR(CRF_FAC4_NB) "," CRF_STR(CRF_FAC4_NB) ",false,true>";
            if ((rc = ensure_lds((const void *)k, lds, m4n, "fac pair"))) return rc;
            hipLaunchKernelGGL(k, grid, dim3(kFac4Threads), lds, st, pf, pb);
        }
    }
    else if (g3 && ml) {
        auto *k = crf_fac_pair_kernel<FLAG, kFac3Threads, kFac3NCH, CRF_FAC3_NB_F, CRF_FAC3_NB_B, true>;
        g_den_kernel = FLAG ? "crf_fac_pair_kernel<true,768,21,4,4,true,false>" : "crf_fac_pair_kernel<false,768,21,4,4,true,false>";
        if ((rc = ensure_lds((const void *)k, lds, m3m, "fac pair"))) return rc;
        hipLaunchKernelGGL(k, grid, dim3(kFac3Threads), lds, st, pf, pb);
    } else if (g3) {
        auto *k = crf_fac_pair_kernel<FLAG, kFac3Threads, kFac3NCH, CRF_FAC3_NB_F, CRF_FAC3_NB_B, false>;
        g_den_kernel = FLAG ? "crf_fac_pair_kernel<true,768,21,4,4,false,false>" : "crf_fac_pair_kernel<false,768,21,4,4,false,false>";
        if ((rc = ensure_lds((const void *)k, lds, m3, "fac pair"))) return rc;
        hipLaunchKernelGGL(k, grid, dim3(kFac3Threads), lds, st, pf, pb);
    } else {
        auto *k = crf_fac_pair_kernel<FLAG, kResThreads, kResNCH, kResBatch, kResBatch, true>;
        g_den_kernel = FLAG ? "crf_fac_pair_kernel<true,512,30,6,6,true,false>" : "crf_fac_pair_kernel<false,512,30,6,6,true,false>";
        if ((rc = ensure_lds((const void *)k, lds, m5, "fac pair"))) return rc;
        hipLaunchKernelGGL(k, grid, dim3(kResThreads), lds, st, pf, pb);
    }
    hipError_t e;
    if ((e = hipGetLastError()) != hipSuccess) { set_error(std::string("crf_fac_pair_kernel: ") + hipGetErrorString(e)); return CRF_ERR_HIP; }
    return CRF_OK;
}

// ... two utterances per workgroup: 2 * ceil(B / 2) workgroups
template <bool FLAG>
static int launch_fac_pair2(const LossParams &lp, size_t lds, hipStream_t st, int *started, int nb = 0, const int *bound = nullptr, int *stage_cnt = nullptr) {
    static LdsMark m2, m2m;
    const FacDev &F = lp.g.fac;
    const bool ml = F.multilane != 0;
    const FacParams pf = fac_params(lp, 0, started, 0, lp.T, nullptr, nb, bound, stage_cnt);
    const FacParams pb = fac_params(lp, 1, started, 0, lp.T, nullptr, nb, bound, stage_cnt);
    const dim3 grid((unsigned)(2 * pf.npair));
    int rc;
#define CRF_LAUNCH_RL2(NCH_, ML_, NAME_, MARK_)                                                                                 \
    {                                                                                                                           \
        static LdsMark MARK_;                                                                                                   \
        auto *k = crf_fac_pair2_kernel<FLAG, kFac3Threads, NCH_, CRF_FAC3_NB2, CRF_FAC3_NB2, ML_, true>;                        \
        g_den_kernel = FLAG ? "crf_fac_pair2_kernel<true,768," NAME_ ",true>" : "crf_fac_pair2_kernel<false,768," NAME_ ",true>"; \
        if ((rc = ensure_lds((const void *)k, lds, MARK_, "fac pair2"))) return rc;                                             \
        hipLaunchKernelGGL(k, grid, dim3(kFac3Threads), lds, st, pf, pb);                                                       \
    }
#define CRF_LAUNCH_P512(ML_, NAME_, MARK_)                                                                                      \
    {                                                                                                                           \
        static LdsMark MARK_;                                                                                                   \
        auto *k = crf_fac_pair2_kernel<FLAG, kResThreads, kResNCH, CRF_FAC5_NB2, CRF_FAC5_NB2, ML_, true>;                      \
        g_den_kernel = FLAG ? "crf_fac_pair2_kernel<true,512,30," NAME_ ",true>" : "crf_fac_pair2_kernel<false,512,30," NAME_ ",true>"; \
        if ((rc = ensure_lds((const void *)k, lds, MARK_, "fac pair2"))) return rc;                                             \
        hipLaunchKernelGGL(k, grid, dim3(kResThreads), lds, st, pf, pb);                                                        \
    }
    if (F.threads == kResThreads) {   // the second layout (HostGraph::facp): 512 threads x 30 chunks, row table, implicit entries
        if (!F.imp || !F.rcl) { set_error("two-utterance kernel on 512 threads: not the second layout"); return CRF_ERR_ARG; }
        if (!ml) CRF_LAUNCH_P512(false, CRF_STR(CRF_FAC5_NB2) "," CRF_STR(CRF_FAC5_NB2) ",false", mp512n)
        else CRF_LAUNCH_P512(true, CRF_STR(CRF_FAC5_NB2) "," CRF_STR(CRF_FAC5_NB2) ",true", mp512m)
    } else
    if (F.rcl == 1 && !ml) CRF_LAUNCH_RL2(kFac3ArcCh, false, "20," CRF_STR(CRF_FAC3_NB2) "," CRF_STR(CRF_FAC3_NB2) ",false", mp20n)
    else if (F.rcl == 1) CRF_LAUNCH_RL2(kFac3ArcCh, true, "20," CRF_STR(CRF_FAC3_NB2) "," CRF_STR(CRF_FAC3_NB2) ",true", mp20m)
    else if (F.rcl == 2 && !ml) CRF_LAUNCH_RL2(kFac3LNCH, false, "21," CRF_STR(CRF_FAC3_NB2) "," CRF_STR(CRF_FAC3_NB2) ",false", mp21n)
    else if (F.rcl == 2) CRF_LAUNCH_RL2(kFac3LNCH, true, "21," CRF_STR(CRF_FAC3_NB2) "," CRF_STR(CRF_FAC3_NB2) ",true", mp21m)
#undef CRF_LAUNCH_RL2
#undef CRF_LAUNCH_P512
    else if (ml) {
        auto *k = crf_fac_pair2_kernel<FLAG, kFac3Threads, kFac3NCH, CRF_FAC3_NB2, CRF_FAC3_NB2, true, false>;
        g_den_kernel = FLAG ? "crf_fac_pair2_kernel<true,768,21," CRF_STR(CRF_FAC3_NB2) "," CRF_STR(CRF_FAC3_NB2) ",true,false>" : "crf_fac_pair2_kernel<false,768,21," CRF_STR(CRF_FAC3_NB2) "," CRF_STR(CRF_FAC3_NB2) ",true,false>";
        if ((rc = ensure_lds((const void *)k, lds, m2m, "fac pair2"))) return rc;
        hipLaunchKernelGGL(k, grid, dim3(kFac3Threads), lds, st, pf, pb);
    } else {
        auto *k = crf_fac_pair2_kernel<FLAG, kFac3Threads, kFac3NCH, CRF_FAC3_NB2, CRF_FAC3_NB2, false, false>;
        g_den_kernel = FLAG ? "crf_fac_pair2_kernel<true,768,21," CRF_STR(CRF_FAC3_NB2) "," CRF_STR(CRF_FAC3_NB2) ",false,false>" : "crf_fac_pair2_kernel<false,768,21," CRF_STR(CRF_FAC3_NB2) "," CRF_STR(CRF_FAC3_NB2) ",false,false>";
        if ((rc = ensure_lds((const void *)k, lds, m2, "fac pair2"))) return rc;
        hipLaunchKernelGGL(k, grid, dim3(kFac3Threads), lds, st, pf, pb);
    }
    hipError_t e;
    if ((e = hipGetLastError()) != hipSuccess) { set_error(std::string("crf_fac_pair2_kernel: ") + hipGetErrorString(e)); return CRF_ERR_HIP; }
    return CRF_OK;
}

}  // namespace crf

using namespace crf;

extern "C" {

int64_t crf_workspace_bytes(const crf_graph *g, int64_t B, int64_t T, int64_t V, int64_t max_label_len) {
    const int64_t Sc = rup64((int)(2 * max_label_len + 1));
    return ws_layout(g ? g->h : nullptr, B, T, V, Sc).total;
}

int crf_den_kernels(const crf_graph *g, int64_t B, int64_t T, int64_t V) {
    if (!g || !g->h) { set_error("null graph"); return -1; }
    const WsLayout w = ws_layout(g->h, B, T, V, 64);
    return w.bat ? 3 : w.fac ? 2 : w.res ? 1 : 0;
}

// Stage bounds of the staged grad pass (crf_loss_fwd_bwd; crf_debug_stage_plan shows them to the tests): bound[0] = 0 < bound[1] < ... < bound[nstage] = T,
// stage k = the recursions' iterations [bound[k-1], bound[k]).  Returns nstage; *gd_piece = the length of the equal pieces.
static int plan_grad_stages(int64_t T, bool segmode, int stages_env, int pieces, int *bound, int *gd_piece_out) {
    int nstage = 1, gd_piece = 0;
    bound[0] = 0;
    bound[1] = (int)T;
    if (T >= 256) {
        const int half = (int)((T / 2 + kGDFrames - 1) / kGDFrames * kGDFrames);
        int piece, first = half;
        if (stages_env > 0 || segmode) {
            const int nshort = std::max(1, std::min(std::min(pieces, kMaxStages - 2), (int)(T - half) / 32));
            piece = ((int)T - half + nshort - 1) / nshort;
            piece = (piece + kGDFrames - 1) / kGDFrames * kGDFrames;
        } else {
            // (round 5, one grad launch for all stages: a stage costs the grad pass nothing any more and the recursions one drain + barrier;
            // pieces of 48 .. 96 iterations all give 2.71 ms where 128 gives 2.75 and round 4's per-stage launches 2.82: profiles/round5_ab_grad_one_launch.txt)
            piece = opt_on(kOpt_gd_stage_launches) ? 128 : 80;
            while ((kMaxStages - 4) * piece < (int)T - half && piece < (int)T) piece += opt_on(kOpt_gd_stage_launches) ? 128 : 16;   // the stage counters cover T - half
            const int piece_env = opt(kOpt_piece, 0);
            if (piece_env > 0) piece = (piece_env + kGDFrames - 1) / kGDFrames * kGDFrames;
            const int body = std::min((int)T - half, (kMaxStages - 2) * piece);   // (a CRF_PIECE too small for the counters)
            first = std::max(half, ((int)T - body + kGDFrames - 1) / kGDFrames * kGDFrames);
            const int fs = opt(kOpt_first_shift, 0);
            if (fs > 0) first = std::min((int)T - kGDFrames, first + fs * kGDFrames);
        }
        nstage = 1;
        gd_piece = piece;
        bound[1] = first;
        while (bound[nstage] < T && nstage < kMaxStages - 1) { bound[nstage + 1] = std::min((int)T, bound[nstage] + piece); ++nstage; }
        bound[nstage] = (int)T;
        // taper: the last stage is what is left to do when the recursions have ended; with one grad launch for all stages (its workgroups
        // wait themselves) a stage costs the grad pass nothing and the recursions one drain + barrier, so the last pieces are halved down
        // to `taper` iterations: ..., piece, piece / 2, piece / 4, ..., taper
        const int taper = stages_env > 0 || segmode || opt_on(kOpt_gd_stage_launches) ? 0 : (opt(kOpt_taper, 32) + kGDFrames - 1) / kGDFrames * kGDFrames;
        if (taper > 0 && taper < piece && nstage >= 3) {
            int desc[kMaxStages + 8], n = 0, pos = (int)T;             // stage ends from the last one backwards
            desc[n++] = pos;
            for (int q = taper; q < piece && n < 8; q *= 2) { pos -= q; desc[n++] = pos; }
            while (pos - piece > first + kGDFrames && n < kMaxStages + 6) { pos -= piece; desc[n++] = pos; }
            if (pos > first && n + 1 <= kMaxStages - 1) {               // (the piece behind `first` takes what is left: 16 .. piece + 16 iterations)
                nstage = n + 1;
                bound[1] = first;
                for (int k = 0; k < n; ++k) bound[2 + k] = desc[n - 1 - k];
            }
        }
    }
    *gd_piece_out = gd_piece;
    return nstage;
}

// The one-launch grad pass's grid (crf_grad_den_kernel, gd_persist): first block of the candidates of the stages `stage` .. nstage, frames per
// workgroup in each; returns the number of workgroups
static int64_t plan_grad_grid(const int *bound, int nstage, int stage, int64_t B, int gd_piece, int *poff, int *fpb) {
    int64_t tot = 0;
    const int sub_env = opt(kOpt_gd_sub, 16);              // frames per workgroup in a last stage shorter than `piece` (16: whole blocks)
    const int fsub = sub_env == 8 ? 8 : sub_env == 4 ? 4 : sub_env == 2 ? 2 : kGDFrames;
    for (int k = stage; k <= nstage; ++k) {
        poff[k] = (int)tot;
        fpb[k] = (k == nstage && bound[k] - bound[k - 1] < gd_piece) ? fsub : kGDFrames;   // (the LAST stage only)
        tot += 2 * (int64_t)((bound[k] - bound[k - 1] + kGDFrames - 1) / kGDFrames + 3) * (kGDFrames / fpb[k]) * B;
    }
    poff[nstage + 1] = (int)tot;
    return tot;
}

static int loss_impl(const crf_graph *g, const float *logp, int fused, int in_dtype, const int32_t *labels, const int32_t *lab_off,
                     const int32_t *lx, const int32_t *ly, int64_t B, int64_t T, int64_t V,
                     int64_t max_label_len, float c_den, float c_ctc, float *grad, float *loss,
                     float *costs_den, float *costs_beta, float *costs_ctc, int32_t *invalid, void *ws,
                     int64_t ws_bytes, void *stream_);

int crf_loss_fwd_bwd(const crf_graph *g, const float *logp, const int32_t *labels, const int32_t *lab_off,
                     const int32_t *lx, const int32_t *ly, int64_t B, int64_t T, int64_t V,
                     int64_t max_label_len, float c_den, float c_ctc, float *grad, float *loss,
                     float *costs_den, float *costs_beta, float *costs_ctc, int32_t *invalid, void *ws,
                     int64_t ws_bytes, void *stream_) {
    return loss_impl(g, logp, 0, 0, labels, lab_off, lx, ly, B, T, V, max_label_len, c_den, c_ctc, grad, loss, costs_den, costs_beta,
                     costs_ctc, invalid, ws, ws_bytes, stream_);
}

int crf_loss_fwd_bwd_logits(const crf_graph *g, const void *logits, int dtype, const int32_t *labels, const int32_t *lab_off,
                            const int32_t *lx, const int32_t *ly, int64_t B, int64_t T, int64_t V,
                            int64_t max_label_len, float c_den, float c_ctc, float *grad, float *loss,
                            float *costs_den, float *costs_beta, float *costs_ctc, int32_t *invalid, void *ws,
                            int64_t ws_bytes, void *stream_) {
    if (dtype < 0 || dtype > 2) { set_error("crf_loss_fwd_bwd_logits: dtype must be 0 (f32), 1 (bf16) or 2 (f16)"); return CRF_ERR_ARG; }
    if (c_ctc == 0.f) { set_error("crf_loss_fwd_bwd_logits: the fused log_softmax needs the numerator pass (c_ctc != 0)"); return CRF_ERR_UNSUPPORTED; }
    return loss_impl(g, (const float *)logits, 1, dtype, labels, lab_off, lx, ly, B, T, V, max_label_len, c_den, c_ctc, grad, loss,
                     costs_den, costs_beta, costs_ctc, invalid, ws, ws_bytes, stream_);
}

static int loss_impl(const crf_graph *g, const float *logp, int fused, int in_dtype, const int32_t *labels, const int32_t *lab_off,
                     const int32_t *lx, const int32_t *ly, int64_t B, int64_t T, int64_t V,
                     int64_t max_label_len, float c_den, float c_ctc, float *grad, float *loss,
                     float *costs_den, float *costs_beta, float *costs_ctc, int32_t *invalid, void *ws,
                     int64_t ws_bytes, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const bool den = c_den != 0.f, ctc = c_ctc != 0.f;
    if (!logp || !lx || !grad || !loss || !ws) { set_error("null argument"); return CRF_ERR_ARG; }
    if (B <= 0 || T <= 0 || V <= 0 || B * T > INT32_MAX) { set_error("bad B/T/V"); return CRF_ERR_ARG; }
    if (!den && !ctc) { set_error("c_den and c_ctc are both zero"); return CRF_ERR_ARG; }
    if (den && (!g || !g->h)) { set_error("denominator requested without a graph"); return CRF_ERR_ARG; }
    if (ctc && (!labels || !lab_off || !ly || max_label_len < 0)) { set_error("numerator requested without labels"); return CRF_ERR_ARG; }
    const HostGraph *h = den ? g->h : nullptr;
    if (den && V <= h->dev.max_label) {
        set_error("den_lm has label " + std::to_string(h->dev.max_label) + " but log_probs has only V=" + std::to_string(V) + " classes");
        return CRF_ERR_ARG;
    }
    if (V > kEpRegs * kChainThreads) { set_error("V > 8192 not supported by this build"); return CRF_ERR_UNSUPPORTED; }
    const int Sc = rup64((int)(2 * (ctc ? max_label_len : 0) + 1));
    if (ctc && 2 * max_label_len + 1 > kCtcRegs * kCtcThreads) { set_error("label length > 2047 not supported by this build"); return CRF_ERR_UNSUPPORTED; }
    const WsLayout w = ws_layout(h, B, T, V, Sc);
    if (ws_bytes < w.total) { set_error("workspace too small: need " + std::to_string(w.total)); return CRF_ERR_WORKSPACE; }
    const bool res = den && w.res, gv = den && w.gv, fac = den && w.fac, bat = den && w.bat;
    if (fac && T * std::max<int64_t>(V, std::max(w.Rq, w.Rb)) >= ((int64_t)1 << 32)) {   // (the factored frame loop adds 32-bit row offsets)
        set_error("T * max(V, row length) >= 2^32 not supported by the factored kernels"); return CRF_ERR_UNSUPPORTED;
    }
    size_t lds_chain = 0;
    if (den && !res && !bat) lds_chain = std::max(chain_lds_bytes(h, (int)V, Sc, 0, gv), chain_lds_bytes(h, (int)V, Sc, 1, gv));
    if (res && !fac) lds_chain = std::max(res_lds_bytes(h, (int)V, 0, h->res_rows_cu_f), res_lds_bytes(h, (int)V, 1, h->res_rows_cu_b));
    if (fac) lds_chain = w.p2mode ? std::max(fac2u_lds_bytes(w.p2mode == 2 ? h->facp : h->dev.fac, (int)V, 0), fac2u_lds_bytes(w.p2mode == 2 ? h->facp : h->dev.fac, (int)V, 1))
                                  : std::max(fac_lds_bytes(h, (int)V, 0), fac_lds_bytes(h, (int)V, 1));
    if (ctc) lds_chain = std::max(lds_chain, chain_lds_bytes(h, (int)V, Sc, 2));
    const int gnc_all = den ? std::max(std::max(h->dev.NC, h->dev.res.NC), std::max(h->dev.fac.ok ? h->dev.fac.NC : 0, h->facp.ok ? h->facp.NC : 0)) : 0;
    // the generic grad kernel stages the two rows of a frame in LDS when they fit, else gathers them from L2
    const bool grad_stage = !den || bat || ((size_t)rup64((int)w.Rq) + rup64((int)w.Rb) + rup64(gnc_all) + 2 * (size_t)rup64((int)V)) * 4 <= 150 * 1024;
    const size_t lds_grad = ((den && !bat && grad_stage ? (size_t)rup64((int)w.Rq) + rup64((int)w.Rb) : 0) + rup64(bat ? 0 : gnc_all) + 2 * (size_t)rup64((int)V)) * sizeof(float);
    if (lds_chain > 160 * 1024 || lds_grad > 160 * 1024) {
        set_error("graph too large for this build (states=" + std::to_string(h ? h->S : 0) + ")");
        return CRF_ERR_UNSUPPORTED;
    }

    LossParams p{};
    if (den) p.g = h->dev;
    // the factored layout this call works with: the main one, or -- two utterances per workgroup on 512 threads -- the second one
    // (ws_layout has sized the rows for it); everything below reads it through p.g.fac / FX
    const FacDev *FX = (den && w.fac) ? (w.p2mode == 2 ? &h->facp : &h->dev.fac) : nullptr;
    if (FX) p.g.fac = *FX;
    p.logp = logp; p.labels = labels; p.lab_off = lab_off; p.lx = lx; p.ly = ly;
    p.B = (int)B; p.T = (int)T; p.V = (int)V; p.Sc = Sc;
    p.c_den = c_den; p.c_ctc = c_ctc;
    char *base = (char *)ws;
    p.ep = (float *)(base + w.off_ep); p.mx = (float *)(base + w.off_mx);
    p.fused = fused; p.in_dtype = in_dtype;
    p.moff = fused ? (float *)(base + w.off_moff) : p.mx; p.inv_s = (float *)(base + w.off_invs);
    p.Q = (float *)(base + w.off_Q); p.BP = (float *)(base + w.off_BP);
    p.Rq = (int)w.Rq; p.Rb = (int)w.Rb; p.res = fac ? 2 : res ? 1 : 0; p.grad_stage = grad_stage ? 1 : 0;
    if (fac) {
        const FacDev &F = *FX;
        p.gq = F.gq; p.gb = F.gb; p.gchunk = F.chunk_off; p.glab = F.lab_chunk_off; p.gNC = F.NC;
    } else if (den) {
        p.gq = res ? h->dev.res.gq : h->dev.perm; p.gb = res ? h->dev.res.gb : h->dev.perm;
        p.gchunk = res ? h->dev.res.chunk_off : h->dev.chunk_off; p.glab = res ? h->dev.res.lab_chunk_off : h->dev.lab_chunk_off;
        p.gNC = res ? h->dev.res.NC : h->dev.NC;
    }
    if (res && !fac) { p.res_lds_rows_f = h->res_rows_cu_f; p.res_lds_rows_b = h->res_rows_cu_b; }
    p.EQ = (int *)(base + w.off_EQ); p.EB = (int *)(base + w.off_EB);
    p.CA = (double *)(base + w.off_CA); p.CB = (double *)(base + w.off_CB);
    p.ECA = (int *)(base + w.off_ECA); p.ECB = (int *)(base + w.off_ECB);
    p.ctc_zc = (double *)(base + w.off_pb);
    p.cb_mxs = p.ctc_zc + B;
    float *pb = (float *)(p.cb_mxs + B);
    p.cb_part = pb + 8 * B; p.cb_F = (int *)(pb + 8 * B + (int64_t)kResMaxK * B);
    p.redo = (int *)(pb + 16 * B);   // [2][B]
    p.redo_ctc = (int *)(pb + 18 * B);   // [B]
    p.ctc_logdom = (int *)(pb + 19 * B); // [B]
    p.ctc_bad = (int *)(base + w.off_cbad);
    // CRF_ROBUST: 0 = never run the robust fallback, 1 = every utterance takes it (tests, or "safe mode"); default: the
    // utterances the fast kernels flag
    const int robust_env = opt(kOpt_robust, -1);   // (read per call: tests switch it)
    p.force_redo = (den && robust_env == 1) ? 1 : 0;
    p.force_redo_ctc = (ctc && (robust_env == 1 || opt_on(kOpt_robust_ctc))) ? 1 : 0;
    p.ctc_tilt = std::min(400, std::max(0, opt(kOpt_ctc_tilt, 100)));
    p.xch = (unsigned long long *)(base + w.off_xch);
    p.err = (int *)(base + w.off_xch + w.xch_bytes);
    p.Row0 = (float *)(base + w.off_row0);
    p.dump = (float *)(base + w.off_dump); p.dump_stride = (int)w.dump_stride;
    p.gvec = (float *)(base + w.off_gvec);
    p.gvec_stride = w.gvec_stride;
    p.den_zs = pb; p.den_ez = (int *)(pb + B); p.ctc_ez = (int *)(pb + 2 * B);
    p.cost_alpha = pb + 3 * B; p.cost_beta = pb + 4 * B; p.cost_ctc = pb + 5 * B; p.invalid = (int *)(pb + 6 * B);
    p.grad = grad; p.loss = loss; p.out_den = costs_den; p.out_beta = costs_beta; p.out_ctc = costs_ctc;
    p.out_invalid = invalid;

    hipError_t e;
#define LAUNCH_CHECK(what)                                                                         \
    if ((e = hipGetLastError()) != hipSuccess) { set_error(std::string(what) + ": " + hipGetErrorString(e)); return CRF_ERR_HIP; }

    const int64_t frames = B * T;
    // Two streams at most: the caller's and one side stream of this (device, caller stream)'s context.  Forward and
    // backward recursions are one grid each (denominator pair, numerator pair); the two grids are independent.
    const bool serial_env = opt_on(kOpt_serial_chains);
    DevCtx *cx = nullptr;
    int rc;
    if ((rc = get_ctx(stream, &cx))) return rc;
    std::lock_guard<std::mutex> call_lock(cx->mu);
    // A context whose probe found no stream beside the caller's (a device shared with another busy process at that moment can make the
    // two single-wave probe kernels miss each other) asks again every 256 calls instead of staying on the serial schedule for good.
    // (at most three more probes, at calls 256, 1 024 and 4 096: a process that cannot have a second queue at all -- GPU_MAX_HW_QUEUES=1 --
    // must not pay a dozen candidates' time-outs every 256 steps for the rest of the run; never while the caller's stream is being captured)
    if (!cx->side && cx->flags && !opt_on(kOpt_no_side_stream) && !opt_on(kOpt_trust_side) && cx->reprobes < 3 &&
        cx->call_id + 1 == (256 << (2 * cx->reprobes))) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(stream, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
        ++cx->reprobes;
        if (cap == hipStreamCaptureStatusNone) {
            (void)hipStreamSynchronize(stream);
            cx->side = find_beside(stream, cx->flags, cx->dev, &cx->side_kind, &cx->side_tries);
            snprintf(cx->side_desc, sizeof(cx->side_desc), "%s (candidate %d, found at call %d)", kSideNames[cx->side_kind], cx->side_tries, cx->call_id + 1);
        }
    }
    const bool serial = serial_env || !cx->side;              // no side stream: everything in order on the caller's stream
    hipStream_t side = serial ? stream : cx->side;
    // error word, start counter and stage counters live in fine-grained memory (get_ctx); the prep kernel clears them
    const bool have_flags = cx->flags != nullptr;
    p.clear = nullptr; p.nclear = 0;
    p.call_id = ++cx->call_id; p.ctc_seen = cx->seen;
    g_call_streams = 1;
    g_side_desc = cx->side_desc;
    if (have_flags) { p.err = cx->flags; p.clear = cx->flags; p.nclear = 64; }
    g_last_err_word = p.err;
    static LdsMark lds_mark_grad;
    if ((rc = ensure_lds((const void *)crf_grad_kernel, lds_grad, lds_mark_grad, "grad"))) return rc;
    bool forked = false, side_used = false;
    bool ctc_pass1 = false;   // the numerator's log-domain fallback has run in this call (staged schedule)
    auto fork_side = [&]() -> int {   // the side stream starts behind everything queued on the caller's stream so far
        if (serial || forked) return CRF_OK;
        if ((e = hipEventRecord(cx->fork, stream)) != hipSuccess || (e = hipStreamWaitEvent(side, cx->fork, 0)) != hipSuccess) {
            set_error(std::string("fork: ") + hipGetErrorString(e)); return CRF_ERR_HIP;
        }
        forked = side_used = true;
        if (g_call_streams < 2) g_call_streams = 2;
        return CRF_OK;
    };
    auto join_side = [&]() -> int {   // the caller's stream continues behind everything queued on the side stream so far
        if (serial || !side_used) return CRF_OK;
        if ((e = hipEventRecord(cx->join, side)) != hipSuccess || (e = hipStreamWaitEvent(stream, cx->join, 0)) != hipSuccess) {
            set_error(std::string("join: ") + hipGetErrorString(e)); return CRF_ERR_HIP;
        }
        forked = false;
        return CRF_OK;
    };
    int *started = p.err + 1;   // workgroups of the den kernels that hold a CU (cleared with the error word)
    int ncu_dev = 256;
    (void)hipDeviceGetAttribute(&ncu_dev, hipDeviceAttributeMultiprocessorCount, cx->dev);
    // The denominator half of the grad pass has a streaming kernel (index pairs in registers, rows
    // prefetched); it needs 16-bit row indices, rows of <= kGDRowRegs*256 floats and <= 2 chunks per thread.
    const int gnc = den ? (fac ? FX->NC : res ? h->dev.res.NC : h->dev.NC) : 0;
    const int gcap = fac ? FX->chunk_cap : kChunk;       // entries per chunk of the grad pass's pair lists
    const bool gd_wide = den && (w.Rq > 4 * kGDRowRegs * kGDThreads || w.Rb > 4 * kGDRowRegs * kGDThreads);   // 512-thread grad workgroups
    const bool fast_den = den && w.Rq <= 8 * kGDRowRegs * kGDThreads && w.Rb <= 8 * kGDRowRegs * kGDThreads && w.Rq % 4 == 0 && w.Rb % 4 == 0 &&
                          gnc <= 4 * kGDThreads && V <= kGDEpRegs * kGDThreads && !opt_on(kOpt_no_fast_grad);
    // numerator half of the grad pass: streaming kernel when the vocabulary fits its registers
    const bool fast_ctc = ctc && V <= kGCVRegs * kGCThreads && 2 * max_label_len + 1 <= kGCRegs * kGCThreads &&
                          !opt_on(kOpt_no_fast_grad);
    // Factored den kernels: 2B workgroups, one CU each.  While that is at most half of the chip, everything else
    // runs BESIDE them on the other half, on the side stream: numerator chains, their grad half, and the den half of
    // the grad pass.  The den half of the grad pass needs rows of BOTH recursions, which work towards each other; it
    // is released in stages: the recursions bump a counter at every stage bound, the grad launch of stage k is queued
    // behind a STREAM-level wait on that counter and takes the 16-frame blocks the stage completed.
    // (A grad pass that SPINS on progress counters of the running den kernels is faster still on a quiet device,
    // but with many launches queued ahead the den kernels were observed to stop for seconds while the waiting
    // workgroups kept their queue busy -- one kernel must never wait for another.)
    const bool no_overlap = opt_on(kOpt_no_overlap);   // diagnostics
    // how stage k of the grad pass is released: a stream-level wait on a counter the running recursions bump
    // (default), or -- CRF_SEGMENTS=1, and automatically if hipStreamWaitValue32 is refused -- by cutting the
    // recursions into one launch per stage with an event after each (~40 us per relaunch, state parked in HBM)
    static std::atomic<bool> use_segments{false};      // set when hipStreamWaitValue32 is refused
    const bool segmode = use_segments.load() || opt_on(kOpt_segments);
    const int stages_env = opt(kOpt_stages, 0);
    const int pieces = stages_env > 0 ? stages_env : (segmode ? 4 : 12);   // measured: 4 / 8 / 12 pieces -> call 4.06 / 3.98 / 3.93 ms (flags)
    // two utterances per workgroup (use_fac_pair2): the den grid is 2 * ceil(B / 2) workgroups instead of 2 B
    const bool pair2 = fac && w.p2mode != 0;   // (pair2_mode, decided with the workspace layout: the rows are sized for the layout it names)
    const int64_t den_wgs = pair2 ? 2 * ((B + 1) / 2) : 2 * B;
    // (the two-utterance kernel has no segment relaunches: where stream-level waits are refused it runs unstaged)
    const bool staged = fac && FX->K == 1 && ctc && fast_den && fast_ctc && !serial && !no_overlap && have_flags && !(pair2 && segmode) && den_wgs * 100 <= (int64_t)ncu_dev * opt(kOpt_stage_fill, 75);   // (B = 80: 4.16 -> 3.56 ms, B = 96: 4.37 -> 4.26, B = 112 at 90 %: 5.33 -> 5.57)
    // Stage bounds.  Nothing can be released before the two recursions have met, so the first stage ends at half of
    // the frames or later; after that a piece of `piece` iterations releases 2 * piece / 16 frame blocks per
    // utterance.  The grad pass has half of the chip and is bandwidth-bound there (~2 TB/s against the 2.2 TB/s the
    // two recursions produce), with a fixed cost per stage (launch, the workgroups' set-up, partial rounds); what is
    // left when the recursions end is its backlog plus the last stage.  Measured (B=64, T=1500, recursions 2.95 ms):
    // pieces of 32 / 48 / 64 / 96 / 128 / 192 / 256 / 384 iterations -> step 3.88 / 3.62 / 3.42 / 3.38 / 3.33 / 3.35 /
    // 3.41 / 3.53 ms; pieces that shrink towards the end (256,192,128,96,64 ...) were no better than equal ones.
    // CRF_PIECE / CRF_STAGES override (segment mode: 4 pieces, each relaunch costs ~40 us).
    int bound[kMaxStages + 1] = {0};
    int nstage = 1, gd_piece = 0;
    bound[1] = (int)T;
    if (staged && T >= 256) nstage = plan_grad_stages(T, segmode, stages_env, pieces, bound, &gd_piece);
    p.gd_nb = nstage + 1;
    for (int k = 0; k <= nstage && k < 16; ++k) p.gd_bound[k] = bound[k];
    float *fstate = (float *)(base + w.off_state), *bstate = fstate + B * w.state_stride;
    const size_t lds_fac = !fac ? 0 : pair2 ? std::max(fac2u_lds_bytes(*FX, (int)V, 0), fac2u_lds_bytes(*FX, (int)V, 1))
                                              : std::max(fac_lds_bytes(h, (int)V, 0), fac_lds_bytes(h, (int)V, 1));
    const size_t lds_ctc = chain_lds_bytes(h, (int)V, Sc, 2);

    const dim3 ggrid((unsigned)((T + kGradFrames - 1) / kGradFrames), (unsigned)B);
    auto launch_grad_den = [&](hipStream_t st, int stage, bool persist = false) -> int {
        p.gd_stage = stage;
        const size_t l = ((size_t)rup64((int)w.Rq + 1) + rup64((int)w.Rb + 1) + 4 * rup64((int)V) + kGDFrames + rup64(gnc)) * sizeof(float);
        dim3 gg((unsigned)((T + kGDFrames - 1) / kGDFrames), (unsigned)B);
        p.gd_nf = 0;
        p.gd_persist = 0;
        const bool full_grid = opt_on(kOpt_gd_full_grid);
        if (persist) {   // the stages `stage` .. nstage in one launch (see the kernel): 2 * nf candidates per utterance and stage, stage-major
            p.gd_persist = 1;
            p.gd_cnt = cx->flags + 16;
            p.gd_target = (int)(2 * B);
            const int64_t tot = plan_grad_grid(p.gd_bound, p.gd_nb - 1, stage, B, gd_piece, p.gd_poff, p.gd_fpb);
            gg = dim3((unsigned)tot, 1);
        } else if (stage > 1 && !full_grid) {   // (stage 1 is the middle of every utterance: all blocks are candidates)
            p.gd_nf = (p.gd_bound[stage] - p.gd_bound[stage - 1] + kGDFrames - 1) / kGDFrames + 3;
            if (2 * p.gd_nf < (int)gg.x) gg.x = (unsigned)(2 * p.gd_nf); else p.gd_nf = 0;
        }
        static LdsMark set1, set2, set3, set5;
        int r2;
        if (gcap == 8) {              // chunk lists cut at 8 entries (many labels with few pairs each): 512 threads, two chunks each
            static LdsMark set7;
            if ((r2 = ensure_lds((const void *)crf_grad_den_kernel<2, 2, 2 * kGDThreads, 8>, l, set7, "grad den"))) return r2;
            hipLaunchKernelGGL((crf_grad_den_kernel<2, 2, 2 * kGDThreads, 8>), gg, dim3(2 * kGDThreads), l, st, p);
        } else if (gnc > 2 * kGDThreads) {   // more than 512 label chunks (graphs over hundreds of classes: V = 500 has ~8 pairs per label,
                                      // one chunk each): 512 threads with two chunks each
            static LdsMark set6;
            if ((r2 = ensure_lds((const void *)crf_grad_den_kernel<2, 2, 2 * kGDThreads>, l, set6, "grad den"))) return r2;
            hipLaunchKernelGGL((crf_grad_den_kernel<2, 2, 2 * kGDThreads>), gg, dim3(2 * kGDThreads), l, st, p);
        } else if (gd_wide && CRF_X_GDW2 && w.Rq <= 32 * kGDThreads && w.Rb <= 32 * kGDThreads) {
            // rows of 5121 .. 8192 floats: 512 threads with four row registers each, held to 128 VGPRs so that a CU takes TWO workgroups (the
            // five-register form below compiles to 148 VGPRs: one workgroup, eight waves, per CU -- the estimated S = 6836 graph ran on that)
            static LdsMark set8;
            if ((r2 = ensure_lds((const void *)crf_grad_den_kernel<1, 2, 2 * kGDThreads, kChunk, 4, 4>, l, set8, "grad den"))) return r2;
            hipLaunchKernelGGL((crf_grad_den_kernel<1, 2, 2 * kGDThreads, kChunk, 4, 4>), gg, dim3(2 * kGDThreads), l, st, p);
        } else if (gd_wide) {   // rows of more than 5120 floats: 512 threads per workgroup (one chunk per thread up to 512 chunks)
            if ((r2 = ensure_lds((const void *)crf_grad_den_kernel<1, 2, 2 * kGDThreads>, l, set5, "grad den"))) return r2;
            hipLaunchKernelGGL((crf_grad_den_kernel<1, 2, 2 * kGDThreads>), gg, dim3(2 * kGDThreads), l, st, p);
        } else if (gnc <= kGDThreads && V <= kGDThreads) {   // small vocabulary
            if ((r2 = ensure_lds((const void *)crf_grad_den_kernel<1, 1>, l, set3, "grad den"))) return r2;
            hipLaunchKernelGGL((crf_grad_den_kernel<1, 1>), gg, dim3(kGDThreads), l, st, p);
        } else if (gnc <= kGDThreads) {
            if ((r2 = ensure_lds((const void *)crf_grad_den_kernel<1, kGDEpRegs>, l, set1, "grad den"))) return r2;
            hipLaunchKernelGGL((crf_grad_den_kernel<1, kGDEpRegs>), gg, dim3(kGDThreads), l, st, p);
        } else {
            if ((r2 = ensure_lds((const void *)crf_grad_den_kernel<2, kGDEpRegs>, l, set2, "grad den"))) return r2;
            hipLaunchKernelGGL((crf_grad_den_kernel<2, kGDEpRegs>), gg, dim3(kGDThreads), l, st, p);
        }
        if ((e = hipGetLastError()) != hipSuccess) { set_error(std::string("crf_grad_den_kernel: ") + hipGetErrorString(e)); return CRF_ERR_HIP; }
        return CRF_OK;
    };
    auto launch_grad_ctc = [&](int phase, hipStream_t st) -> int {  // phase 2: subtract from the den half; 0: plain CTC (writes)
        p.grad_phase = phase;
        if (fast_ctc) {
            const size_t l = (size_t)4 * rup64((int)V) * sizeof(float) + kGCFrames * sizeof(double) + 64;
            const dim3 gg((unsigned)((T + kGCFrames - 1) / kGCFrames), (unsigned)B);
            const int Sxm = 2 * (int)max_label_len + 1;
            if (Sxm <= 2 * kGCThreads) hipLaunchKernelGGL(crf_grad_ctc_kernel<2>, gg, dim3(kGCThreads), l, st, p);
            else if (Sxm <= 4 * kGCThreads) hipLaunchKernelGGL(crf_grad_ctc_kernel<4>, gg, dim3(kGCThreads), l, st, p);
            else hipLaunchKernelGGL(crf_grad_ctc_kernel<kGCRegs>, gg, dim3(kGCThreads), l, st, p);
        } else {
            hipLaunchKernelGGL(crf_grad_kernel, ggrid, dim3(kGradThreads), lds_grad, st, p);
        }
        if ((e = hipGetLastError()) != hipSuccess) { set_error(std::string("crf_grad(ctc): ") + hipGetErrorString(e)); return CRF_ERR_HIP; }
        return CRF_OK;
    };
    // Numerator fallback: utterances with frames the grad pass marked (or whose scaled chain lost its mass) redo their chains in the
    // log domain, then the marked frames' posteriors are subtracted from the rows; the other utterances' workgroups leave at once --
    // two near-empty launches.  Pass 1 runs right behind the numerator's grad half on the side stream of the staged schedule, i.e.
    // BESIDE the denominator recursions (V = 500: 1.2 ms that followed the call's last grad launch); pass 2 at the end of every call
    // takes what is marked and was not redone in pass 1.
    auto launch_robust_ctc_chains = [&](hipStream_t st, int pass) -> int {
        static LdsMark mrc[4];
        int r2;
        const int64_t ni = (2 * max_label_len + 1 + kCtcThreads - 1) / kCtcThreads;
        const int nri = ni <= 1 ? 0 : ni <= 2 ? 1 : ni <= 4 ? 2 : 3;
        const void *fn = nri == 0 ? (const void *)crf_robust_ctc_kernel<1> : nri == 1 ? (const void *)crf_robust_ctc_kernel<2>
                       : nri == 2 ? (const void *)crf_robust_ctc_kernel<4> : (const void *)crf_robust_ctc_kernel<kCtcRegs>;
        if ((r2 = ensure_lds(fn, lds_ctc, mrc[nri], "robust ctc"))) return r2;
        p.ctc_pass = pass;
        switch (nri) {
            case 0: hipLaunchKernelGGL(crf_robust_ctc_kernel<1>, dim3((unsigned)(2 * B)), dim3(kCtcThreads), lds_ctc, st, p); break;
            case 1: hipLaunchKernelGGL(crf_robust_ctc_kernel<2>, dim3((unsigned)(2 * B)), dim3(kCtcThreads), lds_ctc, st, p); break;
            case 2: hipLaunchKernelGGL(crf_robust_ctc_kernel<4>, dim3((unsigned)(2 * B)), dim3(kCtcThreads), lds_ctc, st, p); break;
            default: hipLaunchKernelGGL(crf_robust_ctc_kernel<kCtcRegs>, dim3((unsigned)(2 * B)), dim3(kCtcThreads), lds_ctc, st, p); break;
        }
        if ((e = hipGetLastError()) != hipSuccess) { set_error(std::string("crf_robust_ctc_kernel: ") + hipGetErrorString(e)); return CRF_ERR_HIP; }
        return CRF_OK;
    };
    auto launch_robust_ctc_fix = [&](hipStream_t st, int pass) -> int {
        p.ctc_pass = pass;
        hipLaunchKernelGGL(crf_robust_ctc_fix_kernel, dim3((unsigned)((T + kGCFrames - 1) / kGCFrames), (unsigned)B), dim3(kGradThreads),
                           (size_t)rup64((int)V) * sizeof(float), st, p);
        if ((e = hipGetLastError()) != hipSuccess) { set_error(std::string("crf_robust_ctc_fix_kernel: ") + hipGetErrorString(e)); return CRF_ERR_HIP; }
        return CRF_OK;
    };
    auto launch_robust_ctc = [&](hipStream_t st, int pass) -> int {
        const int r2 = launch_robust_ctc_chains(st, pass);
        return r2 ? r2 : launch_robust_ctc_fix(st, pass);
    };
    // forward log Z = backward log Z?  (crf_den_check_kernel: behind every launch of the recursions, on their stream -- in the staged schedule
    // it runs while the side stream finishes the grad pass -- and in front of the fallback kernels, which take what it flags)
    auto launch_den_check = [&](hipStream_t st) -> int {
        if (!den || robust_env == 0) return CRF_OK;
        hipLaunchKernelGGL(crf_den_check_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, st, p, (res || fac) ? 1 : 0);
        if ((e = hipGetLastError()) != hipSuccess) { set_error(std::string("crf_den_check_kernel: ") + hipGetErrorString(e)); return CRF_ERR_HIP; }
        return CRF_OK;
    };
    // the denominator recursions of the whole batch on `st` (every layout; both directions per launch)
    auto launch_den = [&](hipStream_t st, bool with_check = true) -> int {
        prof_mark(1, false, st); prof_mark(2, false, st);
        int r2 = CRF_OK;
        const CoresGuard cores((fac && FX->K > 1) || (res && !fac && h->dev.res.K > 1), cx->dev, st);
        if (fac && FX->K > 1) {
            const int grp = std::max(1, ncu_dev / 4);
            for (int b0 = 0; b0 < (int)B && !r2; b0 += grp) r2 = launch_fac2_pair(p, lds_fac, st, b0, std::min(grp, (int)B - b0));
        } else if (fac && pair2) {
            r2 = launch_fac_pair2<false>(p, lds_fac, st, started);
        } else if (fac) {
            r2 = launch_fac_pair<false>(p, lds_fac, st, started, 0, (int)T, fstate, bstate);
        } else if (res) {
            // K CUs per utterance and direction exchange the state vector through L2 every frame.  With K > 1 every
            // workgroup of a launch must be resident at once (its peers spin on it): groups of at most CUs/(2K) utterances.
            const int K = h->dev.res.K;
            const int grp = K > 1 ? std::max(1, ncu_dev / (2 * K)) : (int)B;
            const size_t l = std::max(res_lds_bytes(h, (int)V, 0, h->res_rows_cu_f), res_lds_bytes(h, (int)V, 1, h->res_rows_cu_b));
            for (int b0 = 0; b0 < (int)B && !r2; b0 += grp) r2 = launch_res_pair(p, l, b0, std::min(grp, (int)B - b0), st);
        } else if (gv) {
            r2 = launch_den_pair<true>(p, std::max(chain_lds_bytes(h, (int)V, Sc, 0, true), chain_lds_bytes(h, (int)V, Sc, 1, true)), st);
            return (r2 || !with_check) ? r2 : launch_den_check(st);
        } else {
            r2 = launch_den_pair<false>(p, std::max(chain_lds_bytes(h, (int)V, Sc, 0), chain_lds_bytes(h, (int)V, Sc, 1)), st);
            return (r2 || !with_check) ? r2 : launch_den_check(st);
        }
        prof_mark(1, true, st); prof_mark(2, true, st);
        return (r2 || !with_check) ? r2 : launch_den_check(st);
    };

    // Three streams (round 5, switch grad_par3; OFF): the numerator half of the grad pass (side stream) and the staged den half (third stream)
    // run BESIDE each other and both ADD into gradient rows the prep kernel has zeroed (0 + x + y: two addends per element, the same bits
    // in either order).  The idea: on one stream the stages queue behind the numerator chains and their grad half, and graphs whose
    // recursions are shorter than that (S = 513: recursions 1.25 ms, step 1.92) wait for them.  Measured SLOWER everywhere
    // (profiles/round5_ab_three_streams.txt: metric 2.79 -> 2.85 ms, S = 513 1.92 -> 2.01, V = 217 3.39 -> 3.51, estimated S = 3 006 2.44 -> 2.78):
    // the stage workgroups then share the free CUs with the numerator chains -- a serial fp64 latency chain whose frames get longer -- and
    // what the stages gain by starting early the chains lose.  Not when a recent call needed the numerator's log-domain fallback (the
    // third stream then carries its chains), nor in segment mode.
    const int aux_env0 = opt(kOpt_aux_stream, -1);
    const int seen0 = cx->seen ? *(volatile int *)cx->seen : 0;
    const bool ctc_wants_aux = aux_env0 >= 0 ? aux_env0 != 0 : (seen0 > 0 && p.call_id - seen0 <= 16);
    const bool par3 = staged && !segmode && cx->aux != nullptr && !(robust_env != 0 && ctc_wants_aux) && opt(kOpt_grad_par3, 0) != 0;
    p.zero_grad = par3 ? 1 : 0;
    for (bool &u : g_prof.used) u = false;
    prof_mark(7, false, stream);
    prof_mark(0, false, stream);
    if (V <= 256) hipLaunchKernelGGL(crf_prep_kernel<16>, dim3((unsigned)((frames + 15) / 16)), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL(crf_prep_kernel<64>, dim3((unsigned)((frames + 3) / 4)), dim3(256), 0, stream, p);
    prof_mark(0, true, stream);
    LAUNCH_CHECK("crf_prep_kernel");
    if (res && (w.xch_bytes > 0 || !have_flags)) {  // exchange granules (tags) and the error word start at zero in every call
        if ((e = hipMemsetAsync(p.xch, 0, (size_t)w.xch_bytes + 256 + 8 * (size_t)B, stream)) != hipSuccess) { set_error("hipMemsetAsync(xch)"); return CRF_ERR_HIP; }
    }

    if (bat) {
        // Utterance-minor denominator (large graphs): one launch per frame on the caller's stream, forward step of
        // frame j and backward step of frame T - j together; the numerator pair runs beside them on the side stream.
        BatchParams bp{};
        bp.g = h->dev.bat; bp.start_lin = h->dev.start_lin; bp.end_lin = h->dev.end_lin;
        bp.S = h->dev.S; bp.P = h->dev.P; bp.B = (int)B; bp.Bp = (int)w.Bp; bp.T = (int)T; bp.V = (int)V; bp.max_label = h->dev.max_label;
        bp.lx = lx; bp.ep = p.ep; bp.moff = p.moff;
        bp.ept = (float *)(base + w.off_ept); bp.Af = (float *)(base + w.off_Af); bp.Zb = (float *)(base + w.off_Zb);
        bp.Q = p.Q; bp.BP = p.BP;
        unsigned *bsm = (unsigned *)(base + w.off_bsm);
        bp.mxf = bsm; bp.mxb = bsm + 3 * w.Bp; bp.Ef = (int *)(bsm + 6 * w.Bp); bp.Fb = (int *)(bsm + 7 * w.Bp);
        bp.zs = (float *)(bsm + 8 * w.Bp); bp.zb = (float *)(bsm + 9 * w.Bp);
        bp.den_zs = p.den_zs; bp.cost_alpha = p.cost_alpha; bp.cost_beta = p.cost_beta; bp.den_ez = p.den_ez; bp.redo = p.redo;
        bp.grad = grad; bp.c_den = c_den;
        if (ctc) {
            if ((rc = fork_side())) return rc;
            if ((rc = launch_ctc_pair(p, lds_ctc, side, max_label_len))) return rc;
        }
        const unsigned ngrp = (unsigned)(w.Bp / w.UL);
        bp.ngrp = (int)ngrp;
        // one task per wave, ONE round of workgroups (a second round with a fraction of the device doubled the launch):
        // the tasks wanted per direction follow from the occupancy the runtime reports, shared by the combos
        const int64_t ncombo = 2 * (int64_t)ngrp;
        const bool bfac = stream_fac(g->h, w.UL);                  // factored streams (T o LM graphs, groups of >= 32 utterances)
        int wg_cu = 0;
        {
            const void *fn = w.UL == 64 ? (bfac ? (const void *)crf_batch_frame_kernel<64, 4, true> : (const void *)crf_batch_frame_kernel<64, 4>)
                           : w.UL == 32 ? (bfac ? (const void *)crf_batch_frame_kernel<32, 4, true> : (const void *)crf_batch_frame_kernel<32, 4>)
                           : w.UL == 16 ? (bfac ? (const void *)crf_batch_frame_kernel<16, 4, true> : (const void *)crf_batch_frame_kernel<16, 4>)
                           : (bfac ? (const void *)crf_batch_frame_kernel<8, 4, true> : (const void *)crf_batch_frame_kernel<8, 4>);
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&wg_cu, fn, kBatThreads, 0) != hipSuccess || wg_cu < 1) { (void)hipGetLastError(); wg_cu = 2; }
        }
        // ... times 70 %: a launch is bound by the L2s and the fabric, not by the CUs, and fewer, longer tasks pay the task set-up
        // (three dependent trips to a cold L2) less often.  Measured, S = 16 385 / B = 64 (repeatable to 0.3 %): 100 / 85 / 70 /
        // 60 / 55 / 45 / 35 % -> 30.7 / 29.5 / 28.8 / 30.6 / 31.8 / 28.5 / 31.3 ms per step (the dips: workgroups per XCD just
        // above a multiple of its 32 CUs); config #5 at B = 8: 100 / 70 / 50 % -> 145.3 / 143.6 / 152.6 ms.  CRF_BAT_FILL overrides.
        const int fill_env = opt(kOpt_bat_fill, 0);
        const int64_t fill = fill_env > 0 && fill_env <= 100 ? fill_env : 70;
        const int want = (int)std::max<int64_t>(16, (int64_t)ncu_dev * wg_cu * kBatWaves * 15 / 16 * fill / 100 / ncombo);
        const StreamDev *sdv = nullptr;
        if ((rc = ensure_stream_tables(g->h, w.UL, want, &sdv))) return rc;
        if ((sdv->fac != 0) != bfac) { set_error("arc streams: factored / plain mismatch"); return CRF_ERR_ARG; }
        bp.SX = h->dev.S + sdv->NU; bp.x_start = sdv->x_start;
        bp.st = *sdv;
        // 8 * nslot workgroups (block b -> XCD b % 8, slot b / 8): every combo gets at least one wave per task of its arc
        // stream (crf_batch_frame_kernel: a combo has nslot * nk or about nslot / ncx workgroups)
        const int64_t tasks_max = std::max({(int64_t)sdv->f.ntasks, (int64_t)sdv->b.ntasks, (int64_t)1});
        const int64_t wg_combo = (tasks_max + kBatWaves - 1) / kBatWaves + ((sdv->f.nrest > 64 || sdv->b.nrest > 64) ? (std::max(sdv->f.nrest, sdv->b.nrest) + 4 * kBatWaves - 1) / (4 * kBatWaves) : 0);
        const unsigned nslot = (unsigned)(ncombo < 8 ? (wg_combo + (8 / ncombo) - 1) / (8 / ncombo) : wg_combo * ((ncombo + 7) / 8));
        const unsigned G = 8 * nslot;
        prof_mark(1, false, stream); prof_mark(2, false, stream);
#define CRF_BAT_UL(KERNEL, GRID, ...)                                                                        \
        switch (w.UL) {                                                                                       \
            case 64: hipLaunchKernelGGL(KERNEL<64>, GRID, dim3(kBatThreads), 0, stream, __VA_ARGS__); break;  \
            case 32: hipLaunchKernelGGL(KERNEL<32>, GRID, dim3(kBatThreads), 0, stream, __VA_ARGS__); break;  \
            case 16: hipLaunchKernelGGL(KERNEL<16>, GRID, dim3(kBatThreads), 0, stream, __VA_ARGS__); break;  \
            default: hipLaunchKernelGGL(KERNEL<8>, GRID, dim3(kBatThreads), 0, stream, __VA_ARGS__); break;   \
        }
        CRF_BAT_UL(crf_batch_transpose_kernel, dim3((unsigned)((V + 63) / 64), (unsigned)T, ngrp), bp);
        hipLaunchKernelGGL(crf_batch_init_kernel, dim3((unsigned)(((int64_t)bp.SX * w.Bp + kBatThreads - 1) / kBatThreads)), dim3(kBatThreads), 0, stream, bp);
        LAUNCH_CHECK("crf_batch_init_kernel");
        g_den_kernel = w.UL == 64 ? (bfac ? "crf_batch_frame_kernel<64,4,true>" : "crf_batch_frame_kernel<64,4,false>")
                     : w.UL == 32 ? (bfac ? "crf_batch_frame_kernel<32,4,true>" : "crf_batch_frame_kernel<32,4,false>")
                     : w.UL == 16 ? (bfac ? "crf_batch_frame_kernel<16,4,true>" : "crf_batch_frame_kernel<16,4,false>")
                     : (bfac ? "crf_batch_frame_kernel<8,4,true>" : "crf_batch_frame_kernel<8,4,false>");
        for (int j = 0; j <= (int)T; ++j) {
            bp.j = j;
            switch (w.UL) {   // (4: batches of gathers in flight per wave; 2 measured 6 % slower, 8 needs more registers than a wave has)
                case 64: if (bfac) hipLaunchKernelGGL((crf_batch_frame_kernel<64, 4, true>), dim3(G), dim3(kBatThreads), 0, stream, bp);
                         else hipLaunchKernelGGL((crf_batch_frame_kernel<64, 4>), dim3(G), dim3(kBatThreads), 0, stream, bp);
                         break;
                case 32: if (bfac) hipLaunchKernelGGL((crf_batch_frame_kernel<32, 4, true>), dim3(G), dim3(kBatThreads), 0, stream, bp);
                         else hipLaunchKernelGGL((crf_batch_frame_kernel<32, 4>), dim3(G), dim3(kBatThreads), 0, stream, bp);
                         break;
                case 16: if (bfac) hipLaunchKernelGGL((crf_batch_frame_kernel<16, 4, true>), dim3(G), dim3(kBatThreads), 0, stream, bp);
                         else hipLaunchKernelGGL((crf_batch_frame_kernel<16, 4>), dim3(G), dim3(kBatThreads), 0, stream, bp);
                         break;
                default: if (bfac) hipLaunchKernelGGL((crf_batch_frame_kernel<8, 4, true>), dim3(G), dim3(kBatThreads), 0, stream, bp);
                         else hipLaunchKernelGGL((crf_batch_frame_kernel<8, 4>), dim3(G), dim3(kBatThreads), 0, stream, bp);
                         break;
            }
        }
        LAUNCH_CHECK("crf_batch_frame_kernel");
        CRF_BAT_UL(crf_batch_zsum_kernel, dim3((unsigned)((h->dev.S + 255) / 256), 1, ngrp), bp);
        hipLaunchKernelGGL(crf_batch_cost_kernel, dim3((unsigned)B), dim3(kBatThreads), 0, stream, bp);
        prof_mark(1, true, stream); prof_mark(2, true, stream);
        if ((rc = launch_den_check(stream))) return rc;
        prof_mark(5, false, stream);
        CRF_BAT_UL(crf_batch_grad_kernel, dim3((unsigned)T, 1, ngrp), bp);
#undef CRF_BAT_UL
        LAUNCH_CHECK("crf_batch_grad_kernel");
        if ((rc = join_side())) return rc;
        if (ctc && (rc = launch_grad_ctc(2, stream))) return rc;
        prof_mark(5, true, stream);
    } else if (staged) {
        // caller's stream: the denominator pair.  Side stream, behind a short bounded start gate: numerator pair, its
        // grad half (writes -c_ctc * gamma_ctc), then the den half of the grad pass stage by stage (adds gamma_den).
        if ((rc = fork_side())) return rc;
        prof_mark(1, false, stream); prof_mark(2, false, stream);
        if (pair2) {
            if ((rc = launch_fac_pair2<true>(p, lds_fac, stream, started, nstage + 1, bound, cx->flags + 16))) return rc;
        } else if (!segmode) {
            if ((rc = launch_fac_pair<true>(p, lds_fac, stream, started, 0, (int)T, fstate, bstate, nstage + 1, bound, cx->flags + 16))) return rc;
        } else {
            for (int k = 0; k < nstage; ++k) {
                if ((rc = launch_fac_pair<false>(p, lds_fac, stream, started, bound[k], bound[k + 1], fstate, bstate))) return rc;
                if ((e = hipEventRecord(cx->ev[k], stream)) != hipSuccess) { set_error(std::string("hipEventRecord(segment): ") + hipGetErrorString(e)); return CRF_ERR_HIP; }
            }
        }
        prof_mark(1, true, stream); prof_mark(2, true, stream);
        if ((rc = launch_den_check(stream))) return rc;
        // hold the numerator back (briefly, bounded) until the den workgroups have their CUs
        hipLaunchKernelGGL(crf_gate_kernel, dim3(1), dim3(1), 0, side, started, (int)den_wgs);
        if ((rc = launch_ctc_pair(p, lds_ctc, side, max_label_len))) return rc;
        prof_mark(5, false, side);
        if ((rc = launch_grad_ctc(par3 ? 3 : 0, side))) return rc;
        if (par3) {
            // the den half of the grad pass on the THIRD stream, stage by stage behind the same stream-level waits, adding with atomics;
            // the side stream goes on with the numerator (fallback chains, if any) and takes the third stream back in behind it
            if ((e = hipStreamWaitEvent(cx->aux, cx->fork, 0)) != hipSuccess) { set_error(std::string("hipStreamWaitEvent(aux fork): ") + hipGetErrorString(e)); return CRF_ERR_HIP; }
            p.grad_den_acc = 2;
            for (int k = 0; k < nstage; ++k) {
                if ((e = hipStreamWaitValue32(cx->aux, cx->flags + 16 + k + 1, (uint32_t)(2 * B), hipStreamWaitValueGte, 0xffffffffu)) != hipSuccess) {
                    // not available here: from the next call on, segments (and no third stream).  This call: wait for the recursions to END
                    (void)hipGetLastError();
                    use_segments = true;
                    if ((e = hipEventRecord(cx->ev[0], stream)) != hipSuccess || (e = hipStreamWaitEvent(cx->aux, cx->ev[0], 0)) != hipSuccess) {
                        set_error(std::string("hipStreamWaitEvent: ") + hipGetErrorString(e)); return CRF_ERR_HIP;
                    }
                }
                if ((rc = launch_grad_den(cx->aux, k + 1))) return rc;
            }
            if ((e = hipEventRecord(cx->ev_b, cx->aux)) != hipSuccess) { set_error(std::string("hipEventRecord(aux): ") + hipGetErrorString(e)); return CRF_ERR_HIP; }
            g_call_streams = 3;
            ctc_pass1 = robust_env != 0;
            if (ctc_pass1 && (rc = launch_robust_ctc_chains(side, 1))) return rc;
            if ((e = hipStreamWaitEvent(side, cx->ev_b, 0)) != hipSuccess) { set_error(std::string("hipStreamWaitEvent(aux): ") + hipGetErrorString(e)); return CRF_ERR_HIP; }
            // (the marked frames' posteriors are subtracted behind BOTH halves: a plain read-modify-write of rows nobody adds to any more)
            if (ctc_pass1 && (rc = launch_robust_ctc_fix(side, 1))) return rc;
            prof_mark(5, true, side);
            if ((rc = join_side())) return rc;
        } else {
        // Numerator fallback, pass 1.  The chains of the marked utterances can take as long as the scaled ones did (T = 3 000, L = 500,
        // every utterance marked: 3 ms): on the third stream they run beside the grad stages instead of in front of them, and the
        // marked frames' posteriors are subtracted behind the last stage (the stages ADD gamma_den: the order does not matter).
        // Enqueued BEFORE the stage waits of the side stream: whatever hardware queues the three streams share, the chains' packets
        // precede the wait for their event.
        bool aux_fix = false;
        ctc_pass1 = robust_env != 0;
        if (robust_env != 0) {
            // (the third stream costs the call ~20 us of event traffic whether or not an utterance is marked -- B = 64, T = 1 500: 3.172 ->
            // 3.192 ms -- so it is taken when one of this context's last 16 calls ran the log-domain chains: they write the call's
            // number to a pinned host word, read here without any synchronisation; `aux_stream` 1 / 0 forces it on / off)
            const int aux_env = opt(kOpt_aux_stream, -1);
            const int seen = cx->seen ? *(volatile int *)cx->seen : 0;
            const bool want_aux = aux_env >= 0 ? aux_env != 0 : (seen > 0 && p.call_id - seen <= 16);
            if (cx->aux && want_aux && hipEventRecord(cx->ev_a, side) == hipSuccess && hipStreamWaitEvent(cx->aux, cx->ev_a, 0) == hipSuccess) {
                if ((rc = launch_robust_ctc_chains(cx->aux, 1))) return rc;
                if ((e = hipEventRecord(cx->ev_b, cx->aux)) != hipSuccess) { set_error(std::string("hipEventRecord(aux): ") + hipGetErrorString(e)); return CRF_ERR_HIP; }
                aux_fix = true;
                g_call_streams = 3;
            } else {
                (void)hipGetLastError();
                if ((rc = launch_robust_ctc_chains(side, 1))) return rc;
            }
        }
        p.grad_den_acc = 1;
        // (one launch for the stages 2 ..: behind stage 1's wait -- every recursion has run half of its frames, every den workgroup is resident)
        const bool gd_one = !segmode && nstage >= 3 && !opt_on(kOpt_gd_stage_launches);
        for (int k = 0; k < nstage; ++k) {
            if (gd_one && k == 1) { if ((rc = launch_grad_den(side, 2, true))) return rc; break; }
            if (!segmode) {
                if ((e = hipStreamWaitValue32(side, cx->flags + 16 + k + 1, (uint32_t)(2 * B), hipStreamWaitValueGte, 0xffffffffu)) != hipSuccess) {
                    // not available here: from the next call on, segments.  This call: wait for the recursions to END
                    (void)hipGetLastError();
                    use_segments = true;
                    if ((e = hipEventRecord(cx->ev[0], stream)) != hipSuccess || (e = hipStreamWaitEvent(side, cx->ev[0], 0)) != hipSuccess) {
                        set_error(std::string("hipStreamWaitEvent: ") + hipGetErrorString(e)); return CRF_ERR_HIP;
                    }
                }
            } else if ((e = hipStreamWaitEvent(side, cx->ev[k], 0)) != hipSuccess) {
                set_error(std::string("hipStreamWaitEvent(segment): ") + hipGetErrorString(e)); return CRF_ERR_HIP;
            }
            if ((rc = launch_grad_den(side, k + 1))) return rc;
        }
        // The marked frames' posteriors are subtracted at ONE place whichever stream ran the chains -- behind the last stage -- so that
        // the order of the float additions into `grad` (and with it every bit of the gradient) does not depend on the unsynchronised
        // hint above (round-3 advisor: with the chains on the side stream the subtraction used to precede the stages).
        if (aux_fix && (e = hipStreamWaitEvent(side, cx->ev_b, 0)) != hipSuccess) { set_error(std::string("hipStreamWaitEvent(aux): ") + hipGetErrorString(e)); return CRF_ERR_HIP; }
        if (ctc_pass1 && (rc = launch_robust_ctc_fix(side, 1))) return rc;
        prof_mark(5, true, side);
        if ((rc = join_side())) return rc;   // the last grad launch is behind every stage of the recursions
        }
    } else if (den && ctc && !serial) {
        // Denominator pair on the caller's stream, numerator pair beside it on the side stream -- unless the den
        // workgroups own every CU (register-resident layouts with 2B (x K) >= CUs): then the numerator recursions run
        // beside the DEN HALF of the grad pass instead (HBM-bound, small workgroups that share CUs happily).
        const int ctc_after_env = opt(kOpt_ctc_after, -1);
        // (factored, one CU per recursion: the den grid leaves ncu - den_wgs CUs free -- B = 96: 64 of them -- and the numerator
        // chains, four workgroups to a CU, run there beside it, behind the start gate so that the den workgroups get their CUs first)
        const bool fac1 = fac && FX->K == 1;
        // (round 5: not only when the den grid owns EVERY CU.  Between the staged schedule's limit -- three quarters of the CUs -- and a full device the
        // chains ran beside the recursions on the few CUs they leave, 2 B chain workgroups on 256 - 2 B CUs, and took longer than the recursions: B = 100 /
        // 104 / 112 / 120 4.40 / 4.54 / 4.82 / 5.16 ms per step; behind them, beside the den half of the grad pass: 4.28 / 4.38 / 4.48 / 4.61,
        // profiles/round5_ab_grad_one_launch.txt)
        const bool after = ctc_after_env >= 0 ? ctc_after_env != 0 : fac1 ? den_wgs * 100 > (int64_t)ncu_dev * opt(kOpt_stage_fill, 75) : (res && h->dev.res.K > 1);
        if (!after && fac1 && have_flags) {
            if ((rc = fork_side())) return rc;
            if ((rc = launch_den(stream))) return rc;
            hipLaunchKernelGGL(crf_gate_kernel, dim3(1), dim3(1), 0, side, started, (int)den_wgs);
            if ((rc = launch_ctc_pair(p, lds_ctc, side, max_label_len))) return rc;
            if ((rc = join_side())) return rc;
            prof_mark(5, false, stream);
            if (fast_den) {
                if ((rc = launch_grad_den(stream, 0))) return rc;
                if ((rc = launch_grad_ctc(2, stream))) return rc;
            } else {
                p.grad_phase = 0;
                hipLaunchKernelGGL(crf_grad_kernel, ggrid, dim3(kGradThreads), lds_grad, stream, p);
                LAUNCH_CHECK("crf_grad_kernel");
            }
            prof_mark(5, true, stream);
        } else if (!after) {
            if ((rc = fork_side())) return rc;
            if ((rc = launch_ctc_pair(p, lds_ctc, side, max_label_len))) return rc;
            if ((rc = launch_den(stream))) return rc;
            if ((rc = join_side())) return rc;
            prof_mark(5, false, stream);
            if (fast_den) {
                if ((rc = launch_grad_den(stream, 0))) return rc;
                if ((rc = launch_grad_ctc(2, stream))) return rc;
            } else {
                p.grad_phase = 0;
                hipLaunchKernelGGL(crf_grad_kernel, ggrid, dim3(kGradThreads), lds_grad, stream, p);
                LAUNCH_CHECK("crf_grad_kernel");
            }
            prof_mark(5, true, stream);
        } else {
            // (the consistency check BEHIND the fork: the numerator chains -- a latency chain that wants its workgroups resident at once -- are
            // released by the end of the recursions and get the CUs first; with the check in front of the fork the grad launch below, which
            // follows it on this stream without an event in between, filled the device first: B = 128 ctc chains 1.5 -> 2.4 ms)
            if ((rc = launch_den(stream, false))) return rc;
            if ((rc = fork_side())) return rc;            // numerator pair starts when the den recursions have drained
            if ((rc = launch_den_check(stream))) return rc;
            if ((rc = launch_ctc_pair(p, lds_ctc, side, max_label_len))) return rc;
            prof_mark(5, false, stream);
            if (fast_den) {
                if ((rc = launch_grad_den(stream, 0))) return rc;
            } else {
                p.grad_phase = 1;
                hipLaunchKernelGGL(crf_grad_kernel, ggrid, dim3(kGradThreads), lds_grad, stream, p);
                LAUNCH_CHECK("crf_grad_kernel(den)");
            }
            if ((rc = join_side())) return rc;
            if ((rc = launch_grad_ctc(2, stream))) return rc;
            prof_mark(5, true, stream);
        }
    } else {
        // one stream: den only (gpu_den), numerator only (gpu_ctc / WARP_CTC_LOSS), or no side stream available
        if (den && (rc = launch_den(stream))) return rc;
        if (ctc && (rc = launch_ctc_pair(p, lds_ctc, stream, max_label_len))) return rc;
        prof_mark(5, false, stream);
        if (den && fast_den) {
            if ((rc = launch_grad_den(stream, 0))) return rc;
            if (ctc && (rc = launch_grad_ctc(2, stream))) return rc;
        } else if (!den && fast_ctc) {
            if ((rc = launch_grad_ctc(0, stream))) return rc;
        } else {
            p.grad_phase = 0;
            hipLaunchKernelGGL(crf_grad_kernel, ggrid, dim3(kGradThreads), lds_grad, stream, p);
            LAUNCH_CHECK("crf_grad_kernel");
        }
        prof_mark(5, true, stream);
    }
    bool fin_folded = false;
    if (den && robust_env != 0) {
        // Fallback for utterances whose scaled-fp32 recursion lost all its mass: redone in a per-frame log-shifted form
        // (crf_robust_den_kernel).  Workgroups of unflagged utterances leave at once -- two near-empty launches per call.
        static LdsMark mr, mrg, mg;
        const bool rgv = w.gv_robust;
        const size_t lr = robust_lds_bytes(h, (int)V, rgv);
        if (rgv) {
            if ((rc = ensure_lds((const void *)crf_robust_den_kernel<true>, lr, mrg, "robust den"))) return rc;
            hipLaunchKernelGGL(crf_robust_den_kernel<true>, dim3((unsigned)(2 * B)), dim3(kChainThreads), lr, stream, p);
        } else {
            if ((rc = ensure_lds((const void *)crf_robust_den_kernel<false>, lr, mr, "robust den"))) return rc;
            hipLaunchKernelGGL(crf_robust_den_kernel<false>, dim3((unsigned)(2 * B)), dim3(kChainThreads), lr, stream, p);
        }
        LAUNCH_CHECK("crf_robust_den_kernel");
        const size_t lg = (size_t)rup64(h->dev.NC) * 4 + (size_t)rup64((int)V) * 12 + 64;
        if ((rc = ensure_lds((const void *)crf_robust_grad_kernel, lg, mg, "robust grad"))) return rc;
        // (the call's sums in this launch unless the numerator's second fallback pass, which rewrites costs, is still to come)
        fin_folded = !(ctc && robust_env != 0 && !ctc_pass1) && !opt_on(kOpt_no_fin_fold) && !g_prof.on;
        p.fin_fold = fin_folded ? 1 : 0;
        hipLaunchKernelGGL(crf_robust_grad_kernel, dim3(16, (unsigned)B), dim3(kGradThreads), lg, stream, p);
        p.fin_fold = 0;
        LAUNCH_CHECK("crf_robust_grad_kernel");
    }
    // (the staged schedule's pass 1 ran behind the only kernel that marks frames: nothing is left for a second pass there)
    if (ctc && robust_env != 0 && !ctc_pass1 && (rc = launch_robust_ctc(stream, 2))) return rc;
    prof_mark(6, false, stream);
    if (!fin_folded) hipLaunchKernelGGL(crf_finalize_kernel, dim3(1), dim3(256), 0, stream, p);
    prof_mark(6, true, stream);
    prof_mark(7, true, stream);
    g_prof.have = g_prof.on;
    LAUNCH_CHECK("crf_finalize_kernel");
#undef LAUNCH_CHECK
    return CRF_OK;
}

int crf_debug_stage_plan(int64_t T, int64_t B, int32_t *out, int n_out) {
    if (!out || n_out < 4 + 4 * (kMaxStages + 2) || T < 1 || B < 1) { set_error("crf_debug_stage_plan: out needs 4 + 4 * 18 ints"); return CRF_ERR_ARG; }
    int bound[kMaxStages + 2] = {0}, poff[kMaxStages + 2] = {0}, fpb[kMaxStages + 2] = {0}, gd_piece = 0;
    const bool segmode = opt_on(kOpt_segments);
    const int stages_env = opt(kOpt_stages, 0);
    const int nstage = plan_grad_stages(T, segmode, stages_env, stages_env > 0 ? stages_env : (segmode ? 4 : 12), bound, &gd_piece);
    const bool one = !segmode && nstage >= 3 && !opt_on(kOpt_gd_stage_launches);
    const int64_t tot = one ? plan_grad_grid(bound, nstage, 2, B, gd_piece, poff, fpb) : 0;
    out[0] = nstage; out[1] = gd_piece; out[2] = one ? 1 : 0; out[3] = (int32_t)tot;
    for (int k = 0; k < kMaxStages + 2; ++k) {
        out[4 + k] = bound[k];
        out[4 + (kMaxStages + 2) + k] = poff[k];
        out[4 + 2 * (kMaxStages + 2) + k] = fpb[k];
        out[4 + 3 * (kMaxStages + 2) + k] = k >= 1 && k <= nstage ? (bound[k] - bound[k - 1] + kGDFrames - 1) / kGDFrames + 3 : 0;   // candidates per run (nfc)
    }
    return CRF_OK;
}

int crf_stage_i32(int32_t *dst_dev, const int32_t *src_pinned_host, int64_t n, void *stream) {
    if (n <= 0) return CRF_OK;
    if (!dst_dev || !src_pinned_host) { set_error("crf_stage_i32: null pointer"); return CRF_ERR_ARG; }
    hipLaunchKernelGGL(crf_stage_i32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dst_dev, src_pinned_host, n);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error(std::string("crf_stage_i32: ") + hipGetErrorString(e)); return CRF_ERR_HIP; }
    return CRF_OK;
}

int crf_timing_read(unsigned long long *out, int n) {
#ifdef CRF_TIMING
    if (!out || n <= 0) return 0;
    if (n > 16384) n = 16384;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tm), (size_t)n * sizeof(unsigned long long)) != hipSuccess) return 0;
    return n;
#else
    (void)out; (void)n;
    return 0;  // not a timing build
#endif
}

int crf_last_fallback_counts(int32_t *out2, void *stream) {
    if (!out2) { set_error("crf_last_fallback_counts: null pointer"); return CRF_ERR_ARG; }
    out2[0] = out2[1] = -1;
    if (!g_last_err_word) { set_error("crf_last_fallback_counts: no call in this thread yet"); return CRF_ERR_ARG; }
    hipError_t e = hipStreamSynchronize((hipStream_t)stream);
    if (e == hipSuccess) e = hipMemcpy(out2, g_last_err_word + kFlagFallback, 2 * sizeof(int32_t), hipMemcpyDeviceToHost);
    if (e != hipSuccess) { set_error(std::string("crf_last_fallback_counts: ") + hipGetErrorString(e)); return CRF_ERR_HIP; }
    return CRF_OK;
}

const char *crf_build_switches(void) {
    return "LAG=" CRF_STR(CRF_X_LAG) " KCLATE=" CRF_STR(CRF_X_KCLATE) " PRIO=" CRF_STR(CRF_X_PRIO) " EARLY=" CRF_STR(CRF_X_EARLY)
           " GFIRST=" CRF_STR(CRF_X_GFIRST) " GCHK=" CRF_STR(CRF_X_GCHK) " CTCSUM=" CRF_STR(CRF_X_CTCSUM) " GDEARLY=" CRF_STR(CRF_X_GDEARLY) " GDMOVE=" CRF_STR(CRF_X_GDMOVE) " GDW2=" CRF_STR(CRF_X_GDW2)
#ifdef CRF_TIMING
           " TIMING=1"
#endif
        ;
}

const char *crf_last_den_kernel(void) { return g_den_kernel; }
int crf_last_call_streams(void) { return g_call_streams; }
const char *crf_last_side_stream(void) { return g_side_desc; }

void crf_profile_enable(int on) { g_prof.on = on != 0; if (!on) g_prof.have = false; }

int crf_profile_read(float *ms_out, int n) {
    if (!ms_out || n <= 0 || !g_prof.have) return 0;
    int w = 0;
    for (int s = 0; s < 8 && s < n; ++s, ++w) {
        ms_out[s] = -1.f;
        if (!g_prof.used[s]) continue;
        if (hipEventSynchronize(g_prof.ev[2 * s + 1]) != hipSuccess) continue;
        float ms = -1.f;
        if (hipEventElapsedTime(&ms, g_prof.ev[2 * s], g_prof.ev[2 * s + 1]) == hipSuccess) ms_out[s] = ms;
    }
    return w;
}

}  // extern "C"
