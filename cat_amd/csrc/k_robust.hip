// cat_amd/csrc/k_robust.hip -- log-domain fallbacks, forward/backward consistency check, finalize
// (one translation unit per kernel family, built in parallel by cat_amd/build.py; the explicit instantiations at the end are the
//  ones the host side in crf_host.hip launches -- a missing one is a link error, -Wl,-z,defs)
#include "crf_device.h"
#include "crf_kernels_decl.h"

namespace crf {

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
// The fallback's ROWS (ln q_t[p], ln b_{t+1}[dst_p]) in the workspace's 32-bit row entries: FIXED-POINT relative to the row's maximum, q = round((x - M) / step) with
// step = max(M - min finite x, 1) * 2^-30 kept per frame in the exponent words EQ / EB (which the redone utterance's fast rows no longer need) -- an absolute
// error of step / 2 on a logarithm whatever its distance from the maximum (3e-6 for a row spanning 3 000 nats).  Rounds 5 / 6 stored (float)(x - M): 6e-8 * |x - M|, and a
// pair 800 nats below its row's forward maximum that the backward row makes the frame's dominant one came out 1.2e-4 off (fuzz campaign 3, seed 116: sigma 20, T = 195).
__device__ __forceinline__ float robust_step(double mx, double mn) { return (float)(fmax(mx > -INFINITY ? mx - mn : 0.0, 1.0) * 0x1p-30); }
__device__ __forceinline__ int robust_enc(double x, double M, float step) { return x > -INFINITY ? (int)rint((x - M) / (double)step) : (int)0x80000000; }
__device__ __forceinline__ double robust_dec(int q, float step) { return q != (int)0x80000000 ? (double)q * (double)step : -INFINITY; }
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
// Rows for crf_robust_grad_kernel: lq_t[p] and lb_t[p] as 32-bit fixed-point logarithms relative to their frame's maximum, step per frame in EQ / EB (robust_enc; the constants cancel in the frame's
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
        int *Qrow = (int *)(p.Q + (bt0 + t) * p.Rq);
        double smax = -INFINITY, smin = INFINITY;
        for (int i = sl0; i < sl1; ++i) {
            const int j = __builtin_amdgcn_readfirstlane(g.fwd.wave_slices[i]);
            const int o = __builtin_amdgcn_readfirstlane(g.fwd.slice_off[j]);
            const int w2 = __builtin_amdgcn_readfirstlane(g.fwd.slice_w2[j]);
            const int r = j * kWave + lane;
            const double lq = g.pair_meta[r].x >= 0 ? ell_row_lse(g.fwd.arcs + o + lane, w2, Xc) : -INFINITY;
            Ql[r] = lq;
            smax = fmax(smax, lq);
            if (lq > -INFINITY) smin = fmin(smin, lq);
        }
        const double M = block_max_d(smax, red, tid);       // (its barriers also publish Ql and Dv)
        const float step = robust_step(M, -block_max_d(-smin, red, tid));
        if (tid == 0) p.EQ[bt0 + t] = __float_as_int(step);
        for (int r = tid; r < Pr; r += kChainThreads) Qrow[r] = robust_enc(Ql[r], M, step);
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
        int *BProw = (int *)(p.BP + (bt0 + lx - 1) * p.Rb);
        double smax = -INFINITY, smin = INFINITY;
        for (int r = tid; r < Pr; r += kChainThreads) {
            const int2 meta = g.pair_meta[r];
            const double lb = (meta.x >= 0 && g.end_lin[meta.x] > 0.f) ? log((double)g.end_lin[meta.x]) : -INFINITY;
            BPst[Pr + r] = lb;
            smax = fmax(smax, lb);
            if (lb > -INFINITY) smin = fmin(smin, lb);
        }
        const double M = block_max_d(smax, red, tid);
        const float step = robust_step(M, -block_max_d(-smin, red, tid));
        if (tid == 0) p.EB[bt0 + lx - 1] = __float_as_int(step);
        for (int r = tid; r < Pr; r += kChainThreads) {
            const double lb = BPst[Pr + r];
            BProw[r] = robust_enc(lb, M, step);
            Z[r] = lb > -INFINITY ? Dv[g.pair_meta[r].y & 0xffff] + lb : -INFINITY;
        }
    }
    __syncthreads();
    double zm = -INFINITY;
    float zs = 0.f;
    double Mprev = 0.0;
    float step_prev = 1.f;
    const int sl0 = g.bwd.wave_off[wave], sl1 = g.bwd.wave_off[wave + 1];
    for (int i = 0; i < lx; ++i) {
        const int t = lx - 1 - i;
        const double *Zc = Z + (size_t)(i & 1) * Pr;
        double *Zn = Z + (size_t)((i + 1) & 1) * Pr;
        double *BPc = BPst + (size_t)(i & 1) * Pr;
        if (t >= 1) load_drow(p, bt0 + t - 1, Dv, tid);      // (its last readers are behind the previous iteration's closing barrier)
        if (i > 0) {  // log b_{t+1}[dst_p], staged by the previous iteration -> BP[b][t]
            const double *BPp = BPst + (size_t)((i - 1) & 1) * Pr;
            int *BProw = (int *)(p.BP + (bt0 + t) * p.Rb);
            if (tid == 0) p.EB[bt0 + t] = __float_as_int(step_prev);
            for (int r = tid; r < Pr; r += kChainThreads) BProw[r] = robust_enc(BPp[r], Mprev, step_prev);
        }
        for (int r = tid; r < Pr; r += kChainThreads) BPc[r] = -INFINITY;   // (pairs into states without a backward row)
        __syncthreads();
        double smax = -INFINITY, smin = INFINITY;
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
                if (lb > -INFINITY) smin = fmin(smin, lb);
                if (meta.y == 1) BPc[meta.z] = lb;
                else for (int pi = g.st_pair_off[s2]; pi < g.st_pair_off[s2 + 1]; ++pi) BPc[g.st_pairs[pi]] = lb;
            }
        }
        if (t >= 1) {
            Mprev = block_max_d(smax, red, tid);            // (barriers: BPc complete)
            step_prev = robust_step(Mprev, -block_max_d(-smin, red, tid));
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
        const int *Qr = (const int *)(p.Q + (bt0 + t) * p.Rq), *Br = (const int *)(p.BP + (bt0 + t) * p.Rb);   // ln q_t[p], ln b_{t+1}[dst_p]: fixed-point relative to the frame's maxima (robust_enc)
        const float sq = __int_as_float(p.EQ[bt0 + t]), sb = __int_as_float(p.EB[bt0 + t]);
        auto chunk_lse = [&](int c) {
            double m = -INFINITY;
            float sm = 0.f;
            for (int j = g.chunk_off[c]; j < g.chunk_off[c + 1]; ++j) { const int r = g.perm[j]; lse_add(m, sm, robust_dec(Qr[r], sq) + robust_dec(Br[r], sb)); }
            return lse_value(m, sm);
        };
        // the chunks' sums relative to the LARGEST of them, so that the float in the LDS is near 0 -- 1e-7 absolute -- where it matters (relative to the rows' two
        // maxima the frame's dominant pair itself can sit hundreds of nats down: fp32 then keeps 6e-5).  Two passes over the rows: this is the fallback.
        double cmx = -INFINITY;
        for (int c = tid; c < NC; c += kGradThreads) cmx = fmax(cmx, chunk_lse(c));
        cmx = bmax(cmx);
        for (int c = tid; c < NC; c += kGradThreads) {
            const double v = chunk_lse(c);
            csum[c] = v > -INFINITY ? (float)(v - cmx) : -INFINITY;
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


// ---- explicit instantiations ----
template __global__ void crf_robust_den_kernel<false>(LossParams);
template __global__ void crf_robust_den_kernel<true>(LossParams);
template __global__ void crf_robust_ctc_kernel<1>(LossParams);
template __global__ void crf_robust_ctc_kernel<2>(LossParams);
template __global__ void crf_robust_ctc_kernel<4>(LossParams);
template __global__ void crf_robust_ctc_kernel<kCtcRegs>(LossParams);

}  // namespace crf
