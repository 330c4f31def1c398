// cat_amd/csrc/k_chain.hip -- prep, metadata staging, the streaming denominator chains, the numerator (CTC) chains and their consistency check
// (one translation unit per kernel family, built in parallel by cat_amd/build.py; the explicit instantiations at the end are the
//  ones the host side in crf_host.hip launches -- a missing one is a link error, -Wl,-z,defs)
#include "crf_device.h"
#include "crf_kernels_decl.h"

namespace crf {

// ---------------------------------------------------------------------------------------------
// prep: e[b][t][v] = exp(logp[b][t][v] - max_v) * 2^kEpExp, mx[b][t] = max_v   (one wave per frame)
// ---------------------------------------------------------------------------------------------


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

// Sum of one ELL row per lane: sum_k x[idx_k] * w_k over n 16-byte elements (2 arcs each) that are
// kWave elements apart.  n is wave-uniform; the host pads n to a multiple of 4 whenever n > 2, so
// the stream is consumed in groups of four elements with the NEXT group already in flight (4 KiB per
// wave, 64 KiB per CU): the L2 latency of the arc stream overlaps the LDS gathers of the group
// that has landed.  hipcc folds a source-level prefetch loop back into load->wait->use, so the
// group loads are issued from inline asm (invisible to its scheduler) and waited for explicitly; two
// register sets alternate.  Loads return in order, so compiler-issued loads/stores in between only
// make either side's waits more conservative (cdna_hip_programming.md 5.7).
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
// (LDS carve: CtcLds / ctc_carve, crf_device.h -- shared with the log-domain fallback)
// ---------------------------------------------------------------------------------------------
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


// Do the two numerator chains agree?  The posteriors of frame 0 -- A_0 is the chain's start, exact; Bx_0 the END of the backward chain; Z the
// end of the forward chain -- sum to one iff neither chain has lost the path mass on its way.  The rescaled fp64 rows end 2^-1074 below
// their frame's maximum: with network outputs a hundred nats apart per frame a chain can drop the states of the eventually dominant
// alignment, and every frame BEHIND the loss then looks consistent (its posteriors sum to one -- over the surviving alignments) while
// being wrong (tests/test_gpu_fuzz.py, round 5).  Such an utterance is redone WHOLE in the log domain (redo_ctc = 2), decided here, in
// front of the grad pass, so that every block of it sees the same verdict.  One workgroup per utterance.
constexpr double kCtcCheckTol = 1e-6;
__global__ __launch_bounds__(256) void crf_ctc_check_kernel(LossParams p) {
    __shared__ double red[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int lx = p.lx[b];
    if (lx <= 0 || p.invalid[b] || p.redo_ctc[b] == 2) return;
    const double zc = p.ctc_zc[b];
    if (!(zc > 0.0)) return;
    const int64_t bt0 = (int64_t)b * p.T;
    const int Sx = 2 * p.ly[b] + 1;
    const double invc = 1.0 / zc;
    // frame 0 (A_0 exact, Bx_0 the end of the backward chain, Z the end of the forward chain) and -- round 6 -- the quarter points: at frame t the sum pairs the
    // forward chain's first t frames with the backward chain's last lx - t, so a set of alignments that BOTH chains drop (the forward one before t, the backward one
    // behind it -- frame 0 then sees the same reduced mass on both sides) is missing on one side only.  A quarter point whose own products sit at the bottom of the
    // fp64 range says nothing (ctc_frame_factor marks such frames for the per-frame fix) and is skipped; frame 0 in that state sends the utterance to the log domain.
    bool bad = false;
    for (int q = 0; q < 4 && !bad; ++q) {
        const int t = (int)((int64_t)lx * q / 4);
        if (q > 0 && t == (int)((int64_t)lx * (q - 1) / 4)) continue;
        const int e = p.ctc_ez[b] - p.ECA[bt0 + t] - p.ECB[bt0 + t];
        const bool unsafe = e + ilogb(invc) > kCtcSafeExp;       // (uniform)
        if (unsafe) { bad = q == 0; continue; }
        const double fc = ldexp(invc, e);
        const double *Ar = p.CA + (bt0 + t) * p.Sc, *Br = p.CB + (bt0 + t) * p.Sc;
        double part = 0.0;
        for (int s = tid; s < Sx; s += 256) part += Ar[s] * Br[s] * fc;
        part = wave_sum_d(part);
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6] = part;
        __syncthreads();
        const double tot = red[0] + red[1] + red[2] + red[3];
        // (1e-6 since round 6: two fp64 chains over the same emissions agree to ~1e-12; round 5's 1e-3 let a chain that had lost 2e-4 of the mass pass -- posterior
        //  rows summing to 1.0002, gradient 2e-4 off: fuzz campaign 1, fused case 11, sigma 20, T = 288)
        if (!(fabs(tot - 1.0) <= kCtcCheckTol)) bad = true;
    }
    if (tid == 0 && bad) atomicMax(&p.redo_ctc[b], 2);
}


// ---- explicit instantiations ----
template __global__ void crf_prep_kernel<16>(LossParams);
template __global__ void crf_prep_kernel<64>(LossParams);
template __global__ void crf_den_pair_kernel<false>(LossParams);
template __global__ void crf_den_pair_kernel<true>(LossParams);
template __global__ void crf_ctc_pair_kernel<1>(LossParams);
template __global__ void crf_ctc_pair_kernel<2>(LossParams);
template __global__ void crf_ctc_pair_kernel<4>(LossParams);
template __global__ void crf_ctc_pair_kernel<kCtcRegs>(LossParams);

}  // namespace crf
