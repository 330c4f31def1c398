// cat_amd/csrc/crf_host.hip -- the host side of the CTC-CRF loss: workspace layout, streams and schedule, kernel launches, the C-ABI entry
// points of include/ctc_crf_hip.h.  The kernels live in one translation unit per family (cat_amd/build.py compiles them in parallel):
//   k_chain.hip      prep (e' = exp(logp - rowmax) * 2^64), metadata staging, streaming denominator chains, numerator (CTC) chains in fp64
//   k_fac_*.hip      register-resident denominator recursions on the factored layout of T o LM graphs (k_fac_body.h): the dominant kernels
//   k_res.hip        register-resident recursions, generic layout over K compute units
//   k_batch.hip      utterance-minor kernels for graphs beyond the registers (one launch per frame, or one persistent launch)
//   k_grad.hip       the grad pass (label posteriors from the stored rows; numerator half; fused combine)
//   k_robust.hip     log-domain fallbacks, forward/backward consistency check, finalize
// What is computed (semantics of the reference):
//   denominator  den_calculate.cu:63-261   alpha/beta over the den graph, logZ, arc posteriors -> labels
//   numerator    gpu_ctc_kernels.h:87-458  CTC alpha/beta on log-probs (blank 0), label posteriors
//   combine      ctc_crf/__init__.py:78-87 grad = c_den*gamma_den - c_ctc*gamma_ctc, loss likewise
// How: linear domain with an exact per-frame power-of-two rescale instead of per-arc log1p(exp()) (den_calculate.cu:29-35);
// the den graph factored through "pairs" p = (destination state, label):
//     forward   q_t[p] = sum_{arcs k in p} a_t[src_k] * w_k,   a_{t+1}[dst_p] = e_t[lab_p] * q_t[p]
//     backward  b_t[s] = sum_{arcs k out of s} w_k * z_t[pair_k],   z_t[p] = e_t[lab_p] * b_{t+1}[dst_p]
//     posterior gamma_den[t][v] = e_t[v] * sum_{p: lab_p = v} q_t[p] * b_{t+1}[dst_p] / Z     (no arc pass: two stored rows per frame)
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

#include "crf_device.h"
#include "crf_kernels_decl.h"

namespace crf {

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
    w.off_bsm = o; o = al(o + (w.bat ? (12 * w.Bp + 640) * 4 : 0));   // mxf[3], mxb[3], Ef, Fb, zs, zb | grid barrier words [512], time-out word (persistent launch)
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
    int *hostw = nullptr;     // pinned, mapped host words: [0] = seen (when there is a third stream), [4] = the one-launch grad pass has timed out
    bool gd_fallback = false; // ... and this context has gone back to one grad launch per stage
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
    {
        void *hp = nullptr;
        if (hipHostMalloc(&hp, 64, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess) { c->hostw = (int *)hp; for (int i = 0; i < 16; ++i) c->hostw[i] = 0; }
        else (void)hipGetLastError();
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
        if (c->aux && c->hostw) c->seen = c->hostw;
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
static int plan_grad_stages(int64_t T, bool segmode, int stages_env, int pieces, int *bound, int *gd_piece_out, int stage_launches = -1) {
    // stage_launches: one grad launch per stage (round 4's schedule) -- the switch gd_stage_launches, or a context whose one-launch grad pass has
    // timed out once (DevCtx::gd_fallback)
    const bool per_stage = stage_launches >= 0 ? stage_launches != 0 : opt_on(kOpt_gd_stage_launches);
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
            piece = per_stage ? 128 : 80;
            while ((kMaxStages - 4) * piece < (int)T - half && piece < (int)T) piece += per_stage ? 128 : 16;   // the stage counters cover T - half
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
        const int taper = stages_env > 0 || segmode || per_stage ? 0 : (opt(kOpt_taper, 32) + kGDFrames - 1) / kGDFrames * kGDFrames;
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
    // (rows of up to 12 288 floats since round 6 -- 512 threads x 6 row registers: a den_lm with a state per seen bigram history has ~5.3 k forward rows
    // at 72 tokens, Rq = Rb = 10 752, 5 % beyond the 10 240 of five registers, and took the generic grad kernel: 10.8 ms of a 17.5 ms step)
    const bool gd_wide6 = den && (w.Rq > 8 * kGDRowRegs * kGDThreads || w.Rb > 8 * kGDRowRegs * kGDThreads);
    const bool fast_den = den && w.Rq <= 8 * 6 * kGDThreads && w.Rb <= 8 * 6 * kGDThreads && w.Rq % 4 == 0 && w.Rb % 4 == 0 && (!gd_wide6 || (gnc <= 2 * kGDThreads && gcap != 8)) &&
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
    // The one-launch grad pass's workgroups wait inside the kernel for the recursions' stage counters (bounded: ~2 s of wall clock).  A time-out --
    // a den launch that aborted, a device shared with a process whose kernels keep the recursions off their CUs -- is written to a pinned host word;
    // a context that has seen one goes back to one grad launch per stage behind stream-level waits (round 4's schedule: nothing waits on the
    // GPU) for good (round-5 advisor)
    if (cx->hostw && !cx->gd_fallback && ((volatile int *)cx->hostw)[4] != 0) {
        cx->gd_fallback = true;
        fprintf(stderr, "[ctc_crf_hip] the one-launch grad pass timed out waiting for the denominator recursions in an earlier call (its loss was NaN): "
                        "this context now launches the grad pass stage by stage behind stream-level waits\n");
    }
    const bool per_stage = opt_on(kOpt_gd_stage_launches) || cx->gd_fallback;
    p.gd_timeout_host = cx->hostw ? cx->hostw + 4 : nullptr;
    if (staged && T >= 256) nstage = plan_grad_stages(T, segmode, stages_env, pieces, bound, &gd_piece, per_stage ? 1 : 0);
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
        } else if (gd_wide6) {        // rows of 10 241 .. 12 288 floats: 512 threads with six row registers each
            static LdsMark set9;
            if ((r2 = ensure_lds((const void *)crf_grad_den_kernel<1, 2, 2 * kGDThreads, kChunk, 6, 1>, l, set9, "grad den"))) return r2;
            hipLaunchKernelGGL((crf_grad_den_kernel<1, 2, 2 * kGDThreads, kChunk, 6, 1>), gg, dim3(2 * kGDThreads), l, st, p);
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
        bp.bar = bsm + 12 * w.Bp; bp.err = (int *)(bsm + 12 * w.Bp + 512);
        bp.den_zs = p.den_zs; bp.cost_alpha = p.cost_alpha; bp.cost_beta = p.cost_beta; bp.den_ez = p.den_ez; bp.redo = p.redo;
        bp.grad = grad; bp.c_den = c_den;
        const unsigned ngrp = (unsigned)(w.Bp / w.UL);
        bp.ngrp = (int)ngrp;
        // one task per wave, ONE round of workgroups (a second round with a fraction of the device doubled the launch):
        // the tasks wanted per direction follow from the occupancy the runtime reports, shared by the combos
        const int64_t ncombo = 2 * (int64_t)ngrp;
        const bool bfac = stream_fac(g->h, w.UL);                  // factored streams (T o LM graphs, groups of >= 32 utterances)
        // ALL frames in one persistent launch (round 6) when its grid is co-resident by the runtime's own count -- switch bat_persist: 0 =
        // one launch per frame (rounds 2 - 5; also the fallback), 1 = persistent (default)
        const bool want_persist = opt(kOpt_bat_persist, 1) != 0;
        auto bat_fn = [&](bool persist) -> const void * {
#define CRF_BAT_FN(K) (w.UL == 64 ? (bfac ? (const void *)K<64, 4, true> : (const void *)K<64, 4, false>)    \
                     : w.UL == 32 ? (bfac ? (const void *)K<32, 4, true> : (const void *)K<32, 4, false>)    \
                     : w.UL == 16 ? (bfac ? (const void *)K<16, 4, true> : (const void *)K<16, 4, false>)    \
                                  : (bfac ? (const void *)K<8, 4, true> : (const void *)K<8, 4, false>))
            return persist ? CRF_BAT_FN(crf_batch_persist_kernel) : CRF_BAT_FN(crf_batch_frame_kernel);
#undef CRF_BAT_FN
        };
        int wg_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&wg_cu, bat_fn(want_persist), kBatThreads, 0) != hipSuccess || wg_cu < 1) { (void)hipGetLastError(); wg_cu = want_persist ? 0 : 2; }
        bool persist = want_persist && wg_cu >= 1;
        if (want_persist && !persist) wg_cu = 2;
        // ... times 70 %: a launch is bound by the L2s and the fabric, not by the CUs, and fewer, longer tasks pay the task set-up
        // (three dependent trips to a cold L2) less often.  Measured, S = 16 385 / B = 64 (repeatable to 0.3 %): 100 / 85 / 70 /
        // 60 / 55 / 45 / 35 % -> 30.7 / 29.5 / 28.8 / 30.6 / 31.8 / 28.5 / 31.3 ms per step (the dips: workgroups per XCD just
        // above a multiple of its 32 CUs); config #5 at B = 8: 100 / 70 / 50 % -> 145.3 / 143.6 / 152.6 ms.  CRF_BAT_FILL overrides.
        const int fill_env = opt(kOpt_bat_fill, 0);
        const int64_t fill = fill_env > 0 && fill_env <= 100 ? fill_env : 70;
        const int64_t slots = (int64_t)ncu_dev * wg_cu;            // workgroups the device holds at once
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
        if (persist && (int64_t)G > slots) persist = false;        // (the grid barrier needs every workgroup resident)
        // The numerator chains run BESIDE the per-frame launches (side stream) but BEHIND the persistent launch, beside the grad pass: a
        // co-resident grid sized for the device's slots must not share them -- with the chains' 2 B workgroups on the CUs, G = 416 of 512
        // slots no longer fitted at once (S = 12 289, B = 64: a CU that holds a chain workgroup has LDS for one workgroup of this kernel,
        // not two), the rest of the grid waited for the chains, and the runtime time-sliced the queues: 1.7 ms per frame instead of 13 us
        // and barrier time-outs (profiles/round6_ab_persistent_batch.txt)
        if (ctc && !persist) {
            if ((rc = fork_side())) return rc;
            if ((rc = launch_ctc_pair(p, lds_ctc, side, max_label_len))) return rc;
        }
        if (opt_on(kOpt_verbose)) fprintf(stderr, "[ctc_crf_hip] utterance-minor: UL %d, %d combos, tasks %d / %d, rest rows %d / %d, grid %u of %lld slots (%d per CU), %s\n", (int)w.UL, (int)ncombo,
                                          sdv->f.ntasks, sdv->b.ntasks, sdv->f.nrest, sdv->b.nrest, G, (long long)slots, wg_cu, persist ? "one persistent launch" : "one launch per frame");
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
        static const char *const kBatNames[2][4][2] = {
            {{"crf_batch_frame_kernel<8,4,false>", "crf_batch_frame_kernel<8,4,true>"}, {"crf_batch_frame_kernel<16,4,false>", "crf_batch_frame_kernel<16,4,true>"},
             {"crf_batch_frame_kernel<32,4,false>", "crf_batch_frame_kernel<32,4,true>"}, {"crf_batch_frame_kernel<64,4,false>", "crf_batch_frame_kernel<64,4,true>"}},
            {{"crf_batch_persist_kernel<8,4,false>", "crf_batch_persist_kernel<8,4,true>"}, {"crf_batch_persist_kernel<16,4,false>", "crf_batch_persist_kernel<16,4,true>"},
             {"crf_batch_persist_kernel<32,4,false>", "crf_batch_persist_kernel<32,4,true>"}, {"crf_batch_persist_kernel<64,4,false>", "crf_batch_persist_kernel<64,4,true>"}}};
        g_den_kernel = kBatNames[persist ? 1 : 0][w.UL == 64 ? 3 : w.UL == 32 ? 2 : w.UL == 16 ? 1 : 0][bfac ? 1 : 0];
        {
            void *args[] = {(void *)&bp};
            const void *fn = bat_fn(persist);
            if (persist) {
                // (co-resident grids of two callers must not interleave: each could become partially resident and wait for the rest)
                CoresGuard guard(true, cx->dev, stream);
                bp.j = 0;
                if ((e = hipLaunchKernel(fn, dim3(G), dim3(kBatThreads), args, 0, stream)) != hipSuccess) { set_error(std::string("crf_batch_persist_kernel: ") + hipGetErrorString(e)); return CRF_ERR_HIP; }
            } else {
                for (int j = 0; j <= (int)T; ++j) {   // (4: batches of gathers in flight per wave; 2 measured 6 % slower, 8 needs more registers than a wave has)
                    bp.j = j;
                    if ((e = hipLaunchKernel(fn, dim3(G), dim3(kBatThreads), args, 0, stream)) != hipSuccess) { set_error(std::string("crf_batch_frame_kernel: ") + hipGetErrorString(e)); return CRF_ERR_HIP; }
                }
            }
        }
        LAUNCH_CHECK("crf_batch_frame_kernel");
        if (ctc && persist) {
            if ((rc = fork_side())) return rc;
            if ((rc = launch_ctc_pair(p, lds_ctc, side, max_label_len))) return rc;
        }
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
        const bool gd_one = !segmode && nstage >= 3 && !per_stage;
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
