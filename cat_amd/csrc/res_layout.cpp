// cat_amd/csrc/res_layout.cpp -- host-side builder of the REGISTER-RESIDENT layout of the denominator
// graph (crf_internal.h: ResDev).  One recursion (forward or backward) of one utterance is split over
// K compute units; every thread keeps its share of the arc list in VGPRs for the whole kernel, so the
// per-frame work is LDS gathers + FMAs only and nothing is streamed from L2 (the streaming kernels of
// crf_kernels.hip remain the fallback for graphs that do not fit).
//
// No reference counterpart: the reference keeps one flat arc array per direction in global memory and
// re-reads it in every one of its T kernel launches (den_calculate.cu:309-355, 75-103, 189-227).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>

#include "../../include/ctc_crf_hip.h"
#include "crf_internal.h"

namespace crf {
namespace {

typedef std::vector<std::vector<std::pair<int, float>>> Rows;

// geometry of one resident workgroup: threads, waves, chunks per thread (registers), words per thread
struct Geom { int threads, waves, nch, words, maxsl, multilane; };   // maxsl: slices per wave (0 = any number); multilane: long rows on adjacent lanes
// Cost model of the row spreading for SMALL graphs ("SMALL graphs" below): a frame costs max(kSpreadSimdWeight x the busiest SIMD's chunk sum,
// kSpreadWaveWeight x the heaviest wave's chunks) -- ~49 cycles per chunk of a saturated SIMD's sum against ~70 per chunk of a wave alone on its SIMD (a batch of
// gathers is an LDS round trip; tools/ubench_issue.py, profiles/round4_ubench_issue.txt).  Calibrated on S = 513 (5 slices on 16 waves: 1.26 -> 0.92 us per frame,
// round 4) and checked there and on a second small graph against the unspread layout by tests/test_gpu_parity.py::test_small_graphs_spread_rows.
constexpr int kSpreadSimdWeight = 5, kSpreadWaveWeight = 7;
// Geometry choice, 1024 x 15 against 768 x 20 / 21 (threads x chunks): graphs with more than kLongRowArcs forward arcs in rows longer than a lane's registers, those
// being more than 1 / kLongRowShareDen of all, keep the 768-thread geometries.  Two data points (den_lm estimated from text, profiles/round4_r4y_point_estimated.txt):
// S = 3 006, 14.9 k of 20.5 k arcs in such rows, is faster on 1024 threads (recursions 2.28 -> 2.13 ms), S = 6 836, 27.3 k of 49.0 k, slower (3.41 -> 3.51): the
// threshold sits between them; `fac_threads=768/1024` overrides.
constexpr size_t kLongRowArcs = 20000, kLongRowShareDen = 5;
constexpr Geom kGeomRes{kResThreads, kResWaves, kResNCH, kResWords, 0, 0};
constexpr Geom kGeomFac512{kResThreads, kResWaves, kResNCH, kResWords, 0, 1};   // factored layout, 512 threads (row constants in LDS)
constexpr Geom kGeomFac3{kFac3Threads, kFac3Threads / kWave, kFac3ArcCh, kFac3NCH * 6, kFac3MaxSl, 1};
// ... and 768 threads with the row constants in an LDS table that the kernel reads one slice AHEAD (crf_kernels.hip, RL): any
// number of slices per wave up to the ten that the 3-bit fields of wave_info.w hold -- graphs with many short rows (a den_lm
// estimated from text: 4 000 rows = 63 slices at 100 k arcs) keep three waves per SIMD instead of falling to 512 threads
constexpr Geom kGeomFac3L{kFac3Threads, kFac3Threads / kWave, kFac3ArcCh, kFac3ArcCh * 6, 10, 1};
// ... the same with ALL 21 chunk slots holding arcs (no row constants in the 21st): 5 % more arc slots, which is what keeps a
// 104 k-arc graph with a few rows of 150 - 500 arcs (V = 143 ... 500 classes) on ONE CU per recursion.  Tried only when the
// 20-chunk table geometry does not take the graph: measured 0.13 - 0.3 us per frame slower on the graphs that fit both
// (S = 513: 2.04 -> 2.52 ms, estimated S = 1 212: 2.04 -> 2.48 ms)
constexpr Geom kGeomFac3L21{kFac3Threads, kFac3Threads / kWave, kFac3LNCH, kFac3LNCH * 6, 10, 1};

// 1024 threads (four waves per SIMD at <= 128 VGPRs), 15 chunks of arcs per thread, row constants in the LDS table: as many arc
// slots as 768 x 20, and as many registers per wave left beside the arcs (128 - 90 against 168 - 126)
constexpr Geom kGeomFac4L{kFac4Threads, kFac4Threads / kWave, kFac4NCH, kFac4NCH * 6, 10, 1};
// 512 threads x 30 chunks with the row constants in the LDS table and implicit entries: the TWO-UTTERANCE kernel's layout (level 6,
// HostGraph::facp).  Two waves per SIMD own 256 registers each: 180 hold arcs, and what a second set of accumulators, 8-byte gather
// results and two epilogues need fits beside them -- the 768-thread version of that kernel (168 registers) spilled 26 - 39 dwords.
constexpr Geom kGeomFac512L{kResThreads, kResWaves, kResNCH, kResWords, 10, 1};

struct DirOut {
    std::vector<unsigned> arcs;   // [K][kResWords][kResThreads]
    std::vector<uint4> wave_info; // [K][kResWaves]
    std::vector<int> row_of;      // rid -> input row (-1 padding)
    std::vector<int> rid_of_row;  // input row -> rid
    std::vector<int> cu_row_off;  // [K+1]
    int64_t slots = 0, conflicts = 0;
    int est_cost = 0;             // max over CUs and waves of (chunks + kEpiCost * slices): the frame-time estimate
    int simd_cost = 0;            // ... and the busiest SIMD's sum of it (waves w, w + 4, ... share a SIMD)
};

inline int chunks_of(size_t deg) { return std::max(1, (int)((deg + kResW - 1) / kResW)); }

// Step 1: decide where every row lives (CU, wave, slice, lane) from the row LENGTHS only.
struct SliceAt { int k, w, c0, len, rid0; std::vector<int> rows; int ord; int lg = 0; };   // ord: number of the slice within its wave;
                                                      // lg > 0: every row of the slice is cut into 2^lg pieces on 2^lg adjacent lanes
// Waves [0, g_emis_waves) of a factored workgroup also stage the next frame's emission row (crf_kernels.hip fac_chain_body, `pre_w`): the
// request at the frame top, and in the tail a wait for it that -- vmcnt counts in order -- includes the acknowledgement of the wave's own
// write-through row stores, ~400 cycles in which that wave alone stands in front of the frame barrier (timing build, round 5: wave 0 arrived
// 270 cycles behind every other wave of the forward recursion, 6 % of the frame).  The placement charges those waves kEmisCost chunks and
// hands them the LIGHTEST slice list of their SIMD.  Set by build_factored around its place_rows calls (0: all waves alike).
static thread_local int g_emis_waves = 0;

bool place_rows_piece(const Rows &rows, const std::vector<int> &row_cu, int K, DirOut *o, std::vector<SliceAt> *slices, const Geom &gm, int piece, bool spread = false) {
    o->arcs.assign((size_t)K * gm.words * gm.threads, 0u);
    o->wave_info.assign((size_t)K * gm.waves, uint4{0u, 0u, 0u, 0u});
    o->rid_of_row.assign(rows.size(), -1);
    o->row_of.clear();
    o->cu_row_off.assign((size_t)K + 1, 0);
    o->est_cost = 0;
    o->simd_cost = 0;
    slices->clear();
    for (int k = 0; k < K; ++k) {
        std::vector<int> mine;
        for (size_t r = 0; r < rows.size(); ++r) if (row_cu[r] == k) mine.push_back((int)r);
        std::stable_sort(mine.begin(), mine.end(), [&](int a, int b) { return rows[a].size() > rows[b].size(); });
        // Rows too long for one lane (more than gm.nch chunks; n-gram LMs: the low-order history states are entered
        // from hundreds of histories): cut into 2^lg equal pieces on 2^lg ADJACENT lanes of one slice; the kernel adds
        // the pieces with a butterfly over those lanes before the row epilogue (every lane of the group then holds
        // the row's sum; all but the first are ordinary padding rows as far as their outputs are concerned).  One lg
        // per slice.  Factored layouts only (gm.multilane): the generic layout splits long rows into sub-rows with
        // virtual copies of their entry instead (build_resident) -- fine for a long tail, hopeless for an n-gram LM.
        std::vector<std::vector<int>> lanes;                  // per slice: row of each lane (-1 = none)
        std::vector<int> len, lgs;
        {
            auto lg_of = [&](int r) {
                int lg = 0;
                const int n = chunks_of(rows[r].size());
                if (!gm.multilane || n <= (spread ? piece : gm.nch)) return 0;   // (spread: rows that would fit a lane are cut too -- place_rows, small graphs)
                // pieces of at most `piece` chunks (place_rows tries several sizes)
                while (lg < 6 && (n + (1 << lg) - 1) / (1 << lg) > piece) ++lg;
                return lg;
            };
            size_t i = 0;
            while (i < mine.size()) {
                const int lg = lg_of(mine[i]), per = kWave >> lg;
                std::vector<int> ln(kWave, -1);
                int mx = 1, n = 0;
                while (i < mine.size() && n < per && lg_of(mine[i]) == lg) {
                    for (int q = 0; q < (1 << lg); ++q) ln[(size_t)n * (1 << lg) + q] = mine[i];
                    mx = std::max(mx, (chunks_of(rows[mine[i]].size()) + (1 << lg) - 1) / (1 << lg));
                    ++n; ++i;
                }
                lanes.push_back(ln); len.push_back(mx); lgs.push_back(lg);
            }
        }
        const int nsl = (int)lanes.size();
        // slices (sorted by length) -> waves with the register capacity as bin size.  What is balanced is
        // the TIME of a wave, measured on the MI355X (tools/timing_probe.py) as roughly linear in its chunks
        // plus a fixed price per slice end (row epilogue: emission lookup, stores, publish) worth several
        // chunks -- a wave with nine 1-chunk slices ran 1.5x longer than one with two 15-chunk slices, and the
        // frame waits for the slowest wave.  Step 1: best-fit-decreasing finds a feasible packing (the
        // registers are ~97% full for the benchmark graph, so balance-first heuristics do not even fit).
        // Step 2: a deterministic annealing over single moves and pair swaps lowers the maximum wave cost.
        const int kEpiCost = opt(kOpt_res_epi, 4);  // slice end, in chunks
        std::vector<int> wave_of(nsl, -1), load(gm.waves, 0), cnt(gm.waves, 0);
        bool packed = true;
        for (int strategy = 0; strategy < 2; ++strategy) {   // (second try: best fit also where the slices per wave are limited -- tight bins)
            packed = true;
            std::fill(wave_of.begin(), wave_of.end(), -1); std::fill(load.begin(), load.end(), 0); std::fill(cnt.begin(), cnt.end(), 0);
            for (int j = 0; j < nsl && packed; ++j) {
                int best = -1;
                for (int w = 0; w < gm.waves; ++w) {
                    if (load[w] + len[j] > gm.nch || (gm.maxsl && cnt[w] >= gm.maxsl)) continue;
                    // with a limit on the slices per wave: longest-first onto the LEAST loaded wave (best fit fills a
                    // wave's slice count with long slices and strands the short ones)
                    if (best < 0 || ((gm.maxsl && strategy == 0) ? load[w] < load[best] : load[w] > load[best])) best = w;
                }
                if (best < 0) { packed = false; break; }
                wave_of[j] = best; load[best] += len[j]; cnt[best]++;
            }
            if (packed || !gm.maxsl) break;
        }
        std::vector<std::vector<int>> lists(gm.waves);
        std::vector<int> cost(gm.waves, 0);
        if (packed) {
            const int kSimd0 = (gm.maxsl && K == 1) ? opt(kOpt_res_simd0, 0) : 0;
            const int kEmisCost = opt(kOpt_res_emis, 8);   // (metric graph: 0 / 3 / 5 / 8 / 12 -> recursions 2.576 / 2.585 / 2.561 / 2.550 / 2.577 ms; V = 217: 2.690 -> 2.625: profiles/round5_ab_emission_waves.txt)
            const int nem = (gm.maxsl && K == 1) ? std::min(g_emis_waves, gm.waves / 2) : 0;
            auto wcost = [&](int w) { return load[w] + kEpiCost * cnt[w] + (w < nem ? kEmisCost : 0); };
            // objective: (max cost, sum of squares) lexicographically, folded into one number
            const bool by_simd = gm.maxsl && gm.waves % 4 == 0 && !opt_on(kOpt_res_no_simd_order);
            auto objective = [&]() {
                int64_t mx = 0, sq = 0;
                if (by_simd) {   // waves w, w + 4, w + 8 share a SIMD and take turns on it: what the frame waits for is the busiest SIMD
                    int64_t g[4] = {kSimd0, 0, 0, 0};
                    for (int w = 0; w < gm.waves; ++w) { const int64_t c = wcost(w); g[w & 3] += c; sq += c * c; }
                    for (int i = 0; i < 4; ++i) mx = std::max(mx, g[i]);
                    return mx * 1000000 + sq;
                }
                for (int w = 0; w < gm.waves; ++w) { const int64_t c = wcost(w); mx = std::max(mx, c); sq += c * c; }
                return mx * 1000000 + sq;
            };
            uint64_t rng = 0x9E3779B97F4A7C15ull ^ ((uint64_t)nsl << 32) ^ (uint64_t)k;
            auto rnd = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; };
            int64_t cur = objective(), best_obj = cur;
            std::vector<int> best_assign = wave_of;
            const int iters = nsl > 1 ? 60000 : 0;
            for (int it = 0; it < iters; ++it) {
                const double temp = 4.0e6 * (1.0 - (double)it / iters);  // in objective units: a few chunks of max cost at the start
                const int ja = (int)(rnd() % (uint64_t)nsl), a_ = wave_of[ja];
                if (rnd() & 1) {  // move
                    const int b_ = (int)(rnd() % gm.waves);
                    if (b_ == a_ || load[b_] + len[ja] > gm.nch || (gm.maxsl && cnt[b_] >= gm.maxsl)) continue;
                    load[a_] -= len[ja]; cnt[a_]--; load[b_] += len[ja]; cnt[b_]++; wave_of[ja] = b_;
                    const int64_t nw = objective();
                    if (nw <= cur || (double)(rnd() % 1000000) / 1e6 < std::exp(-(double)(nw - cur) / std::max(temp, 1.0))) cur = nw;
                    else { load[b_] -= len[ja]; cnt[b_]--; load[a_] += len[ja]; cnt[a_]++; wave_of[ja] = a_; }
                } else {          // swap
                    const int jb = (int)(rnd() % (uint64_t)nsl), b_ = wave_of[jb];
                    if (b_ == a_ || len[ja] == len[jb]) continue;
                    const int d = len[ja] - len[jb];
                    if (load[b_] + d > gm.nch || load[a_] - d > gm.nch) continue;
                    load[a_] -= d; load[b_] += d; wave_of[ja] = b_; wave_of[jb] = a_;
                    const int64_t nw = objective();
                    if (nw <= cur || (double)(rnd() % 1000000) / 1e6 < std::exp(-(double)(nw - cur) / std::max(temp, 1.0))) cur = nw;
                    else { load[a_] += d; load[b_] -= d; wave_of[ja] = a_; wave_of[jb] = b_; }
                }
                if (cur < best_obj) { best_obj = cur; best_assign = wave_of; }
            }
            wave_of = best_assign;
            std::fill(load.begin(), load.end(), 0);
            std::fill(cnt.begin(), cnt.end(), 0);
            for (int j = 0; j < nsl; ++j) { lists[wave_of[j]].push_back(j); load[wave_of[j]] += len[j]; cnt[wave_of[j]]++; }
            for (int w = 0; w < gm.waves; ++w) cost[w] = wcost(w);
            if (by_simd) {
                // Within a SIMD the OLDEST wave is issued first (timing build: the last wave of a SIMD ends ~1000 cycles
                // after the first, later still when it is a heavy one): heaviest list to the lowest wave id.
                for (int g = 0; g < 4; ++g) {
                    std::vector<int> ws;
                    for (int w = g; w < gm.waves; w += 4) ws.push_back(w);
                    // (round 5: the waves that stage emissions -- ids below nem -- come LAST in that order and so get the lightest lists: with
                    // issue priorities by progress the age of a wave matters less than the ~400 cycles its tail waits for the row stores)
                    if (nem > 0 && kEmisCost > 0) std::stable_partition(ws.begin(), ws.end(), [&](int w) { return w >= nem; });
                    std::vector<int> order = ws;
                    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cost[a] - (a < nem ? kEmisCost : 0) > cost[b] - (b < nem ? kEmisCost : 0); });
                    std::vector<std::vector<int>> nl;
                    std::vector<int> nload, ncnt, ncost;
                    for (int w : order) { nl.push_back(lists[w]); nload.push_back(load[w]); ncnt.push_back(cnt[w]); ncost.push_back(cost[w]); }
                    for (size_t i = 0; i < ws.size(); ++i) { lists[ws[i]] = nl[i]; load[ws[i]] = nload[i]; cnt[ws[i]] = ncnt[i]; cost[ws[i]] = wcost(ws[i]); }
                }
            }
            if (opt_on(kOpt_verbose)) {
                fprintf(stderr, "[res_layout] lens:");
                for (int j = 0; j < nsl; ++j) fprintf(stderr, " %d", len[j]);
                fprintf(stderr, "\n[res_layout] K=%d cu=%d:", K, k);
                for (int w = 0; w < gm.waves; ++w) fprintf(stderr, " w%d(%dch,%zusl)", w, load[w], lists[w].size());
                int simd[4] = {0, 0, 0, 0}, tot = 0;
                for (int w = 0; w < gm.waves; ++w) { simd[w & 3] += cost[w]; tot += cost[w]; }
                fprintf(stderr, " | cost (chunks + %d per slice): total %d, busiest SIMD %d\n", kEpiCost, tot, std::max(std::max(simd[0], simd[1]), std::max(simd[2], simd[3])));
            }
        }
        if (!packed) {
            if (opt_on(kOpt_verbose)) {
                int tot = 0;
                for (int j = 0; j < nsl; ++j) tot += len[j];
                fprintf(stderr, "[res_layout] K=%d cu=%d does not fit: %d slices, %d chunks in all, capacity %d waves x %d chunks", K, k, nsl, tot, gm.waves, gm.nch);
                if (gm.maxsl) fprintf(stderr, " (%d slices per wave)", gm.maxsl);
                fprintf(stderr, "\n");
                fprintf(stderr, "[res_layout]   slice lengths (lg):"); for (int j = 0; j < nsl; ++j) fprintf(stderr, " %d(%d)", len[j], lgs[j]); fprintf(stderr, "\n");
            }
            return false;  // does not fit with this K
        }
        o->est_cost = std::max(o->est_cost, *std::max_element(cost.begin(), cost.end()));
        for (int g4 = 0; g4 < 4; ++g4) {
            int c = 0;
            for (int w = g4; w < gm.waves; w += 4) c += cost[w];
            o->simd_cost = std::max(o->simd_cost, c);
        }
        int rid = o->cu_row_off[k];
        int nlong = 0, own_off = 0;                             // multi-lane rows so far; owner lane (within its group) of the current one
        for (int w = 0; w < gm.waves; ++w) {
            unsigned ends = 0;
            int c0 = 0;
            const int wave_row0 = rid;
            int ord = 0;
            unsigned lgbits = 0;
            // wave_info.w has ten 3-bit fields: slices of multi-lane rows come first in their wave (the kernels read lg = 0 for
            // slice numbers >= 10); a wave with more than ten of them does not fit this geometry
            if (lists[w].size() > 10) std::stable_partition(lists[w].begin(), lists[w].end(), [&](int j) { return lgs[(size_t)j] > 0; });
            {
                int nml = 0;
                for (int j : lists[w]) nml += lgs[(size_t)j] > 0;
                if (nml > 10) return false;
            }
            for (int j : lists[w]) {
                ends |= 1u << (c0 + len[j] - 1);
                if (ord < 10) lgbits |= (unsigned)lgs[j] << (3 * ord);
                SliceAt sl{k, w, c0, len[j], rid, {}, ord++, lgs[j]};
                for (int lane = 0; lane < kWave; ++lane) {
                    const int r = lanes[j][lane];
                    // the lane that owns the row's outputs: after the butterfly every lane of the group holds the row's sum, so any
                    // of them will do -- and with implicit entries (factored 768-thread layouts) the row id IS the LDS position of
                    // the row's entry: "the first lane" would put the long rows of an n-gram LM, its most gathered entries, on
                    // banks 0 and 16 only (measured by pack_arcs' model: extra LDS cycles per frame 1 919 -> see DESIGN).  Rotate.
                    const int gmask = (1 << lgs[j]) - 1;
                    bool first = r >= 0 && (lane & gmask) == 0;
                    if (gm.maxsl && lgs[j] > 0 && r >= 0 && !opt_on(kOpt_res_owner_first)) {
                        if ((lane & gmask) == 0) { own_off = (nlong + (nlong >> lgs[j])) & gmask; ++nlong; }
                        first = (lane & gmask) == own_off;
                    }
                    sl.rows.push_back(r);
                    o->row_of.push_back(first ? r : -1);
                    if (first) o->rid_of_row[r] = rid + lane;
                }
                slices->push_back(sl);
                rid += kWave;
                c0 += len[j];
            }
            o->wave_info[(size_t)k * gm.waves + w] = uint4{ends, (unsigned)c0, (unsigned)wave_row0, lgbits};
        }
        o->cu_row_off[(size_t)k + 1] = rid;
    }
    return true;
}

// Rows longer than a lane's registers are cut into pieces on adjacent lanes (multilane geometries), and the piece size is a
// trade: short pieces pack well but every 64 lanes of pieces are a slice of their own with a row epilogue of its own (16 lanes
// per row: four rows per epilogue), long pieces mean fewer slices and fewer padding rows but may not pack at all.  The cost
// model (chunks + kEpiCost per slice, busiest SIMD) follows the measured frame time closely (den_lm of 40 000 sentences:
// model 1.40 x the benchmark graph's frame, measured 1.38 x), so try a few sizes and keep the cheapest packing.
static bool place_rows_sized(const Rows &rows, const std::vector<int> &row_cu, int K, DirOut *o, std::vector<SliceAt> *slices, const Geom &gm) {
    const int half = (gm.nch + 1) / 2;
    bool any_long = false;
    if (gm.multilane)
        for (auto &r : rows) if (chunks_of(r.size()) > gm.nch) { any_long = true; break; }
    const int piece_env = opt(kOpt_res_piece, 0);
    if (!any_long || piece_env > 0) return place_rows_piece(rows, row_cu, K, o, slices, gm, piece_env > 0 ? std::min(piece_env, gm.nch) : half);
    bool have = false;
    DirOut best;
    std::vector<SliceAt> best_slices;
    for (int piece : {gm.nch * 4 / 5, gm.nch * 7 / 10, gm.nch * 3 / 5, half}) {
        DirOut cand;
        std::vector<SliceAt> cs;
        if (!place_rows_piece(rows, row_cu, K, &cand, &cs, gm, piece)) continue;
        if (!have || cand.simd_cost < best.simd_cost) { best = std::move(cand); best_slices = std::move(cs); have = true; }
    }
    if (!have) return false;
    *o = std::move(best);
    *slices = std::move(best_slices);
    return true;
}
bool place_rows(const Rows &rows, const std::vector<int> &row_cu, int K, DirOut *o, std::vector<SliceAt> *slices, const Geom &gm = kGeomRes) {
    if (!place_rows_sized(rows, row_cu, K, o, slices, gm)) {
        // Eight waves of 30 chunks (the two-utterance kernel's geometry) pack worse than sixteen of 15: a row of 16 - 30 chunks fits a lane
        // and makes a slice as long as a whole wave.  Cut such rows too (pieces of half a lane and less), as the 1024-thread geometry has to.
        if (!(gm.multilane && gm.maxsl && gm.nch >= 2 * kFac4NCH && opt(kOpt_res_piece, 0) <= 0)) return false;
        bool have = false;
        for (int piece : {gm.nch / 2, gm.nch * 2 / 5, gm.nch / 3, gm.nch / 4}) {
            DirOut cand;
            std::vector<SliceAt> cs;
            if (!place_rows_piece(rows, row_cu, K, &cand, &cs, gm, piece, true)) continue;
            if (!have || cand.simd_cost < o->simd_cost) { *o = std::move(cand); *slices = std::move(cs); have = true; }
        }
        return have;
    }
    // (kSpreadSimdWeight / kSpreadWaveWeight, at the head of this file: the two cycle counts of the comparison below, in units of 10 cycles per chunk)
    // SMALL graphs (fewer slices than waves: most waves of the workgroup would have no rows at all while a few walk 10 - 15
    // chunks one batch after the other -- S = 513: 5 slices on 16 waves, 1.15 us per frame of which the arcs need a quarter):
    // cut the rows into 2 or 4 pieces on adjacent lanes although they would fit one, so that every wave has a short list.
    // What a frame then costs is no longer the busiest SIMD's sum but one wave's serial chain (a batch of gathers is an LDS
    // round trip: ~70 cycles per chunk for a wave alone on its SIMD against ~49 per chunk of a saturated SIMD's sum,
    // tools/ubench_issue.py): compare max(5 x busiest SIMD, 7 x heaviest wave).
    if (gm.multilane && K == 1 && opt(kOpt_res_piece, 0) <= 0 && !opt_on(kOpt_res_no_spread) && (int)slices->size() < gm.waves) {
        auto frame_est = [](const DirOut &d) { return std::max<int64_t>((int64_t)d.simd_cost * kSpreadSimdWeight, (int64_t)d.est_cost * kSpreadWaveWeight); };
        int maxlen = 1;
        for (auto &r : rows) maxlen = std::max(maxlen, std::min(gm.nch, chunks_of(r.size())));
        for (int div : {2, 4}) {
            const int piece = std::max(2, (maxlen + div - 1) / div);
            DirOut cand;
            std::vector<SliceAt> cs;
            if (!place_rows_piece(rows, row_cu, K, &cand, &cs, gm, piece, true)) continue;
            if (frame_est(cand) < frame_est(*o)) { *o = std::move(cand); *slices = std::move(cs); }
        }
    }
    return true;
}

// Grad pass (crf_grad_den_kernel): thread c of a workgroup walks the <= kChunk (Q position, BP position) pairs of chunk c, so
// at step j the 64 lanes of a wave gather Qs[q_j of their chunk] and then Bs[b_j of their chunk] from the rows staged in LDS --
// positions scattered by the layouts, ~3.5 lanes of a half-wave on the busiest bank (bank = index mod 32, halves of 32 lanes).
// The order of a chunk's pairs is free (one sum per chunk): arrange it, wave by wave and step by step, so that each lane takes
// the remaining pair whose two banks are least used at that step (equal address = broadcast, free).
// Returns the LDS cycles per frame (busiest bank per half-wave gather, summed) before and after.
void arrange_grad_pairs(std::vector<int> *gq, std::vector<int> *gb, const std::vector<int> &gchunk, int64_t *before, int64_t *after) {
    const int NC = (int)gchunk.size() - 1;
    auto load_of = [](const std::vector<int> (&bk)[32]) { size_t m = 0; for (auto &v : bk) m = std::max(m, v.size()); return (int64_t)m; };
    auto add = [](std::vector<int> (&bk)[32], int a) { auto &v = bk[a & 31]; if (std::find(v.begin(), v.end(), a) == v.end()) v.push_back(a); };
    auto cost_with = [](const std::vector<int> (&bk)[32], int a) { const auto &v = bk[a & 31]; return (int)v.size() + (std::find(v.begin(), v.end(), a) == v.end() ? 1 : 0); };
    *before = *after = 0;
    const bool on = !opt_on(kOpt_no_grad_arrange);
    for (int c0 = 0; c0 < NC; c0 += 32) {                       // one half-wave: chunks [c0, c0 + 32)
        const int nl = std::min(32, NC - c0);
        std::vector<std::vector<std::pair<int, int>>> rem(nl);
        int maxlen = 0;
        for (int l = 0; l < nl; ++l) {
            for (int i = gchunk[c0 + l]; i < gchunk[c0 + l + 1]; ++i) rem[l].push_back({(*gq)[i], (*gb)[i]});
            maxlen = std::max(maxlen, (int)rem[l].size());
        }
        for (int j = 0; j < maxlen; ++j) {                       // as listed
            std::vector<int> bq[32], bb[32];
            for (int l = 0; l < nl; ++l) if (j < (int)rem[l].size()) { add(bq, rem[l][j].first); add(bb, rem[l][j].second); }
            *before += load_of(bq) + load_of(bb);
        }
        std::vector<std::vector<std::pair<int, int>>> out(nl);
        for (int j = 0; j < maxlen; ++j) {
            std::vector<int> bq[32], bb[32];
            std::vector<int> order;
            for (int l = 0; l < nl; ++l) if (!rem[l].empty()) order.push_back(l);
            std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return rem[a].size() < rem[b].size(); });   // fewest choices first
            for (int l : order) {
                size_t best = 0;
                int bc = 1 << 30;
                for (size_t i = 0; i < rem[l].size(); ++i) {
                    const int cq = cost_with(bq, rem[l][i].first), cb = cost_with(bb, rem[l][i].second);
                    const int c = on ? std::max(cq, cb) * 64 + cq + cb : (int)i;
                    if (c < bc) { bc = c; best = i; }
                }
                add(bq, rem[l][best].first); add(bb, rem[l][best].second);
                out[l].push_back(rem[l][best]);
                rem[l].erase(rem[l].begin() + (long)best);
            }
            *after += load_of(bq) + load_of(bb);
        }
        for (int l = 0; l < nl; ++l)
            for (size_t i = 0; i < out[l].size(); ++i) { (*gq)[gchunk[c0 + l] + (int)i] = out[l][i].first; (*gb)[gchunk[c0 + l] + (int)i] = out[l][i].second; }
    }
}

// Step 2: put the arcs (gather indices already renumbered) into the (chunk, slot) positions of their
// slice.  One position = ONE ds_read_b32 gather per wave, serviced in two 32-lane halves over 32 banks
// (bank = index mod 32, MI355X_MICROARCH.md LDS table): the order of a row's arcs is free.
//   pass 1 (greedy): position by position, lanes with the fewest arcs left choose first, each takes the
//           arc whose bank is least used in its half (equal index = broadcast, free);
//   pass 2 (local search): swap two arcs of one lane between two positions whenever that lowers the LDS
//           cycles (busiest bank) of the two gathers.
// Padding gathers (weight 0) broadcast the address of a real lane of their half.
// `stride`: bytes per gather index (4: one float per gather; 8: an entry PAIR per gather, ds_read_b64, whose
// bank class is again index mod 32 -- 64 banks, two per lane).
// `dup_n`, `dup_off` (factored layout): entries [0, dup_n) exist TWICE in the gather vector, the second copy at
// entry + dup_off, i.e. on other banks (dup_off mod 32 != 0): every gather of such an entry may read either copy.
// With one copy the best order of a lane's arcs still leaves (busiest bank of the half-wave over the slice) - (slice
// length) extra cycles -- 44 % on the benchmark graph; with a choice of two banks per arc nearly all of it goes.
void pack_arcs(const Rows &rows, const std::vector<SliceAt> &slices, DirOut *o, int stride = 4, const Geom &gm = kGeomRes,
               int dup_n = 0, int dup_off = 0) {
    const bool arrange = !opt_on(kOpt_no_bank_arrange);
    auto alt = [&](int idx) { return idx < 0 ? -1 : idx < dup_n ? idx + dup_off : (idx >= dup_off && idx < dup_off + dup_n) ? idx - dup_off : -1; };
    for (const SliceAt &sl : slices) {
        const int NI = sl.len * kResW;
        std::vector<std::vector<std::pair<int, float>>> rem(kWave);
        for (int lane = 0; lane < kWave; ++lane)
            if (sl.rows[lane] >= 0) {
                const auto &row = rows[sl.rows[lane]];
                const int np = 1 << sl.lg, q = lane & (np - 1);
                for (size_t i = (size_t)q; i < row.size(); i += (size_t)np) rem[lane].push_back(row[i]);   // piece q of np
            }
        // place[ins][lane] = arc (index, weight) or index -1
        std::vector<std::vector<std::pair<int, float>>> place(NI, std::vector<std::pair<int, float>>(kWave, {-1, 0.f}));
        std::vector<std::vector<int>> cnt(NI, std::vector<int>(2 * 32, 0));  // lanes per (half, bank)
        if (!arrange) {
            for (int ins = 0; ins < NI; ++ins)
                for (int half = 0; half < 2; ++half) {
                    int occupant[32];
                    for (int b = 0; b < 32; ++b) occupant[b] = -1;
                    int lanes[32];
                    for (int l = 0; l < 32; ++l) lanes[l] = half * 32 + l;
                    std::stable_sort(lanes, lanes + 32, [&](int a, int b) { return rem[a].size() < rem[b].size(); });
                    for (int li = 0; li < 32; ++li) {
                        const int lane = lanes[li];
                        auto &rv = rem[lane];
                        if (rv.empty()) continue;
                        size_t best = 0;
                        int best_cost = 1 << 30;
                        bool best_alt = false;
                        for (size_t q = 0; q < (arrange ? rv.size() : (size_t)1) && best_cost; ++q)
                            for (int c = 0; c < 2; ++c) {
                                const int idx = c ? alt(rv[q].first) : rv[q].first;
                                if (idx < 0 || (c && !arrange)) continue;
                                const int bank = idx & 31;
                                const int cost = occupant[bank] == idx ? 0 : cnt[ins][half * 32 + bank];
                                if (cost < best_cost) { best_cost = cost; best = q; best_alt = c != 0; if (!cost) break; }
                            }
                        if (best_alt) rv[best].first = alt(rv[best].first);
                        place[ins][lane] = rv[best];
                        const int bank = rv[best].first & 31;
                        if (occupant[bank] != rv[best].first) cnt[ins][half * 32 + bank]++;
                        if (occupant[bank] < 0) occupant[bank] = rv[best].first;
                        rv.erase(rv.begin() + (long)best);
                    }
                }
        } else {
            // pass 1: per half-wave, (a) choose the copy of every arc so that the 32 banks carry equal numbers of
            // this slice's gathers, (b) colour the bipartite multigraph lanes x banks with NI colours (Koenig: possible
            // when no lane and no bank has more than NI edges; alternating-path recolouring) -- colour = position, so
            // every gather is conflict-free -- and (c) put the edges of over-full banks into their lanes' free
            // positions, the k-th surplus edge of every bank into the same positions where possible (a gather
            // costs its BUSIEST bank: two banks with two addresses each cost one extra cycle, not two).
            for (int half = 0; half < 2; ++half) {
                struct Edge { int lane, idx, bank, col; float w; bool over; };
                std::vector<Edge> ed;
                int load[32] = {0};
                for (int l = 0; l < 32; ++l)
                    for (auto &a : rem[half * 32 + l]) {
                        Edge e{half * 32 + l, a.first, a.first & 31, -1, a.second, false};
                        const int al = alt(e.idx);
                        if (al >= 0 && load[al & 31] < load[e.bank]) { e.idx = al; e.bank = al & 31; }
                        ++load[e.bank];
                        ed.push_back(e);
                    }
                // (a) balance: move one arc along a CHAIN of banks (an arc of bank u whose other copy is on bank v is
                // an edge u -> v) from a bank to one that carries at least two fewer -- breadth-first, until no bank has one
                for (int iter = 0; iter < 4096; ++iter) {
                    int order[32];
                    for (int b = 0; b < 32; ++b) order[b] = b;
                    std::stable_sort(order, order + 32, [&](int x, int y) { return load[x] > load[y]; });
                    bool moved = false;
                    for (int oi = 0; oi < 32 && !moved; ++oi) {
                        const int b0 = order[oi];
                        int via[32];                              // edge (index into ed) that reached the bank
                        for (int b = 0; b < 32; ++b) via[b] = -2;
                        via[b0] = -1;
                        std::vector<int> queue{b0};
                        int found = -1;
                        for (size_t qi = 0; qi < queue.size() && found < 0; ++qi) {
                            const int u = queue[qi];
                            for (size_t i = 0; i < ed.size() && found < 0; ++i) {
                                if (ed[i].bank != u) continue;
                                const int al = alt(ed[i].idx);
                                if (al < 0) continue;
                                const int v = al & 31;
                                if (via[v] != -2) continue;
                                via[v] = (int)i;
                                if (load[v] + 2 <= load[b0]) found = v; else queue.push_back(v);
                            }
                        }
                        if (found < 0) continue;
                        for (int v = found; v != b0;) {           // move one arc along every edge of the chain
                            Edge &e = ed[via[v]];
                            const int u = e.bank;
                            --load[u]; e.idx = alt(e.idx); e.bank = e.idx & 31; ++load[e.bank];
                            v = u;
                        }
                        moved = true;
                    }
                    if (!moved) break;
                }
                for (int b = 0; b < 32; ++b) {                   // surplus edges of over-full banks
                    int surplus = load[b] - NI;
                    for (size_t i = ed.size(); i-- > 0 && surplus > 0;)
                        if (ed[i].bank == b) { ed[i].over = true; --surplus; }
                }
                if (opt(kOpt_verbose, 0) >= 2) {
                    int mx = 0, nover = 0, tot = 0; for (int b = 0; b < 32; ++b) { mx = std::max(mx, load[b]); tot += load[b]; }
                    for (auto &e : ed) nover += e.over;
                    int nalt = 0; for (auto &e : ed) nalt += alt(e.idx) >= 0;
                    fprintf(stderr, "[pack] w%d c0=%d half %d: NI=%d edges=%d (avg/bank %.1f) max bank %d surplus %d with-alt %d\n", sl.w, sl.c0, half, NI, tot, tot / 32.0, mx, nover, nalt);
                }
                std::vector<int> lane_c((size_t)32 * NI, -1), bank_c((size_t)32 * NI, -1);   // edge at (node, colour)
                auto LC = [&](int lane, int c) -> int & { return lane_c[(size_t)(lane & 31) * NI + c]; };
                auto BC = [&](int bank, int c) -> int & { return bank_c[(size_t)bank * NI + c]; };
                for (size_t i = 0; i < ed.size(); ++i) {         // (b)
                    Edge &e = ed[i];
                    if (e.over) continue;
                    int a = -1, b = -1, both = -1;
                    for (int c = 0; c < NI; ++c) {
                        const bool fl = LC(e.lane, c) < 0, fb = BC(e.bank, c) < 0;
                        if (fl && fb) { both = c; break; }
                        if (fl && a < 0) a = c;
                        if (fb && b < 0) b = c;
                    }
                    if (both < 0) {
                        // colour a is free at the lane, b at the bank: swap a <-> b along the path that starts with the
                        // bank's a-edge (it cannot end in this lane, which has no a-edge), then a is free at both
                        std::vector<int> path;
                        int node = e.bank, col = a;
                        bool at_bank = true;
                        for (;;) {
                            const int e2 = at_bank ? BC(node, col) : LC(node, col);
                            if (e2 < 0) break;
                            path.push_back(e2);
                            node = at_bank ? ed[e2].lane : ed[e2].bank;
                            at_bank = !at_bank;
                            col = col == a ? b : a;
                        }
                        for (int e2 : path) { LC(ed[e2].lane, ed[e2].col) = -1; BC(ed[e2].bank, ed[e2].col) = -1; }
                        for (int e2 : path) { ed[e2].col = ed[e2].col == a ? b : a; LC(ed[e2].lane, ed[e2].col) = e2; BC(ed[e2].bank, ed[e2].col) = e2; }
                        both = a;
                    }
                    e.col = both;
                    LC(e.lane, both) = (int)i;
                    BC(e.bank, both) = (int)i;
                }
                std::vector<int> extra(NI, 0);                   // (c): positions that already cost a second cycle
                for (size_t i = 0; i < ed.size(); ++i) {
                    Edge &e = ed[i];
                    if (!e.over) continue;
                    int best = -1, best_key = 1 << 30;
                    for (int c = 0; c < NI; ++c) {
                        if (LC(e.lane, c) >= 0) continue;
                        int same = 0;                             // addresses already on this bank in position c
                        for (size_t j = 0; j < ed.size(); ++j) if (ed[j].col == c && ed[j].bank == e.bank && ed[j].idx != e.idx) ++same;
                        const int cost_after = std::max(extra[c], same);   // extra cycles of position c afterwards
                        const int key = (cost_after - extra[c]) * 1024 + (extra[c] ? 0 : 1) * 32 + same;
                        if (key < best_key) { best_key = key; best = c; }
                    }
                    if (best < 0) continue;                       // cannot happen: a lane has at most NI arcs
                    int same = 0;
                    for (size_t j = 0; j < ed.size(); ++j) if (ed[j].col == best && ed[j].bank == e.bank && ed[j].idx != e.idx) ++same;
                    extra[best] = std::max(extra[best], same);
                    e.col = best;
                    LC(e.lane, best) = (int)i;
                }
                for (auto &e : ed) if (e.col >= 0) place[e.col][e.lane] = {e.idx, e.w};
            }
        }
        // pass 2: local search on the TRUE cost.  A half-wave gather takes as many LDS cycles as its busiest
        // bank has distinct addresses, so what counts is sum over (position, half) of that maximum -- not the
        // number of colliding lanes (ten 2-way collisions in one gather cost one extra cycle, the same ten
        // spread over ten gathers cost ten).  Move: swap two arcs of one lane between two positions; accepted
        // if the two positions together get cheaper, or stay equal while the collisions shrink (plateau moves
        // that later allow a maximum to drop).  Deterministic order, bounded passes.
        if (arrange) {
            auto half_cost = [&](int ins, int h, int *coll) {  // cycles of one half-wave gather, #colliding lanes
                int n[32] = {0}, first[32], mx = 1, c = 0;
                for (int l = 0; l < 32; ++l) {
                    const int idx = place[ins][h + l].first;
                    if (idx < 0) continue;
                    const int bk = idx & 31;
                    if (n[bk] == 0) { first[bk] = idx; n[bk] = 1; continue; }
                    // distinct addresses per bank: count exactly (rows are short lists, 32 lanes)
                    bool seen = false;
                    for (int l2 = 0; l2 < l && !seen; ++l2) seen = place[ins][h + l2].first == idx;
                    if (!seen) { ++n[bk]; ++c; mx = std::max(mx, n[bk]); }
                    (void)first;
                }
                if (coll) *coll = c;
                return mx;
            };
            std::vector<int> hc(2 * NI), hcoll(2 * NI);
            for (int ins = 0; ins < NI; ++ins)
                for (int hh = 0; hh < 2; ++hh) hc[2 * ins + hh] = half_cost(ins, hh * 32, &hcoll[2 * ins + hh]);
            for (int pass = 0; pass < (dup_n ? 12 : 6); ++pass) {
                bool any = false;
                for (int i1 = 0; i1 < NI; ++i1)
                    for (int hh = 0; hh < 2; ++hh) {
                        if (hc[2 * i1 + hh] <= 1) continue;  // conflict-free gather
                        const int h = hh * 32;
                        for (int l = 0; l < 32; ++l) {
                            const int lane = h + l;
                            if (place[i1][lane].first < 0) continue;
                            const int b1 = place[i1][lane].first & 31;
                            // only lanes that sit in a contended bank of this gather are worth moving
                            int same = 0;
                            for (int l2 = 0; l2 < 32; ++l2) {
                                const int idx2 = place[i1][h + l2].first;
                                if (idx2 >= 0 && (idx2 & 31) == b1 && idx2 != place[i1][lane].first) ++same;
                            }
                            if (!same) continue;
                            if (alt(place[i1][lane].first) >= 0) {   // the other copy of this entry
                                const int keep = place[i1][lane].first, cb = hcoll[2 * i1 + hh];
                                place[i1][lane].first = alt(keep);
                                int c1;
                                const int n1 = half_cost(i1, h, &c1);
                                if (n1 < hc[2 * i1 + hh] || (n1 == hc[2 * i1 + hh] && c1 < cb)) {
                                    hc[2 * i1 + hh] = n1; hcoll[2 * i1 + hh] = c1;
                                    any = true;
                                    if (hc[2 * i1 + hh] <= 1) break;
                                    continue;
                                }
                                place[i1][lane].first = keep;
                            }
                            for (int i2 = 0; i2 < NI; ++i2) {
                                if (i2 == i1 || place[i2][lane].first < 0) continue;
                                if (((place[i2][lane].first ^ place[i1][lane].first) & 31) == 0) continue;
                                const int before = hc[2 * i1 + hh] + hc[2 * i2 + hh], cb = hcoll[2 * i1 + hh] + hcoll[2 * i2 + hh];
                                std::swap(place[i1][lane], place[i2][lane]);
                                int c1, c2;
                                const int n1 = half_cost(i1, h, &c1), n2 = half_cost(i2, h, &c2);
                                if (n1 + n2 < before || (n1 + n2 == before && c1 + c2 < cb)) {
                                    hc[2 * i1 + hh] = n1; hc[2 * i2 + hh] = n2; hcoll[2 * i1 + hh] = c1; hcoll[2 * i2 + hh] = c2;
                                    any = true;
                                    break;
                                }
                                std::swap(place[i1][lane], place[i2][lane]);
                            }
                            if (hc[2 * i1 + hh] <= 1) break;
                        }
                    }
                if (!any) break;
            }
            // cnt[][] is used below for the statistics: rebuild it as distinct addresses per bank
            for (int ins = 0; ins < NI; ++ins)
                for (int hb = 0; hb < 64; ++hb) cnt[ins][hb] = 0;
            for (int ins = 0; ins < NI; ++ins)
                for (int hh = 0; hh < 2; ++hh)
                    for (int l = 0; l < 32; ++l) {
                        const int idx = place[ins][hh * 32 + l].first;
                        if (idx < 0) continue;
                        bool seen = false;
                        for (int l2 = 0; l2 < l && !seen; ++l2) seen = place[ins][hh * 32 + l2].first == idx;
                        if (!seen) cnt[ins][hh * 32 + (idx & 31)]++;
                    }
        }
        for (int ins = 0; ins < NI; ++ins) {
            const int c = sl.c0 + ins / kResW, slot = ins % kResW;
            for (int half = 0; half < 2; ++half) {
                int first_real = -1;
                for (int l = 0; l < 32; ++l) if (place[ins][half * 32 + l].first >= 0) { first_real = half * 32 + l; break; }
                {   // extra LDS cycles of this half-wave gather: the busiest bank serves one address per cycle
                    int mx = 1;
                    for (int b = 0; b < 32; ++b) mx = std::max(mx, cnt[ins][half * 32 + b]);
                    o->conflicts += mx - 1;
                }
                for (int l = 0; l < 32; ++l) {
                    const int lane = half * 32 + l;
                    const bool real = place[ins][lane].first >= 0;
                    const int off16 = real ? place[ins][lane].first * stride : (first_real >= 0 ? place[ins][first_real].first * stride : 0);
                    const float wv = real ? place[ins][lane].second : 0.f;
                    const size_t t = (size_t)sl.w * kWave + lane;
                    unsigned &iw = o->arcs[((size_t)sl.k * gm.words + (size_t)c * 6 + (slot >> 1)) * gm.threads + t];
                    iw |= (unsigned)(off16 & 0xffff) << ((slot & 1) * 16);
                    unsigned wb;
                    memcpy(&wb, &wv, 4);
                    o->arcs[((size_t)sl.k * gm.words + (size_t)c * 6 + 2 + slot) * gm.threads + t] = wb;
                }
            }
            o->slots += kWave;
        }
    }
}

// balance states over K CUs by the chunk load of the rows keyed by each state
std::vector<int> assign_owner(int S, int K, const std::vector<int64_t> &load) {
    std::vector<int> order(S), owner(S, 0);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return load[a] > load[b]; });
    std::vector<int64_t> tot(K, 0);
    std::vector<int> cnt(K, 0);
    for (int s : order) {
        int best = 0;
        for (int k = 1; k < K; ++k)
            if (tot[k] < tot[best] || (tot[k] == tot[best] && cnt[k] < cnt[best])) best = k;
        owner[s] = best; tot[best] += load[s]; cnt[best]++;
    }
    return owner;
}

template <typename T>
int up(HostGraph *h, const std::vector<T> &v, const T **out) {
    if (h->device < 0) { *out = nullptr; return CRF_OK; }
    void *d = nullptr;
    size_t bytes = std::max<size_t>(v.size(), 1) * sizeof(T);
    hipError_t e = hipMalloc(&d, bytes);
    if (e != hipSuccess) { set_error(std::string("hipMalloc: ") + hipGetErrorString(e)); return CRF_ERR_HIP; }
    h->allocs.push_back(d);
    if (!v.empty() && (e = hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice)) != hipSuccess) {
        set_error(std::string("hipMemcpy: ") + hipGetErrorString(e)); return CRF_ERR_HIP;
    }
    *out = (const T *)d;
    return CRF_OK;
}

}  // namespace

int build_resident(HostGraph *h, int S, int P, const std::vector<int> &pair_dst, const std::vector<int> &pair_lab,
                   const Rows &in_arcs_of_pair, const Rows &out_arcs_of_state, const std::vector<float> &start_lin,
                   const std::vector<float> &end_lin, const std::vector<int> &label_sorted_pairs) {
    ResDev &R = h->dev.res;
    R = ResDev{};
    if (opt_on(kOpt_no_resident)) return CRF_OK;
    // Rows may be SPLIT into sub-rows (pieces of <= thr arcs, thr a multiple of the chunk width).
    // Forward: a pair with many in-arcs becomes several sub-rows with the same (dst, label); everything
    // downstream is linear in q (a_{t+1}[dst] += e'*q, gamma = sum q*b), so sub-rows are simply separate
    // pairs.  Backward: a state's sub-rows produce partial b values that are summed by the LDS atomics
    // into z_{t-1} and, in the grad pass, by listing every (forward sub-row, backward sub-row) product.
    // Splitting evens out row lengths: less slice padding and a tight fit into the per-wave register budget.
    const int max_lab = *std::max_element(pair_lab.begin(), pair_lab.end());
    static const int kSplit[] = {1 << 30, 96, 64, 48, 32, 16, 8};
    struct Dir {
        DirOut o;
        std::vector<int> sub_of, sub_j, owner, ex_cnt;  // sub-row -> input row / index among its key's sub-rows
        std::vector<int> gbase, gmap;                   // gather entry g, copy i -> new index gmap[gbase[g] + i]
        int G = 0, has_nx = 0;
        bool ok = false;
    };
    // rows[r]: arcs (gather entry, weight); key[r]: the state that owns the row (CU assignment); tgt[r]: the
    // gather entry the row produces (-1: none).  Several rows may target the same entry (forward: pairs
    // with different labels into one state; any row cut into sub-rows): they are partial sums, and each
    // gets its own virtual copy of the entry.  gkey[g]: the state whose owner produces entry g.
    auto try_dir = [&](const Rows &rows, const std::vector<int> &key, const std::vector<int> &tgt, int K,
                       const std::vector<int> &gkey, std::vector<int> *goff) -> Dir {
        Dir best;
        Rows best_sub;
        std::vector<int> best_subtgt, best_nparts;
        std::vector<SliceAt> best_slices;
        bool have = false, any_ragged = false;
        // attempt 0: no splitting; attempt 1: split only the rows that would sit in a ragged last slice
        // (fewer than 64 rows) of their CU into 2-chunk pieces, so that they fill leftover register space
        // instead of claiming a whole slice; attempts 2..: global thresholds
        std::vector<char> ragged(rows.size(), 0);
        for (int attempt = 0; attempt < 1 + (int)(sizeof(kSplit) / sizeof(kSplit[0])); ++attempt) {
            const int thr = attempt <= 1 ? (1 << 30) : kSplit[attempt - 1];
            if (attempt == 1 && !any_ragged) continue;  // attempt 0 placed everything: nothing is ragged
            Dir d;
            Rows sub;
            std::vector<int> subkey, subtgt;
            for (size_t r = 0; r < rows.size(); ++r) {
                const auto &a = rows[r];
                const int rthr = (attempt == 1 && ragged[r]) ? 2 * kResW : thr;
                const int parts = std::max(1, (int)((a.size() + (size_t)rthr - 1) / (size_t)rthr));
                size_t per = (a.size() + parts - 1) / parts;
                per = (per + kResW - 1) / kResW * kResW;
                for (int q = 0; q < parts; ++q) {
                    const size_t lo = std::min(a.size(), q * per), hi = std::min(a.size(), (q + 1) * per);
                    if (q > 0 && lo >= hi) break;
                    d.sub_of.push_back((int)r);
                    sub.emplace_back(a.begin() + (long)lo, a.begin() + (long)hi);
                    subkey.push_back(key[r]);
                    subtgt.push_back(tgt[r]);
                }
            }
            // A key (state) with n > 1 sub-rows gets n VIRTUAL copies of every gather entry it produces:
            // sub-row i writes copy i, and every arc that reads the entry is replicated to read all n copies.
            // Each entry then has exactly one producing row -- no LDS atomics (they are lane-serial, ~2.5 clk
            // per lane) and every value is final, publishable and usable for the frame maximum in its row's
            // epilogue.  Splits are rare (long-tail rows), so the replicated arcs are few.
            std::vector<int> nparts(gkey.size(), 0);
            d.sub_j.assign(sub.size(), 0);
            for (size_t r = 0; r < sub.size(); ++r) if (subtgt[r] >= 0) d.sub_j[r] = nparts[subtgt[r]]++;
            d.gbase.assign(gkey.size() + 1, 0);
            for (size_t g = 0; g < gkey.size(); ++g) d.gbase[g + 1] = d.gbase[g] + std::max(1, nparts[g]);
            d.G = d.gbase[gkey.size()];
            if (d.G > 16383) continue;  // 16-bit LDS byte offsets
            for (auto &row : sub) {
                std::vector<std::pair<int, float>> ex;
                for (auto &a : row)
                    for (int i = d.gbase[a.first]; i < d.gbase[a.first + 1]; ++i) ex.push_back({i, a.second});
                row.swap(ex);
            }
            std::vector<int64_t> load(S, 0);
            for (size_t r = 0; r < sub.size(); ++r) load[subkey[r]] += chunks_of(sub[r].size());
            d.owner = assign_owner(S, K, load);
            std::vector<int> cu(sub.size());
            for (size_t r = 0; r < sub.size(); ++r) cu[r] = d.owner[subkey[r]];
            std::vector<SliceAt> slices;
            if (!place_rows(sub, cu, K, &d.o, &slices)) {
                if (attempt == 0) {
                    // For attempt 1: every CU has (rows mod 64) rows too many for full slices.  Split THOSE
                    // finely so they fill leftover register space instead of claiming a slice of their own --
                    // choosing the rows whose entries are read by the fewest arcs (ideally none, e.g. the
                    // start state), because a split key's readers are replicated.
                    std::vector<int64_t> refs(S, 0);
                    for (const auto &row : rows)
                        for (const auto &a : row) refs[gkey[a.first]]++;
                    for (int k = 0; k < K; ++k) {
                        std::vector<int> mine;
                        for (size_t r = 0; r < sub.size(); ++r) if (cu[r] == k) mine.push_back((int)r);
                        std::stable_sort(mine.begin(), mine.end(), [&](int a, int b) { return refs[subkey[a]] < refs[subkey[b]]; });
                        for (size_t i = 0; i < mine.size() % kWave; ++i) { ragged[d.sub_of[mine[i]]] = 1; any_ragged = true; }
                    }
                }
                continue;
            }
            // keep the attempt with the lowest frame-time estimate (splitting long rows removes slice padding
            // and lets the waves balance, at the price of more slice ends and replicated arcs); ties and
            // near-ties go to the earlier attempt = fewer splits
            if (!have || d.o.est_cost + 1 < best.o.est_cost) {
                best = std::move(d);
                best_sub = std::move(sub); best_subtgt = std::move(subtgt); best_slices = std::move(slices); best_nparts = std::move(nparts);
                have = true;
            }
        }
        if (!have) return best;
        {
            Dir &d = best;
            Rows &sub = best_sub;
            std::vector<int> &subtgt = best_subtgt, &nparts = best_nparts;
            std::vector<SliceAt> &slices = best_slices;
            // Numbering of the (virtual) gather entries, CU by CU: first the produced entries IN THE ORDER
            // OF THEIR PRODUCING ROW's id -- a slice's 64 epilogues then publish 64 consecutive granules (one
            // coalesced 512-byte store) and write 64 consecutive LDS words; last the entries nobody produces
            // (e.g. the start state: no in-arcs), which stay 0 after the first frame and are never exchanged.
            d.gmap.assign(d.G, -1);
            std::vector<int> off(K + 1, 0);
            d.ex_cnt.assign(2 * K, 0);  // [k] produced entries, [K + k] unused (no shared entries any more)
            for (int k = 0, n = 0; k < K; ++k) {
                const int before = n;
                // EVERY row id owns the entry own_off[k] + (rid - cu_row_off[k]) -- also padding rows and rows that
                // produce nothing (they publish zeros into entries nobody reads).  The kernel then needs no
                // per-row "which entry" lookup in the row epilogue, only the row's label.
                for (int rid = d.o.cu_row_off[k]; rid < d.o.cu_row_off[k + 1]; ++rid, ++n) {
                    const int r = d.o.row_of[rid];
                    if (r >= 0 && subtgt[r] >= 0) d.gmap[d.gbase[subtgt[r]] + d.sub_j[r]] = n;
                }
                d.ex_cnt[k] = n - before;
                for (size_t g = 0; g < gkey.size(); ++g)
                    if (d.owner[gkey[g]] == k && nparts[g] == 0) d.gmap[d.gbase[g]] = n++;
                off[k + 1] = n;
                if (k == K - 1) d.G = n;  // virtual entries + the rows that own an unused one
            }
            if (d.G > 16383) { d.ok = false; return d; }
            for (auto &row : sub)
                for (auto &a : row) a.first = d.gmap[a.first];
            pack_arcs(sub, slices, &d.o);
            *goff = off;
            d.ok = true;
        }
        return best;
    };
    std::vector<int> fkey(P), state_id(S), gkey_f(S), gkey_b(P);
    for (int p = 0; p < P; ++p) { fkey[p] = pair_dst[p]; gkey_b[p] = pair_dst[p]; }
    std::iota(state_id.begin(), state_id.end(), 0);
    gkey_f = state_id;
    const int min_k = std::max(1, opt(kOpt_res_mink, 1));  // experiments
    for (int K = 1; K <= kResMaxK; K *= 2) {
        if (K < min_k) continue;
        std::vector<int> xoff, zoff;
        // forward rows = pairs, taken in label-sorted order so that sub-rows come out label-sorted; each
        // produces (a partial sum of) x[dst]
        Rows fin(P);
        std::vector<int> fin_key(P), fin_tgt(P);
        for (int j = 0; j < P; ++j) {
            const int p = label_sorted_pairs[j];
            fin[j] = in_arcs_of_pair[p]; fin_key[j] = pair_dst[p]; fin_tgt[j] = pair_dst[p];
        }
        Dir F = try_dir(fin, fin_key, fin_tgt, K, gkey_f, &xoff);
        if (!F.ok) continue;
        // backward rows = states; a state entered with n > 1 labels is listed n times (same arcs), once per
        // pair it produces z for, so that every row writes exactly one entry (only the first copy feeds the
        // grad pass and logZ); a state nobody enters (e.g. the start state) produces nothing
        std::vector<std::vector<int>> pairs_into(S);
        for (int p = 0; p < P; ++p) pairs_into[pair_dst[p]].push_back(p);
        Rows bin;
        std::vector<int> bin_key, bin_tgt, bin_dup;
        for (int s = 0; s < S; ++s)
            for (int dd = 0; dd < std::max<int>(1, (int)pairs_into[s].size()); ++dd) {
                bin.push_back(out_arcs_of_state[s]);
                bin_key.push_back(s);
                bin_tgt.push_back(pairs_into[s].empty() ? -1 : pairs_into[s][dd]);
                bin_dup.push_back(dd);
            }
        Dir Bk = try_dir(bin, bin_key, bin_tgt, K, gkey_b, &zoff);
        if (!Bk.ok) continue;
        const DirOut &fo = F.o, &bo = Bk.o;
        const int NRf = (int)F.sub_of.size(), NRb = (int)Bk.sub_of.size();
        auto pair_of_sub = [&](int r) { return label_sorted_pairs[F.sub_of[r]]; };
        (void)NRf;

        // ---- row metadata and side tables
        const int Rf = fo.cu_row_off[K], Rb = bo.cu_row_off[K];
        const int kNoLab = -1;  // padding rows / rows that produce nothing: the kernels scale them by 0
        std::vector<int> flab(Rf, kNoLab), blab(Rb, kNoLab);
        auto xof = [&](int s, int j) { return F.gmap[F.gbase[s] + j]; };   // forward x entry of (state, copy)
        auto zof = [&](int p, int j) { return Bk.gmap[Bk.gbase[p] + j]; };  // backward z entry of (pair, copy)
        const int Gf = F.G, Gb = Bk.G;
        for (int r = 0; r < Rf; ++r)
            if (fo.row_of[r] >= 0) flab[r] = pair_lab[pair_of_sub(fo.row_of[r])];
        std::vector<std::vector<int>> bsubs_of(S);  // backward sub-rows of the FIRST copy of each state
        for (int r = 0; r < NRb; ++r) if (bin_dup[Bk.sub_of[r]] == 0) bsubs_of[bin_key[Bk.sub_of[r]]].push_back(r);
        std::vector<float> brow_start(Rb, 0.f), brow_end(Rb, 0.f);
        for (int r = 0; r < Rb; ++r) {
            const int sr = bo.row_of[r];
            if (sr < 0) continue;
            const int in = Bk.sub_of[sr], s = bin_key[in], p = bin_tgt[in], j = Bk.sub_j[sr];
            if (p >= 0) blab[r] = pair_lab[p];
            if (bin_dup[in] == 0) {  // logZ and the end weight are taken from the first copy only
                brow_start[r] = start_lin[s];
                brow_end[r] = j == 0 ? end_lin[s] : 0.f;
            }
        }
        std::vector<int2> bcsr;
        std::vector<float> x_start(Gf, 0.f), x_end(Gf, 0.f), z_end(Gb, 0.f);
        std::vector<int> z_lab(Gb, 0);
        for (int s = 0; s < S; ++s)
            for (int j = 0; j < F.gbase[s + 1] - F.gbase[s]; ++j) {  // x_0 = start weight once, end weight on every copy
                x_start[xof(s, j)] = j == 0 ? start_lin[s] : 0.f;
                x_end[xof(s, j)] = end_lin[s];
            }
        for (int p = 0; p < P; ++p)
            for (int j = 0; j < Bk.gbase[p + 1] - Bk.gbase[p]; ++j) {
                z_lab[zof(p, j)] = pair_lab[p];
                z_end[zof(p, j)] = j == 0 ? end_lin[pair_dst[p]] : 0.f;
            }
        // grad pass list: every (forward sub-row, backward sub-row of its dst state), label-sorted, cut into
        // chunks of <= kChunk entries within one label
        std::vector<int> gq, gb, glabel, gchunk{0}, glab((size_t)max_lab + 2, 0);
        for (int r = 0; r < NRf; ++r) {
            const int p = pair_of_sub(r);
            for (int bs : bsubs_of[pair_dst[p]]) { gq.push_back(fo.rid_of_row[r]); gb.push_back(bo.rid_of_row[bs]); glabel.push_back(pair_lab[p]); }
        }
        {
            const int NL = (int)gq.size();
            int r = 0;
            for (int v = 0; v <= max_lab; ++v) {
                glab[v] = (int)gchunk.size() - 1;
                int e = r;
                while (e < NL && glabel[e] == v) ++e;
                for (int c = r; c < e; c += kChunk) gchunk.push_back(std::min(e, c + kChunk));
                r = e;
            }
            glab[(size_t)max_lab + 1] = (int)gchunk.size() - 1;
        }
        {
            int64_t gcb = 0, gca = 0;
            arrange_grad_pairs(&gq, &gb, gchunk, &gcb, &gca);
        }
        R.K = K;
        R.f.R = Rf; R.f.G = Gf; R.b.R = Rb; R.b.G = Gb;
        R.NC = (int)gchunk.size() - 1;
        for (int k = 0; k < K; ++k) {
            h->res_rows_cu_f = std::max(h->res_rows_cu_f, fo.cu_row_off[k + 1] - fo.cu_row_off[k]);
            h->res_rows_cu_b = std::max(h->res_rows_cu_b, bo.cu_row_off[k + 1] - bo.cu_row_off[k]);
        }
        h->res_stats.K = K;
        h->res_stats.slots_f = fo.slots; h->res_stats.slots_b = bo.slots;
        h->res_stats.conflicts_f = fo.conflicts; h->res_stats.conflicts_b = bo.conflicts;
        int rc;
        if ((rc = up(h, fo.arcs, &R.f.arcs)) || (rc = up(h, fo.wave_info, &R.f.wave_info)) ||
            (rc = up(h, flab, &R.f.row_lab)) || (rc = up(h, fo.cu_row_off, &R.f.cu_row_off)) ||
            (rc = up(h, xoff, &R.f.own_off)) || (rc = up(h, F.ex_cnt, &R.f.ex_cnt)) || (rc = up(h, Bk.ex_cnt, &R.b.ex_cnt)) || (rc = up(h, bo.arcs, &R.b.arcs)) ||
            (rc = up(h, bo.wave_info, &R.b.wave_info)) || (rc = up(h, blab, &R.b.row_lab)) ||
            (rc = up(h, bo.cu_row_off, &R.b.cu_row_off)) || (rc = up(h, zoff, &R.b.own_off)) ||
            (rc = up(h, x_start, &R.x_start)) || (rc = up(h, x_end, &R.x_end)) || (rc = up(h, z_lab, &R.z_lab)) ||
            (rc = up(h, z_end, &R.z_end)) || (rc = up(h, brow_start, &R.brow_start)) ||
            (rc = up(h, brow_end, &R.brow_end)) || (rc = up(h, bcsr, &R.bcsr)) || (rc = up(h, gq, &R.gq)) ||
            (rc = up(h, gb, &R.gb)) || (rc = up(h, gchunk, &R.chunk_off)) || (rc = up(h, glab, &R.lab_chunk_off)))
            return rc;
        ResHostCopy &C = h->rh;
        C.farcs = fo.arcs; C.barcs = bo.arcs; C.fwi = fo.wave_info; C.bwi = bo.wave_info; C.flab = flab; C.blab = blab;
        C.fcu = fo.cu_row_off; C.bcu = bo.cu_row_off; C.fown = xoff; C.bown = zoff; C.z_lab = z_lab; C.gq = gq; C.gb = gb; C.gchunk = gchunk; C.glab = glab;
        C.x_start = x_start; C.x_end = x_end; C.z_end = z_end; C.brow_start = brow_start; C.brow_end = brow_end;
        return CRF_OK;
    }
    return CRF_OK;  // K stays 0: not resident
}

// =================================================================================================
// Factored layout (crf_internal.h: FacDev).
// =================================================================================================
// rows -> K CUs: longest first onto the CU with the fewest chunks so far (ties: fewest rows)
static std::vector<int> deal_rows(const Rows &rows, int K) {
    std::vector<int> cu(rows.size(), 0);
    if (K <= 1) return cu;
    std::vector<int> ord(rows.size());
    for (size_t i = 0; i < rows.size(); ++i) ord[i] = (int)i;
    std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return rows[(size_t)a].size() > rows[(size_t)b].size(); });
    std::vector<int64_t> load((size_t)K, 0), cnt((size_t)K, 0);
    for (int r : ord) {
        int best = 0;
        for (int k = 1; k < K; ++k)
            if (load[(size_t)k] < load[(size_t)best] || (load[(size_t)k] == load[(size_t)best] && cnt[(size_t)k] < cnt[(size_t)best])) best = k;
        cu[(size_t)r] = best; load[(size_t)best] += chunks_of(rows[(size_t)r].size()); cnt[(size_t)best]++;
    }
    return cu;
}

static int build_factored_impl(HostGraph *h, int S, int P, const std::vector<int> &pair_dst, const std::vector<int> &pair_lab,
                               const Rows &in_arcs_of_pair, const Rows &out_arcs_of_state, const std::vector<float> &start_lin,
                               const std::vector<float> &end_lin, int level, bool *retry_next, int dup_mask, int *new_mask, bool short_only = false, bool *long_bail = nullptr, int K = 1) {
    const bool second = level == 6;                          // the two-utterance kernel's layout: HostGraph::facp / fhp
    FacDev &F = second ? h->facp : h->dev.fac;
    F = FacDev{};
    if (opt_on(kOpt_no_factored) || opt_on(kOpt_no_resident)) return CRF_OK;
    const bool verbose = opt_on(kOpt_verbose);
    auto give_up = [&](const char *why) { if (verbose) fprintf(stderr, "[fac_layout] not used: %s\n", why); return CRF_OK; };
    // ---- precondition: every state is entered with at most one label (true for T o LM: a state of the
    // composition remembers the last token), so "pair" and "destination state" are the same thing
    std::vector<int> pair_of(S, -1);
    for (int p = 0; p < P; ++p) {
        if (pair_of[pair_dst[p]] >= 0) return give_up("a state is entered with several labels");
        pair_of[pair_dst[p]] = p;
    }
    auto wbits = [](float w) { unsigned b; memcpy(&b, &w, 4); return b; };

    // ---- 1. match states that feed the same rows with the same weights
    std::vector<int> mate(S, -1);
    {
        std::vector<std::pair<uint64_t, int>> cand;  // (s1 << 32 | s2) -> count, via sort
        std::vector<uint64_t> keys;
        for (int p = 0; p < P; ++p) {
            std::vector<std::pair<unsigned, int>> a;
            for (auto &x : in_arcs_of_pair[p]) a.push_back({wbits(x.second), x.first});
            std::sort(a.begin(), a.end());
            for (size_t i = 0; i < a.size();) {
                size_t j = i;
                while (j < a.size() && a[j].first == a[i].first) ++j;
                if (j - i >= 2 && j - i <= 64)
                    for (size_t u = i; u < j; ++u)
                        for (size_t v = u + 1; v < j; ++v)
                            if (a[u].second != a[v].second)
                                keys.push_back((uint64_t)std::min(a[u].second, a[v].second) << 32 | (unsigned)std::max(a[u].second, a[v].second));
                i = j;
            }
        }
        std::sort(keys.begin(), keys.end());
        for (size_t i = 0; i < keys.size();) {
            size_t j = i;
            while (j < keys.size() && keys[j] == keys[i]) ++j;
            cand.push_back({keys[i], (int)(j - i)});
            i = j;
        }
        std::stable_sort(cand.begin(), cand.end(), [](const std::pair<uint64_t, int> &x, const std::pair<uint64_t, int> &y) { return x.second > y.second; });
        for (auto &c : cand) {
            const int s1 = (int)(c.first >> 32), s2 = (int)(c.first & 0xffffffffu);
            if (c.second < 2) break;
            if (mate[s1] < 0 && mate[s2] < 0) { mate[s1] = s2; mate[s2] = s1; }
        }
    }
    int64_t nmatched = 0;
    for (int s = 0; s < S; ++s) if (mate[s] > s) ++nmatched;
    if (nmatched * 4 < S) return give_up("fewer than half of the states pair up");

    // ---- 2. forward: pair SUMS.  In T o LM the row of (g, blank) has exactly two in-arcs, from (g, blank) and
    // (g, token), with one weight w: a_{t+1}[(g,blank)] = e'[blank] * w * (a_t[(g,blank)] + a_t[(g,token)]).
    // So the recursion keeps, per matched pair (tail state s1 with such a row, main state s2), three entries:
    //     U = a[s1] + a[s2]   (what every other row reads of the pair: ONE gather, ONE weight for two arcs)
    //     L = a[s2], A = a[s1] (read by the few arcs that need one of them alone; logZ at the end)
    // and the row of s2 updates all three in its epilogue: L' = e'[l2] * rowsum, A' = e'[l1] * w * U, U' = A' + L'
    // (no subtraction anywhere) -- the row of s1 disappears.  A matched pair without that structure is
    // un-matched again (generic graphs: everything below still works, there is just nothing to save).
    const int max_lab = *std::max_element(pair_lab.begin(), pair_lab.end());
    std::vector<int> tail_of(S, -1);   // main state -> its tail state
    std::vector<float> tail_w(S, 0.f);
    for (int s = 0; s < S; ++s) {
        const int m = mate[s];
        if (m < 0 || m < s) continue;
        auto self_pair = [&](int t, int o, float *w) {   // is the row of t exactly {t: w, o: w}?
            const int p = pair_of[t];
            if (p < 0 || in_arcs_of_pair[p].size() != 2) return false;
            const auto &a = in_arcs_of_pair[p];
            if (wbits(a[0].second) != wbits(a[1].second)) return false;
            if (!((a[0].first == t && a[1].first == o) || (a[0].first == o && a[1].first == t))) return false;
            *w = a[0].second;
            return true;
        };
        float w = 0.f;
        if (pair_of[m] >= 0 && self_pair(s, m, &w)) { tail_of[m] = s; tail_w[m] = w; }
        else if (pair_of[s] >= 0 && self_pair(m, s, &w)) { tail_of[s] = m; tail_w[s] = w; }
        else { mate[s] = mate[m] = -1; }
    }
    nmatched = 0;
    for (int s = 0; s < S; ++s) if (tail_of[s] >= 0) ++nmatched;
    if (nmatched * 4 < S) return give_up("fewer than half of the states form (tail, main) pairs");
    std::vector<char> is_tail(S, 0);
    for (int s = 0; s < S; ++s) if (tail_of[s] >= 0) is_tail[tail_of[s]] = 1;
    // Geometry: 768 threads (3 waves per SIMD at <= 168 VGPRs; 20 chunks of arcs and the constants of up to 3 rows per
    // thread) when both directions fit it, else 512 threads x 30 chunks (2 waves per SIMD); CRF_FAC_THREADS=512 forces
    // the latter.
    const bool lvl_table = level == 1 || level == 3 || level == 4 || level == 6;   // row constants in the LDS table: 20 chunks of arcs per thread (1) or 21 (3)
    const Geom *gm = level == 0 ? &kGeomFac3 : level == 1 ? &kGeomFac3L : level == 3 ? &kGeomFac3L21 : level == 4 ? &kGeomFac4L : level == 6 ? &kGeomFac512L : &kGeomFac512;
    const bool allow3 = level != 2 && level != 6; // a larger geometry is left to try
    const bool rcregs = level == 0;               // row constants in registers
    const bool implicit = gm->maxsl > 0;   // entries numbered by row id, row constants in registers (below)
    // entries (512-thread layout): [U of every pair][sink][L of every pair][A of every pair][plain states]
    std::vector<int> entU(S, -1), ent(S, -1);   // entU: by main state; ent: a[s] itself (L, A or plain)
    // ... and a SECOND copy of the U entries (and the sink, which padding rows write) behind everything, on
    // other banks: [U][sink][L][A][plain] [pad] [U'][sink'] -- see pack_arcs
    const int bank_shift = opt(kOpt_fac_bank_shift, 5) & 31;
    // second copy of the gathered entries: per direction (dup_mask bit 0 forward, bit 1 backward); a direction that turns out
    // not to fit with it clears its bit in *new_mask and the caller builds again
    const bool env_nodup = opt_on(kOpt_fac_no_dup);
    const bool no_dupf = !(dup_mask & 1) || env_nodup || K > 1, no_dupb = !(dup_mask & 2) || env_nodup || K > 1;   // (two CUs: one copy, the peers' entries are fetched into it)
    *new_mask = dup_mask;
    int nent = 0;
    for (int s = 0; s < S; ++s) if (tail_of[s] >= 0) entU[s] = nent++;
    int nU = nent;
    int sink = nent++;
    for (int s = 0; s < S; ++s) if (tail_of[s] >= 0) ent[s] = nent++;
    for (int s = 0; s < S; ++s) if (tail_of[s] >= 0) ent[tail_of[s]] = nent++;
    for (int s = 0; s < S; ++s) if (ent[s] < 0) ent[s] = nent++;
    const int64_t nsolo = 0;
    // rows: every pair except the tail rows
    std::vector<int> main_rows;
    for (int p = 0; p < P; ++p) if (!is_tail[pair_dst[p]]) main_rows.push_back(p);
    Rows fsub(main_rows.size());
    for (size_t i = 0; i < main_rows.size(); ++i) {
        std::vector<std::pair<int, float>> a = in_arcs_of_pair[main_rows[i]];
        std::vector<char> used(a.size(), 0);
        for (size_t u = 0; u < a.size(); ++u) {
            if (used[u]) continue;
            used[u] = 1;
            const int s = a[u].first;
            const int mn = tail_of[s] >= 0 ? s : (is_tail[s] ? mate[s] : -1);   // main state of s's pair, if any
            size_t v = a.size();
            if (mn >= 0) {
                const int other = s == mn ? tail_of[mn] : mn;
                for (size_t q = u + 1; q < a.size(); ++q)
                    if (!used[q] && a[q].first == other && wbits(a[q].second) == wbits(a[u].second)) { v = q; break; }
            }
            if (v < a.size()) { used[v] = 1; fsub[i].push_back({entU[mn], a[u].second}); }
            else fsub[i].push_back({ent[s], a[u].second});
        }
    }
    if (level == 4) {
        // 1024 threads: measured faster (2 - 6 %) on graphs whose rows fit a lane or nearly all do (V = 217 / 500: 3 - 8 % of the arcs
        // in longer rows), SLOWER on den_lm estimated from text, where most arcs sit in rows of hundreds of arcs on many lanes
        // (S = 3 006: 2.51 -> 2.64 ms per step, S = 6 836: 4.83 -> 5.26): those keep the 768-thread geometries
        size_t long_arcs = 0, all_arcs = 0;
        for (auto &r : fsub) { all_arcs += r.size(); if (chunks_of(r.size()) > gm->nch) long_arcs += r.size(); }
        if (opt_on(kOpt_verbose)) fprintf(stderr, "[fac_layout] 1024 threads: %zu of %zu forward arcs in rows longer than a lane\n", long_arcs, all_arcs);
        // (round 4, with issue priorities by progress: S = 3 006 -- 14.9 k of 20.5 k forward arcs in such rows -- is now FASTER on 1024
        // threads, recursions 2.28 -> 2.13 ms; S = 6 836 -- 27.3 k of 49.0 k -- still slower, 3.41 -> 3.51: the share alone does not
        // decide, the number of multi-lane slices a wave has to finish per frame does)
        if (opt(kOpt_fac_threads, 0) != 1024 && long_arcs * kLongRowShareDen > all_arcs && long_arcs > kLongRowArcs) { *retry_next = true; return CRF_OK; }
    }
    if (short_only)   // graphs with rows longer than a lane's registers (every den_lm estimated from text) run 2-3 % faster with the
        for (auto &r : fsub)   // row constants in the LDS table (level 1; measured): leave them to it
            if (chunks_of(r.size()) > gm->nch) { if (long_bail) *long_bail = true; *retry_next = true; return CRF_OK; }
    struct EmisGuard { ~EmisGuard() { g_emis_waves = 0; } } emis_guard;   // (reset on every way out of this function)
    g_emis_waves = (max_lab + kWave) / kWave;                            // waves that hold emissions at V = max label + 1, i.e. ceil(V / 64): place_rows_piece
    DirOut fo;
    std::vector<SliceAt> fslices;
    if (!place_rows(fsub, deal_rows(fsub, K), K, &fo, &fslices, *gm)) {
        if (allow3) { *retry_next = true; return CRF_OK; }
        return give_up("forward rows do not fit one CU");
    }
    const int Rf = fo.cu_row_off[(size_t)K];
    if (implicit) {
        // 768-thread layout: the entries of a row are where its row id says -- U at rid, L at Rf + rid, A at 2 Rf + rid
        // (then the states nobody enters, a zero entry for padding gathers, and the second copy of the U entries) -- so
        // a row epilogue needs no table to find them: what is left of the row constants (two labels, the tail weight)
        // lives in two registers per slice.  The epilogue was a chain of three LDS round trips (constants -> the
        // values they point to -> emissions), 70 % of the frame loop in the timing build.
        std::vector<int> remap(nent, -1);
        std::vector<int> nentU(S, -1), nentS(S, -1);
        for (size_t r = 0; r < main_rows.size(); ++r) {
            const int rid = fo.rid_of_row[r], s = pair_dst[main_rows[r]];
            nentS[s] = Rf + rid;
            if (tail_of[s] >= 0) { nentU[s] = rid; nentS[tail_of[s]] = 2 * Rf + rid; }
        }
        int nx = 3 * Rf;
        for (int s = 0; s < S; ++s) if (nentS[s] < 0) nentS[s] = nx++;
        const int nsink = nx++;
        for (int s = 0; s < S; ++s) { remap[ent[s]] = nentS[s]; if (entU[s] >= 0) remap[entU[s]] = nentU[s]; }
        remap[sink] = nsink;
        for (auto &row : fsub) for (auto &a : row) a.first = remap[a.first];
        ent = nentS; entU = nentU; sink = nsink; nent = nx; nU = Rf;
    }
    int fdup = 0;                                        // entries between the two copies
    if (!no_dupf) { fdup = nent; while ((fdup & 31) != bank_shift) ++fdup; }
    const int Gf = fdup ? fdup + nU + 1 : nent;
    if ((size_t)Gf * 4 > 65536) {
        if (!no_dupf) { *new_mask = dup_mask & ~1; return CRF_OK; }   // first without the second copy of the U entries
        if (allow3) { *retry_next = true; return CRF_OK; }
        return give_up("forward gather vector > 64 KiB");
    }
    pack_arcs(fsub, fslices, &fo, 4, *gm, fdup ? nU : 0, fdup);
    // two CUs: the U entries cross every frame as one contiguous range per CU; of the L and A entries (and the plain states, which
    // sit in the L range) only those that a row of the OTHER CU gathers -- in T o LM the L entry of a couple is read by its own
    // row alone (the token's self-loop) -- as a list per CU; everything once more after the last frame, for logZ.
    std::vector<int> xlist;                      // entries CU 0 fetches, then those CU 1 fetches
    int xlist_off[3] = {0, 0, 0};
    if (K > 1 && implicit) {
        auto cu_of_rid = [&](int rid) { int k = 0; while (k + 1 < K && rid >= fo.cu_row_off[(size_t)k + 1]) ++k; return k; };
        std::vector<std::vector<int>> need((size_t)K);
        for (size_t r = 0; r < fsub.size(); ++r) {
            const int rid = fo.rid_of_row[r];
            if (rid < 0) continue;
            const int mycu = cu_of_rid(rid);
            for (auto &a : fsub[r])
                if (a.first >= Rf && a.first < 3 * Rf && cu_of_rid(a.first % Rf) != mycu) need[(size_t)mycu].push_back(a.first);
        }
        for (int k = 0; k < K && k < 2; ++k) {
            auto &v = need[(size_t)k];
            std::sort(v.begin(), v.end());
            v.erase(std::unique(v.begin(), v.end()), v.end());
            xlist_off[k] = (int)xlist.size();
            xlist.insert(xlist.end(), v.begin(), v.end());
        }
        xlist_off[2] = (int)xlist.size();
        if (verbose) fprintf(stderr, "[fac_layout] two CUs: of %d L / A entries per CU, CU 0 fetches %d and CU 1 %d every frame\n", 2 * (Rf / K), xlist_off[1] - xlist_off[0], xlist_off[2] - xlist_off[1]);
    }
    if (xlist.empty()) xlist.push_back(0);
    const int NT = 0;
    std::vector<int> fpos(P, -1);   // position of pair p in the Q row: main rows [0, Rf), their tails [Rf, 2 Rf)
    std::vector<int4> frow_meta(Rf, int4{sink * 4, (sink * 4) | ((sink * 4) << 16), 0, 0});   // padding rows: all to the sink
    for (int rid = 0; rid < Rf; ++rid) {
        const int r = fo.row_of[rid];
        if (r < 0) continue;
        const int p = main_rows[r], s = pair_dst[p];
        fpos[p] = rid;
        if (tail_of[s] >= 0) {
            const int t = tail_of[s], pt = pair_of[t];
            fpos[pt] = Rf + rid;
            frow_meta[rid] = int4{(entU[s] * 4) | (pair_lab[p] << 16), (ent[s] * 4) | ((ent[t] * 4) << 16), (int)wbits(tail_w[s]), pair_lab[pt]};
        } else {
            frow_meta[rid] = int4{(sink * 4) | (pair_lab[p] << 16), (ent[s] * 4) | ((sink * 4) << 16), 0, 0};
        }
    }
    if (rcregs)   // row constants of the slice number `ord` of a wave: words (kFac3ArcCh * 6 + 2 * ord) and the next of its threads
        for (const SliceAt &sl : fslices)
            for (int lane = 0; lane < kWave; ++lane) {
                const int rid = sl.rid0 + lane, r = fo.row_of[rid];
                if (r < 0) continue;
                const int4 m = frow_meta[rid];
                const size_t t = (size_t)sl.w * kWave + lane, w0 = (size_t)kFac3ArcCh * 6 + 2 * (size_t)sl.ord;
                fo.arcs[w0 * gm->threads + t] = (((unsigned)m.x >> 16) * 4u) | (((unsigned)m.w * 4u) << 16);   // BYTE offsets of the two emissions: main label * 4 | tail label * 4 << 16
                fo.arcs[(w0 + 1) * gm->threads + t] = (unsigned)m.z;                             // tail weight (0: no tail)
            }
    const int Rq = 2 * Rf;
    std::vector<float> x_start(Gf, 0.f), x_end(Gf, 0.f);
    for (int s = 0; s < S; ++s) {
        x_start[ent[s]] = start_lin[s];
        x_end[ent[s]] = end_lin[s];
        if (tail_of[s] >= 0) x_start[entU[s]] = start_lin[s] + start_lin[tail_of[s]];
    }
    if (fdup) for (int u = 0; u < nU; ++u) x_start[fdup + u] = x_start[u];   // (implicit layout: nU = Rf, unused U slots are 0)
    std::vector<int4> ftail(1, int4{0, 0, 0, 0});
    std::vector<int> tail_rows;   // (statistics only)
    for (int s = 0; s < S; ++s) if (is_tail[s]) tail_rows.push_back(s);

    // ---- 4. backward rows: matched states with common out-arcs and at most one extra arc each share a row
    struct BRow { int s0, s1; std::vector<std::pair<int, float>> arcs; int e0 = -1, e1 = -1; float w0 = 0.f, w1 = 0.f; };  // arcs: (pair, w)
    std::vector<BRow> brow;
    int64_t nfused = 0;
    // States that no arc enters are never gathered: their b_t is needed once, at t = 0, for logZ of the backward
    // recursion (costs_beta).  They get no row; the kernel adds start * sum_arcs w * z_0 after its last frame (bx list).
    // In T o LM that is the start state -- and with 2^k histories the row it does not take is the one that would have
    // opened another slice of 64 rows for itself.
    std::vector<int> bx_idx;
    std::vector<float> bx_w;
    float bx_se = 0.f;
    std::vector<char> no_row(S, 0);
    for (int s = 0; s < S; ++s)
        if (pair_of[s] < 0 && mate[s] < 0) { no_row[s] = 1; bx_se += start_lin[s] * end_lin[s]; }
    // ... and the states the forward matching left alone may still share a backward row (common out-arcs, at most one
    // extra arc each): (history 0, blank) and (history 0, token) in T o LM, whose forward structure the start state spoils
    std::vector<int> bmate = mate;
    {
        std::vector<int> alone;
        for (int s = 0; s < S; ++s) if (mate[s] < 0 && !no_row[s]) alone.push_back(s);
        if (alone.size() <= 64)
            for (size_t i = 0; i < alone.size(); ++i)
                for (size_t j = i + 1; j < alone.size() && bmate[alone[i]] < 0; ++j) {
                    const int s = alone[i], m = alone[j];
                    if (bmate[m] >= 0) continue;
                    std::vector<std::pair<int, unsigned>> a, b, common;
                    for (auto &x : out_arcs_of_state[s]) a.push_back({x.first, wbits(x.second)});
                    for (auto &x : out_arcs_of_state[m]) b.push_back({x.first, wbits(x.second)});
                    std::sort(a.begin(), a.end());
                    std::sort(b.begin(), b.end());
                    std::set_intersection(a.begin(), a.end(), b.begin(), b.end(), std::back_inserter(common));
                    if (!common.empty() && a.size() - common.size() <= 1 && b.size() - common.size() <= 1) { bmate[s] = m; bmate[m] = s; }
                }
    }
    {
        std::vector<char> done(S, 0);
        for (int s = 0; s < S; ++s) {
            if (done[s] || no_row[s]) continue;
            const int m = bmate[s];
            bool fused = false;
            if (m > s) {
                std::vector<std::pair<int, unsigned>> a, b;
                for (auto &x : out_arcs_of_state[s]) a.push_back({x.first, wbits(x.second)});
                for (auto &x : out_arcs_of_state[m]) b.push_back({x.first, wbits(x.second)});
                std::sort(a.begin(), a.end());
                std::sort(b.begin(), b.end());
                std::vector<std::pair<int, unsigned>> common, ea, eb;
                std::set_intersection(a.begin(), a.end(), b.begin(), b.end(), std::back_inserter(common));
                std::set_difference(a.begin(), a.end(), common.begin(), common.end(), std::back_inserter(ea));
                std::set_difference(b.begin(), b.end(), common.begin(), common.end(), std::back_inserter(eb));
                if (ea.size() <= 1 && eb.size() <= 1 && !common.empty()) {
                    BRow r{s, m, {}, -1, -1, 0.f, 0.f};
                    for (auto &c : common) { float w; memcpy(&w, &c.second, 4); r.arcs.push_back({c.first, w}); }
                    if (!ea.empty()) { r.e0 = ea[0].first; memcpy(&r.w0, &ea[0].second, 4); }
                    if (!eb.empty()) { r.e1 = eb[0].first; memcpy(&r.w1, &eb[0].second, 4); }
                    // which of the two states is output 0 alternates: the z entries 2*rid + output are what the
                    // rows gather, and in a T o LM graph nearly all arcs enter the "token" state of a pair -- with
                    // a fixed order every gather would hit an odd entry, i.e. half of the LDS banks
                    if (brow.size() & 1) { std::swap(r.s0, r.s1); std::swap(r.e0, r.e1); std::swap(r.w0, r.w1); }
                    brow.push_back(r);
                    done[s] = done[m] = 1;
                    fused = true;
                    ++nfused;
                }
            }
            if (!fused) { brow.push_back(BRow{s, -1, out_arcs_of_state[s], -1, -1, 0.f, 0.f}); done[s] = 1; }
        }
    }
    // Rows in label order (rows of equal length keep it through the placement): the 32 lanes of a half-wave then look up
    // the emission of one or two labels in their epilogue -- a broadcast -- instead of up to 32 (bank conflicts).  The
    // forward rows are pairs, which are label-sorted already.
    {
        auto key = [&](const BRow &r) {
            const int l0 = pair_of[r.s0] >= 0 ? pair_lab[pair_of[r.s0]] : -1, l1 = (r.s1 >= 0 && pair_of[r.s1] >= 0) ? pair_lab[pair_of[r.s1]] : -1;
            return std::max(l0, l1);
        };
        std::stable_sort(brow.begin(), brow.end(), [&](const BRow &a, const BRow &b) { return key(a) < key(b); });
    }
    Rows bsub(brow.size());
    for (size_t i = 0; i < brow.size(); ++i) bsub[i] = brow[i].arcs;   // pair ids for now; lengths are all placement needs
    if (level == 4) {   // (the same test as on the forward rows above: a graph whose BACKWARD rows are mostly multi-lane keeps 768 threads too)
        size_t long_arcs = 0, all_arcs = 0;
        for (auto &r : bsub) { all_arcs += r.size(); if (chunks_of(r.size()) > gm->nch) long_arcs += r.size(); }
        if (opt(kOpt_fac_threads, 0) != 1024 && long_arcs * kLongRowShareDen > all_arcs && long_arcs > kLongRowArcs) { *retry_next = true; return CRF_OK; }
    }
    if (short_only)
        for (auto &r : bsub)
            if (chunks_of(r.size()) > gm->nch) { if (long_bail) *long_bail = true; *retry_next = true; return CRF_OK; }
    DirOut bo;
    std::vector<SliceAt> bslices;
    if (!place_rows(bsub, deal_rows(bsub, K), K, &bo, &bslices, *gm)) {
        if (allow3) { *retry_next = true; return CRF_OK; }   // both directions then use the larger per-thread budget
        return give_up("backward rows do not fit one CU");
    }
    const int Rb = bo.cu_row_off[(size_t)K], Gb0 = 2 * Rb + 2, zsink = 2 * Rb;
    int bdup = 0;                                        // second copy of every z entry: [z (2 Rb)][sink pair] [pad] [z'][sink']
    if (!no_dupb) { bdup = Gb0; while ((bdup & 31) != bank_shift) ++bdup; }
    const int Gb = bdup ? bdup + Gb0 : Gb0;
    if ((size_t)Gb * 4 > 65536) {
        if (!no_dupb) { *new_mask = dup_mask & ~2; return CRF_OK; }
        return give_up("backward gather vector > 64 KiB");
    }
    std::vector<int> zpos(S, -1);   // BP / z position of state s: 2*rid + output
    for (int rid = 0; rid < Rb; ++rid) {
        const int r = bo.row_of[rid];
        if (r < 0) continue;
        zpos[brow[r].s0] = 2 * rid;
        if (brow[r].s1 >= 0) zpos[brow[r].s1] = 2 * rid + 1;
    }
    auto zof_pair = [&](int p) { return zpos[pair_dst[p]]; };
    for (auto &row : bsub)
        for (auto &a : row) a.first = zof_pair(a.first);
    for (int s = 0; s < S; ++s)   // rowless states: b_0[s] * start[s] = start[s] * sum over their arcs of w * z_0[pair]
        if (no_row[s] && start_lin[s] != 0.f)
            for (auto &a : out_arcs_of_state[s]) { bx_idx.push_back(zof_pair(a.first)); bx_w.push_back(a.second * start_lin[s]); }
    if (bx_idx.empty()) { bx_idx.push_back(zsink); bx_w.push_back(0.f); }
    pack_arcs(bsub, bslices, &bo, 4, *gm, bdup ? 2 * Rb : 0, bdup);
    const int noLab = -1;
    std::vector<int4> brow_meta(Rb, int4{(zsink * 4) | ((zsink * 4) << 16), 0, 0, (noLab & 0xffff) | (int)((unsigned)noLab << 16)});
    std::vector<float> brow_start((size_t)2 * Rb, 0.f), brow_end((size_t)2 * Rb, 0.f), z_end(Gb, 0.f);
    std::vector<int> z_lab(Gb, -1);
    for (int rid = 0; rid < Rb; ++rid) {
        const int r = bo.row_of[rid];
        if (r < 0) continue;
        const BRow &br = brow[r];
        const int o0 = br.e0 >= 0 ? zof_pair(br.e0) : zsink, o1 = br.e1 >= 0 ? zof_pair(br.e1) : zsink;
        const int l0 = pair_of[br.s0] >= 0 ? pair_lab[pair_of[br.s0]] : noLab;
        const int l1 = (br.s1 >= 0 && pair_of[br.s1] >= 0) ? pair_lab[pair_of[br.s1]] : noLab;
        brow_meta[rid] = int4{(o0 * 4) | ((o1 * 4) << 16), (int)wbits(br.w0), (int)wbits(br.w1), (l0 & 0xffff) | (l1 << 16)};
        brow_start[2 * rid] = start_lin[br.s0]; brow_end[2 * rid] = end_lin[br.s0];
        z_lab[2 * rid] = l0; z_end[2 * rid] = end_lin[br.s0];
        if (br.s1 >= 0) {
            brow_start[2 * rid + 1] = start_lin[br.s1]; brow_end[2 * rid + 1] = end_lin[br.s1];
            z_lab[2 * rid + 1] = l1; z_end[2 * rid + 1] = end_lin[br.s1];
        }
    }

    if (bdup) for (int z = 0; z < 2 * Rb; ++z) { z_lab[bdup + z] = z_lab[z]; z_end[bdup + z] = z_end[z]; }
    if (rcregs)   // row constants in registers: {extra-arc z offsets, labels (0xffff = none)}; the two extra weights stay in LDS
        for (const SliceAt &sl : bslices)
            for (int lane = 0; lane < kWave; ++lane) {
                const int rid = sl.rid0 + lane;
                const int4 m = brow_meta[rid];
                const size_t t = (size_t)sl.w * kWave + lane, w0 = (size_t)kFac3ArcCh * 6 + 2 * (size_t)sl.ord;
                bo.arcs[w0 * gm->threads + t] = (unsigned)m.x;
                {   // byte offsets of the two emissions (0xffff = no label)
                    const unsigned l0 = (unsigned)m.w & 0xffffu, l1 = (unsigned)m.w >> 16;
                    bo.arcs[(w0 + 1) * gm->threads + t] = (l0 == 0xffffu ? 0xffffu : l0 * 4u) | ((l1 == 0xffffu ? 0xffffu : l1 * 4u) << 16);
                }
            }

    // ---- 5. grad pass list: one (Q position, BP position) per pair, label-sorted (pairs already are), chunked
    std::vector<int> gq, gb, gchunk{0}, glab((size_t)max_lab + 2, 0);
    for (int p = 0; p < P; ++p) { gq.push_back(fpos[p]); gb.push_back(zpos[pair_dst[p]]); }
    // Chunks of at most `cap` entries within one label: kChunk = 32, or 8 when the labels have few pairs each (hundreds of classes
    // over a graph of this size: V = 500 -> ~8 pairs per label): the grad kernel's threads walk every slot of a chunk, and chunks of
    // 32 slots a quarter full spend most of their gathers on padding.  (8 only when the chunk count then fits 512 threads x 2.)
    int cap = kChunk;
    auto count_chunks = [&](int c_) { int n = 0, r = 0; for (int v = 0; v <= max_lab; ++v) { int e = r; while (e < P && pair_lab[e] == v) ++e; n += (e - r + c_ - 1) / c_; r = e; } return n; };
    {
        const int n32 = count_chunks(kChunk), n8 = count_chunks(8);
        if (n32 > 0 && (int64_t)P * 5 < (int64_t)n32 * kChunk * 2 && n8 <= 1024) cap = 8;   // chunks of 32 would be less than 40 % full
    }
    {
        int r = 0;
        for (int v = 0; v <= max_lab; ++v) {
            glab[v] = (int)gchunk.size() - 1;
            int e = r;
            while (e < P && pair_lab[e] == v) ++e;
            for (int c = r; c < e; c += cap) gchunk.push_back(std::min(e, c + cap));
            r = e;
        }
        glab[(size_t)max_lab + 1] = (int)gchunk.size() - 1;
    }
    for (int p = 1; p < P; ++p) if (pair_lab[p] < pair_lab[p - 1]) return give_up("pairs are not label-sorted");
    int64_t gcb = 0, gca = 0;
    arrange_grad_pairs(&gq, &gb, gchunk, &gcb, &gca);
    if (verbose) fprintf(stderr, "[fac_layout] grad pass: LDS cycles per frame for the pair gathers %lld as listed, %lld arranged (%d chunks)\n", (long long)gcb, (long long)gca, (int)gchunk.size() - 1);

    if (!second) h->fac_stats = FacBuildStats{0, nmatched, nsolo, (int64_t)tail_rows.size(), fo.slots, bo.slots, nfused, Gf, Gb};   // (ok: set with F.ok below)
    if (verbose)
        fprintf(stderr, "[fac_layout] matched pairs %lld, solo slots %lld, tail rows %zu (NT=%d), fwd rows %d slots %lld (extra LDS cycles %lld), bwd rows %d (fused %lld) slots %lld (extra %lld), Gf=%d Gb=%d\n",
                (long long)nmatched, (long long)nsolo, tail_rows.size(), NT, Rf, (long long)fo.slots, (long long)fo.conflicts, Rb, (long long)nfused, (long long)bo.slots, (long long)bo.conflicts, Gf, Gb);
    {   // LDS of the recursion kernels (crf_kernels.hip fac_lds_bytes): two state vectors + the row constants + two emission rows
        // (budgeted for the graph's own label range, at least 256: a call with more classes than fit falls back to the other
        // kernel families, crf_kernels.hip use_factored).  Too much with the second copy of the gathered entries: build that
        // direction again without it.  (level 1: the forward table has 8 bytes per row, and both have 64 rows of slack.)
        const int V0 = std::max(max_lab + 1, 256);
        const size_t esz = second ? 8 : 4;        // (the two-utterance kernel keeps float2 state vectors and emission rows: fac2u_lds_bytes)
        const size_t tail = ((size_t)2 * ((V0 + 1 + 63) / 64 * 64) + 4 * (size_t)gm->waves + 16) * esz + 256;
        auto need = [&](int G, int R, int rb) { return (size_t)2 * ((G + 63) / 64 * 64) * esz + (size_t)(R + (lvl_table ? 64 : 0)) * rb + tail; };
        auto cu_rows = [&](const DirOut &o) { int m = 0; for (int k = 0; k < K; ++k) m = std::max(m, o.cu_row_off[(size_t)k + 1] - o.cu_row_off[(size_t)k]); return m; };   // (two CUs: a CU's table holds its own rows)
        const bool f_ok = need(Gf, K > 1 ? cu_rows(fo) : Rf, lvl_table ? 8 : 16) <= (size_t)160 * 1024, b_ok = need(Gb, K > 1 ? cu_rows(bo) : Rb, 16) <= (size_t)160 * 1024;
        if (!f_ok || !b_ok) {
            int m = dup_mask;
            if (!f_ok && !no_dupf) m &= ~1;
            if (!b_ok && !no_dupb) m &= ~2;
            if (m != dup_mask) { *new_mask = m; return CRF_OK; }
            // (one CU per recursion: the table of ALL rows beside the two vectors -- a den_lm with a state per seen bigram history, S ~ 10 k at
            // 72 tokens, has few arcs per row and fails HERE, not on the arc slots; with two CUs per recursion a CU's table holds its own rows
            // only: let the planner go on to that geometry instead of leaving the graph to the generic layout -- round 6: S = 10 608 / A = 78 k
            // ran on the generic K = 2 layout at 7.1 ms of recursions, S = 10 632 / A = 111 k on generic K = 4 at 13.6 ms, where the factored
            // two-CU layout takes S = 10 636 / A = 147 k at 6.4 ms)
            if (K == 1 && allow3 && retry_next) { if (verbose) fprintf(stderr, "[fac_layout] one CU per recursion: state vectors and row constants exceed the LDS\n"); *retry_next = true; return CRF_OK; }
            return give_up("state vectors and row constants exceed the LDS");
        }
    }
    F.nbx = (int)bx_idx.size(); F.bx_se = bx_se;
    F.multilane = 0;
    for (auto &wi : fo.wave_info) if (wi.w) F.multilane = 1;
    for (auto &wi : bo.wave_info) if (wi.w) F.multilane = 1;
    F.f.R = Rf; F.f.G = Gf; F.b.R = Rb; F.b.G = Gb; F.f.dup = fdup * 4; F.b.dup = bdup * 4;
    F.NT = NT; F.Rq = Rq; F.Rbp = 2 * Rb; F.NC = (int)gchunk.size() - 1; F.chunk_cap = cap; F.threads = gm->threads; F.imp = implicit ? 1 : 0; F.rcl = level == 1 ? 1 : level == 3 ? 2 : (level == 4 || level == 6) ? 1 : 0; F.K = K;
    for (int k = 0; k < 3; ++k) F.xlist_off[k] = xlist_off[k];
    for (int k = 0; k <= 2; ++k) { F.f.cu_row[k] = fo.cu_row_off[(size_t)std::min(k, K)]; F.b.cu_row[k] = bo.cu_row_off[(size_t)std::min(k, K)]; }
    int rc;
    if ((rc = up(h, fo.arcs, &F.f.arcs)) || (rc = up(h, fo.wave_info, &F.f.wave_info)) || (rc = up(h, bo.arcs, &F.b.arcs)) ||
        (rc = up(h, bo.wave_info, &F.b.wave_info)) || (rc = up(h, frow_meta, &F.frow_meta)) ||
        (rc = up(h, x_start, &F.x_start)) || (rc = up(h, x_end, &F.x_end)) || (rc = up(h, brow_meta, &F.brow_meta)) ||
        (rc = up(h, z_lab, &F.z_lab)) || (rc = up(h, z_end, &F.z_end)) || (rc = up(h, brow_start, &F.brow_start)) ||
        (rc = up(h, brow_end, &F.brow_end)) || (rc = up(h, bx_idx, &F.bx_idx)) || (rc = up(h, bx_w, &F.bx_w)) || (rc = up(h, gq, &F.gq)) || (rc = up(h, gb, &F.gb)) ||
        (rc = up(h, gchunk, &F.chunk_off)) || (rc = up(h, glab, &F.lab_chunk_off)) || (rc = up(h, xlist, &F.xlist)))
        return rc;
    F.ok = 1;
    if (!second) h->fac_stats.ok = 1;
    FacHostCopy &C = second ? h->fhp : h->fh;
    C.farcs = fo.arcs; C.barcs = bo.arcs; C.fwi = fo.wave_info; C.bwi = bo.wave_info; C.frow_meta = frow_meta; C.brow_meta = brow_meta;
    C.x_start = x_start; C.x_end = x_end; C.z_end = z_end; C.brow_start = brow_start; C.brow_end = brow_end; C.bx_w = bx_w;
    C.start_lin = start_lin; C.end_lin = end_lin; C.z_lab = z_lab; C.bx_idx = bx_idx; C.xlist = xlist; C.words = gm->words;
    C.gq = gq; C.gb = gb; C.gchunk = gchunk; C.glab = glab;
    return CRF_OK;
}


int build_factored(HostGraph *h, int S, int P, const std::vector<int> &pair_dst, const std::vector<int> &pair_lab,
                   const Rows &in_arcs_of_pair, const Rows &out_arcs_of_state, const std::vector<float> &start_lin,
                   const std::vector<float> &end_lin) {
    // Geometry: 768 threads x 21 chunks (3 waves per SIMD at <= 168 VGPRs) when both directions fit it, else
    // 512 threads x 30 chunks (2 waves per SIMD); CRF_FAC_THREADS=512 forces the latter.
    // ... then 768 threads with the row constants in LDS (any number of slices per wave; CRF_FAC_RCL=1 starts there,
    // CRF_FAC_NO_RCL=1 skips it).  Each geometry first with the second copy of the gathered entries, then without it in the direction(s) that do not fit.
    const int thr = opt(kOpt_fac_threads, 0);          // 0: the planner's order; 512 / 768 / 1024: that geometry (first)
    const bool no_rcl = opt_on(kOpt_fac_no_rcl);
    const bool from_rcl = opt_on(kOpt_fac_rcl);
    int rc = CRF_OK;
    struct Try { int level; bool short_only; int K = 1; };
    std::vector<Try> plan;
    bool dflt = false;
    if (thr == 512) plan = {{2, false}};
    else if (opt_on(kOpt_fac_k2)) plan = thr == 1024 ? std::vector<Try>{{4, false, 2}, {2, false}} : std::vector<Try>{{1, false, 2}, {2, false}};   // (tests: two CUs per recursion for any T o LM graph; with fac_threads = 1024 on that geometry)
    else if (opt(kOpt_fac_rcl, 0) == 2) plan = {{3, false}, {2, false}};   // (tests: the 21-chunk table geometry for any T o LM graph)
    else if (from_rcl) plan = {{1, false}, {3, false}, {2, false}};
    else if (no_rcl) plan = {{0, false}, {2, false}};
    else {
        // 768 threads: level 0 (row constants in registers) only for graphs without long rows -- unless level 1 (LDS table) does not
        // take them; then the 21-chunk table geometry, two CUs per recursion (table geometry), 512 threads.  In front of all of
        // them since round 3: 1024 threads x 15 chunks with the table (level 4: four waves per SIMD at 128 VGPRs -- the same
        // registers per wave are left beside the arcs as at 768 x 21 -- measured 2 - 6 % faster on every graph that fits it)
        // (Two CUs per recursion stay on 768 threads: built on 1024 x 15 in round 4 -- `fac_k2` with `fac_threads` = 1024, geometry 5 --
        // and measured SLOWER, H = 3 072 / 4 096 recursions 4.13 -> 4.31 / 5.84 -> 6.3 ms: the hand-off's polls and granule stores are per
        // wave, and there are a third more waves.)
        plan = {{0, true}, {1, false}, {0, false}, {3, false}, {1, false, 2}, {2, false}};
        dflt = true;
        if (thr != 768) plan.insert(plan.begin(), Try{4, false});
    }
    const bool no_k2 = opt_on(kOpt_fac_no_k2);
    bool long_bail = false;
    for (const Try &t : plan) {
        if (t.level == 0 && !t.short_only && dflt && !long_bail) continue;   // level 0 has been tried in full already
        if (t.K > 1 && no_k2) continue;
        bool retry = false;
        int mask = 3, nm = 3;
        for (;;) {
            rc = build_factored_impl(h, S, P, pair_dst, pair_lab, in_arcs_of_pair, out_arcs_of_state, start_lin, end_lin, t.level, &retry, mask, &nm, t.short_only, &long_bail, t.K);
            if (rc != CRF_OK || retry || nm == mask) break;
            mask = nm;
        }
        if (rc != CRF_OK || !retry) break;
    }
    // The two-utterance kernel's own layout (512 threads x 30 chunks, table, implicit entries) beside a main layout on one CU per
    // recursion that is not a 768-thread one (those the two-utterance kernel takes as they are).  Does not fit / no structure: facp.ok = 0.
    h->facp = FacDev{};
    if (rc == CRF_OK && h->dev.fac.ok && h->dev.fac.K == 1 && h->dev.fac.threads != kFac3Threads && !opt_on(kOpt_no_facp) && thr != 512) {
        bool retry = false;
        int mask = 3, nm = 3;
        for (;;) {
            rc = build_factored_impl(h, S, P, pair_dst, pair_lab, in_arcs_of_pair, out_arcs_of_state, start_lin, end_lin, 6, &retry, mask, &nm, false, nullptr, 1);
            if (rc != CRF_OK || retry || nm == mask) break;
            mask = nm;
        }
        if (rc != CRF_OK) { h->facp = FacDev{}; rc = CRF_OK; }   // (the second layout is optional: a failure while building it must not fail a graph whose main layout stands)
        if (opt_on(kOpt_verbose)) fprintf(stderr, "[fac_layout] second layout (two utterances per workgroup, 512 x 30): %s\n", h->facp.ok ? "built" : "not available");
    }
    return rc;
}


// CPU emulation of the factored recursion kernels' DATA FLOW on the layout tables (tests; no GPU): the packed arc words of
// every CU / wave / lane, slice ends, the butterfly over multi-lane rows, row constants (registers or table), implicit or
// tabulated entries, the second copy, the rowless states, and for two CUs per recursion a private vector per CU that only
// receives what the kernel fetches (the peer's U range, its list of L / A entries; everything the kernel does not fetch is
// NaN, so a gather of an entry that never crosses poisons the result).  fp64, no rescaling, random emissions, T frames.
// out3 = {plain forward sum over the graph's own tables, factored forward, factored backward}; they must agree.
static double plain_forward(const HostGraph *h, const std::vector<std::vector<double>> &e);
int debug_emulate_factored(const HostGraph *h, int T, unsigned seed, double *out3, int which) {
    const FacDev &F = which ? h->facp : h->dev.fac;
    const FacHostCopy &C = which ? h->fhp : h->fh;
    out3[0] = out3[1] = out3[2] = 0.0;
    if (!F.ok || C.words == 0) { set_error("no factored layout"); return CRF_ERR_UNSUPPORTED; }
    auto fail = [&](const std::string &why) { set_error("factored layout emulation: " + why); return CRF_ERR_ARG; };
    const int S = (int)h->S, V = h->dev.max_label + 1, NTH = F.threads, NW = NTH / kWave, K = F.K, words = C.words;
    const bool implicit = F.imp != 0, rcregs = implicit && !F.rcl;
    uint64_t rng = 0x9E3779B97F4A7C15ull ^ seed;
    auto rnd = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return 0.5 + (double)(rng % 1000003) / 1000003.0; };
    std::vector<std::vector<double>> e((size_t)T, std::vector<double>((size_t)V + 1, 0.0));   // e[t][V] = 0: "no label"
    for (auto &row : e) for (int v = 0; v < V; ++v) row[(size_t)v] = rnd();
    auto fl = [](unsigned b) { float f; memcpy(&f, &b, 4); return (double)f; };
    const double nan = std::nan("");
    if (h->h_src.empty() && h->A > 0) { set_error("the graph's arcs were not kept (more than 2^20)"); return CRF_ERR_UNSUPPORTED; }
    out3[0] = plain_forward(h, e);
    // rows as the recursions store them for the grad pass: Q[t] = {row sums of the main rows, w * U of their tail rows},
    // BP[t] = b_{t+1} of the (up to) two states of every backward row; BP[T - 1] = the end weights
    std::vector<std::vector<double>> Qrows((size_t)T, std::vector<double>((size_t)2 * F.f.R, 0.0)), BProws((size_t)T, std::vector<double>((size_t)2 * F.b.R, 0.0));
    for (int r = 0; r < 2 * F.b.R; ++r) BProws[(size_t)T - 1][(size_t)r] = (double)C.brow_end[(size_t)r];
    // ---- the factored layout, one direction
    auto run = [&](int dir, double *result) -> int {
        const FacDirDev &L = dir == 0 ? F.f : F.b;
        const std::vector<unsigned> &A = dir == 0 ? C.farcs : C.barcs;
        const std::vector<uint4> &WI = dir == 0 ? C.fwi : C.bwi;
        const int G = L.G, R = L.R, dup = L.dup / 4;
        if ((int)A.size() != K * words * NTH || (int)WI.size() != K * NW) return fail("table sizes");
        std::vector<std::vector<double>> X[2];
        for (int q = 0; q < 2; ++q) X[q].assign((size_t)K, std::vector<double>((size_t)G, 0.0));
        std::vector<double> b0rows((size_t)2 * R, 0.0);
        for (int k = 0; k < K; ++k)
            for (int z = 0; z < G; ++z)
                X[0][(size_t)k][(size_t)z] = dir == 0 ? (double)C.x_start[(size_t)z]
                                                      : e[(size_t)T - 1][(size_t)(C.z_lab[(size_t)z] < 0 ? V : C.z_lab[(size_t)z])] * (double)C.z_end[(size_t)z];
        const int produced = dir == 0 ? 3 * R : 2 * R;          // entries [0, produced) are written by rows
        for (int i = 0; i < T; ++i) {
            const int par = i & 1, t = dir == 0 ? i : T - 1 - i;
            const std::vector<double> &em = dir == 0 ? e[(size_t)t] : e[(size_t)(t >= 1 ? t - 1 : 0)];
            for (int k = 0; k < K; ++k) {
                const std::vector<double> &src = X[par][(size_t)k];
                std::vector<double> &dst = X[1 - par][(size_t)k];
                if (K > 1) for (int z = 0; z < produced && z < G; ++z) dst[(size_t)z] = nan;
                for (int w = 0; w < NW; ++w) {
                    const uint4 wi = WI[(size_t)k * NW + w];
                    const unsigned ends = wi.x, lgbits = wi.w;
                    const int nch = (int)wi.y;
                    int row = (int)wi.z, slice = 0;
                    if (nch > (rcregs ? kFac3ArcCh : words / 6)) return fail("a wave uses more chunks than the geometry has");
                    double acc[kWave];
                    for (double &x : acc) x = 0.0;
                    for (int c = 0; c < nch; ++c) {
                        for (int lane = 0; lane < kWave; ++lane) {
                            const size_t base = (size_t)k * words * NTH + (size_t)(w * kWave + lane);
                            auto W = [&](int j) { return A[base + (size_t)(6 * c + j) * NTH]; };
                            const unsigned i01 = W(0), i23 = W(1);
                            const unsigned offs[4] = {i01 & 0xffffu, i01 >> 16, i23 & 0xffffu, i23 >> 16};
                            for (int q = 0; q < 4; ++q) {
                                if (offs[q] % 4 || (int)(offs[q] / 4) >= G) return fail("a gather offset outside the vector");
                                const double wq = fl(W(2 + q));
                                if (wq != 0.0) acc[lane] += wq * src[offs[q] / 4];     // (padding gathers: weight 0)
                            }
                        }
                        if (!(ends >> c & 1u)) continue;
                        const unsigned lg = slice < 10 ? (lgbits >> (3 * slice)) & 7u : 0u;
                        double tot[kWave];
                        for (int lane = 0; lane < kWave; ++lane) {
                            const int g0 = lane & ~((1 << lg) - 1);
                            double sacc = 0.0;
                            for (int q = 0; q < (1 << lg); ++q) sacc += acc[g0 + q];
                            tot[lane] = sacc;
                        }
                        for (int lane = 0; lane < kWave; ++lane) {
                            const int rid = row + lane;
                            if (rid < L.cu_row[k] || rid >= L.cu_row[k + 1]) return fail("a wave's row outside its CU's range");
                            if (dir == 0) {
                                const int4 m = C.frow_meta[(size_t)rid];
                                unsigned lab_m = (unsigned)m.x >> 16, lab_t = (unsigned)m.w;
                                double tw = fl((unsigned)m.z);
                                if (rcregs) {   // the register copies of the constants must say the same
                                    const size_t base = (size_t)k * words * NTH + (size_t)(w * kWave + lane);
                                    const unsigned k0 = A[base + (size_t)(kFac3ArcCh * 6 + 2 * slice) * NTH], k1 = A[base + (size_t)(kFac3ArcCh * 6 + 2 * slice + 1) * NTH];
                                    if (slice >= kFac3MaxSl) return fail("more slices in a wave than row-constant registers");
                                    if (k0 != ((lab_m * 4u) | ((lab_t * 4u) << 16)) || k1 != (unsigned)m.z) return fail("forward row constants in registers differ from the table");
                                }
                                const int uo = implicit ? rid : (m.x & 0xffff) / 4, lo = implicit ? R + rid : (m.y & 0xffff) / 4, ao = implicit ? 2 * R + rid : (int)((unsigned)m.y >> 16) / 4;
                                if (implicit && tw != 0.0 && ((m.x & 0xffff) / 4 != rid || (m.y & 0xffff) / 4 != R + rid || (int)((unsigned)m.y >> 16) / 4 != 2 * R + rid))
                                    return fail("implicit entries differ from the row table");
                                const double uold = src[(size_t)uo], rv = tot[lane], qt = tw * uold;
                                Qrows[(size_t)t][(size_t)rid] = rv; Qrows[(size_t)t][(size_t)R + rid] = qt;
                                const double Lp = em[(size_t)lab_m] * rv, Ap = em[(size_t)lab_t] * qt, Up = Lp + Ap;
                                dst[(size_t)uo] = Up; dst[(size_t)lo] = Lp; dst[(size_t)ao] = Ap;
                                if (dup) dst[(size_t)uo + dup] = Up;
                            } else {
                                const int4 m = C.brow_meta[(size_t)rid];
                                const unsigned o0 = (unsigned)m.x & 0xffffu, o1 = (unsigned)m.x >> 16, l0 = (unsigned)m.w & 0xffffu, l1 = (unsigned)m.w >> 16;
                                if (rcregs) {
                                    const size_t base = (size_t)k * words * NTH + (size_t)(w * kWave + lane);
                                    const unsigned k0 = A[base + (size_t)(kFac3ArcCh * 6 + 2 * slice) * NTH], k1 = A[base + (size_t)(kFac3ArcCh * 6 + 2 * slice + 1) * NTH];
                                    if (slice >= kFac3MaxSl) return fail("more slices in a wave than row-constant registers");
                                    const unsigned want1 = (l0 == 0xffffu ? 0xffffu : l0 * 4u) | ((l1 == 0xffffu ? 0xffffu : l1 * 4u) << 16);
                                    if (k0 != (unsigned)m.x || k1 != want1) return fail("backward row constants in registers differ from the table");
                                }
                                const double z0 = src[o0 / 4], z1 = src[o1 / 4];
                                const double bv0 = fl((unsigned)m.y) * z0 + tot[lane], bv1 = fl((unsigned)m.z) * z1 + tot[lane];
                                if (t == 0) { b0rows[(size_t)2 * rid] = bv0; b0rows[(size_t)2 * rid + 1] = bv1; }
                                else { BProws[(size_t)t - 1][(size_t)2 * rid] = bv0; BProws[(size_t)t - 1][(size_t)2 * rid + 1] = bv1; }
                                const double zv0 = em[(size_t)(l0 == 0xffffu ? V : l0)] * bv0, zv1 = em[(size_t)(l1 == 0xffffu ? V : l1)] * bv1;
                                dst[(size_t)2 * rid] = zv0; dst[(size_t)2 * rid + 1] = zv1;
                                if (dup) { dst[(size_t)2 * rid + dup] = zv0; dst[(size_t)2 * rid + 1 + dup] = zv1; }
                            }
                        }
                        for (double &x : acc) x = 0.0;
                        row += kWave; ++slice;
                    }
                }
            }
            if (K > 1) {   // what the kernel fetches from the peer
                for (int k = 0; k < K; ++k) {
                    const std::vector<double> &peer = X[1 - par][(size_t)(1 - k)];
                    std::vector<double> &mine = X[1 - par][(size_t)k];
                    const int p0 = L.cu_row[1 - k], p1 = L.cu_row[2 - k];
                    if (dir == 0) {
                        for (int z = p0; z < p1; ++z) mine[(size_t)z] = peer[(size_t)z];
                        if (i == T - 1) { if (k == 0) for (int z = p0; z < p1; ++z) { mine[(size_t)R + z] = peer[(size_t)R + z]; mine[(size_t)2 * R + z] = peer[(size_t)2 * R + z]; } }
                        else if (!opt_on(kOpt_emu_drop_list)) for (int j = F.xlist_off[k]; j < F.xlist_off[k + 1]; ++j) {   // (negative control of the test)
                            const int en = C.xlist[(size_t)j], rid = en % R;
                            if (en < R || en >= 3 * R || rid < p0 || rid >= p1) return fail("a listed entry is not an L / A entry of the peer");
                            mine[(size_t)en] = peer[(size_t)en];
                        }
                    } else {
                        for (int z = 2 * p0; z < 2 * p1; ++z) mine[(size_t)z] = peer[(size_t)z];
                    }
                }
            }
            if (dir == 0 && i == 0)   // entries no row produces still hold a_0 in the buffer frame 0 read from: cleared
                for (int k = 0; k < K; ++k) for (int z = 0; z < G; ++z) if (C.x_start[(size_t)z] != 0.f) X[0][(size_t)k][(size_t)z] = 0.0;
        }
        double sum = 0.0;
        if (dir == 0) {
            for (int z = 0; z < G; ++z) if (C.x_end[(size_t)z] != 0.f) sum += X[T & 1][0][(size_t)z] * (double)C.x_end[(size_t)z];
        } else {
            for (int r = 0; r < 2 * R; ++r) if (C.brow_start[(size_t)r] != 0.f) sum += (double)C.brow_start[(size_t)r] * b0rows[(size_t)r];
            const std::vector<double> &Xl = X[(T - 1) & 1][0];
            for (int a2 = 0; a2 < F.nbx; ++a2) sum += (double)C.bx_w[(size_t)a2] * Xl[(size_t)C.bx_idx[(size_t)a2]];
        }
        *result = sum;
        return CRF_OK;
    };
    int rc = run(0, &out3[1]);
    if (!rc) rc = run(1, &out3[2]);
    if (rc) return rc;
    // ---- the grad pass's lists: for every frame, sum over the label-sorted (Q position, BP position) pairs of
    // e_t[label] * Q[t][q] * BP[t][b] is the total path mass again
    const int NC = (int)C.gchunk.size() - 1;
    for (int t = 0; t < T; ++t) {
        double tot = 0.0;
        for (int v = 0; v < V && v + 1 < (int)C.glab.size(); ++v)
            for (int c = C.glab[(size_t)v]; c < C.glab[(size_t)v + 1]; ++c) {
                if (c < 0 || c >= NC) return fail("label -> chunk table");
                if (C.gchunk[(size_t)c + 1] - C.gchunk[(size_t)c] > kChunk) return fail("a grad chunk is longer than kChunk");
                double sc = 0.0;
                for (int j = C.gchunk[(size_t)c]; j < C.gchunk[(size_t)c + 1]; ++j) {
                    const int q = C.gq[(size_t)j], b = C.gb[(size_t)j];
                    if (q < 0 || q >= 2 * F.f.R || b < 0 || b >= 2 * F.b.R) return fail("a grad pair outside the rows");
                    sc += Qrows[(size_t)t][(size_t)q] * BProws[(size_t)t][(size_t)b];
                }
                tot += e[(size_t)t][(size_t)v] * sc;
            }
        if (opt_on(kOpt_emu_verbose)) fprintf(stderr, "[emu] frame %d: pair-list mass %.12g, path mass %.12g\n", t, tot, out3[0]);
        if (!(std::fabs(tot - out3[0]) <= 1e-9 * std::fabs(out3[0]))) return fail("the grad pass's pair lists do not give the path mass at frame " + std::to_string(t));
    }
    return CRF_OK;
}


// plain recursion over the graph's arcs (the emulations' reference): sum over end states after T frames of emissions e
static double plain_forward(const HostGraph *h, const std::vector<std::vector<double>> &e) {
    const int S = (int)h->S;
    std::vector<double> a(h->h_start.begin(), h->h_start.end()), an((size_t)S);
    for (size_t t = 0; t < e.size(); ++t) {
        std::fill(an.begin(), an.end(), 0.0);
        for (size_t k = 0; k < h->h_src.size(); ++k) an[(size_t)h->h_dst[k]] += e[t][(size_t)h->h_lab[k]] * (double)h->h_w[k] * a[(size_t)h->h_src[k]];
        a.swap(an);
    }
    double z = 0.0;
    for (int s2 = 0; s2 < S; ++s2) z += a[(size_t)s2] * (double)h->h_end[(size_t)s2];
    return z;
}

// CPU emulation of the GENERIC register-resident kernels' data flow on the layout tables (tests; no GPU): every row -- pair
// rows forward, state copies backward -- sums its packed arcs and produces ONE entry, own_off[k] + (row - cu_row_off[k]);
// every produced entry crosses to the peers, so one vector per direction serves all CUs here.
int debug_emulate_resident(const HostGraph *h, int T, unsigned seed, double *out3) {
    const ResDev &R = h->dev.res;
    const ResHostCopy &C = h->rh;
    out3[0] = out3[1] = out3[2] = 0.0;
    if (R.K < 1 || C.farcs.empty()) { set_error("no generic register-resident layout"); return CRF_ERR_UNSUPPORTED; }
    if (h->h_src.empty() && h->A > 0) { set_error("the graph's arcs were not kept (more than 2^20)"); return CRF_ERR_UNSUPPORTED; }
    auto fail = [&](const std::string &why) { set_error("generic layout emulation: " + why); return CRF_ERR_ARG; };
    const int V = h->dev.max_label + 1, K = R.K;
    uint64_t rng = 0x9E3779B97F4A7C15ull ^ seed;
    auto rnd = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return 0.5 + (double)(rng % 1000003) / 1000003.0; };
    std::vector<std::vector<double>> e((size_t)T, std::vector<double>((size_t)V + 1, 0.0));
    for (auto &row : e) for (int v = 0; v < V; ++v) row[(size_t)v] = rnd();
    auto fl = [](unsigned b) { float f; memcpy(&f, &b, 4); return (double)f; };
    out3[0] = plain_forward(h, e);
    std::vector<std::vector<double>> Qrows((size_t)T, std::vector<double>((size_t)R.f.R, 0.0)), BProws((size_t)T, std::vector<double>((size_t)R.b.R, 0.0));
    for (int r = 0; r < R.b.R; ++r) BProws[(size_t)T - 1][(size_t)r] = (double)C.brow_end[(size_t)r];
    auto run = [&](int dir, double *result) -> int {
        const ResDirDev &L = dir == 0 ? R.f : R.b;
        const std::vector<unsigned> &A = dir == 0 ? C.farcs : C.barcs;
        const std::vector<uint4> &WI = dir == 0 ? C.fwi : C.bwi;
        const std::vector<int> &lab = dir == 0 ? C.flab : C.blab, &cu = dir == 0 ? C.fcu : C.bcu, &own = dir == 0 ? C.fown : C.bown;
        const int G = L.G, NTH = kResThreads, NW = kResWaves, words = kResWords;
        if ((int)A.size() != K * words * NTH || (int)WI.size() != K * NW || (int)cu.size() < K + 1 || (int)own.size() < K + 1) return fail("table sizes");
        std::vector<double> X[2] = {std::vector<double>((size_t)G, 0.0), std::vector<double>((size_t)G, 0.0)}, b0rows((size_t)L.R, 0.0);
        std::vector<char> produced((size_t)G, 0);
        for (int z = 0; z < G; ++z) X[0][(size_t)z] = dir == 0 ? (double)C.x_start[(size_t)z] : e[(size_t)T - 1][(size_t)(C.z_lab[(size_t)z] < 0 ? V : C.z_lab[(size_t)z])] * (double)C.z_end[(size_t)z];
        for (int i = 0; i < T; ++i) {
            const int par = i & 1, t = dir == 0 ? i : T - 1 - i;
            const std::vector<double> &em = dir == 0 ? e[(size_t)t] : e[(size_t)(t >= 1 ? t - 1 : 0)];
            const std::vector<double> &src = X[par];
            std::vector<double> &dst = X[1 - par];
            for (int k = 0; k < K; ++k)
                for (int w = 0; w < NW; ++w) {
                    const uint4 wi = WI[(size_t)k * NW + w];
                    const unsigned ends = wi.x;
                    const int nch = (int)wi.y;
                    int row = (int)wi.z;
                    if (nch > kResNCH) return fail("a wave uses more chunks than a thread has");
                    double acc[kWave];
                    for (double &x : acc) x = 0.0;
                    for (int c = 0; c < nch; ++c) {
                        for (int lane = 0; lane < kWave; ++lane) {
                            const size_t base = (size_t)k * words * NTH + (size_t)(w * kWave + lane);
                            auto W = [&](int j) { return A[base + (size_t)(6 * c + j) * NTH]; };
                            const unsigned i01 = W(0), i23 = W(1);
                            const unsigned offs[4] = {i01 & 0xffffu, i01 >> 16, i23 & 0xffffu, i23 >> 16};
                            for (int q = 0; q < 4; ++q) {
                                if (offs[q] % 4 || (int)(offs[q] / 4) >= G) return fail("a gather offset outside the vector");
                                const double wq = fl(W(2 + q));
                                if (wq != 0.0) acc[lane] += wq * src[offs[q] / 4];
                            }
                        }
                        if (!(ends >> c & 1u)) continue;
                        for (int lane = 0; lane < kWave; ++lane) {
                            const int rid = row + lane;
                            if (rid < cu[(size_t)k] || rid >= cu[(size_t)k + 1]) return fail("a wave's row outside its CU's range");
                            const int en = own[(size_t)k] + (rid - cu[(size_t)k]);
                            if (en < 0 || en >= G) return fail("a produced entry outside the vector");
                            if (i == 0) { if (produced[(size_t)en]) return fail("two rows produce one entry"); produced[(size_t)en] = 1; }
                            const double rv = acc[lane];
                            const int l = lab[(size_t)rid];
                            if (dir == 0) Qrows[(size_t)t][(size_t)rid] = rv;
                            else if (t == 0) b0rows[(size_t)rid] = rv;
                            else BProws[(size_t)t - 1][(size_t)rid] = rv;
                            dst[(size_t)en] = em[(size_t)(l < 0 || l > V ? V : l)] * rv;
                        }
                        for (double &x : acc) x = 0.0;
                        row += kWave;
                    }
                }
            if (dir == 0 && i == 0) for (int z = 0; z < G; ++z) if (C.x_start[(size_t)z] != 0.f) X[0][(size_t)z] = 0.0;
        }
        double sum = 0.0;
        if (dir == 0) for (int z = 0; z < G; ++z) sum += X[T & 1][(size_t)z] * (double)C.x_end[(size_t)z];
        else for (int r = 0; r < L.R; ++r) sum += (double)C.brow_start[(size_t)r] * b0rows[(size_t)r];
        *result = sum;
        return CRF_OK;
    };
    int rc = run(0, &out3[1]);
    if (!rc) rc = run(1, &out3[2]);
    if (rc) return rc;
    const int NC = (int)C.gchunk.size() - 1;
    for (int t = 0; t < T; ++t) {
        double tot = 0.0;
        for (int v = 0; v < V && v + 1 < (int)C.glab.size(); ++v)
            for (int c = C.glab[(size_t)v]; c < C.glab[(size_t)v + 1]; ++c) {
                if (c < 0 || c >= NC) return fail("label -> chunk table");
                double sc = 0.0;
                for (int j = C.gchunk[(size_t)c]; j < C.gchunk[(size_t)c + 1]; ++j) {
                    const int q = C.gq[(size_t)j], b = C.gb[(size_t)j];
                    if (q < 0 || q >= R.f.R || b < 0 || b >= R.b.R) return fail("a grad pair outside the rows");
                    sc += Qrows[(size_t)t][(size_t)q] * BProws[(size_t)t][(size_t)b];
                }
                tot += e[(size_t)t][(size_t)v] * sc;
            }
        if (!(std::fabs(tot - out3[0]) <= 1e-9 * std::fabs(out3[0]))) return fail("the grad pass's pair lists do not give the path mass at frame " + std::to_string(t));
    }
    return CRF_OK;
}

}  // namespace crf
