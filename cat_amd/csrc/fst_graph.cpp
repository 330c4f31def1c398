// cat_amd/csrc/fst_graph.cpp -- host side of the denominator graph: OpenFst-binary reader (no OpenFst)
// and the "graph compiler" that turns the arc list into the tables the gfx950 kernels stream.
//
// Replaces reference src/ctc_crf/gpu_den/fst_read.cc:11-62 (ReadFst on top of OpenFst 1.6.7, which
// is not vendored) and the table construction + upload of Init, den_calculate.cu:288-392.
// Conventions are the reference's (fst_read.cc:40-60): label = ilabel-1, weight = -cost,
// end_weight = -Final, start_weight[start] = 0.  Epsilon input labels are rejected (the reference
// would read logits[-1], fst_read.cc:55-56).
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <numeric>

#include "../../include/ctc_crf_hip.h"
#include "crf_internal.h"

namespace crf {

static thread_local std::string g_err;
void set_error(const std::string &msg) { g_err = msg; }

// debug / experiment switches (crf_internal.h: CRF_OPTS)
static std::atomic<int> g_opt[kOptCount];
static std::atomic<unsigned long long> g_opt_set{0};   // bit k: switch k has a value
static_assert(kOptCount <= 64, "one bit per switch");
static const char *const kOptName[kOptCount] = {
#define CRF_OPT_NAME(name, doc) #name,
    CRF_OPTS(CRF_OPT_NAME)
#undef CRF_OPT_NAME
};
static const char *const kOptDoc[kOptCount] = {
#define CRF_OPT_DOC(name, doc) doc,
    CRF_OPTS(CRF_OPT_DOC)
#undef CRF_OPT_DOC
};
int opt(Opt k, int dflt) { return (g_opt_set.load(std::memory_order_relaxed) >> (int)k & 1ull) ? g_opt[k].load(std::memory_order_relaxed) : dflt; }
int opt_set(const char *key, int value, bool unset) {
    if (!key) { set_error("crf_debug_set: null key"); return CRF_ERR_ARG; }
    for (int k = 0; k < kOptCount; ++k)
        if (!strcmp(key, kOptName[k])) {
            if (unset) g_opt_set.fetch_and(~(1ull << k));
            else { g_opt[k].store(value); g_opt_set.fetch_or(1ull << k); }
            return CRF_OK;
        }
    set_error(std::string("crf_debug_set: unknown switch '") + key + "'");
    return CRF_ERR_ARG;
}
const char *opt_list() {
    static const std::string s = [] {
        std::string r;
        for (int k = 0; k < kOptCount; ++k) r += std::string(kOptName[k]) + ": " + kOptDoc[k] + "\n";
        return r;
    }();
    return s.c_str();
}
const char *last_error_cstr() { return g_err.c_str(); }

#define HIP_TRY(expr)                                                                             \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            set_error(std::string(#expr) + ": " + hipGetErrorString(e_));                         \
            return CRF_ERR_HIP;                                                                   \
        }                                                                                         \
    } while (0)

namespace {
struct Cursor {
    const unsigned char *p;
    size_t n, off = 0;
    bool ok = true;
    template <typename T>
    T get() {
        T v{};
        if (off + sizeof(T) > n) { ok = false; return v; }
        memcpy(&v, p + off, sizeof(T));
        off += sizeof(T);
        return v;
    }
    std::string str() {
        int32_t len = get<int32_t>();
        if (!ok || len < 0 || off + (size_t)len > n) { ok = false; return {}; }
        std::string s((const char *)p + off, (size_t)len);
        off += (size_t)len;
        return s;
    }
    void skip_symtab() {  // OpenFst SymbolTable binary: magic, name, available_key, size, entries
        get<int32_t>(); str(); get<int64_t>();
        int64_t cnt = get<int64_t>();
        for (int64_t i = 0; ok && i < cnt; ++i) { str(); get<int64_t>(); }
    }
};
}  // namespace

int read_fst_file(const char *path, int64_t *S, std::vector<int32_t> *src, std::vector<int32_t> *dst,
                  std::vector<int32_t> *lab, std::vector<float> *w, std::vector<float> *start_w,
                  std::vector<float> *end_w) {
    FILE *f = fopen(path, "rb");
    if (!f) { set_error(std::string("cannot open den_lm: ") + path); return CRF_ERR_IO; }
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<unsigned char> buf((size_t)std::max(0L, sz));
    size_t got = buf.empty() ? 0 : fread(buf.data(), 1, buf.size(), f);
    fclose(f);
    if (got != buf.size()) { set_error(std::string("short read: ") + path); return CRF_ERR_IO; }
    Cursor c{buf.data(), buf.size()};
    if (c.get<uint32_t>() != 0x7eb2fdd6u) { set_error(std::string(path) + ": not an OpenFst binary (bad magic)"); return CRF_ERR_FORMAT; }
    std::string fsttype = c.str(), arctype = c.str();
    if (fsttype != "vector" || arctype != "standard") {
        set_error(std::string(path) + ": need fst type vector/standard, got " + fsttype + "/" + arctype);
        return CRF_ERR_FORMAT;
    }
    c.get<int32_t>();  // version
    int32_t flags = c.get<int32_t>();
    c.get<uint64_t>();  // properties
    int64_t start = c.get<int64_t>();
    int64_t ns = c.get<int64_t>();
    c.get<int64_t>();  // narcs (0 in VectorFst headers; arcs are counted per state)
    if (flags & 1) c.skip_symtab();
    if (flags & 2) c.skip_symtab();
    if (!c.ok || ns <= 0 || ns > INT32_MAX || start < 0 || start >= ns) {
        set_error(std::string(path) + ": corrupt header / no start state");
        return CRF_ERR_FORMAT;
    }
    *S = ns;
    start_w->assign((size_t)ns, -INFINITY);
    end_w->assign((size_t)ns, -INFINITY);
    (*start_w)[(size_t)start] = 0.f;
    for (int64_t s = 0; s < ns; ++s) {
        float fin = c.get<float>();
        int64_t na = c.get<int64_t>();
        if (!c.ok || na < 0) { set_error(std::string(path) + ": truncated state record"); return CRF_ERR_FORMAT; }
        if (fin != INFINITY) (*end_w)[(size_t)s] = -fin;
        for (int64_t k = 0; k < na; ++k) {
            int32_t il = c.get<int32_t>();
            c.get<int32_t>();
            float cost = c.get<float>();
            int32_t nx = c.get<int32_t>();
            if (!c.ok) { set_error(std::string(path) + ": truncated arc record"); return CRF_ERR_FORMAT; }
            if (il <= 0) { set_error(std::string(path) + ": epsilon / negative input label on an arc (den_lm must be epsilon-free)"); return CRF_ERR_FORMAT; }
            if (nx < 0 || nx >= ns) { set_error(std::string(path) + ": arc to a non-existent state"); return CRF_ERR_FORMAT; }
            src->push_back((int32_t)s); dst->push_back(nx); lab->push_back(il - 1); w->push_back(-cost);
        }
    }
    if (src->size() > (size_t)INT32_MAX) { set_error("too many arcs"); return CRF_ERR_UNSUPPORTED; }
    return CRF_OK;
}

namespace {

struct EllHost {
    std::vector<uint4> arcs;
    std::vector<int> slice_off, slice_w2, wave_off, wave_slices;
    std::vector<int> row_of;  // row position -> original row id (-1 padding)
    int64_t padded_arcs = 0;
    int64_t conflict_cycles = 0;  // gathers that land on an already-used bank (extra LDS cycles)
};

// rows[r] = list of (idx, w).  Rows are sorted by degree (descending, stable) and cut into slices
// of 64; each slice is as wide as its first (longest) row, rounded up to an even arc count.
EllHost build_ell(const std::vector<std::vector<std::pair<int, float>>> &rows) {
    EllHost e;
    const bool arrange = !opt_on(kOpt_no_bank_arrange);
    const int R = (int)rows.size();
    std::vector<int> order(R);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return rows[a].size() > rows[b].size(); });
    const int nsl = (R + kWave - 1) / kWave;
    e.row_of.assign((size_t)nsl * kWave, -1);
    for (int i = 0; i < R; ++i) e.row_of[i] = order[i];
    e.slice_off.resize(nsl);
    e.slice_w2.resize(nsl);
    for (int j = 0; j < nsl; ++j) {
        int wmax = (int)rows[order[(size_t)j * kWave]].size();
        int w2 = (wmax + 1) / 2;
        if (w2 > 2) w2 = (w2 + 3) & ~3;  // ell_row_sum streams groups of four elements
        e.slice_off[j] = (int)e.arcs.size();
        e.slice_w2[j] = w2;
        e.arcs.resize(e.arcs.size() + (size_t)w2 * kWave, uint4{0u, 0u, 0u, 0u});
        // Column assignment.  Column c of the slice is ONE ds_read_b32 gather per wave, serviced in
        // two 32-lane halves over 32 banks (bank = idx mod 32, MI355X_MICROARCH.md LDS table).  The
        // order of a row's arcs is free, so pick it greedily per column: lanes with the fewest arcs
        // left choose first, each takes the arc whose bank is least used in its half (equal idx =
        // broadcast, free).  Summation order changes, the sum is the same up to fp32 rounding.
        std::vector<std::vector<std::pair<int, float>>> rem(kWave);
        for (int lane = 0; lane < kWave; ++lane) {
            int r = e.row_of[(size_t)j * kWave + lane];
            if (r >= 0) rem[lane] = rows[r];
        }
        for (int c = 0; c < 2 * w2; ++c) {
            for (int half = 0; half < 2; ++half) {
                int used[32];
                int occupant[32];
                for (int b = 0; b < 32; ++b) { used[b] = 0; occupant[b] = -1; }
                int lanes[32];
                for (int l = 0; l < 32; ++l) lanes[l] = half * 32 + l;
                std::stable_sort(lanes, lanes + 32, [&](int a, int b) { return rem[a].size() < rem[b].size(); });
                for (int li = 0; li < 32; ++li) {
                    const int lane = lanes[li];
                    auto &rv = rem[lane];
                    if (rv.empty()) continue;
                    size_t best = 0;
                    int best_cost = 1 << 30;
                    for (size_t k = 0; k < (arrange ? rv.size() : (size_t)1); ++k) {
                        const int bank = rv[k].first & 31;
                        const int cost = (occupant[bank] == rv[k].first) ? 0 : used[bank];
                        if (cost < best_cost) { best_cost = cost; best = k; if (cost == 0) break; }
                    }
                    const auto arc = rv[best];
                    rv.erase(rv.begin() + (long)best);
                    const int bank = arc.first & 31;
                    if (occupant[bank] != arc.first) { used[bank]++; if (occupant[bank] < 0) occupant[bank] = arc.first; }
                    e.conflict_cycles += (best_cost > 0) ? 1 : 0;
                    uint4 &a = e.arcs[(size_t)e.slice_off[j] + (size_t)(c / 2) * kWave + lane];
                    uint32_t wb;
                    memcpy(&wb, &arc.second, 4);
                    if (c & 1) { a.z = (uint32_t)arc.first; a.w = wb; }
                    else { a.x = (uint32_t)arc.first; a.y = wb; }
                }
            }
        }
        // padding arcs have weight 0; give them the index the first lane of their half reads in the same
        // column, so the padded gather is a broadcast and costs no extra bank cycle
        for (int c = 0; c < 2 * w2; ++c)
            for (int half = 0; half < 2; ++half) {
                uint32_t bidx = 0;
                bool have = false;
                for (int l = 0; l < 32 && !have; ++l) {
                    const uint4 &a = e.arcs[(size_t)e.slice_off[j] + (size_t)(c / 2) * kWave + half * 32 + l];
                    const uint32_t wbits = (c & 1) ? a.w : a.y;
                    if (wbits != 0u) { bidx = (c & 1) ? a.z : a.x; have = true; }
                }
                for (int l = 0; l < 32; ++l) {
                    uint4 &a = e.arcs[(size_t)e.slice_off[j] + (size_t)(c / 2) * kWave + half * 32 + l];
                    if (c & 1) { if (a.w == 0u) a.z = bidx; }
                    else { if (a.y == 0u) a.x = bidx; }
                }
            }
        e.padded_arcs += (int64_t)w2 * 2 * kWave;
    }
    // longest-processing-time assignment of slices (already sorted by width) to the waves
    std::vector<int64_t> load(kChainWaves, 0);
    std::vector<std::vector<int>> lists(kChainWaves);
    for (int j = 0; j < nsl; ++j) {
        int best = 0;
        for (int wv = 1; wv < kChainWaves; ++wv)
            if (load[wv] < load[best]) best = wv;
        lists[best].push_back(j);
        load[best] += e.slice_w2[j] + 4;  // +4: per-slice fixed cost (row epilogue)
    }
    e.wave_off.assign(kChainWaves + 1, 0);
    for (int wv = 0; wv < kChainWaves; ++wv) {
        e.wave_off[wv + 1] = e.wave_off[wv] + (int)lists[wv].size();
        for (int j : lists[wv]) e.wave_slices.push_back(j);
    }
    return e;
}

template <typename T>
int upload(HostGraph *h, const std::vector<T> &v, const T **out) {
    void *d = nullptr;
    size_t bytes = std::max<size_t>(v.size(), 1) * sizeof(T);
    HIP_TRY(hipMalloc(&d, bytes));
    h->allocs.push_back(d);
    if (!v.empty()) HIP_TRY(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = (const T *)d;
    return CRF_OK;
}

// One direction's arc stream (crf_internal.h: StreamDirDev).  rows / row_st / arcs: the BatchDev tables of that direction.
struct StreamHost { std::vector<int4> tasks, meta; std::vector<int2> recs; std::vector<int> rest; };
// a stream row: NM descriptor words and its records
struct SRow { int4 m[3]; const int2 *recs; int n; };
// rows (most records first) -> bundles of AL rows -> tasks of at most `maxb` bundles and about task_steps steps
static int build_stream_rows(int AL, int UL, int task_steps, int maxb, int NM, const std::vector<SRow> &srows, StreamHost *sh) {
    constexpr int kB = 4;                                 // steps per batch (crf_kernels.hip: kStreamBatch)
    std::vector<int4> tasks, meta;
    std::vector<int2> recs;
    const int nbund = ((int)srows.size() + AL - 1) / AL;
    // what a bundle costs a task besides its gather batches: its row epilogues (stores, ring refill), in batches of 4 steps -- the last tasks of a combo hold the
    // shortest rows, i.e. the most bundles per gather step, and were the last to reach the frame's end (profiles/round6_ab_persistent_batch.txt: 15 us where
    // the first tasks take 10); switch bat_epi, default kBatEpiDefault
    const int epi = std::max(0, opt(kOpt_bat_epi, kBatEpiDefault)) * kB;
    int task_b0 = 0, task_batch0 = 0, task_steps_now = 0;
    auto close_task = [&](int b1) {
        if (b1 == task_b0) return;
        const int nbat = (int)(recs.size() / ((size_t)AL * kB)) - task_batch0;
        tasks.push_back(int4{task_batch0, nbat, task_b0, b1 - task_b0});
        task_b0 = b1; task_batch0 += nbat; task_steps_now = 0;
    };
    for (int b = 0; b < nbund; ++b) {
        int len = 0;
        for (int aj = 0; aj < AL; ++aj) {
            const int i = b * AL + aj;
            if (i < (int)srows.size()) len = std::max(len, srows[(size_t)i].n);
        }
        const int nbat = (len + kB - 1) / kB;             // every row of the bundle padded to whole batches
        if (task_steps_now > 0 && (task_steps_now + nbat * kB + epi > task_steps || b - task_b0 >= maxb)) close_task(b);
        for (int aj = 0; aj < AL; ++aj) {
            const int i = b * AL + aj;
            for (int q = 0; q < NM; ++q) meta.push_back(i < (int)srows.size() ? srows[(size_t)i].m[q] : int4{-1, 0, 0, 0});   // (padding row: state -1)
        }
        for (int bt = 0; bt < nbat; ++bt)
            for (int aj = 0; aj < AL; ++aj)
                for (int k = 0; k < kB; ++k) {
                    const int st = bt * kB + k, i = b * AL + aj;
                    int2 rcd{0, 0};
                    if (i < (int)srows.size() && st < srows[(size_t)i].n) {
                        rcd = srows[(size_t)i].recs[st];
                        // the kernels want entry * UL (UL utterances per entry), below 2^29 so that the byte offset fits 31 bits
                        if (rcd.x < 0 || (int64_t)rcd.x * UL >= (1ll << 29)) { set_error("arc stream: index does not fit"); return CRF_ERR_ARG; }
                        rcd.x *= UL;
                    }
                    if (k == 0 && bt == nbat - 1) rcd.x |= (int)0x80000000u;   // the bundle ends with this batch
                    recs.push_back(rcd);
                }
        task_steps_now += nbat * kB + epi;
    }
    close_task(nbund);
    for (int k = 0; k < 1024; ++k) recs.push_back(int2{0, 0});            // the kernels stage whole 4 KB chunks, one chunk ahead
    sh->tasks = std::move(tasks); sh->meta = std::move(meta); sh->recs = std::move(recs);
    return CRF_OK;
}
static int build_stream_host(int AL, int UL, int task_steps, const std::vector<int4> &rows, const std::vector<int> &row_st,
                             const std::vector<int2> &arcs, StreamHost *sh) {
    std::vector<SRow> srows;
    std::vector<int> rest;
    for (int r = 0; r < (int)rows.size(); ++r) {
        const int4 &d = rows[(size_t)r];
        if ((d.w & 0x40000000) && d.y > d.x)               // (rows are most-arcs-first already)
            srows.push_back(SRow{{int4{row_st[(size_t)r], d.z, d.w & 0xffff, 0}, int4{-1, 0, 0, 0}, int4{0, 0, 0, 0}}, arcs.data() + d.x, d.y - d.x});
        else rest.push_back(r);
    }
    const int rc = build_stream_rows(AL, UL, task_steps, 8, 1, srows, sh);   // (8: kStreamBundles)
    sh->rest = std::move(rest);
    return rc;
}
// ... and of the factored rows (three descriptor words per row)
static int build_stream_host_fac(int AL, int UL, int task_steps, const std::vector<FacRowH> &rows, const std::vector<int> &rest, StreamHost *sh) {
    std::vector<SRow> srows;
    for (const FacRowH &r : rows)
        srows.push_back(SRow{{int4{r.st0, r.pr0, r.lab0, 0}, int4{r.st1, r.pr1, r.lab1, 0}, int4{r.x0, r.w0, r.x1, r.w1}}, r.recs.data(), (int)r.recs.size()});
    const int rc = build_stream_rows(AL, UL, task_steps, stream_max_bundles(UL, true), 3, srows, sh);
    sh->rest = rest;
    return rc;
}

static int build_stream_dir(HostGraph *h, int AL, int UL, int task_steps, const std::vector<int4> &rows, const std::vector<int> &row_st,
                            const std::vector<int2> &arcs, StreamDirDev *out) {
    StreamHost sh;
    int rc;
    if ((rc = build_stream_host(AL, UL, task_steps, rows, row_st, arcs, &sh))) return rc;
    out->ntasks = (int)sh.tasks.size(); out->nrest = (int)sh.rest.size();
    if ((rc = upload(h, sh.tasks, &out->tasks)) || (rc = upload(h, sh.recs, &out->recs)) || (rc = upload(h, sh.meta, &out->meta)) || (rc = upload(h, sh.rest, &out->rest))) return rc;
    return CRF_OK;
}

// Host-side check of the arc streams (tests; no GPU): every row with one entering pair and at least one arc is in exactly one
// bundle of exactly one task, its records are its arcs (index * UL, weight bits) in order followed by null records, the
// end-of-bundle flag sits in the first record of the bundle's last batch for every lane group and nowhere else, a task has
// at most 8 bundles, every other row is in `rest`.  out: {tasks, rest rows, steps, arc records that are not padding}.
static int check_stream_host(int AL, int UL, const StreamHost &sh, const std::vector<int4> &rows, const std::vector<int> &row_st,
                             const std::vector<int2> &arcs, int64_t *out) {
    constexpr int kB = 4;
    auto fail = [&](const char *why) { set_error(std::string("arc stream check: ") + why); return CRF_ERR_ARG; };
    std::vector<int> row_of_state_pair;                   // rows are identified by (state, pair id)
    std::vector<char> seen(rows.size(), 0);
    std::map<std::pair<int, int>, int> row_at;
    for (int r = 0; r < (int)rows.size(); ++r) if ((rows[(size_t)r].w & 0x40000000) && rows[(size_t)r].y > rows[(size_t)r].x) row_at[{row_st[(size_t)r], rows[(size_t)r].z}] = r;
    int64_t steps = 0, real = 0;
    int next_bundle = 0;
    for (const int4 &t : sh.tasks) {
        if (t.w < 1 || t.w > 8) return fail("a task has no or more than 8 bundles");
        if (t.z != next_bundle) return fail("the tasks do not cover the bundles in order");
        next_bundle += t.w;
        int batch = t.x;
        for (int b = 0; b < t.w; ++b) {
            // length of this bundle: up to the batch that carries the flag
            int nbat = 0;
            for (;;) {
                if (batch + nbat >= t.x + t.y) return fail("a bundle runs past its task");
                const int2 first = sh.recs[((size_t)(batch + nbat) * AL + 0) * kB];
                ++nbat;
                if (first.x < 0) break;
            }
            for (int aj = 0; aj < AL; ++aj) {
                const int4 m = sh.meta[(size_t)(t.z + b) * AL + aj];
                int n = 0, a0 = 0;
                if (m.x >= 0) {
                    auto it = row_at.find({m.x, m.y});
                    if (it == row_at.end()) return fail("a descriptor names a row that is not a stream row");
                    const int r = it->second;
                    if (seen[(size_t)r]) return fail("a row appears twice");
                    seen[(size_t)r] = 1;
                    if ((rows[(size_t)r].w & 0xffff) != m.z) return fail("label of a descriptor");
                    n = rows[(size_t)r].y - rows[(size_t)r].x; a0 = rows[(size_t)r].x;
                    if (n > nbat * kB) return fail("a row is longer than its bundle");
                }
                for (int st = 0; st < nbat * kB; ++st) {
                    const int2 rc = sh.recs[((size_t)(batch + st / kB) * AL + aj) * kB + st % kB];
                    const bool flag = rc.x < 0, want_flag = st == (nbat - 1) * kB;
                    if (flag != want_flag) return fail("end-of-bundle flag");
                    const int idx = rc.x & 0x7fffffff;
                    if (st < n) {
                        if (idx != arcs[(size_t)a0 + st].x * UL || rc.y != arcs[(size_t)a0 + st].y) return fail("a record is not its arc");
                        ++real;
                    } else if (idx != 0 || rc.y != 0) return fail("padding record is not null");
                }
            }
            batch += nbat; steps += (int64_t)nbat * kB;
        }
        if (batch != t.x + t.y) return fail("batches of a task");
    }
    int64_t nrest = 0;
    std::vector<char> in_rest(rows.size(), 0);
    for (int r : sh.rest) { if (r < 0 || r >= (int)rows.size() || in_rest[(size_t)r]) return fail("rest list"); in_rest[(size_t)r] = 1; ++nrest; }
    for (int r = 0; r < (int)rows.size(); ++r) {
        const bool simple = (rows[(size_t)r].w & 0x40000000) && rows[(size_t)r].y > rows[(size_t)r].x;
        if (simple != (seen[(size_t)r] != 0) || simple == (in_rest[(size_t)r] != 0)) return fail("a row is in neither or both of stream and rest");
    }
    out[0] += (int64_t)sh.tasks.size(); out[1] += nrest; out[2] += steps; out[3] += real;
    return CRF_OK;
}

int upload_ell(HostGraph *h, const EllHost &e, EllDev *d) {
    int rc;
    if ((rc = upload(h, e.arcs, &d->arcs))) return rc;
    if ((rc = upload(h, e.slice_off, &d->slice_off))) return rc;
    if ((rc = upload(h, e.slice_w2, &d->slice_w2))) return rc;
    if ((rc = upload(h, e.wave_off, &d->wave_off))) return rc;
    if ((rc = upload(h, e.wave_slices, &d->wave_slices))) return rc;
    d->nslices = (int)e.slice_off.size();
    return CRF_OK;
}

}  // namespace

// Factored rows of the utterance-minor kernels (crf_internal.h: StreamDev) from the BatchDev tables.  Structure detection as
// in the register-resident factored layout (res_layout.cpp build_factored), restated on these tables:
//   tail t, main o:  t has one entering pair whose two arcs come from t itself and from o with the same weight bits; both
//                    states have exactly one entering pair; a state is in at most one such couple;
//   forward rows:    every other one-pair row, the arcs from a couple's two states with equal weights merged into one record
//                    reading U = S + k; the main row carries its tail as second output;
//   backward rows:   a couple whose out-arcs are common except at most one each shares a row.
// Anything that does not fit stays what it was (one-pair rows: plain stream rows; the others: the row-at-a-time path).
// Used when at least half of the states are in couples.
static void build_batch_factored(HostGraph *h, int S, const std::vector<int4> &frow, const std::vector<int> &frow_d, const std::vector<int2> &farcs,
                                 const std::vector<int4> &brow, const std::vector<int> &brow_s, const std::vector<int2> &barcs,
                                 const std::vector<float> &start_lin) {
    FacBatchH &F = h->fb;
    F = FacBatchH();
    if (opt_on(kOpt_bat_no_fac)) return;
    std::vector<int> fr_of(S, -1), br_of(S, -1);
    for (int r = 0; r < S; ++r) { fr_of[(size_t)frow_d[(size_t)r]] = r; br_of[(size_t)brow_s[(size_t)r]] = r; }
    auto fsimple = [&](int s) { const int4 &d = frow[(size_t)fr_of[(size_t)s]]; return (d.w & 0x40000000) && d.y > d.x; };
    auto bsimple = [&](int s) { const int4 &d = brow[(size_t)br_of[(size_t)s]]; return (d.w & 0x40000000) && d.y > d.x; };
    std::vector<int> main_of(S, -1), tail_of(S, -1), uidx(S, -1), tailw(S, 0);
    int NU = 0;
    for (int t = 0; t < S; ++t) {
        if (!fsimple(t)) continue;
        const int4 &d = frow[(size_t)fr_of[(size_t)t]];
        if (d.y - d.x != 2) continue;
        const int2 a = farcs[(size_t)d.x], b = farcs[(size_t)d.x + 1];
        if (a.y != b.y) continue;
        const int o = a.x == t ? b.x : b.x == t ? a.x : -1;
        if (o < 0 || o == t || !fsimple(o)) continue;
        if (main_of[(size_t)t] >= 0 || tail_of[(size_t)t] >= 0 || main_of[(size_t)o] >= 0 || tail_of[(size_t)o] >= 0) continue;
        main_of[(size_t)t] = o; tail_of[(size_t)o] = t; tailw[(size_t)o] = a.y;
    }
    // (a main state must not itself look like a tail that was skipped: nothing to check -- couples are disjoint by construction)
    for (int o = 0; o < S; ++o) if (tail_of[(size_t)o] >= 0) uidx[(size_t)o] = NU++;
    if ((int64_t)NU * 4 < S) return;
    auto couple_main = [&](int s) { return tail_of[(size_t)s] >= 0 ? s : main_of[(size_t)s]; };   // main state of s's couple, -1: none
    // forward
    for (int r = 0; r < S; ++r) {
        const int s = frow_d[(size_t)r];
        const int4 &d = frow[(size_t)r];
        if (!((d.w & 0x40000000) && d.y > d.x)) { F.frest.push_back(r); continue; }
        if (main_of[(size_t)s] >= 0) continue;                        // a tail: computed by its main row
        FacRowH row;
        row.st0 = s; row.pr0 = d.z; row.lab0 = d.w & 0xffff;
        if (tail_of[(size_t)s] >= 0) {
            const int t = tail_of[(size_t)s];
            const int4 &dt = frow[(size_t)fr_of[(size_t)t]];
            row.st1 = t; row.pr1 = dt.z; row.lab1 = dt.w & 0xffff; row.x0 = S + uidx[(size_t)s]; row.w0 = tailw[(size_t)s];
        }
        std::map<std::pair<int, int>, std::vector<int>> by;          // (couple's main, weight bits) -> arcs
        for (int k = d.x; k < d.y; ++k) by[{couple_main(farcs[(size_t)k].x), farcs[(size_t)k].y}].push_back(k);
        std::vector<char> used((size_t)(d.y - d.x), 0);
        for (int k = d.x; k < d.y; ++k) {
            if (used[(size_t)(k - d.x)]) continue;
            used[(size_t)(k - d.x)] = 1;
            const int src = farcs[(size_t)k].x, mn = couple_main(src);
            int partner = -1;
            if (mn >= 0) {
                const int other = src == mn ? tail_of[(size_t)mn] : mn;
                for (int q : by[{mn, farcs[(size_t)k].y}]) if (!used[(size_t)(q - d.x)] && farcs[(size_t)q].x == other) { partner = q; break; }
            }
            if (partner >= 0) { used[(size_t)(partner - d.x)] = 1; row.recs.push_back(int2{S + uidx[(size_t)mn], farcs[(size_t)k].y}); }
            else row.recs.push_back(farcs[(size_t)k]);
        }
        F.recs_f += (int64_t)row.recs.size();
        F.frows.push_back(std::move(row));
    }
    // backward
    std::vector<char> done(S, 0);
    for (int r = 0; r < S; ++r) {
        const int s = brow_s[(size_t)r];
        const int4 &d = brow[(size_t)r];
        if (!((d.w & 0x40000000) && d.y > d.x)) { F.brest.push_back(r); continue; }
        if (done[(size_t)s]) continue;
        done[(size_t)s] = 1;
        FacRowH row;
        row.st0 = s; row.pr0 = d.z; row.lab0 = d.w & 0xffff;
        const int mn = couple_main(s), m = mn < 0 ? -1 : (s == mn ? tail_of[(size_t)mn] : mn);
        bool fused = false;
        if (m >= 0 && !done[(size_t)m] && bsimple(m)) {
            const int4 &dm = brow[(size_t)br_of[(size_t)m]];
            std::vector<std::pair<int, int>> a, b, common, ea, eb;
            for (int k = d.x; k < d.y; ++k) a.push_back({barcs[(size_t)k].x, barcs[(size_t)k].y});
            for (int k = dm.x; k < dm.y; ++k) b.push_back({barcs[(size_t)k].x, barcs[(size_t)k].y});
            std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
            std::set_intersection(a.begin(), a.end(), b.begin(), b.end(), std::back_inserter(common));
            std::set_difference(a.begin(), a.end(), common.begin(), common.end(), std::back_inserter(ea));
            std::set_difference(b.begin(), b.end(), common.begin(), common.end(), std::back_inserter(eb));
            if (!common.empty() && ea.size() <= 1 && eb.size() <= 1) {
                fused = true;
                done[(size_t)m] = 1;
                row.st1 = m; row.pr1 = dm.z; row.lab1 = dm.w & 0xffff;
                if (!ea.empty()) { row.x0 = ea[0].first; row.w0 = ea[0].second; }
                if (!eb.empty()) { row.x1 = eb[0].first; row.w1 = eb[0].second; }
                for (auto &c : common) row.recs.push_back(int2{c.first, c.second});
            }
        }
        if (!fused) for (int k = d.x; k < d.y; ++k) row.recs.push_back(barcs[(size_t)k]);
        F.recs_b += (int64_t)row.recs.size();
        F.brows.push_back(std::move(row));
    }
    auto by_len = [](const FacRowH &x, const FacRowH &y) { return x.recs.size() > y.recs.size(); };
    std::stable_sort(F.frows.begin(), F.frows.end(), by_len);
    std::stable_sort(F.brows.begin(), F.brows.end(), by_len);
    F.x_start.assign((size_t)S + NU, 0.f);
    for (int s2 = 0; s2 < S; ++s2) {
        F.x_start[(size_t)s2] = start_lin[(size_t)s2];
        if (tail_of[(size_t)s2] >= 0) F.x_start[(size_t)S + uidx[(size_t)s2]] = start_lin[(size_t)s2] + start_lin[(size_t)tail_of[(size_t)s2]];
    }
    F.NU = NU;
    F.ok = 1;
}

// Host check of the factored rows (tests; no GPU): on random vectors, one step of both recursions through the factored rows
// (U entries, folded tail rows, fused backward rows) equals the step through the plain tables, pair by pair and state by
// state (fp64, 1e-12).  out: {NU, forward records, backward records, plain arcs}.
int debug_check_facbatch(const HostGraph *h, int64_t *out4) {
    const FacBatchH &F = h->fb;
    for (int i = 0; i < 4; ++i) out4[i] = 0;
    if (!F.ok) return CRF_OK;
    auto fl = [](int b) { float f; memcpy(&f, &b, 4); return (double)f; };
    const int S = (int)h->S, P = (int)h->P;
    uint64_t rng = 88172645463325252ull;
    auto rnd = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (double)(rng % 1000003) / 1000003.0 + 0.01; };
    std::vector<double> a((size_t)S), z((size_t)P), x((size_t)S + F.NU, 0.0);
    for (auto &v : a) v = rnd();
    for (auto &v : z) v = rnd();
    // U entries from the couples: found through the forward rows' second outputs
    for (int s = 0; s < S; ++s) x[(size_t)s] = a[(size_t)s];
    for (const FacRowH &r : F.frows) if (r.st1 >= 0) x[(size_t)r.x0] = a[(size_t)r.st0] + a[(size_t)r.st1];
    auto fail = [&](const char *why) { set_error(std::string("factored batch rows: ") + why); return CRF_ERR_ARG; };
    std::vector<double> q_plain((size_t)P, -1.0), q_fac((size_t)P, -1.0), b_plain((size_t)S, -1.0), b_fac((size_t)S, -1.0);
    std::vector<char> in_rest_f((size_t)S, 0), in_rest_b((size_t)S, 0);
    for (int r : F.frest) in_rest_f[(size_t)r] = 1;
    for (int r : F.brest) in_rest_b[(size_t)r] = 1;
    for (int r = 0; r < S; ++r) {
        const int4 &d = h->hb_frow[(size_t)r];
        if (in_rest_f[(size_t)r]) continue;
        double acc = 0.0;
        for (int k = d.x; k < d.y; ++k) acc += fl(h->hb_farcs[(size_t)k].y) * a[(size_t)h->hb_farcs[(size_t)k].x];
        q_plain[(size_t)d.z] = acc;
    }
    for (const FacRowH &r : F.frows) {
        double acc = 0.0;
        for (const int2 &c : r.recs) acc += fl(c.y) * x[(size_t)c.x];
        if (q_fac[(size_t)r.pr0] >= 0.0) return fail("a pair is produced twice (forward)");
        q_fac[(size_t)r.pr0] = acc;
        if (r.st1 >= 0) { if (q_fac[(size_t)r.pr1] >= 0.0) return fail("a pair is produced twice (forward, tail)"); q_fac[(size_t)r.pr1] = fl(r.w0) * x[(size_t)r.x0]; }
    }
    for (int p2 = 0; p2 < P; ++p2) {
        if ((q_plain[(size_t)p2] < 0.0) != (q_fac[(size_t)p2] < 0.0)) return fail("forward rows do not cover the one-pair rows");
        if (q_plain[(size_t)p2] >= 0.0 && std::fabs(q_plain[(size_t)p2] - q_fac[(size_t)p2]) > 1e-9 * std::fabs(q_plain[(size_t)p2])) return fail("a forward row's sum differs");
    }
    for (int r = 0; r < S; ++r) {
        const int4 &d = h->hb_brow[(size_t)r];
        if (in_rest_b[(size_t)r]) continue;
        double acc = 0.0;
        for (int k = d.x; k < d.y; ++k) acc += fl(h->hb_barcs[(size_t)k].y) * z[(size_t)h->hb_barcs[(size_t)k].x];
        b_plain[(size_t)h->hb_brow_s[(size_t)r]] = acc;
    }
    for (const FacRowH &r : F.brows) {
        double acc = 0.0;
        for (const int2 &c : r.recs) acc += fl(c.y) * z[(size_t)c.x];
        if (b_fac[(size_t)r.st0] >= 0.0) return fail("a state is produced twice (backward)");
        b_fac[(size_t)r.st0] = acc + fl(r.w0) * z[(size_t)r.x0];
        if (r.st1 >= 0) { if (b_fac[(size_t)r.st1] >= 0.0) return fail("a state is produced twice (backward, mate)"); b_fac[(size_t)r.st1] = acc + fl(r.w1) * z[(size_t)r.x1]; }
    }
    for (int s = 0; s < S; ++s) {
        if ((b_plain[(size_t)s] < 0.0) != (b_fac[(size_t)s] < 0.0)) return fail("backward rows do not cover the one-pair rows");
        if (b_plain[(size_t)s] >= 0.0 && std::fabs(b_plain[(size_t)s] - b_fac[(size_t)s]) > 1e-9 * std::fabs(b_plain[(size_t)s])) return fail("a backward row's sum differs");
    }
    // descriptors: pairs and labels as in the plain tables
    std::vector<int> fr_of((size_t)S, -1), br_of((size_t)S, -1);
    for (int r = 0; r < S; ++r) { fr_of[(size_t)h->hb_frow_d[(size_t)r]] = r; br_of[(size_t)h->hb_brow_s[(size_t)r]] = r; }
    auto desc_ok = [&](const std::vector<int4> &rows, const std::vector<int> &of, int st, int pr, int lab) {
        const int4 &d = rows[(size_t)of[(size_t)st]];
        return (d.w & 0x40000000) && d.z == pr && (d.w & 0xffff) == lab;
    };
    for (const FacRowH &r : F.frows) if (!desc_ok(h->hb_frow, fr_of, r.st0, r.pr0, r.lab0) || (r.st1 >= 0 && !desc_ok(h->hb_frow, fr_of, r.st1, r.pr1, r.lab1))) return fail("forward descriptor");
    for (const FacRowH &r : F.brows) if (!desc_ok(h->hb_brow, br_of, r.st0, r.pr0, r.lab0) || (r.st1 >= 0 && !desc_ok(h->hb_brow, br_of, r.st1, r.pr1, r.lab1))) return fail("backward descriptor");
    out4[0] = F.NU; out4[1] = F.recs_f; out4[2] = F.recs_b; out4[3] = h->A;
    return CRF_OK;
}

// steps per task: the longer direction's steps (rows padded to whole batches of 4) over the tasks wanted; CRF_BAT_TASK overrides
static int stream_task_steps(const HostGraph *h, int AL, int want, bool fac = false) {
    const int epi = std::max(0, opt(kOpt_bat_epi, kBatEpiDefault)) * 4;   // (build_stream_rows: what a bundle's row epilogues cost, in steps)
    auto steps_of = [&](const std::vector<int4> &rows) {
        int64_t n = 0;
        for (const int4 &d : rows) if ((d.w & 0x40000000) && d.y > d.x) n += (d.y - d.x + 3) / 4 * 4 + epi;
        return (n + AL - 1) / AL;                         // (rows of a bundle have about the same length)
    };
    auto steps_fac = [&](const std::vector<FacRowH> &rows) {
        int64_t n = 0;
        for (const FacRowH &r : rows) n += ((int64_t)r.recs.size() + 3) / 4 * 4 + epi;
        return (n + AL - 1) / AL;
    };
    const int64_t steps = fac ? std::max(steps_fac(h->fb.frows), steps_fac(h->fb.brows)) : std::max(steps_of(h->hb_frow), steps_of(h->hb_brow));
    int task_steps = (int)std::min<int64_t>(4096, std::max<int64_t>(64, ((steps + want - 1) / want + 3) / 4 * 4));
    if (opt(kOpt_bat_task, 0) > 0) task_steps = std::max(8, opt(kOpt_bat_task, 0));
    return task_steps;
}

int debug_check_decode(int nslot, int ncombo) {
    if (nslot < 1 || ncombo < 1) { set_error("debug_check_decode: bad arguments"); return CRF_ERR_ARG; }
    std::map<std::pair<int, int>, int> seen;
    std::vector<int> nch((size_t)ncombo, -1);
    for (int b = 0; b < 8 * nslot; ++b) {
        int combo, chunk, nchunk;
        bat_decode(b, 8 * nslot, ncombo, &combo, &chunk, &nchunk);
        if (combo < 0 || combo >= ncombo) continue;      // (more XCDs than combos' residues can fill: a block without work)
        if (chunk < 0 || chunk >= nchunk) { set_error("bat_decode: chunk out of range"); return CRF_ERR_ARG; }
        if (nch[(size_t)combo] >= 0 && nch[(size_t)combo] != nchunk) { set_error("bat_decode: blocks of a combo disagree on its chunk count"); return CRF_ERR_ARG; }
        nch[(size_t)combo] = nchunk;
        if (seen[{combo, chunk}]++) { set_error("bat_decode: a (combo, chunk) is taken twice"); return CRF_ERR_ARG; }
    }
    for (int c = 0; c < ncombo; ++c) {
        if (nch[(size_t)c] < 1) { set_error("bat_decode: a combo has no workgroup"); return CRF_ERR_ARG; }
        for (int k = 0; k < nch[(size_t)c]; ++k) if (!seen.count({c, k})) { set_error("bat_decode: a chunk of a combo is missing"); return CRF_ERR_ARG; }
    }
    return CRF_OK;
}

// Structure check of a stream against the rows it was cut from (any descriptor width): bundle i holds rows [i * AL, (i + 1) * AL)
// in order, records = the row's records (entry * UL) followed by null records up to the bundle's whole batches, the end flag in
// the first record of the bundle's last batch for every lane group and nowhere else, descriptors word for word, a task has
// at most maxb bundles and the tasks cover the bundles in order.  out: {tasks, 0, steps, records that are not padding}.
static int check_stream_rows(int AL, int UL, int NM, int maxb, const StreamHost &sh, const std::vector<SRow> &srows, int64_t *out) {
    constexpr int kB = 4;
    auto fail = [&](const char *why) { set_error(std::string("arc stream check (rows): ") + why); return CRF_ERR_ARG; };
    int next_bundle = 0;
    int64_t steps = 0, real = 0;
    for (const int4 &t : sh.tasks) {
        if (t.w < 1 || t.w > maxb) return fail("a task has no or too many bundles");
        if (t.z != next_bundle) return fail("the tasks do not cover the bundles in order");
        next_bundle += t.w;
        int batch = t.x;
        for (int b = 0; b < t.w; ++b) {
            int nbat = 0;
            for (;;) {
                if (batch + nbat >= t.x + t.y) return fail("a bundle runs past its task");
                const int2 first = sh.recs[((size_t)(batch + nbat) * AL + 0) * kB];
                ++nbat;
                if (first.x < 0) break;
            }
            for (int aj = 0; aj < AL; ++aj) {
                const int i = (t.z + b) * AL + aj;
                const bool have = i < (int)srows.size();
                for (int q = 0; q < NM; ++q) {
                    const int4 m = sh.meta[((size_t)(t.z + b) * AL + aj) * NM + q], w = have ? srows[(size_t)i].m[q] : int4{-1, 0, 0, 0};
                    if (m.x != w.x || m.y != w.y || m.z != w.z || m.w != w.w) return fail("a descriptor word");
                }
                const int n = have ? srows[(size_t)i].n : 0;
                if (n > nbat * kB) return fail("a row is longer than its bundle");
                for (int st = 0; st < nbat * kB; ++st) {
                    const int2 rc = sh.recs[((size_t)(batch + st / kB) * AL + aj) * kB + st % kB];
                    if ((rc.x < 0) != (st == (nbat - 1) * kB)) return fail("end-of-bundle flag");
                    const int idx = rc.x & 0x7fffffff;
                    if (st < n) {
                        if (idx != srows[(size_t)i].recs[st].x * UL || rc.y != srows[(size_t)i].recs[st].y) return fail("a record is not its row's");
                        ++real;
                    } else if (idx != 0 || rc.y != 0) return fail("padding record is not null");
                }
            }
            batch += nbat; steps += (int64_t)nbat * kB;
        }
        if (batch != t.x + t.y) return fail("batches of a task");
    }
    if (next_bundle != ((int)srows.size() + AL - 1) / AL) return fail("bundles missing");
    out[0] += (int64_t)sh.tasks.size(); out[2] += steps; out[3] += real;
    return CRF_OK;
}

// Builds the arc streams of both directions on the host and checks them (check_stream_host); no device needed.
int debug_check_streams(const HostGraph *h, int UL, int want, int64_t *out4) {
    const bool fac = UL < 0;                             // -UL: the factored streams (graphs with factored rows)
    if (fac) UL = -UL;
    if (!h || !(UL == 8 || UL == 16 || UL == 32 || UL == 64) || want < 1 || !out4) { set_error("debug_check_streams: bad arguments"); return CRF_ERR_ARG; }
    const int AL = 256 / UL, task_steps = stream_task_steps(h, AL, want, fac);
    for (int i = 0; i < 4; ++i) out4[i] = 0;
    if (fac) {
        if (!h->fb.ok) return CRF_OK;
        for (int dir = 0; dir < 2; ++dir) {
            const std::vector<FacRowH> &rows = dir == 0 ? h->fb.frows : h->fb.brows;
            std::vector<SRow> srows;
            for (const FacRowH &r : rows)
                srows.push_back(SRow{{int4{r.st0, r.pr0, r.lab0, 0}, int4{r.st1, r.pr1, r.lab1, 0}, int4{r.x0, r.w0, r.x1, r.w1}}, r.recs.data(), (int)r.recs.size()});
            StreamHost sh;
            int rc = build_stream_host_fac(AL, UL, task_steps, rows, dir == 0 ? h->fb.frest : h->fb.brest, &sh);
            if (!rc) rc = check_stream_rows(AL, UL, 3, stream_max_bundles(UL, true), sh, srows, out4);
            if (rc) return rc;
            out4[1] += (int64_t)sh.rest.size();
        }
        return CRF_OK;
    }
    for (int dir = 0; dir < 2; ++dir) {
        const std::vector<int4> &rows = dir == 0 ? h->hb_frow : h->hb_brow;
        const std::vector<int> &st = dir == 0 ? h->hb_frow_d : h->hb_brow_s;
        const std::vector<int2> &arcs = dir == 0 ? h->hb_farcs : h->hb_barcs;
        StreamHost sh;
        int rc = build_stream_host(AL, UL, task_steps, rows, st, arcs, &sh);
        if (!rc) rc = check_stream_host(AL, UL, sh, rows, st, arcs, out4);
        if (rc) return rc;
    }
    return CRF_OK;
}

bool stream_fac(const HostGraph *h, int UL) { (void)UL; return h && h->fb.ok && !opt_on(kOpt_bat_no_fac); }

int ensure_stream_tables(HostGraph *h, int UL, int want, const StreamDev **out) {
    const int AL = 256 / std::max(UL, 1);                 // lane groups: a lane takes 4 utterances, UL / 4 lanes a row
    static std::mutex mu;
    if (!h || !(UL == 8 || UL == 16 || UL == 32 || UL == 64) || want < 1 || !h->dev.bat.ok) { set_error("ensure_stream_tables: bad arguments"); return CRF_ERR_ARG; }
    std::lock_guard<std::mutex> lock(mu);
    const bool fac = stream_fac(h, UL);
    for (StreamDev *sd : h->streams)
        if (sd->AL == AL && sd->want == want && (sd->fac != 0) == fac) { *out = sd; return CRF_OK; }
    // factored streams for T o LM graphs
    const int task_steps = stream_task_steps(h, AL, want, fac);
    int prev = 0;
    if (hipGetDevice(&prev) != hipSuccess || hipSetDevice(h->device) != hipSuccess) { set_error("ensure_stream_tables: cannot select the graph's device"); return CRF_ERR_HIP; }
    auto *sd = new StreamDev();
    int rc = CRF_OK;
    if (fac) {
        for (int dir = 0; dir < 2 && !rc; ++dir) {
            StreamHost sh;
            StreamDirDev *o = dir == 0 ? &sd->f : &sd->b;
            rc = build_stream_host_fac(AL, UL, task_steps, dir == 0 ? h->fb.frows : h->fb.brows, dir == 0 ? h->fb.frest : h->fb.brest, &sh);
            if (rc) break;
            o->ntasks = (int)sh.tasks.size(); o->nrest = (int)sh.rest.size();
            if ((rc = upload(h, sh.tasks, &o->tasks)) || (rc = upload(h, sh.recs, &o->recs)) || (rc = upload(h, sh.meta, &o->meta)) || (rc = upload(h, sh.rest, &o->rest))) break;
        }
        if (!rc) rc = upload(h, h->fb.x_start, &sd->x_start);
        sd->fac = 1; sd->NU = h->fb.NU;
    } else {
        rc = build_stream_dir(h, AL, UL, task_steps, h->hb_frow, h->hb_frow_d, h->hb_farcs, &sd->f);
        if (!rc) rc = build_stream_dir(h, AL, UL, task_steps, h->hb_brow, h->hb_brow_s, h->hb_barcs, &sd->b);
        sd->fac = 0; sd->NU = 0; sd->x_start = h->dev.start_lin;
    }
    (void)hipSetDevice(prev);
    if (rc) { delete sd; return rc; }
    sd->AL = AL; sd->want = want; sd->ok = 1;
    h->streams.push_back(sd);
    *out = sd;
    return CRF_OK;
}


// RE-GAUGING of weight-pushed graphs.  A den_lm is CTC topology o LM: the two states (g, blank) and (g, token) of an LM
// history feed the same rows with the same weights, which is what the factored layout lives on (res_layout.cpp:
// build_factored).  A tool that PUSHES weights (w' = w + V(dst) - V(src), final' = final - V, V(start) = 0: every path
// weight unchanged) destroys the equality -- the two weights into a row then differ by the constant V(token state) -
// V(blank state).  Path weights are invariant under ANY such potential, so the compiler may choose its own: states that
// share >= 2 rows with a CONSTANT log-weight difference d get potentials that make the weights equal again (g[s2] = -d;
// w~ = w + g[dst] - g[src], start~ = start + g, end~ = end - g), and the now-equal weights are snapped to the same bits.
// Nothing happens to graphs that need nothing (all differences zero).  Returns true when the weights were changed.
static bool regauge_pushed(int64_t S, int64_t A, const int32_t *src, const int32_t *dst, const int32_t *lab,
                           std::vector<float> &w, std::vector<float> &start_w, std::vector<float> &end_w) {
    if (opt_on(kOpt_no_regauge)) return false;
    std::map<std::pair<int, int>, std::vector<int>> rows;           // (label, dst) -> arcs
    for (int64_t k = 0; k < A; ++k) rows[{(int)lab[k], (int)dst[k]}].push_back((int)k);
    struct Obs { uint64_t key; double d; };
    std::vector<Obs> obs;
    int64_t nobs = 0;
    for (auto &kv : rows) { const int64_t n = (int64_t)kv.second.size(); if (n >= 2 && n <= 512) nobs += n * (n - 1) / 2; }
    if (nobs <= 30000000 && !opt_on(kOpt_regauge_minhash)) {
        // every pair of states that meets in a row (quadratic in the row length: graphs up to a few hundred thousand arcs)
        for (auto &kv : rows) {
            const std::vector<int> &a = kv.second;
            if (a.size() < 2 || a.size() > 512) continue;
            for (size_t u = 0; u < a.size(); ++u)
                for (size_t v = u + 1; v < a.size(); ++v) {
                    int s1 = src[a[u]], s2 = src[a[v]];
                    double d = (double)w[(size_t)a[u]] - (double)w[(size_t)a[v]];
                    if (s1 == s2 || !std::isfinite(d)) continue;
                    if (s1 > s2) { std::swap(s1, s2); d = -d; }
                    obs.push_back({(uint64_t)s1 << 32 | (unsigned)s2, d});
                }
        }
    } else {
        // Large graphs (config #5: 4.3 M arcs in rows of 64 -- 134 M pairs, minutes of sorting): the two states of a history
        // feed the SAME rows but one each, so they are found as near-duplicates -- twelve min-hashes over the set of rows a state
        // feeds (two sets that differ in two of d + 2 elements share a min-hash with probability d / (d + 2)); only states that
        // collide in one of them are compared, arc list against arc list.  O(A log A).
        std::vector<int> rid((size_t)A);
        { int r = 0; for (auto &kv : rows) { for (int k : kv.second) rid[(size_t)k] = r; ++r; } }
        std::vector<int> order((size_t)A);
        std::iota(order.begin(), order.end(), 0);
        std::sort(order.begin(), order.end(), [&](int x, int y) { return src[x] != src[y] ? src[x] < src[y] : rid[(size_t)x] < rid[(size_t)y]; });
        std::vector<int64_t> off((size_t)S + 1, 0);
        for (int64_t k = 0; k < A; ++k) ++off[(size_t)src[k] + 1];
        for (int64_t q = 0; q < S; ++q) off[(size_t)q + 1] += off[(size_t)q];
        constexpr int NH = 12;
        auto mix = [](uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; };
        std::vector<uint64_t> pairs;
        std::vector<std::pair<uint64_t, int>> mh((size_t)S);
        for (int j = 0; j < NH; ++j) {
            for (int64_t q = 0; q < S; ++q) {
                uint64_t m = ~0ull;
                for (int64_t i = off[(size_t)q]; i < off[(size_t)q + 1]; ++i) m = std::min(m, mix((uint64_t)rid[(size_t)order[(size_t)i]] * 0x9e3779b97f4a7c15ull + (uint64_t)j * 0x632be59bd9b4e019ull));
                mh[(size_t)q] = {m, (int)q};
            }
            std::sort(mh.begin(), mh.end());
            for (size_t i = 0; i < mh.size();) {
                size_t e = i;
                while (e < mh.size() && mh[e].first == mh[i].first) ++e;
                if (mh[i].first != ~0ull && e - i >= 2 && e - i <= 16)
                    for (size_t u = i; u < e; ++u)
                        for (size_t v = u + 1; v < e; ++v) pairs.push_back((uint64_t)std::min(mh[u].second, mh[v].second) << 32 | (unsigned)std::max(mh[u].second, mh[v].second));
                i = e;
            }
        }
        std::sort(pairs.begin(), pairs.end());
        pairs.erase(std::unique(pairs.begin(), pairs.end()), pairs.end());
        for (uint64_t key : pairs) {
            const int s1 = (int)(key >> 32), s2 = (int)(key & 0xffffffffu);
            int64_t i = off[(size_t)s1], j = off[(size_t)s2];
            const int64_t ie = off[(size_t)s1 + 1], je = off[(size_t)s2 + 1];
            while (i < ie && j < je) {
                const int a1 = order[(size_t)i], a2 = order[(size_t)j];
                if (rid[(size_t)a1] < rid[(size_t)a2]) ++i;
                else if (rid[(size_t)a1] > rid[(size_t)a2]) ++j;
                else {
                    const double d = (double)w[(size_t)a1] - (double)w[(size_t)a2];
                    if (std::isfinite(d)) obs.push_back({key, d});
                    ++i; ++j;
                }
            }
        }
    }
    std::sort(obs.begin(), obs.end(), [](const Obs &x, const Obs &y) { return x.key < y.key || (x.key == y.key && x.d < y.d); });
    struct Cand { uint64_t key; int n; double d; };
    std::vector<Cand> cand;
    for (size_t i = 0; i < obs.size();) {
        size_t j = i;
        while (j < obs.size() && obs[j].key == obs[i].key) ++j;
        // the difference MOST of the common rows agree on (a pair may also meet in a row where its weights have nothing to
        // do with each other: the self-loop of (g, token) beside the arc from (g, blank) when the history maps to itself)
        size_t best_a = i, best_n = 0;
        for (size_t a = i, b = i; a < j; ++a) {
            while (b < j && obs[b].d - obs[a].d <= 1e-4) ++b;
            if (b - a > best_n) { best_n = b - a; best_a = a; }
        }
        if (best_n >= 2) {
            double sum = 0.0;
            for (size_t a = best_a; a < best_a + best_n; ++a) sum += obs[a].d;
            cand.push_back({obs[i].key, (int)best_n, sum / (double)best_n});
        }
        i = j;
    }
    std::stable_sort(cand.begin(), cand.end(), [](const Cand &x, const Cand &y) { return x.n > y.n; });
    std::vector<int> mate((size_t)S, -1);
    std::vector<double> g((size_t)S, 0.0);
    int64_t moved = 0, matched = 0;
    for (const Cand &c : cand) {
        const int s1 = (int)(c.key >> 32), s2 = (int)(c.key & 0xffffffffu);
        if (mate[s1] >= 0 || mate[s2] >= 0) continue;
        mate[s1] = s2; mate[s2] = s1;
        g[s2] = -c.d;                                     // w(s1 -> d) - g[s1] == w(s2 -> d) - g[s2], g[s1] = 0
        ++matched;
        if (std::fabs(c.d) > 1e-6) ++moved;
    }
    if (opt_on(kOpt_verbose))
        fprintf(stderr, "[regauge] %lld candidate pairs, %lld matched, %lld with a non-zero difference (S = %lld)\n", (long long)cand.size(), (long long)matched, (long long)moved, (long long)S);
    if (moved * 8 < S) return false;                       // not a pushed T o LM graph (or nothing to undo)
    for (int64_t k = 0; k < A; ++k) w[(size_t)k] = (float)((double)w[(size_t)k] + g[dst[k]] - g[src[k]]);
    for (int64_t s = 0; s < S; ++s) {
        if (std::isfinite(start_w[(size_t)s])) start_w[(size_t)s] = (float)((double)start_w[(size_t)s] + g[s]);
        if (std::isfinite(end_w[(size_t)s])) end_w[(size_t)s] = (float)((double)end_w[(size_t)s] - g[s]);
    }
    // Snap the weights of mates in a common row to the same bits -- within the ROUNDING of the pushed weights only (the file
    // holds fp32 sums w + V(dst) - V(src): a few ulp of the weight's magnitude); a pair that differs by more is left alone
    // (it then simply does not factor), so no path weight moves by more than float rounding.
    double max_snap = 0.0;
    for (auto &kv : rows) {
        const std::vector<int> &a = kv.second;
        if (a.size() < 2 || a.size() > 512) continue;   // (quadratic in the row length)
        for (size_t u = 0; u < a.size(); ++u)
            for (size_t v = u + 1; v < a.size(); ++v) {
                if (mate[src[a[u]]] != src[a[v]]) continue;
                const double wu = w[(size_t)a[u]], wv = w[(size_t)a[v]], df = std::fabs(wu - wv);
                if (df <= 4e-6 * std::max(1.0, std::max(std::fabs(wu), std::fabs(wv)))) { w[(size_t)a[v]] = w[(size_t)a[u]]; max_snap = std::max(max_snap, df); }
            }
    }
    if (opt_on(kOpt_verbose)) fprintf(stderr, "[regauge] largest snapped log-weight difference %.3g\n", max_snap);
    return true;
}

int compile_graph(int64_t S, int64_t A, const int32_t *src, const int32_t *dst, const int32_t *lab,
                  const float *w_in, const float *start_in, const float *end_in, int device, HostGraph **out) {
    if (S <= 0 || S > INT32_MAX / 4 || A < 0 || A > INT32_MAX / 4) { set_error("graph size out of range"); return CRF_ERR_UNSUPPORTED; }
    int max_label = 0;
    for (int64_t k = 0; k < A; ++k) {
        if (src[k] < 0 || src[k] >= S || dst[k] < 0 || dst[k] >= S) { set_error("arc endpoint out of range"); return CRF_ERR_ARG; }
        if (lab[k] < 0) { set_error("negative label (epsilon ilabel) on an arc"); return CRF_ERR_FORMAT; }
        max_label = std::max(max_label, (int)lab[k]);
    }
    // the compiler's own potentials for weight-pushed graphs (regauge_pushed): every table below is built from these weights
    std::vector<float> w_v(w_in, w_in + A), start_v(start_in, start_in + S), end_v(end_in, end_in + S);
    const bool regauged = regauge_pushed(S, A, src, dst, lab, w_v, start_v, end_v);
    const float *w = w_v.data(), *start_w = start_v.data(), *end_w = end_v.data();
    // 1. pairs = distinct (dst, label)
    std::map<std::pair<int, int>, int> pair_id;  // (label, dst) -> temp id, ordered => deterministic
    for (int64_t k = 0; k < A; ++k) pair_id.emplace(std::make_pair((int)lab[k], (int)dst[k]), 0);
    const int P = (int)pair_id.size();
    {
        int i = 0;
        for (auto &kv : pair_id) kv.second = i++;
    }
    std::vector<int> tmp_dst(P), tmp_lab(P);
    for (auto &kv : pair_id) { tmp_lab[kv.second] = kv.first.first; tmp_dst[kv.second] = kv.first.second; }
    std::vector<std::vector<std::pair<int, float>>> frows(P);
    std::vector<int> arc_tmp_pair((size_t)A);
    for (int64_t k = 0; k < A; ++k) {
        int tp = pair_id[{(int)lab[k], (int)dst[k]}];
        arc_tmp_pair[(size_t)k] = tp;
        frows[tp].push_back({(int)src[k], expf(w[k])});
    }
    EllHost fe = build_ell(frows);
    const int Pr = (int)fe.row_of.size();
    std::vector<int> pid_of_tmp(P), pair_dst(Pr, -1), pair_lab(Pr, 0);
    for (int r = 0; r < Pr; ++r)
        if (fe.row_of[r] >= 0) { pid_of_tmp[fe.row_of[r]] = r; pair_dst[r] = tmp_dst[fe.row_of[r]]; pair_lab[r] = tmp_lab[fe.row_of[r]]; }
    // 2. backward rows = states, arcs carry the pair id of (dst, label)
    std::vector<std::vector<std::pair<int, float>>> brows((size_t)S);
    for (int64_t k = 0; k < A; ++k) brows[src[k]].push_back({pid_of_tmp[arc_tmp_pair[(size_t)k]], expf(w[k])});
    EllHost be = build_ell(brows);
    const int Sr = (int)be.row_of.size();
    // 3. pairs of each destination state
    std::vector<int> st_pair_off((size_t)S + 1, 0), st_pairs(P);
    for (int r = 0; r < Pr; ++r) if (pair_dst[r] >= 0) st_pair_off[(size_t)pair_dst[r] + 1]++;
    for (int64_t s = 0; s < S; ++s) st_pair_off[s + 1] += st_pair_off[s];
    {
        std::vector<int> fill(st_pair_off.begin(), st_pair_off.end() - 1);
        for (int r = 0; r < Pr; ++r) if (pair_dst[r] >= 0) st_pairs[fill[pair_dst[r]]++] = r;
    }
    // 4. grad-pass tables: pair ids sorted by (label, pair id); chunks of <= kChunk within a label
    std::vector<int> perm;
    perm.reserve(P);
    for (int r = 0; r < Pr; ++r) if (pair_dst[r] >= 0) perm.push_back(r);
    std::stable_sort(perm.begin(), perm.end(), [&](int a, int b) { return pair_lab[a] < pair_lab[b]; });
    std::vector<int> chunk_off{0}, lab_chunk_off((size_t)max_label + 2, 0);
    {
        size_t i = 0;
        for (int v = 0; v <= max_label; ++v) {
            lab_chunk_off[v] = (int)chunk_off.size() - 1;
            size_t j = i;
            while (j < perm.size() && pair_lab[perm[j]] == v) ++j;
            for (size_t c = i; c < j; c += kChunk) chunk_off.push_back((int)std::min(j, c + kChunk));
            i = j;
        }
        lab_chunk_off[(size_t)max_label + 1] = (int)chunk_off.size() - 1;
    }
    std::vector<int2> pair_meta(Pr);
    // label | (1 << 16) when the pair is the only one into its destination state (plain LDS store)
    for (int r = 0; r < Pr; ++r) {
        const bool sole = pair_dst[r] >= 0 && st_pair_off[(size_t)pair_dst[r] + 1] - st_pair_off[pair_dst[r]] == 1;
        pair_meta[r] = int2{pair_dst[r], pair_lab[r] | (sole ? 1 << 16 : 0)};
    }
    std::vector<int4> bwd_row_meta(Sr);
    for (int r = 0; r < Sr; ++r) {
        const int st = be.row_of[r];
        int4 m{-1, 0, 0, 0};
        if (st >= 0) {
            m.x = st;
            m.y = st_pair_off[(size_t)st + 1] - st_pair_off[st];
            if (m.y > 0) { m.z = st_pairs[st_pair_off[st]]; m.w = pair_lab[m.z]; }
        }
        bwd_row_meta[r] = m;
    }
    std::vector<float> start_lin((size_t)S), end_lin((size_t)S);
    for (int64_t s = 0; s < S; ++s) { start_lin[s] = expf(start_w[s]); end_lin[s] = expf(end_w[s]); }
    // 5. utterance-minor ("batch") tables: CSR with one descriptor per row, rows in processing order (most arcs first, so
    //    the long rows of a launch start first); pairs in (label, destination) order = the temporary pair ids
    std::vector<int2> b_farcs((size_t)A), b_barcs((size_t)A);
    std::vector<int4> b_stp((size_t)P), b_frow((size_t)S), b_brow((size_t)S);
    std::vector<int> b_frow_d((size_t)S), b_brow_s((size_t)S), b_lab_off((size_t)max_label + 2, 0);
    {
        auto wbits = [](float x) { int b; memcpy(&b, &x, 4); return b; };
        std::vector<int> st_poff((size_t)S + 1, 0), bst_off((size_t)S + 1, 0);
        for (int tp = 0; tp < P; ++tp) st_poff[(size_t)tmp_dst[tp] + 1]++;
        for (int64_t s = 0; s < S; ++s) st_poff[s + 1] += st_poff[s];
        std::vector<int> fill(st_poff.begin(), st_poff.end() - 1), tmp_at((size_t)P);
        for (int tp = 0; tp < P; ++tp) tmp_at[(size_t)fill[tmp_dst[tp]]++] = tp;       // stp position -> temporary pair id
        int o = 0;
        for (int k = 0; k < P; ++k) {
            const int tp = tmp_at[k];
            b_stp[k] = int4{tp, tmp_lab[tp], o, o + (int)frows[tp].size()};
            for (auto &a : frows[tp]) b_farcs[(size_t)o++] = int2{a.first, wbits(a.second)};
        }
        for (int64_t k = 0; k < A; ++k) bst_off[(size_t)src[k] + 1]++;
        for (int64_t s = 0; s < S; ++s) bst_off[s + 1] += bst_off[s];
        std::vector<int> bfill(bst_off.begin(), bst_off.end() - 1);
        for (int64_t k = 0; k < A; ++k) b_barcs[(size_t)bfill[src[k]]++] = int2{arc_tmp_pair[(size_t)k], wbits(expf(w[k]))};
        std::vector<int> ord((size_t)S);
        std::iota(ord.begin(), ord.end(), 0);
        auto fdeg = [&](int s) { return st_poff[(size_t)s + 1] > st_poff[s] ? b_stp[(size_t)st_poff[(size_t)s + 1] - 1].w - b_stp[(size_t)st_poff[s]].z : 0; };
        std::stable_sort(ord.begin(), ord.end(), [&](int x, int y) { return fdeg(x) > fdeg(y); });
        for (int64_t r = 0; r < S; ++r) {
            const int s0 = ord[(size_t)r], k0 = st_poff[s0], k1 = st_poff[(size_t)s0 + 1];
            b_frow_d[(size_t)r] = s0;
            // one entering pair (every state of a T o LM graph): pair id and label sit in the descriptor itself
            b_frow[(size_t)r] = k1 - k0 == 1 ? int4{b_stp[k0].z, b_stp[k0].w, b_stp[k0].x, b_stp[k0].y | 0x40000000}
                              : k1 > k0 ? int4{b_stp[k0].z, b_stp[(size_t)k1 - 1].w, k0, k1} : int4{0, 0, 0, 0};
        }
        std::iota(ord.begin(), ord.end(), 0);
        std::stable_sort(ord.begin(), ord.end(), [&](int x, int y) { return bst_off[(size_t)x + 1] - bst_off[x] > bst_off[(size_t)y + 1] - bst_off[y]; });
        for (int64_t r = 0; r < S; ++r) {
            const int s0 = ord[(size_t)r];
            b_brow_s[(size_t)r] = s0;
            const int k0 = st_poff[s0], k1 = st_poff[(size_t)s0 + 1];
            b_brow[(size_t)r] = k1 - k0 == 1 ? int4{bst_off[s0], bst_off[(size_t)s0 + 1], b_stp[k0].x, b_stp[k0].y | 0x40000000}
                              : k1 > k0 ? int4{bst_off[s0], bst_off[(size_t)s0 + 1], k0, k1} : int4{bst_off[s0], bst_off[(size_t)s0 + 1], 0, 0};
        }
        for (int tp = 0; tp < P; ++tp) b_lab_off[(size_t)tmp_lab[tp] + 1]++;
        for (int v = 0; v <= max_label; ++v) b_lab_off[(size_t)v + 1] += b_lab_off[v];
    }
    auto *h = new HostGraph();
    h->device = device; h->S = S; h->A = A; h->P = P; h->regauged = regauged ? 1 : 0;
    if (device >= 0) { int n = 0; if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && n > 0) h->ncu = n; else (void)hipGetLastError(); }
    h->fwd_padded_arcs = fe.padded_arcs; h->bwd_padded_arcs = be.padded_arcs;
    h->fwd_conflicts = fe.conflict_cycles; h->bwd_conflicts = be.conflict_cycles;
    for (auto &r : frows) h->max_in_deg = std::max(h->max_in_deg, (int)r.size());
    for (auto &r : brows) h->max_out_deg = std::max(h->max_out_deg, (int)r.size());
    // canonical (label, dst)-ordered pair tables for the register-resident layout
    std::vector<std::vector<std::pair<int, float>>> out_arcs_tmp((size_t)S);
    for (int64_t k = 0; k < A; ++k) out_arcs_tmp[src[k]].push_back({arc_tmp_pair[(size_t)k], expf(w[k])});
    std::vector<int> canon(P);
    std::iota(canon.begin(), canon.end(), 0);
    GraphDev &d0 = h->dev;
    d0.S = (int)S; d0.A = (int)A; d0.P = P; d0.Pr = Pr; d0.Sr = Sr; d0.max_label = max_label;
    d0.NC = (int)chunk_off.size() - 1;
    h->hb_farcs = b_farcs; h->hb_barcs = b_barcs; h->hb_frow = b_frow; h->hb_brow = b_brow;   // (kept: arc streams are cut from them on first use)
    h->hb_frow_d = b_frow_d; h->hb_brow_s = b_brow_s;
    build_batch_factored(h, (int)S, b_frow, b_frow_d, b_farcs, b_brow, b_brow_s, b_barcs, start_lin);
    if (A <= (1 << 20)) {   // (kept for the CPU emulations of the layouts: their reference is the recursion over these arcs)
        h->h_src.assign(src, src + A); h->h_dst.assign(dst, dst + A); h->h_lab.assign(lab, lab + A);
        h->h_w.resize((size_t)A);
        for (int64_t k = 0; k < A; ++k) h->h_w[(size_t)k] = expf(w[k]);
        h->h_start = start_lin; h->h_end = end_lin;
    }
    if (device < 0) {  // host-only compile (diagnostics / CPU tests): tables are built, nothing is uploaded
        h->device = -1;
        int rcr = build_resident(h, (int)S, P, tmp_dst, tmp_lab, frows, out_arcs_tmp, start_lin, end_lin, canon);
        if (!rcr) rcr = build_factored(h, (int)S, P, tmp_dst, tmp_lab, frows, out_arcs_tmp, start_lin, end_lin);
        if (rcr) { delete h; return rcr; }
        *out = h;
        return CRF_OK;
    }
    int prev = 0;
    hipError_t e0 = hipGetDevice(&prev);
    if (e0 != hipSuccess || hipSetDevice(device) != hipSuccess) {
        set_error(std::string("cannot select device ") + std::to_string(device));
        delete h;
        return CRF_ERR_HIP;
    }
    GraphDev &d = h->dev;
    d.S = (int)S; d.A = (int)A; d.P = P; d.Pr = Pr; d.Sr = Sr; d.max_label = max_label;
    d.NC = (int)chunk_off.size() - 1;
    int rc = CRF_OK;
    do {
        if ((rc = upload_ell(h, fe, &d.fwd))) break;
        if ((rc = upload_ell(h, be, &d.bwd))) break;
        if ((rc = upload(h, pair_meta, &d.pair_meta))) break;
        if ((rc = upload(h, bwd_row_meta, &d.bwd_row_meta))) break;
        if ((rc = upload(h, st_pair_off, &d.st_pair_off))) break;
        if ((rc = upload(h, st_pairs, &d.st_pairs))) break;
        if ((rc = upload(h, start_lin, &d.start_lin))) break;
        if ((rc = upload(h, end_lin, &d.end_lin))) break;
        if ((rc = upload(h, perm, &d.perm))) break;
        if ((rc = upload(h, chunk_off, &d.chunk_off))) break;
        if ((rc = upload(h, lab_chunk_off, &d.lab_chunk_off))) break;
        if ((rc = upload(h, b_farcs, &d.bat.farcs)) || (rc = upload(h, b_frow_d, &d.bat.frow_d)) || (rc = upload(h, b_frow, &d.bat.frow)) ||
            (rc = upload(h, b_stp, &d.bat.stp)) || (rc = upload(h, b_barcs, &d.bat.barcs)) || (rc = upload(h, b_brow_s, &d.bat.brow_s)) ||
            (rc = upload(h, b_brow, &d.bat.brow)) || (rc = upload(h, b_lab_off, &d.bat.lab_off))) break;
        d.bat.ok = 1;
        if ((rc = build_resident(h, (int)S, P, tmp_dst, tmp_lab, frows, out_arcs_tmp, start_lin, end_lin, canon))) break;
        if ((rc = build_factored(h, (int)S, P, tmp_dst, tmp_lab, frows, out_arcs_tmp, start_lin, end_lin))) break;
    } while (0);
    (void)hipSetDevice(prev);
    if (rc) {
        for (void *p : h->allocs) (void)hipFree(p);
        delete h;
        return rc;
    }
    *out = h;
    return CRF_OK;
}

}  // namespace crf

extern "C" {

int crf_graph_create_from_arcs(int64_t S, int64_t A, const int32_t *src, const int32_t *dst,
                               const int32_t *lab, const float *w, const float *start_w,
                               const float *end_w, int device, crf_graph **out) {
    if (!out || !start_w || !end_w || (A > 0 && (!src || !dst || !lab || !w))) { crf::set_error("null argument"); return CRF_ERR_ARG; }
    crf::HostGraph *h = nullptr;
    int rc = crf::compile_graph(S, A, src, dst, lab, w, start_w, end_w, device, &h);
    if (rc) return rc;
    *out = new crf_graph{h};
    return CRF_OK;
}

int crf_graph_create(const char *fst_path, int device, crf_graph **out) {
    if (!fst_path || !out) { crf::set_error("null argument"); return CRF_ERR_ARG; }
    int64_t S = 0;
    std::vector<int32_t> src, dst, lab;
    std::vector<float> w, sw, ew;
    int rc = crf::read_fst_file(fst_path, &S, &src, &dst, &lab, &w, &sw, &ew);
    if (rc) return rc;
    return crf_graph_create_from_arcs(S, (int64_t)src.size(), src.data(), dst.data(), lab.data(), w.data(),
                                      sw.data(), ew.data(), device, out);
}

void crf_graph_destroy(crf_graph *g) {
    if (!g) return;
    if (g->h) {
        int prev = 0;
        bool sw = hipGetDevice(&prev) == hipSuccess && hipSetDevice(g->h->device) == hipSuccess;
        for (void *p : g->h->allocs) (void)hipFree(p);
        if (sw) (void)hipSetDevice(prev);
        for (crf::StreamDev *sd : g->h->streams) delete sd;
        delete g->h;
    }
    delete g;
}

int crf_graph_dims(const crf_graph *g, int64_t *S, int64_t *A, int64_t *P, int64_t *max_label) {
    if (!g || !g->h) { crf::set_error("null graph"); return CRF_ERR_ARG; }
    if (S) *S = g->h->S;
    if (A) *A = g->h->A;
    if (P) *P = g->h->P;
    if (max_label) *max_label = g->h->dev.max_label;
    return CRF_OK;
}

int crf_graph_stats(const crf_graph *g, int64_t *out, int n) {
    if (!g || !g->h || !out) { crf::set_error("null argument"); return CRF_ERR_ARG; }
    const crf::HostGraph *h = g->h;
    const int64_t v[27] = {h->S, h->A, h->P, h->dev.Pr, h->dev.Sr, h->fwd_padded_arcs, h->bwd_padded_arcs,
                           h->fwd_conflicts, h->bwd_conflicts, (int64_t)h->max_in_deg * 100000 + h->max_out_deg,
                           h->res_stats.K, h->res_stats.slots_f, h->res_stats.slots_b, h->res_stats.conflicts_f,
                           h->res_stats.conflicts_b, (int64_t)h->dev.res.f.R * 100000 + h->dev.res.b.R,
                           h->fac_stats.ok, h->fac_stats.matched, (int64_t)h->regauged, h->fac_stats.tail,
                           h->fac_stats.slots_f, h->fac_stats.slots_b, h->fac_stats.fused,
                           h->fac_stats.Gf * 100000 + h->fac_stats.Gb,
                           h->fac_stats.ok ? (h->dev.fac.threads == crf::kFac4Threads ? (h->dev.fac.K > 1 ? 5 : 4) : h->dev.fac.threads != crf::kFac3Threads ? 2 : h->dev.fac.K > 1 ? 3 : h->dev.fac.rcl ? 1 : 0) : -1,
                           h->fac_stats.ok ? (h->dev.fac.threads == crf::kFac4Threads ? crf::kFac4NCH : h->dev.fac.threads != crf::kFac3Threads ? crf::kResNCH : h->dev.fac.rcl == 2 ? crf::kFac3LNCH : crf::kFac3ArcCh) : 0,
                           h->facp.ok};
    for (int i = 0; i < n && i < 27; ++i) out[i] = v[i];
    return CRF_OK;
}

int crf_debug_decode_check(int nslot, int ncombo) { return crf::debug_check_decode(nslot, ncombo); }

int crf_debug_fac_emulate(const crf_graph *g, int T, unsigned seed, double *out3) {
    if (!g || !g->h || !out3 || T < 1) { crf::set_error("bad argument"); return CRF_ERR_ARG; }
    return crf::debug_emulate_factored(g->h, T, seed, out3, crf::opt_on(crf::kOpt_emu_facp) ? 1 : 0);
}

int crf_debug_res_emulate(const crf_graph *g, int T, unsigned seed, double *out3) {
    if (!g || !g->h || !out3 || T < 1) { crf::set_error("bad argument"); return CRF_ERR_ARG; }
    return crf::debug_emulate_resident(g->h, T, seed, out3);
}

int crf_debug_facbatch_check(const crf_graph *g, int64_t *out4) {
    if (!g || !g->h || !out4) { crf::set_error("null argument"); return CRF_ERR_ARG; }
    return crf::debug_check_facbatch(g->h, out4);
}

int crf_debug_stream_check(const crf_graph *g, int UL, int want, int64_t *out4) {
    if (!g || !g->h) { crf::set_error("null graph"); return CRF_ERR_ARG; }
    return crf::debug_check_streams(g->h, UL, want, out4);
}

const char *crf_last_error(void) { return crf::last_error_cstr(); }
int crf_debug_set(const char *key, int value) { return crf::opt_set(key, value, false); }
int crf_debug_unset(const char *key) { return crf::opt_set(key, 0, true); }
const char *crf_debug_list(void) { return crf::opt_list(); }
const char *crf_version(void) { return "ctc_crf_hip 0.1.0 (gfx950)"; }

}  // extern "C"
