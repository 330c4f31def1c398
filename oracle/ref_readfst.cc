// oracle/ref_readfst.cc -- TEST INFRASTRUCTURE ONLY.
// Supplies the `ReadFst` symbol that the reference's den_calculate.cu:275-285 declares and
// fst_read.cc:11-62 defines on top of OpenFst (not vendored, cannot be built here).  Own code:
// a plain parser of the OpenFst vector/standard binary layout, filling the output vectors with
// the conventions fst_read.cc:40-60 applies (label = ilabel-1, weight = -cost,
// end_weight = -Final, start_weight[start] = 0).
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

int DEN_NUM_STATES_UNUSED = 0;

namespace {
struct Reader {
    std::vector<unsigned char> buf;
    size_t off = 0;
    template <typename T> T get() {
        T v;
        if (off + sizeof(T) > buf.size()) { fprintf(stderr, "ref_readfst: truncated file\n"); exit(1); }
        memcpy(&v, buf.data() + off, sizeof(T));
        off += sizeof(T);
        return v;
    }
    std::string str() {
        int32_t n = get<int32_t>();
        std::string s((const char *)buf.data() + off, (size_t)n);
        off += (size_t)n;
        return s;
    }
    void skip_symtab() {
        get<int32_t>(); str(); get<int64_t>();
        int64_t n = get<int64_t>();
        for (int64_t i = 0; i < n; ++i) { str(); get<int64_t>(); }
    }
};
}  // namespace

void ReadFst(const char *fst_name, std::vector<std::vector<int> > &alpha_next,
             std::vector<std::vector<int> > &beta_next, std::vector<std::vector<int> > &alpha_ilabel,
             std::vector<std::vector<int> > &beta_ilabel, std::vector<std::vector<float> > &alpha_weight,
             std::vector<std::vector<float> > &beta_weight, std::vector<float> &start_weight,
             std::vector<float> &end_weight, int &num_states, int &num_arcs) {
    FILE *f = fopen(fst_name, "rb");
    if (!f) { fprintf(stderr, "ref_readfst: cannot open %s\n", fst_name); exit(1); }
    Reader r;
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    r.buf.resize((size_t)sz);
    if (fread(r.buf.data(), 1, (size_t)sz, f) != (size_t)sz) { fprintf(stderr, "ref_readfst: short read\n"); exit(1); }
    fclose(f);
    if (r.get<uint32_t>() != 0x7eb2fdd6u) { fprintf(stderr, "ref_readfst: bad magic\n"); exit(1); }
    std::string ft = r.str(), at = r.str();
    if (ft != "vector" || at != "standard") { fprintf(stderr, "ref_readfst: need vector/standard\n"); exit(1); }
    r.get<int32_t>();
    int32_t flags = r.get<int32_t>();
    r.get<uint64_t>();
    int64_t start = r.get<int64_t>();
    int64_t ns = r.get<int64_t>();
    r.get<int64_t>();
    if (flags & 1) r.skip_symtab();
    if (flags & 2) r.skip_symtab();
    num_states = (int)ns;
    num_arcs = 0;
    alpha_next.assign(ns, {}); beta_next.assign(ns, {});
    alpha_ilabel.assign(ns, {}); beta_ilabel.assign(ns, {});
    alpha_weight.assign(ns, {}); beta_weight.assign(ns, {});
    start_weight.assign(ns, -INFINITY);
    end_weight.assign(ns, -INFINITY);
    start_weight[start] = 0.f;
    for (int64_t s = 0; s < ns; ++s) {
        float fin = r.get<float>();
        int64_t na = r.get<int64_t>();
        if (fin != INFINITY) end_weight[s] = -fin;
        for (int64_t k = 0; k < na; ++k) {
            int32_t il = r.get<int32_t>();
            r.get<int32_t>();
            float w = r.get<float>();
            int32_t nx = r.get<int32_t>();
            beta_next[s].push_back(nx);
            alpha_next[nx].push_back((int)s);
            beta_ilabel[s].push_back(il - 1);
            alpha_ilabel[nx].push_back(il - 1);
            beta_weight[s].push_back(-w);
            alpha_weight[nx].push_back(-w);
            ++num_arcs;
        }
    }
}
