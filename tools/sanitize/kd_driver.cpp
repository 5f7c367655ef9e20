// Driver for tools/sanitize/run.sh: the host k-d builder on a fixture mesh, export compared with the reference arrays.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "kd_build.h"
static std::vector<char> slurp(const char *p) { FILE *f = fopen(p, "rb"); if (!f) { perror(p); exit(2); } fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET); std::vector<char> b(n); if (fread(b.data(), 1, n, f) != (size_t)n) exit(2); fclose(f); return b; }
int main(int argc, char **argv) {
    auto v = slurp(argv[1]), f = slurp(argv[2]), want = slurp(argv[3]);
    const int64_t nv = v.size() / 24, nf = f.size() / 12;
    for (int threads : {1, 4, 16}) {
        rt::KdTreeHost t;
        rt::build_kdtree((const double *)v.data(), nv, (const uint32_t *)f.data(), nf, 24, 100000.0, threads, t);
        std::vector<uint32_t> pairs(2 * (size_t)t.n_nodes);
        rt::export_reference(t, pairs.data());
        const bool same = pairs.size() * 4 == want.size() && !memcmp(pairs.data(), want.data(), want.size());
        printf("threads %d: nodes %lld leaves %zu depth %d export %s\n", threads, (long long)t.n_nodes, t.leaves.size(), t.depth, same ? "== reference" : "DIFFERS");
        if (!same) return 1;
    }
    // degenerate inputs
    { rt::KdTreeHost t; double vv[9] = {0,0,0, 1,0,0, 0,1,0}; uint32_t ff[3] = {0,1,2}; rt::build_kdtree(vv, 3, ff, 1, 24, 100000.0, 2, t); printf("one face: nodes %lld\n", (long long)t.n_nodes); }
    { rt::KdTreeHost t; rt::build_kdtree(nullptr, 0, nullptr, 0, 24, 100000.0, 2, t); printf("no faces: nodes %lld\n", (long long)t.n_nodes); }
    return 0;
}
