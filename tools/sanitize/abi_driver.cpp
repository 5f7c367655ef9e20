// Driver for tools/sanitize/run.sh: a host-only context (device = -1) driven through the C ABI - mesh, k-d build, export,
// the calls that must fail cleanly - with the library's host code under AddressSanitizer.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "rt_hip.h"
static std::vector<char> slurp(const char *p) { FILE *f = fopen(p, "rb"); if (!f) { perror(p); exit(2); } fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET); std::vector<char> b(n); if (fread(b.data(), 1, n, f) != (size_t)n) exit(2); fclose(f); return b; }
int main(int argc, char **argv) {
    auto v = slurp(argv[1]), f = slurp(argv[2]), fm = slurp(argv[3]), want = slurp(argv[4]);
    rt_config c; rt_default_config(&c); c.device = -1; c.width = 64; c.height = 48;
    rt_ctx *ctx = nullptr;
    if (rt_create(&c, &ctx)) { printf("create: %s\n", rt_last_error(nullptr)); return 1; }
    const int nv = (int)(v.size() / 24), nf = (int)(f.size() / 12);
    for (int rep = 0; rep < 2; rep++) {
        if (rt_set_mesh(ctx, (const double *)v.data(), nv, (const uint32_t *)f.data(), (const uint32_t *)fm.data(), nf)) { printf("set_mesh: %s\n", rt_last_error(ctx)); return 1; }
        if (rt_build_kdtree(ctx)) { printf("build: %s\n", rt_last_error(ctx)); return 1; }
        int64_t nn = 0, nl = 0;
        rt_kdtree_counts(ctx, &nn, &nl);
        std::vector<uint32_t> nodes(2 * nn); std::vector<int32_t> leaves(4 * nl); double box[6];
        if (rt_export_kdtree_reference(ctx, nodes.data(), leaves.data(), box)) { printf("export: %s\n", rt_last_error(ctx)); return 1; }
        printf("rep %d: nodes %lld leaves %lld export %s\n", rep, (long long)nn, (long long)nl, (nodes.size() * 4 == want.size() && !memcmp(nodes.data(), want.data(), want.size())) ? "== reference" : "DIFFERS");
    }
    // calls that must fail cleanly on a host-only context
    int r1 = rt_trace(ctx); printf("trace on host-only: %d (%s)\n", r1, rt_last_error(ctx));
    uint8_t px[16]; int r2 = rt_read_rgba8(ctx, px); printf("read on host-only: %d\n", r2);
    int32_t id; int r3 = rt_pick(ctx, 1, 1, &id); printf("pick on host-only: %d\n", r3);
    rt_set_mesh(ctx, nullptr, 0, nullptr, nullptr, 0);
    int r4 = rt_build_kdtree(ctx); printf("build of empty mesh: %d\n", r4);
    rt_destroy(ctx);
    rt_config bad = c; bad.width = 0; rt_ctx *b = nullptr; printf("bad config: %d (%s)\n", rt_create(&bad, &b), rt_last_error(nullptr));
    return 0;
}
