#include "reorder.h"
#include <chrono>
#include <cstdio>
#include <vector>
using namespace gespmm;
static std::vector<int32_t> rd(const char* f) { FILE* h = fopen(f, "rb"); fseek(h, 0, SEEK_END); long n = ftell(h); fseek(h, 0, SEEK_SET); std::vector<int32_t> v(n / 4); fread(v.data(), 4, v.size(), h); fclose(h); return v; }
int main(int argc, char** argv) {
    for (const char* name : {"com-amazon-like", "com-amazon-sbm"}) {
        char a[256], b[256]; snprintf(a, 256, "/tmp/%s.rp", name); snprintf(b, 256, "/tmp/%s.ci", name);
        auto rp = rd(a), ci = rd(b); int64_t M = rp.size() - 1, K = M;
        printf("%s storage order: %.3f\n", name, simulate_l2_hits(M, K, rp.data(), ci.data(), nullptr, 8, 6144));
        for (int cap : {64, 128, 256, 512, 1024}) for (int growth : {2, 4, 8}) for (int sweeps : {5, 10}) {
            ClusterOptions o; o.threads = 8; o.first_cap = cap; o.cap_growth = growth; o.sweeps = sweeps;
            std::vector<int32_t> perm(M); ClusterStats st;
            auto t0 = std::chrono::steady_clock::now();
            cluster_rows(M, K, rp.data(), ci.data(), o, perm.data(), &st);
            double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            printf("  cap %4d growth %d sweeps %2d: hits %.3f  levels %d  %.0f ms\n", cap, growth, sweeps,
                   simulate_l2_hits(M, K, rp.data(), ci.data(), perm.data(), 8, 6144), st.levels, ms);
        }
    }
}
