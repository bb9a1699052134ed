#include "reorder.h"
#include <chrono>
#include <cstdio>
#include <random>
#include <vector>
#include <algorithm>
using namespace gespmm;
int main() {
    // a com-Amazon-sized random symmetric-ish graph (structure does not matter for timing)
    const int64_t M = 334863; const int64_t E = 925872;
    std::mt19937_64 rng(1); std::vector<std::pair<int32_t,int32_t>> ed; ed.reserve(2*E);
    for (int64_t i = 0; i < E; ++i) { int32_t a = rng() % M, b = rng() % M; ed.push_back({a,b}); ed.push_back({b,a}); }
    std::sort(ed.begin(), ed.end());
    std::vector<int32_t> rp(M + 1, 0), ci(ed.size());
    for (size_t i = 0; i < ed.size(); ++i) { rp[ed[i].first + 1]++; ci[i] = ed[i].second; }
    for (int64_t i = 0; i < M; ++i) rp[i + 1] += rp[i];
    for (int th : {1, 2, 4, 8, 16, 32}) {
        ClusterOptions o; o.threads = th; std::vector<int32_t> perm(M); ClusterStats st;
        double best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            auto t0 = std::chrono::steady_clock::now();
            cluster_rows(M, M, rp.data(), ci.data(), o, perm.data(), &st);
            best = std::min(best, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        }
        printf("threads %2d: %.0f ms (levels %d)\n", th, best, st.levels);
    }
}
