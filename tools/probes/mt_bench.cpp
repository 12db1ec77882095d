// round 4: phase timing of the numpy-legacy donor draws on the host (twist+temper per word; donors with k = 0 -> the masked
// rejection alone, k = 2 / 5 -> plus the backward walks).  build: hipcc -O3 -std=c++17 tools/probes/mt_bench.cpp -o build_ab/mt_bench
#include "../../stochopy_amd/csrc/sx_mt19937.cpp"
#include <chrono>
#include <cstdio>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    sx_mt *g = sx_mt_create(0);
    // twist throughput
    double best = 1e9; uint32_t sink = 0;
    for (int r = 0; r < 5; ++r) {
        double t0 = now();
        for (int b = 0; b < 40000; ++b) { mt_twist(g); sink ^= g->out[b % 624]; }
        best = std::min(best, now() - t0);
    }
    printf("twist+temper: %.3f ns/word (sink %u)\n", best * 1e9 / (40000.0 * 624), sink);
    std::vector<int32_t> don(5 * 4096);
    for (int k : {0, 2, 5}) {
        best = 1e9;
        for (int r = 0; r < 5; ++r) { double t0 = now(); sx_mt_de_donors(g, 4096, k, don.data()); best = std::min(best, now() - t0); }
        printf("donors k=%d: %.2f ms\n", k, best * 1e3);
    }
}
