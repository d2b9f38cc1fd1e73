#include <cstdio>
#include <random>
#include <vector>
#include "yk_host.h"
int main() {
    std::mt19937 rng(5);
    std::uniform_real_distribution<float> u(-10.f, 10.f), s(0.f, 0.5f);
    for (unsigned method = 0; method < 3; ++method) {
        std::vector<yk::ShapeBounds> b(200000);
        for (auto& x : b) { for (int k = 0; k < 3; ++k) { x.bmin[k] = u(rng); x.bmax[k] = x.bmin[k] + s(rng); } }
        yk::HostBvh out;
        yk::build_bvh(b, 1 + method, method, out);
        std::printf("method %u: %zu nodes depth %u\n", method, out.nodes.size(), out.depth);
    }
}
