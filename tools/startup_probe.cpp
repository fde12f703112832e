// Where the per-process start-up goes: time of every C-ABI call of a minimal run (10^5 positions, 3 states, 10 sweeps).
//   g++ -O2 -std=c++17 -o tools/bin/startup_probe tools/startup_probe.cpp -Iinclude -Lhammlet_amd -lhammlet_hip -Wl,-rpath,$PWD/hammlet_amd
#include <chrono>
#include <cstdio>
#include <vector>
#include "hml.h"
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define STEP(name, call) do { const double t0 = now(); if ((call) != 0) { printf("%s failed: %s\n", name, hml_last_error()); return 1; } printf("%-28s %8.1f ms\n", name, 1e3 * (now() - t0)); } while (0)
int main() {
    const uint64_t T = 100000;
    std::vector<float> x(T);
    const float mu[3] = {-1, 0, 1};
    hml_synth_gauss(x.data(), nullptr, T, 3, mu, 0.2f, 2000.0, 1, 4);
    const double t_start = now();
    hml_ctx* c = nullptr;
    STEP("hml_create", hml_create(&c, 0, 1, 0, nullptr));
    STEP("hml_load_observations", hml_load_observations(c, x.data(), T));
    float prior[4];
    STEP("hml_autoprior", hml_autoprior(c, 0.2f, 0.9f, prior));
    STEP("hml_set_model", hml_set_model(c, 3, prior, 0.5f, 0.5f, 0.5f, 1));
    STEP("hml_sample_prior", hml_sample_prior(c));
    STEP("hml_iterate(1) + sync", (hml_iterate(c, 'F', 1, 1) || hml_sync(c)));
    STEP("hml_iterate(100) + sync", (hml_iterate(c, 'F', 100, 1) || hml_sync(c)));
    uint64_t n = 0; int cols = 0;
    STEP("hml_marginals_rle (count)", hml_marginals_rle(c, &n, &cols, nullptr, nullptr));
    { const double t0 = now(); hml_destroy(c); printf("%-28s %8.1f ms\n", "hml_destroy", 1e3 * (now() - t0)); }
    printf("%-28s %8.1f ms\n", "total", 1e3 * (now() - t_start));
    return 0;
}
