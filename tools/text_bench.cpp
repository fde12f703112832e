// Measures the text reader (hml_text_*) against the statement it replaces, `while ( input >> v )`
// (reference src/wavelet.hpp:131), on a synthetic column of N values written the way measurement files are
// ("%.5f\n" / "%.9g\n").  Prints one JSON line.
//   g++ -O2 -std=c++17 -o tools/bin/text_bench tools/text_bench.cpp -Iinclude -Lhammlet_amd -lhammlet_hip -Wl,-rpath,$PWD/hammlet_amd
//   tools/bin/text_bench [N=100000000] [fmt=%.5f] [cpu_sample=20000000]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "hml.h"

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CHECK(x) do { if ((x) != 0) { fprintf(stderr, "%s: %s\n", #x, hml_last_error()); return 1; } } while (0)

int main(int argc, char** argv) {
    const uint64_t N = argc > 1 ? strtoull(argv[1], nullptr, 10) : 100000000ull;
    const std::string fmt = std::string(argc > 2 ? argv[2] : "%.5f") + "\n";
    const uint64_t cpu_n = argc > 3 ? strtoull(argv[3], nullptr, 10) : 20000000ull;
    std::vector<float> x(N);
    std::vector<int16_t> st;
    const float mu[5] = {-2, -1, 0, 1, 2};
    CHECK(hml_synth_gauss(x.data(), nullptr, N, 5, mu, 0.3f, 5000.0, 3, 16));
    std::string text;
    text.reserve(N * 10);
    {
        char buf[64];
        for (uint64_t i = 0; i < N; ++i) text.append(buf, (size_t)snprintf(buf, sizeof buf, fmt.c_str(), (double)x[i]));
    }
    // (a) from memory through hml_text_feed (includes the copy into the pinned staging buffer)
    std::vector<float> got;
    double t_mem = 0, t_file = 0;
    uint64_t irregular = 0;
    for (int rep = 0; rep < 2; ++rep) {   // second run: pinned allocations and code objects warm
        hml_text* r = nullptr;
        const double t0 = now();
        CHECK(hml_text_open(&r, 0, 0));
        const double t1 = now();
        CHECK(hml_text_feed(r, text.data(), text.size()));
        uint64_t n = 0; int stopped = 0;
        CHECK(hml_text_finish(r, &n, &stopped));
        got.resize(n);
        CHECK(hml_text_values(r, got.data()));
        t_mem = now() - t1;
        if (rep == 0) fprintf(stderr, "open (pinned + device buffers): %.3f s\n", t1 - t0);
        CHECK(hml_text_counters(r, nullptr, &irregular, nullptr));
        hml_text_close(r);
        if (n != N || stopped) { fprintf(stderr, "count mismatch %llu\n", (unsigned long long)n); return 1; }
    }
    // (b) from a file through hml_text_buffer/commit (read() straight into the staging buffer)
    const char* path = "/tmp/hml_text_bench.txt";
    { std::ofstream f(path, std::ios::binary); f.write(text.data(), (std::streamsize)text.size()); }
    {
        hml_text* r = nullptr;
        CHECK(hml_text_open(&r, 0, 0));
        const double t1 = now();
        std::ifstream f(path, std::ios::binary);
        for (;;) {
            char* buf; uint64_t cap;
            CHECK(hml_text_buffer(r, &buf, &cap));
            f.read(buf, (std::streamsize)cap);
            if (f.gcount() <= 0) break;
            CHECK(hml_text_commit(r, (uint64_t)f.gcount()));
        }
        uint64_t n = 0; int stopped = 0;
        CHECK(hml_text_finish(r, &n, &stopped));
        std::vector<float> g2(n);
        CHECK(hml_text_values(r, g2.data()));
        t_file = now() - t1;
        hml_text_close(r);
        if (memcmp(g2.data(), got.data(), n * 4) != 0) { fprintf(stderr, "file/memory mismatch\n"); return 1; }
    }
    if (const char* keep = getenv("HML_TEXT_KEEP")) rename(path, keep); else remove(path);
    // (c) the reference's statement on the first cpu_n values of the same text, one thread
    uint64_t bytes_cpu = 0;
    { uint64_t k = 0; while (bytes_cpu < text.size() && k < cpu_n) { if (text[bytes_cpu] == '\n') ++k; ++bytes_cpu; } }
    std::vector<float> ref;
    ref.reserve(cpu_n);
    double t_cpu;
    {
        std::istringstream input(text.substr(0, bytes_cpu));
        const double t0 = now();
        float v;
        while (input >> v) ref.push_back(v);
        t_cpu = now() - t0;
    }
    if (memcmp(ref.data(), got.data(), ref.size() * 4) != 0) { fprintf(stderr, "VALUES DIFFER from the stream extraction\n"); return 1; }
    printf("{\"values\": %llu, \"bytes\": %zu, \"format\": \"%s\", \"gpu_reader_from_memory_s\": %.4f, \"gpu_reader_from_file_s\": %.4f, "
           "\"gpu_values_per_s\": %.4g, \"gpu_text_GB_per_s\": %.3f, \"irregular_tokens\": %llu, "
           "\"cpu_istream_values\": %zu, \"cpu_istream_s\": %.3f, \"cpu_values_per_s\": %.4g, \"identical_to_stream_extraction\": true}\n",
           (unsigned long long)N, text.size(), argc > 2 ? argv[2] : "%.5f", t_mem, t_file, N / t_mem, text.size() / t_mem / 1e9,
           (unsigned long long)irregular, ref.size(), t_cpu, ref.size() / t_cpu);
    return 0;
}
