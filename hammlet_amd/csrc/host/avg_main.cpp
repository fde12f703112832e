// avg - average of non-overlapping windows of a data stream; same argument, input and output as the reference's tool
// (reference src/tools/avg.cpp:18-42): `avg WINDOW < values`, one line per window (a shorter last window included), the
// window's values added up in float in stream order and divided by their count.  The values are extracted by the GPU
// text reader (hml_text_*: bit-identical to `cin >> v`, including where it stops); the sums run on the host in the
// reference's order.
#include <iostream>
#include <sstream>
#include <stdexcept>
#include <vector>

#include "hml.h"

static void check(int rc) {
    if (rc != 0) throw std::runtime_error(hml_last_error());
}

int main(int argc, const char* argv[]) {
    try {
        if (argc <= 1) throw std::runtime_error("Not enough arguments!");
        std::istringstream ss(argv[1]);
        size_t windowSize = 0;
        ss >> windowSize;
        hml_text* reader = nullptr;
        check(hml_text_open(&reader, 0, 0));
        for (;;) {
            char* buf = nullptr;
            uint64_t cap = 0;
            check(hml_text_buffer(reader, &buf, &cap));
            std::cin.read(buf, (std::streamsize)cap);
            const std::streamsize got = std::cin.gcount();
            if (got <= 0) break;
            check(hml_text_commit(reader, (uint64_t)got));
        }
        uint64_t n = 0;
        int stopped = 0;
        check(hml_text_finish(reader, &n, &stopped));
        std::vector<float> x(n);
        check(hml_text_values(reader, x.data()));
        hml_text_close(reader);
        float sum = 0;
        size_t pos = 0;
        std::string out;
        std::ostringstream line;
        for (uint64_t i = 0; i < n; ++i) {
            sum += x[i];
            pos++;
            if (pos == windowSize) {
                std::cout << sum / pos << "\n";
                pos = 0;
                sum = 0;
            }
        }
        if (pos != 0) std::cout << sum / pos << "\n";
        std::cout.flush();
        return 0;
    } catch (std::exception& e) {
        std::cerr << "avg: " << e.what() << std::endl;
        return 1;
    }
}
