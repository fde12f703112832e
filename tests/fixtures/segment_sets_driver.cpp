// Test fixture (host logic, no GPU): hammlet::MarginalSegmentSets of include/hammlet/Records.hpp - the rule behind the `segments`
// side file (reference src/Records.hpp:208-209, src/StateMarginals.hpp:51-137,204) - fed with the recorded sweeps of a
// `sequences` file (one line per sweep: tab-separated SIZE:STATE runs); prints one `#segments \t queue length` line per sweep.
#include <cstdint>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "hammlet/hammlet.hpp"   // (declares the C ABI the other classes of the header use; nothing of it is called here)

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    std::ifstream in(argv[1]);
    std::string line;
    hammlet::MarginalSegmentSets sets;
    while (std::getline(in, line)) {
        std::vector<uint32_t> runStart;
        std::vector<int16_t> runState;
        uint64_t pos = 0;
        std::istringstream ss(line);
        std::string tok;
        while (ss >> tok) {
            const size_t colon = tok.find(':');
            runStart.push_back((uint32_t)pos);
            runState.push_back((int16_t)std::stoi(tok.substr(colon + 1)));
            pos += std::stoull(tok.substr(0, colon));
        }
        const uint64_t queued = sets.addSweep(runStart, runState, pos);
        std::cout << sets.nrSegments() << "\t" << queued << "\n";
    }
    return 0;
}
