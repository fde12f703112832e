// maxSegmentation - maximum-posterior-margin segmentation of a marginals file; same flags, input and output as the
// reference's post-processing tool (reference src/tools/maxSegmentation.cpp:27-83).  File-level glue only: inside a
// run the same segmentation comes from the device without the file round trip (hml_max_segmentation, `-O X`).
//
// Output rules of the reference tool, kept as they are: one line "LENGTH<TAB>STATE" per run of equal arg-max state
// (first maximum wins, a row without positive counts gives state 0); the running state starts at 0, so a file whose
// first segment has another state begins with the line "0<TAB>0"; empty input gives "0<TAB>0".
#include <cstdint>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>

#include "hammlet/Parser.hpp"

using namespace hammlet;

int main(int argc, const char* argv[]) {
    Parser args(argc, argv);
    args.registerFlags({"-i", "-infile"}, "");
    args.registerFlags({"-h", "--help", "-help"}, "");
    args.parseArgs();
    if (args.isSet("-h")) {
        std::cout << "Given a marginals file (-i) or input from STDIN, computes the maximum posterior margins segmentation, "
                     "combining adjacent segments whenever possible." << std::endl;
        return 0;
    }
    std::ifstream file;
    if (args.isSet("-i")) file.open(args.parse<std::string>("-i"), std::ios::in);
    std::istream& in = args.isSet("-i") ? static_cast<std::istream&>(file) : std::cin;

    uint64_t run_len = 0;
    uint64_t run_state = 0, state = 0;
    std::string line;
    while (std::getline(in, line)) {
        const char* p = line.c_str();
        char* end = nullptr;
        // first field: segment length; remaining fields: counts per state (non-numeric text ends the row)
        uint64_t len = strtoull(p, &end, 10);
        if (end == p) len = 0;
        p = end;
        uint64_t best = 0, column = 0;
        state = 0;
        for (;; ++column) {
            const uint64_t c = strtoull(p, &end, 10);
            if (end == p) break;
            p = end;
            if (c > best) { best = c; state = column; }
        }
        if (state == run_state) run_len += len;
        else {
            std::cout << run_len << "\t" << run_state << std::endl;
            run_len = len;
            run_state = state;
        }
    }
    std::cout << run_len << "\t" << state << std::endl;
    return 0;
}
