// sortStates - states ordered by the absolute value of their last sampled emission mean; same input (a parameters
// file written with -O P) and output as the reference's post-processing script (reference bin/sortStates:1-6:
// tail | awk | sed | sort -k 2 -n -r | awk).  File-level glue; the chain-parallel pooling applies the same idea on the
// device side (hammlet_amd/chains.py: relabel_permutation).
//
// Output: "#state<TAB>mean", then one line "STATE<TAB>MEAN" per state, largest |mean| first; the mean is printed as it
// stands in the file.  Equal keys fall back to the comparison `sort -r` makes on the whole intermediate line
// "STATE<TAB>ABS<TAB>SIGN" (bytewise, reversed).
#include <algorithm>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

int main(int argc, const char* argv[]) {
    if (argc < 2) { std::cerr << "usage: sortStates PARAMETERS_FILE" << std::endl; return 1; }
    std::ifstream in(argv[1]);
    std::string line, last;
    while (std::getline(in, line)) last = line;          // tail -n 1
    std::cout << "#state\tmean" << std::endl;
    std::istringstream fields(last);
    std::vector<std::string> tok;
    for (std::string t; fields >> t;) tok.push_back(t);
    struct Row { std::string state, abs, sign, key_line; double key; };
    std::vector<Row> rows;
    for (size_t i = 0; i < tok.size(); i += 2) {          // odd fields are the means
        Row r;
        r.state = std::to_string(i / 2);
        const std::string& m = tok[i];
        // sed 's/-([^-]+)/\1\t-/g' on "STATE\tMEAN": a leading minus moves behind the number
        if (!m.empty() && m[0] == '-' && m.size() > 1) { r.abs = m.substr(1); r.sign = "-"; }
        else { r.abs = m; r.sign = ""; }
        r.key = strtod(r.abs.c_str(), nullptr);
        r.key_line = r.state + "\t" + r.abs + (r.sign.empty() ? "" : "\t" + r.sign);
        rows.push_back(r);
    }
    std::sort(rows.begin(), rows.end(), [](const Row& a, const Row& b) {
        if (a.key != b.key) return a.key > b.key;         // -n -r on field 2
        return a.key_line > b.key_line;                   // last-resort comparison, reversed
    });
    for (const Row& r : rows) std::cout << r.state << "\t" << r.sign << r.abs << std::endl;
    return 0;
}
