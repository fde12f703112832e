// combineCounts - add and subtract per-position counts of several compressed genome files; same flags, files and
// output as the reference's tool (reference src/tools/combineCounts.cpp:30-205 over src/tools/MappedValues.hpp:20-93).
// Every input PREFIX names three files - PREFIX-size.csv ("refseq<TAB>entries<TAB>cumulative"), PREFIX-pos.csv.gz and
// PREFIX-count.csv.gz (one number per line) - and -i lists prefixes behind "+" or "-".  Per reference sequence the
// (position, signed count) pairs of all inputs are sorted by position and equal positions summed; the result goes to
// OUT-size.csv / OUT-pos.csv.gz / OUT-count.csv.gz with the reference sequences in the order they were first seen.
//
// Behaviour of the reference tool that is kept as it is:
//   * numbers are read with atoi (int; text that is no number counts 0; a file that ends early yields zeros);
//   * entries whose counts cancel stay in the output with count 0; a reference sequence listed with 0 entries comes
//     out with the single entry (0, 0) (sortAddAndCompress resizes to at least one element, MappedValues.hpp:76-93);
//   * -o and the list of -i are read before -h is looked at; -n ends with "-n not implemented yet!" after all inputs
//     were read.
// The reference lets exceptions escape (abort); this tool prints the message and exits with status 1.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "gz_lines.hpp"
#include "hammlet/Parser.hpp"

using namespace hammlet;

namespace {

typedef std::pair<size_t, long> Tally;   // (position, count)

// sort by position, sum equal positions in place (never fewer than one element, like the reference)
void merge_equal_positions(std::vector<Tally>& v) {
    std::sort(v.begin(), v.end(), [](const Tally& a, const Tally& b) { return a.first < b.first; });
    size_t kept = 0;
    for (size_t i = 1; i < v.size(); ++i) {
        if (v[i].first == v[kept].first) v[kept].second += v[i].second;
        else v[++kept] = v[i];
    }
    v.resize(kept + 1);
}

int run(int argc, const char* argv[]) {
    Parser args(argc, argv);
    args.registerFlags({"-i", "-input-prefices"}, "");
    args.registerFlags({"-p", "-pos-suffix"}, "-pos.csv.gz");
    args.registerFlags({"-c", "-count-suffix"}, "-count.csv.gz");
    args.registerFlags({"-s", "-size-suffix"}, "-size.csv");
    args.registerFlags({"-n", "-normalization-prefix"}, "mappability");
    args.registerFlags({"-o", "-out-prefix"});
    args.registerFlags({"-h", "--help", "-help"}, "");
    args.parseArgs();
    const std::string pos_suffix = args.parse<std::string>("-pos-suffix", 0);
    const std::string count_suffix = args.parse<std::string>("-count-suffix", 0);
    const std::string size_suffix = args.parse<std::string>("-size-suffix", 0);
    const std::string out_prefix = args.parse<std::string>("-out-prefix");
    const std::vector<std::string> inputs = args.parseVector<std::string>("-input-prefices");
    if (args.isSet("-h")) {
        std::cout << "Takes lists of file prefices (-i) and adds their counts (use + and - before lists of files), and adds them. Shared suffices for files can be set using -p, -c, and -s, for position, count and size. The output prefix is set using -o. If -n is provided, its argument is used as a prefix for normalization, i.e. counts are multiplied if a position exists (e.g. for mappability correction)."
                  << std::endl;
        return 0;
    }

    std::unordered_map<std::string, std::vector<Tally>> by_refseq;
    std::vector<std::string> seen_order;
    long sign = 1;
    if (inputs[0] != "+" && inputs[0] != "-") throw std::runtime_error("First token of -i must be + or -!");
    for (const std::string& token : inputs) {
        if (token == "+") { sign = 1; continue; }
        if (token == "-") { sign = -1; continue; }
        std::cout << (sign > 0 ? "Adding" : "Subtracting") << " counts for " << token << "*" << std::endl;
        std::ifstream sizes(token + size_suffix);
        GzLines positions, counts;
        positions.open(token + pos_suffix);
        counts.open(token + count_suffix);
        if (!sizes.good()) throw std::runtime_error("Cannot open " + token + size_suffix + "!");
        std::string text, refseq, rest;
        size_t entries = 0;
        while (std::getline(sizes, text)) {
            std::stringstream fields(text);
            fields >> refseq >> entries >> rest;
            if (by_refseq.find(refseq) == by_refseq.end()) {
                seen_order.push_back(refseq);
                by_refseq.insert({refseq, {}});
            }
            std::vector<Tally>& tallies = by_refseq[refseq];
            tallies.reserve(tallies.size() + entries);
            for (size_t i = 0; i < entries; ++i) {
                positions.next(text);
                const size_t pos = (size_t)atoi(text.c_str());
                counts.next(text);
                tallies.push_back(Tally(pos, sign * (long)atoi(text.c_str())));
            }
            merge_equal_positions(tallies);
        }
    }
    if (args.isSet("-n")) throw std::runtime_error("-n not implemented yet!");

    std::cout << "Writing output to " << out_prefix << "*" << std::endl;
    std::ofstream sizes(out_prefix + size_suffix);
    GzOut positions, counts;
    positions.open(out_prefix + pos_suffix);
    counts.open(out_prefix + count_suffix);
    size_t total = 0;
    for (const std::string& refseq : seen_order) {
        const std::vector<Tally>& tallies = by_refseq[refseq];
        total += tallies.size();
        sizes << refseq << "\t" << tallies.size() << "\t" << total << std::endl;
        for (const Tally& t : tallies) {
            positions.write(std::to_string(t.first) + "\n");
            counts.write(std::to_string(t.second) + "\n");
        }
    }
    return 0;
}

}  // namespace

int main(int argc, const char* argv[]) {
    try {
        return run(argc, argv);
    } catch (std::exception& e) {
        std::cout.flush();
        std::cerr << "combineCounts: " << e.what() << std::endl;
        return 1;
    }
}
