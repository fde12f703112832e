// mapLinesToGenome - prepend genome coordinates to the lines of a data file (one line per window of genome positions,
// or per run-length segment of windows); same flags, files and output as the reference's tool (reference
// src/tools/mapLinesToGenome.cpp:29-177 over src/tools/GenomeGetter.hpp:24-118).  File-level glue of the WGS workflow
// (config 5): PREFIX-size.csv lists "refseq<TAB>entries<TAB>cumulative entries", PREFIX-pos.csv.gz the covered
// positions, one per line.
//
// Behaviour of the reference tool that is kept as it is:
//   * output always goes to standard output; -o/-outfile only creates (truncates) the named file
//     (mapLinesToGenome.cpp:70-75 opens it, :139-156 write to cout);
//   * with -b the first tab-separated field is the segment size, read with std::stoi; a line without a tab is its own
//     size field AND its own payload (:110-112);
//   * the last data line may cover fewer genome positions than its window, silently (:163-165); any other mismatch
//     between data and genome ends with the reference's messages ("Data too long for genome!", "Data ended before
//     genome!", "Genome ended before all data was processed!", "Not enough entries in position file!").
// The reference lets these exceptions escape (abort); this tool prints the message and exits with status 1.
#include <cstring>
#include <fstream>
#include <iostream>
#include <limits>
#include <sstream>
#include <stdexcept>
#include <string>

#include "gz_lines.hpp"
#include "hammlet/Parser.hpp"

using namespace hammlet;

namespace {

// The walk over the compressed genome: one call per covered position, in file order (GenomeGetter::next).
class GenomeCursor {
    GzLines sizes_, positions_;
    std::string text_;
    std::string name_;
    size_t pos_ = 0;
    size_t in_refseq_ = 0, done_in_refseq_ = 0, cumulative_ = 0;
    bool fresh_ = false;

public:
    explicit GenomeCursor(const std::string& prefix) {
        if (!sizes_.open(prefix + "-size.csv")) throw std::runtime_error("Cannot read " + prefix + "-size.csv!");
        if (!positions_.open(prefix + "-pos.csv.gz")) throw std::runtime_error("Cannot read " + prefix + "-pos.csv.gz!");
    }
    // advance to the next covered position; false once the size file has no further reference sequence
    bool advance() {
        std::stringstream fields(text_);
        if (done_in_refseq_ == in_refseq_) {
            fresh_ = true;
            if (!sizes_.next(text_)) {
                name_.clear();
                pos_ = 0;
                return false;
            }
            fields.str(text_);
            fields >> name_;
            fields >> in_refseq_;
            fields >> cumulative_;
            done_in_refseq_ = 0;
            fields.clear();
        } else {
            fresh_ = false;
        }
        if (!positions_.next(text_)) throw std::runtime_error("Not enough entries in position file!");
        fields.str(text_);
        fields >> pos_;
        fields.clear();
        ++done_in_refseq_;
        return true;
    }
    const std::string& refseq() const { return name_; }
    size_t pos() const { return pos_; }
    bool new_refseq() const { return fresh_; }
};

int run(int argc, const char* argv[]) {
    Parser args(argc, argv);
    args.registerFlags({"-g", "-genome-prefix"}, "");
    args.registerFlags({"-c", "-coordinates"}, "");
    args.registerFlags({"-w", "-window-size"}, "1");
    args.registerFlags({"-r", "-range"});
    args.registerFlags({"-i", "-infile"}, "");
    args.registerFlags({"-o", "-outfile"}, "");
    args.registerFlags({"-b", "-blocks"}, "");
    args.registerFlags({"-h", "--help", "-help"}, "");
    args.parseArgs();
    if (args.isSet("-h")) {
        std::cout << "Prepend genomic coordinates to lines, separated by tabs. If -c/-coordinates is set, coordinates are refseq:start:inclusiveend, otherwise they are separated by tabs. The input file name is set using -i, otherwise lines are read from STDIN. The PREFIX for the genomes size and position files is set using -g/-genome-prefix, and genomic coordinates are read from PREFIX-size.csv and PREFIX-pos.csv. The window size -w/-window size specifies the number of genome coordinates corresponding to one line in the data, for example if mapping counts have been averaged over adjacent, non-overlapping windows. The last data line may map to less genome positions than the window size, and no warning is issued. If -b/-blocks is set, the first entry in each input line (up to the first tab) is considered the segment size for a run-length encoding, and thus specifies that the lines should be repeated this many times; in that case, the window size is multiplied by that number. If -r/-range is specified, only the first and last genome position (in columns) is printed for each input segment per refseq, instead of mapping to all genome positions. If an INT is provided as argument to -r, this specifies the maximum distance between adjacent positions within a range, otherwise a new range is started. If -o/-outfile is specified, output is written to that path, otherwise it is written to STDOUT."
                  << std::endl;
        return 0;
    }
    const std::string prefix = args.parse<std::string>("-genome-prefix");

    std::ifstream data_file;
    if (args.isSet("-i")) data_file.open(args.parse<std::string>("-i"), std::ios::in);
    std::istream& data = args.isSet("-i") ? static_cast<std::istream&>(data_file) : std::cin;
    std::ofstream named_out;   // created, never written (see the header)
    if (args.isSet("-o")) named_out.open(args.parse<std::string>("-o"), std::ios::out);
    std::ostream& out = std::cout;

    const bool run_lengths = args.isSet("-b");
    const bool ranges = args.isSet("-range");
    GenomeCursor genome(prefix);
    const char* after_refseq = args.isSet("-coordinates") ? ":" : "\t";
    const char* after_start = args.isSet("-coordinates") ? "-" : "\t";
    const size_t window = args.parse<size_t>("-w");
    size_t merge_reach = std::numeric_limits<size_t>::max();   // farthest step between neighbours of one range
    if (ranges && args.nrTokens("-range") > 0) merge_reach = args.parse<size_t>("-range", 0);

    std::string refseq, line;
    size_t first = 0, last = 0, segment = 1;
    while (std::getline(data, line)) {
        if (run_lengths) {
            const size_t tab = line.find_first_of("\t");
            segment = (size_t)std::stoi(line.substr(0, tab));
            line = line.substr(tab + 1);
            if (segment == 0) throw std::runtime_error("Segment size must be positive!");
        }
        size_t to_cover = window * segment;   // genome positions this data line stands for
        if (ranges) {
            if (!genome.advance()) throw std::runtime_error("Genome ended before all data was processed!");
            if (genome.new_refseq()) refseq = genome.refseq();
            first = last = genome.pos();
            for (--to_cover; to_cover > 0; --to_cover) {
                if (!genome.advance()) break;   // the genome may end inside the last window
                if (genome.new_refseq() || genome.pos() - last > merge_reach) {
                    out << refseq << after_refseq << first << after_start << last << "\t" << line << std::endl;
                    refseq = genome.refseq();
                    first = genome.pos();
                }
                last = genome.pos();
            }
            out << refseq << after_refseq << first << after_start << last << "\t" << line << std::endl;
        } else {
            for (; to_cover > 0; --to_cover) {
                if (!genome.advance()) break;
                out << genome.refseq() << after_refseq << genome.pos() << "\t" << line << std::endl;
            }
        }
        // only the last window may be incomplete
        if (to_cover >= window) throw std::runtime_error("Data too long for genome!");
    }
    if (genome.advance()) throw std::runtime_error("Data ended before genome!");
    return 0;
}

}  // namespace

int main(int argc, const char* argv[]) {
    try {
        return run(argc, argv);
    } catch (std::exception& e) {
        std::cout.flush();
        std::cerr << "mapLinesToGenome: " << e.what() << std::endl;
        return 1;
    }
}
