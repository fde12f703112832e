// TEST INFRASTRUCTURE - NOT PART OF THE PRODUCT (see hml_oracle.hpp).
//
// Command-line front end of the CPU restatement.  It accepts the subset of the reference's flags
// (src/main.cpp:33-63) that the parity tests use and writes the same output files
// (src/Records.hpp:99-235), so its files can be compared byte for byte with those of the
// unmodified reference binary oracle/_ref/hammlet.  Extra flags: --rng/--math/--reduce/--chain
// select the device-mode deviations, --raw FILE reads float32 instead of text.
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>

#include "hml_oracle.hpp"

using namespace hml_oracle;

static std::vector<std::string> toks(const std::map<std::string, std::vector<std::string>>& m, const std::string& k,
                                     const std::string& def) {
    auto it = m.find(k);
    if (it != m.end()) return it->second;
    std::vector<std::string> v;
    std::istringstream ss(def);
    std::string t;
    while (ss >> t) v.push_back(t);
    return v;
}

int main(int argc, const char* argv[]) {
    try {
        std::map<std::string, std::vector<std::string>> a;
        std::map<std::string, std::string> alias = {
            {"-input-file", "-f"}, {"-output-pattern", "-o"}, {"-output-data", "-O"}, {"-overwrite", "-w"},
            {"-states", "-s"}, {"-emissions", "-e"}, {"-auto-priors", "-a"}, {"-transitions", "-t"},
            {"-no-self-transitions", "-S"}, {"-initial-dist", "-I"}, {"-random-seed", "-R"},
            {"-iterations", "-i"}, {"-weight-multiplier", "-m"}};
        const char* known[] = {"-f", "-o", "-O", "-w", "-s", "-e", "-a", "-t", "-S", "-I", "-R", "-i", "-m",
                               "--rng", "--math", "--reduce", "--chain", "--raw", "-v"};
        std::string cur;
        for (int i = 1; i < argc; ++i) {
            std::string t = argv[i];
            if (alias.count(t)) t = alias[t];
            bool isflag = false;
            for (auto k : known) if (t == k) isflag = true;
            if (isflag) { cur = t; a[cur]; }
            else {
                if (cur.empty()) throw std::runtime_error("First argument must be a flag!");
                a[cur].push_back(t);
            }
        }
        Config cfg;
        {   // "-s K" or "-s C P [D]" (main.cpp:114-137): K = P^D states with the combinations mapping
            auto st = toks(a, "-s", "3");
            if (st.size() == 1) cfg.K = std::stoi(st[0]);
            else {
                if (st[0] != "C" && st[0] != "combinations") throw std::runtime_error("Unknown mapping type " + st[0] + "!");
                cfg.P = std::stoi(st[1]);
                cfg.D = st.size() >= 3 ? std::stoi(st[2]) : 1;
                long k = 1;
                for (int d = 0; d < cfg.D; ++d) k *= cfg.P;
                cfg.K = (int)k;
            }
        }
        auto e = toks(a, "-e", "normal 0.2 0.9");
        cfg.e_var = std::stof(e[1]); cfg.e_p = std::stof(e[2]);
        if (!a.count("-a")) throw std::runtime_error("Manual theta priors not implemented, use -a!");
        auto t = toks(a, "-t", "0.5 0.5");
        cfg.t_off = std::stof(t[0]); cfg.t_diag = t.size() > 1 ? std::stof(t[1]) : cfg.t_off;
        cfg.self_trans = !a.count("-S");
        cfg.pi_alpha = std::stof(toks(a, "-I", "0.5")[0]);
        cfg.seed = std::stoull(toks(a, "-R", "0")[0]);
        cfg.weight_mult = std::stof(toks(a, "-m", "1")[0]);
        cfg.rng = std::stoi(toks(a, "--rng", "0")[0]);
        cfg.math = std::stoi(toks(a, "--math", "0")[0]);
        cfg.reduce = std::stoi(toks(a, "--reduce", "0")[0]);
        cfg.chain = (uint32_t)std::stoul(toks(a, "--chain", "0")[0]);

        std::string opref, osuff;
        if (!a.count("-o") && a.count("-f")) {
            std::string fn = a["-f"][0];
            size_t i = fn.find_last_of(".");
            opref = fn.substr(0, i) + "-";
            osuff = fn.substr(i);
        } else {
            auto o = toks(a, "-o", "hammlet- .csv");
            opref = o[0]; osuff = o[1];
        }

        std::vector<float> x;
        if (a.count("--raw")) {
            std::ifstream f(a["--raw"][0], std::ios::binary);
            f.seekg(0, std::ios::end);
            size_t n = (size_t)f.tellg() / 4;
            f.seekg(0);
            x.resize(n);
            f.read((char*)x.data(), n * 4);
        } else if (a.count("-f")) {
            for (auto& fn : a["-f"]) {
                std::ifstream f(fn);
                if (!f) throw std::runtime_error("Cannot read from input file " + fn + "!");
                float v;
                while (f >> v) x.push_back(v);
            }
        } else {
            float v;
            while (std::cin >> v) x.push_back(v);
        }

        Oracle o(cfg);
        auto O = toks(a, "-O", "marginals");
        auto has = [&](const char* s, const char* l) { for (auto& z : O) if (z == s || z == l) return true; return false; };
        o.rec_marginals = has("M", "marginals");
        o.rec_sequences = has("S", "sequences");
        o.rec_params = has("P", "parameters");
        o.rec_blocks = has("B", "blocks");
        o.rec_compression = has("C", "compression");
        o.rec_segments = has("G", "segments");

        o.load(x.data(), x.size());
        { std::vector<float>().swap(x); }
        o.autoprior();
        o.init_model();

        auto sch = toks(a, "-i", "M 500 0 S P F 200 0 F 300 3");
        size_t nn = 0;
        for (auto& c : sch) if (c != "P" && c != "S" && c != "D") nn++;
        if (nn % 3 != 0) throw std::runtime_error("Parameters for -i, excluding \"P\", \"S\" and \"D\", must be multiples of 3!");
        for (size_t i = 0; i < sch.size();) {
            const std::string m = sch[i];
            if (m == "P") { o.token_P(); i++; continue; }
            if (m == "S") { o.token_S(); i++; continue; }
            if (m == "D") { o.token_D(); i++; continue; }
            o.token_begin();
            if (i + 2 >= sch.size()) throw std::runtime_error("Incomplete command line for -i!");
            size_t iters = std::stoull(sch[i + 1]), thin = std::stoull(sch[i + 2]);
            i += 3;
            if (m != "F" && m != "M") throw std::runtime_error("Unknown sampling type " + m + "!");
            if (thin > iters) std::cout << "[WARNING] Thinning parameter is larger than number of iterations. No data will be recorded!" << std::endl;
            for (size_t it = 0; it < iters; ++it) {
                bool rec = thin > 0 && ((it + 1) % thin == 0);
                o.sweep(m[0], rec);
            }
        }
        for (uint64_t k = 0; k < o.warn_uniform; ++k) std::cout << "[WARNING] Uniform sampling of forward variables!" << std::endl;

        auto put = [&](const char* type, const std::string& s) {
            std::ofstream f(opref + type + osuff);
            f << s;
        };
        if (o.rec_marginals) put("marginals", o.marginals_text());
        if (o.rec_sequences) put("sequences", o.out_sequences);
        if (o.rec_blocks) put("blocks", o.out_blocks);
        if (o.rec_params) put("parameters", o.out_params);
        if (o.rec_compression) put("compression", o.out_compression);
        if (o.rec_segments) put("segments", o.out_segments);
        return 0;
    } catch (std::exception& e) {
        std::cout << std::flush;
        std::cerr << std::endl << "[ERROR] " << e.what() << std::endl;
        std::cerr << "Terminating HaMMLET. The rest is silence." << std::endl;
        return 1;
    }
}
