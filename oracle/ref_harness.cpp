// TEST INFRASTRUCTURE - stage-level harness around the UNMODIFIED reference headers (compiled from
// /root/reference/src where they lie, output only into oracle/_ref/; see oracle/Makefile).  It calls the
// reference's own functions in src/main.cpp's order (main.cpp:264-352) on one input trace and dumps what each
// stage produced, so that tests/golden/make_stage_golden.py can commit the values as fixtures:
//   maxlet coefficients (wavelet.hpp:97-188), noise estimate (main.cpp:303-311), breakpoint weights
//   (wavelet.hpp:68-93), the integral array (Statistics/IntegralArray.hpp:136-191), the auto prior
//   (AutoPriors.hpp:86-110), and for each requested threshold the block list (Blocks/BreakpointArray.hpp:216-235),
//   the block statistics (IntegralArray.hpp:104-124,198-212) and, for fixed parameters, the per-state emission
//   terms innerProduct - N*logNormalizer (EFD.hpp:23-38, StateSequence/ForwardBackward.hpp:74-76).
// The reference's classes keep these arrays private; this translation unit (and only it) is compiled with
// -fno-access-control to read them.
//
//   ref_harness INPUT.f32 OUTPUT.bin MULT THR [THR ...]
//   ref_harness --parse INPUT.txt OUTPUT.f32     the values the reference's reader extracts from a text file
//                                                (MaxletTransform, wavelet.hpp:97-134: suffstats[i].sum() = v_i)
// OUTPUT.bin: a sequence of records {char name[16]; uint64 count; uint32 elem_size; payload}.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "Tags.hpp"
#include "HMM.hpp"
#include "Parser.hpp"
#include "Emissions.hpp"
#include "Blocks.hpp"
#include "AutoPriors.hpp"
#include "Records.hpp"
#include "wavelet.hpp"
#include "StateSequence.hpp"
#include "Statistics.hpp"
#include "includes.hpp"
#include "utils.hpp"

static FILE* g_out = nullptr;
static void put(const char* name, const void* data, uint64_t count, uint32_t elem) {
    char nm[16] = {0};
    strncpy(nm, name, 15);
    fwrite(nm, 1, 16, g_out);
    fwrite(&count, 8, 1, g_out);
    fwrite(&elem, 4, 1, g_out);
    if (count) fwrite(data, elem, count, g_out);
}

int main(int argc, char** argv) {
    if (argc == 4 && std::string(argv[1]) == "--parse") {
        std::ifstream fin(argv[2]);
        vector<real_t> coeffs;
        vector<SufficientStatistics<Normal>> stats;
        MaxletTransform(fin, coeffs, stats, 1);
        std::vector<float> vals(stats.size());
        for (size_t i = 0; i < stats.size(); ++i) vals[i] = stats[i].sum();
        FILE* f = fopen(argv[3], "wb");
        if (!vals.empty()) fwrite(vals.data(), 4, vals.size(), f);
        fclose(f);
        return 0;
    }
    if (argc < 5) { fprintf(stderr, "usage: ref_harness INPUT.f32 OUTPUT.bin MULT THR [THR ...]\n"); return 2; }
    std::vector<float> x;
    {
        std::ifstream f(argv[1], std::ios::binary);
        f.seekg(0, std::ios::end);
        const size_t n = (size_t)f.tellg() / 4;
        f.seekg(0);
        x.resize(n);
        f.read(reinterpret_cast<char*>(x.data()), n * 4);
    }
    g_out = fopen(argv[2], "wb");
    const real_t mult = (real_t)atof(argv[3]);
    // the reference reads text: %.9g round-trips float32 through `istream >> float`
    std::stringstream text;
    {
        char buf[64];
        for (float v : x) { snprintf(buf, sizeof buf, "%.9g\n", (double)v); text << buf; }
    }
    const size_t nrDataDim = 1;
    vector<real_t> inputValues;
    vector<SufficientStatistics<Normal>> stats;
    MaxletTransform(text, inputValues, stats, nrDataDim, x.size() + 1);
    const size_t T = inputValues.size();
    put("coeffs", inputValues.data(), T, 4);
    double stdEstimate = 0;
    size_t nrDetailCoeffs = 0;
    for (size_t i = 1; i < inputValues.size(); i += 2) { stdEstimate += inputValues[i]; nrDetailCoeffs++; }
    stdEstimate /= nrDetailCoeffs;
    stdEstimate /= 0.797884560802865355879892119868763736951717262329869315331;
    put("sigma", &stdEstimate, 1, 8);
    HaarBreakpointWeights(inputValues);
    for (auto& w : inputValues) w *= mult;
    put("weights", inputValues.data(), T, 4);

    typedef Statistics<IntegralArray, Normal> S;
    typedef Blocks<BreakpointArray> B;
    S ia(stats, nrDataDim);
    {
        std::vector<float> flat(2 * (T + 1));
        for (size_t i = 0; i <= T; ++i) { flat[2 * i] = ia.mStats[i].sum(); flat[2 * i + 1] = ia.mStats[i].sumSq(); }
        put("integral", flat.data(), 2 * (T + 1), 4);
    }
    B waveletBlocks(inputValues);
    Emissions<S, B> y(ia, waveletBlocks);
    {
        vector<real_t> prior = autoPrior((real_t)0.2, (real_t)0.9, y, stdEstimate);
        put("autoprior", prior.data(), prior.size(), 4);
    }
    // fixed parameters for the emission terms: three states
    const real_t means[3] = {-1.0f, 0.25f, 1.5f}, vars[3] = {0.04f, 0.09f, 0.5f};
    for (int a = 4; a < argc; ++a) {
        const real_t thr = (real_t)atof(argv[a]);
        y.createBlocks(thr);
        y.initForward();
        std::vector<uint32_t> starts;
        std::vector<float> sums, sumsqs, E;
        while (y.next()) {
            starts.push_back((uint32_t)y.start());
            sums.push_back(y.suffStat(0).sum());
            sumsqs.push_back(y.suffStat(0).sumSq());
            const real_t N = y.blockSize();
            for (int s = 0; s < 3; ++s) {
                Observation<NormalParam> p(means[s], vars[s]);
                const real_t e = innerProduct(y.suffStat(0), p) - N * logNormalizer(p);
                E.push_back(e);
            }
        }
        starts.push_back((uint32_t)T);
        put("thr", &thr, 1, 4);
        put("starts", starts.data(), starts.size(), 4);
        put("sum", sums.data(), sums.size(), 4);
        put("sumsq", sumsqs.data(), sumsqs.size(), 4);
        put("E", E.data(), E.size(), 4);
    }
    fclose(g_out);
    return 0;
}
