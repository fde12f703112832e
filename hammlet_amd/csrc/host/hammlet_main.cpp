// hammlet - command-line driver of the MI355X-native sampler.  Same flags, same text-stream input, same
// output files and the same error format as the reference's driver (reference src/main.cpp:23-477;
// flag semantics doc/hammlet-manpage.md:33-175); everything between reading the input and writing the
// files runs on the GPU through libhammlet_hip.so.
//
// Extensions (not in the reference): `-O X` writes PREFIXmaxsegmentationSUFFIX; -raw FILE reads float32 values instead of text; -device N selects
// the GPU; -chain N selects the Philox sub-key of an independent chain.
#include <ctime>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "hammlet/Parser.hpp"
#include "hammlet/hammlet.hpp"

using namespace hammlet;
using std::cerr;
using std::cout;
using std::endl;
using std::flush;
using std::string;
using std::vector;

static const char* kHelp =
    "hammlet (MI355X) - Bayesian HMM segmentation with dynamic Haar-wavelet compression\n\n"
    "  -f, -input-file FILE...        input files (default: standard input), whitespace-separated numbers\n"
    "  -raw FILE                      float32 input file (extension)\n"
    "  -o, -output-pattern PRE SUF    output files are PRE{marginals,...}SUF (default: hammlet- .csv)\n"
    "  -O, -output-data M S P B C G   marginals sequences parameters blocks compression segments\n"
    "                    X            maxsegmentation: the maxSegmentation tool's output for the marginals (extension)\n"
    "  -w, -overwrite                 allow overwriting output files\n"
    "  -s, -states K | C P D          number of states (default 3), or P parameters shared by P^D states over D dimensions\n"
    "  -e, -emissions normal VAR P    automatic prior: P(variance < VAR) = P (default normal 0.2 0.9)\n"
    "  -a, -auto-priors               (required) derive emission priors from the data\n"
    "  -t, -transitions OFF [DIAG]    Dirichlet prior of the transition rows (default 0.5 0.5)\n"
    "  -S, -no-self-transitions       do not model within-block self-transitions\n"
    "  -I, -initial-dist ALPHA        Dirichlet prior of the initial distribution (default 0.5)\n"
    "  -R, -random-seed N             seed (default: time)\n"
    "  -i, -iterations SCHEME         tokens: M n t | F n t | S | D | P (default M 500 0 S P F 200 0 F 300 3)\n"
    "  -m, -weight-multiplier F       multiply breakpoint weights (default 1)\n"
    "  -v, -verbose   -g, -arguments   -h, -help\n";

int main(int argc, const char* argv[]) {
    try {
        Parser args(argc, argv);
        args.registerFlags({"-v", "-verbose"});
        args.registerFlags({"-g", "-arguments"});
        args.registerFlags({"-h", "-help", "--help"});
        args.registerFlags({"-f", "-input-file"});
        args.registerFlags({"-o", "-output-pattern"}, "hammlet- .csv");
        args.registerFlags({"-O", "-output-data"}, "marginals");
        args.registerFlags({"-w", "-overwrite"});
        args.registerFlags({"-s", "-states"}, "3");
        args.registerFlags({"-e", "-emissions"}, "normal 0.2 0.9");
        args.registerFlags({"-a", "-auto-priors"});
        args.registerFlags({"-t", "-transitions"}, "0.5 0.5");
        args.registerFlags({"-S", "-no-self-transitions"});
        args.registerFlags({"-I", "-initial-dist"}, "0.5");
        args.registerFlags({"-R", "-random-seed"}, std::to_string(time(0)));
        args.registerFlags({"-i", "-iterations"}, "M 500 0 S P F 200 0 F 300 3");
        args.registerFlags({"-m", "-weight-multiplier"}, "1");
        // extensions (registered last so that `-g` prints the reference's lines first)
        args.registerFlags({"-raw"});
        args.registerFlags({"-device"}, "0");
        args.registerFlags({"-chain"}, "0");
        args.parseArgs();

        if (args.isSet("-g")) args.print();
        const bool verbose = args.isSet("-v");
        const bool overwrite = args.isSet("-w");
        if (args.isSet("-h")) {
            cout << endl << kHelp << endl;
            return 0;
        }

        // output pattern: without -o, "-f name.ext" yields "name-" ".ext"
        string opref, osuff;
        if (!args.isSet("-o") && args.isSet("-f")) {
            const string filename = args.parse<string>("-f");
            const size_t i = filename.find_last_of(".");
            opref = filename.substr(0, i) + "-";
            osuff = filename.substr(i);
        } else {
            opref = args.parse<string>("-o", 0);
            osuff = args.parse<string>("-o", 1);
        }

        const size_t rng_seed = args.parse<size_t>("-R", 0);
        const int device = args.parse<int>("-device");
        const uint32_t chain = args.parse<uint32_t>("-chain");

        // states: "-s K", or "-s C P D": P emission parameters shared by P^D states over D data dimensions whose values
        // follow each other in the input (reference main.cpp:114-137)
        size_t nrParams, nrDataDim = 1;
        if (args.nrTokens("-s") == 1) {
            nrParams = args.parse<size_t>("-s", 0);
        } else {
            const string m = args.parse<string>("-s", 0);
            if (m != "C" && m != "combinations") throw std::runtime_error("Unknown mapping type " + m + "!");
            nrParams = args.parse<size_t>("-s", 1);
            if (args.nrTokens("-s") >= 3) nrDataDim = args.parse<size_t>("-s", 2);
        }
        Mapping mapping(nrDataDim, nrParams, combinations);
        const size_t nrStates = mapping.nrStates();

        // first token = off-diagonal, second = diagonal (reference main.cpp:144-149)
        const real_t trans = args.parse<real_t>("-t", 0);
        real_t selfTrans = trans;
        if (args.nrTokens("-t") > 1) selfTrans = args.parse<real_t>("-t", 1);
        TransitionHyperParam<DirichletParamVector> tau_A(nrStates, trans, selfTrans);
        const bool useSelfTrans = !args.isSet("-S");
        const real_t initialAlpha = args.parse<real_t>("-I", 0);
        InitialHyperParam<DirichletParam> tau_pi(nrStates, initialAlpha);
        const real_t weightMultiplier = args.parse<real_t>("-m");

        vector<vector<real_t>> thetaParams;
        if (!args.isSet("-a")) throw std::runtime_error("Manual theta priors not implemented, use -a!");
        const vector<real_t> thp = args.parseVector<real_t>("-e", 1, 3);
        for (size_t i = 0; i < nrParams; ++i) thetaParams.push_back(thp);

        if (verbose) {
            cout << "Data dimensions: " << nrDataDim << endl;
            cout << "Emission distributions: " << nrParams << endl;
            cout << "States: " << nrStates << endl;
            string scheme;
            for (const string& t : args.tokens("-i")) scheme += (scheme.empty() ? "" : " ") + t;
            cout << "Sampling scheme: " << scheme << endl;
            cout << "Random seed: " << rng_seed << endl;
        }

        Parser outputArgs = args.subparser("-output-data");
        outputArgs.registerFlags({"M", "marginals"});
        outputArgs.registerFlags({"S", "sequences"});
        outputArgs.registerFlags({"P", "parameters"});
        outputArgs.registerFlags({"B", "blocks"});
        outputArgs.registerFlags({"C", "compression"});
        outputArgs.registerFlags({"D", "mapping"});
        outputArgs.registerFlags({"G", "segments"});
        outputArgs.registerFlags({"X", "maxsegmentation"});   // extension
        outputArgs.parseArgs();

        // ---- input
        inputDevice() = device;
        vector<real_t> inputValues;   // the observations (the device computes coefficients, weights and statistics)
        if (args.isSet("-raw")) {
            const string fname = args.parse<string>("-raw");
            std::ifstream fin(fname, std::ios::binary);
            if (!fin) throw std::runtime_error("Cannot read from input file " + fname + "!");
            fin.seekg(0, std::ios::end);
            const size_t n = (size_t)fin.tellg() / sizeof(float);
            fin.seekg(0);
            inputValues.resize(n);
            fin.read(reinterpret_cast<char*>(inputValues.data()), n * sizeof(float));
        } else if (args.isSet("-f")) {
            for (const string& fname : args.parseVector<string>("-f")) {
                if (verbose) cout << "Reading " + fname << endl << flush;
                std::ifstream fin(fname);
                if (!fin) throw std::runtime_error("Cannot read from input file " + fname + "!");
                // upper estimate of the number of values (the reference counts the lines, main.cpp:277): a value and
                // its separator take at least two bytes
                fin.seekg(0, std::ios::end);
                const std::streamoff bytes = fin.tellg();
                fin.seekg(0);
                readValues(fin, inputValues, nrDataDim, bytes > 0 ? (size_t)bytes / 2 + 1 : 0);
            }
        } else {
            if (verbose) cout << "Reading from standard input" << endl << flush;
            readValues(std::cin, inputValues, nrDataDim);
        }
        if (verbose) cout << "Output will be written to " + opref + "*" + osuff << endl << flush;
        // (the reference counts the coefficients, one per position; here the vector still holds the D values of every position)
        if (inputValues.size() % nrDataDim != 0)
            throw std::runtime_error("Input stream did not contain enough values to fill all dimensions at last position!");
        const size_t T = inputValues.size() / nrDataDim;
        if (verbose) cout << "Number of data points: " + std::to_string(T) << endl << flush;

        if (inputValues.empty()) throw std::runtime_error("Cannot compute Haar breakpoint weights, vector is empty!");
        if (verbose) cout << "Calculating Haar breakpoint weights" << endl << flush;

        // the device context: created once every argument has been parsed and the input has been read (and before
        // `records`, whose destructor fetches the marginals from it)
        rng_t RNG(rng_seed, device, chain);
        Transitions<DirichletVector> A(nrStates, RNG);
        Initial<Dirichlet> pi(nrStates, RNG);

        Records records(T, opref, osuff, nrStates);
        records.setRecordStateSequence(outputArgs.isSet("sequences"), overwrite);
        records.setRecordTheta(outputArgs.isSet("parameters"), overwrite);
        records.setRecordBlocks(outputArgs.isSet("blocks"), overwrite);
        records.setRecordCompression(outputArgs.isSet("compression"), overwrite);
        records.setRecordMarginals(outputArgs.isSet("marginals"), overwrite);
        records.setRecordSegments(outputArgs.isSet("segments"), overwrite);
        records.setRecordMaxSegmentation(outputArgs.isSet("maxsegmentation"), overwrite);

        typedef Statistics<IntegralArray, Normal> S;
        typedef Blocks<BreakpointArray> B;
        S ia(inputValues, nrDataDim);          // upload + maxlet transform + weights + integral array (GPU)
        B waveletBlocks(ia);
        if (weightMultiplier != 1) waveletBlocks.scaleWeights(weightMultiplier);
        Emissions<S, B> y(ia, waveletBlocks);
        records.attach(y.ctx());

        const double stdEstimate = ia.noiseEstimate();
        thetaParams[0] = autoPrior(thetaParams[0][0], thetaParams[0][1], y, stdEstimate);
        for (auto& p : thetaParams) p = thetaParams[0];
        ThetaHyperParam<NormalInverseGammaParam> tau_theta(thetaParams);
        Theta<NormalInverseGamma> theta(tau_theta, tau_A, tau_pi, useSelfTrans, RNG);

        // ---- sampling scheme (reference main.cpp:368-454)
        size_t nrTokens = 0;
        for (const string& c : args.tokens("-i"))
            if (c != "P" && c != "S" && c != "D") nrTokens++;
        if (nrTokens % 3 != 0)
            throw std::runtime_error("Parameters for -i, excluding \"P\", \"S\" and \"D\", must be multiples of 3!");
        nrTokens = args.nrTokens("-i");

        bool samplePrior = true;
        bool dynamic = true;
        if (verbose) cout << "Setting block structure to dynamic" << endl << flush;
        for (size_t i = 0; i < nrTokens;) {
            const string method = args.parse<string>("-i", i);
            if (samplePrior) {
                if (verbose) cout << "Sampling prior" << endl << flush;
                hml_check(hml_sample_prior(RNG.ctx()));
                samplePrior = false;
            }
            size_t iterations = 0, thinning = 0;
            if (method == "P") {
                samplePrior = true;
                i++;
                continue;
            } else if (method == "S") {
                if (verbose) cout << "Setting block structure to static" << endl << flush;
                y.createBlocks(theta);
                dynamic = false;
                i++;
                continue;
            } else if (method == "D") {
                if (verbose) cout << "Setting block structure to dynamic" << endl << flush;
                dynamic = true;   // (sampleHMM switches the device back to per-sweep recompression)
                i++;
                continue;
            } else {
                if (i + 2 >= nrTokens) throw std::runtime_error("Incomplete command line for -i!");
                iterations = args.parse<size_t>("-i", i + 1);
                thinning = args.parse<size_t>("-i", i + 2);
                i += 3;
            }
            if (method == "F") {
                if (verbose) cout << "Sampling Forward-Backward" << endl << flush;
                StateSequence<ForwardBackward> q(RNG);
                sampleHMM(y, q, theta, tau_theta, A, tau_A, pi, tau_pi, mapping, iterations, thinning, records, dynamic, useSelfTrans);
            } else if (method == "M") {
                if (verbose) cout << "Sampling mixture" << endl << flush;
                StateSequence<Mixture> q(RNG);
                sampleHMM(y, q, theta, tau_theta, A, tau_A, pi, tau_pi, mapping, iterations, thinning, records, dynamic, useSelfTrans);
            } else {
                throw std::runtime_error("Unknown sampling type " + method + "!");
            }
        }
        hml_check(hml_sync(RNG.ctx()));
        records.close();
        if (verbose) cout << "Exit HaMMLET" << endl << flush;
        return 0;
    } catch (std::exception& e) {
        cout << flush;
        cerr << endl << flush << "[ERROR] " << e.what() << endl;
        cerr << "Terminating HaMMLET. The rest is silence." << endl << flush;
        return 1;
    }
}
