// hammlet - command-line driver of the MI355X-native sampler.  Same flags, same text-stream input, same
// output files and the same error format as the reference's driver (reference src/main.cpp:23-477;
// flag semantics doc/hammlet-manpage.md:33-175); everything between reading the input and writing the
// files runs on the GPU through libhammlet_hip.so.
//
// Extensions (not in the reference): `-O X` writes PREFIXmaxsegmentationSUFFIX; -raw FILE reads float32 values instead of text; -device N selects
// the GPU; -chain N selects the Philox sub-key of an independent chain; -chains N runs N independent chains (sub-keys
// chain .. chain+N-1), chain k on GPU (device + k) mod #GPUs in its own host thread, and pools their recorded marginals
// with one all-reduce over RCCL before PREFIXmarginalsSUFFIX is written (hml_allreduce_marginals); the per-sweep side
// files of chain k >= 1 are PREFIXchainK.{sequences,...}SUFFIX.
#include <condition_variable>
#include <cstdlib>
#include <ctime>
#include <exception>
#include <fstream>
#include <iostream>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
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
    "  -device N  -chain N            GPU and Philox sub-key of the chain (extensions)\n"
    "  -compat                        reference-compatible mode: the reference's mt19937 stream, libm arithmetic and\n"
    "                                 summation orders on the GPU - the same files as the reference for the same -R\n"
    "                                 (univariate models; one lane walks the blocks: for traces up to ~10^6 positions)\n"
    "  -chains N                      N independent chains, one per GPU, marginals pooled over RCCL (extension);\n"
    "                                 chains beyond the number of GPUs share a GPU and the construction it holds.\n"
    "                                 The pooled marginals / maxsegmentation files use common labels (states by\n"
    "                                 ascending mean); PREFIX[chainK.]relabelSUFFIX lists each chain's own label of\n"
    "                                 pooled state 0, 1, ... (its parameters / sequences files keep its own labels)\n"
    "  -v, -verbose   -g, -arguments   -h, -help\n";

// one entry of the sampling scheme (-i)
struct Step {
    string method;
    size_t iterations = 0, thinning = 0;
    bool incomplete = false;
};

// everything a chain needs besides the observations
struct Job {
    size_t T = 0, nrDataDim = 1, nrStates = 0, seed = 0;
    string opref, osuff;
    bool overwrite = false, useSelfTrans = true;
    real_t weightMultiplier = 1, trans = 0.5, selfTrans = 0.5, initialAlpha = 0.5;
    vector<vector<real_t>> thetaParams;
    vector<Step> scheme;
    std::map<string, bool> outputs;
};

// Meeting point of the chain threads of `-chains N` and the main thread: a chain arrives with its context once its
// scheme has run, the main thread pools the marginals of all of them (hml_allreduce_marginals) and lets them go on to
// write their files - or tells them not to when a chain failed.
class Rendezvous {
    std::mutex mMutex;
    std::condition_variable mCv;
    const int mExpected;
    vector<hml_ctx*> mCtx;
    int mAbandoned = 0;
    bool mReleased = false, mOk = false;

public:
    explicit Rendezvous(int n) : mExpected(n), mCtx(n, nullptr) {}
    // chain side: true = the marginals were pooled, write them
    bool arrive(int index, hml_ctx* ctx) {
        std::unique_lock<std::mutex> lock(mMutex);
        mCtx[index] = ctx;
        mCv.notify_all();
        mCv.wait(lock, [&] { return mReleased; });
        return mOk;
    }
    void abandon() {
        std::lock_guard<std::mutex> lock(mMutex);
        ++mAbandoned;
        mCv.notify_all();
    }
    // main side
    vector<hml_ctx*> waitForAll() {
        std::unique_lock<std::mutex> lock(mMutex);
        auto arrived = [&] { int n = 0; for (hml_ctx* c : mCtx) n += c != nullptr; return n; };
        mCv.wait(lock, [&] { return arrived() + mAbandoned >= mExpected; });
        vector<hml_ctx*> out;
        if (mAbandoned == 0) out = mCtx;
        return out;
    }
    void release(bool ok) {
        std::lock_guard<std::mutex> lock(mMutex);
        mReleased = true;
        mOk = ok;
        mCv.notify_all();
    }
};

// Chains of `-chains N` that share a GPU share the construction of the observations (weights, summary, integral arrays -
// hml_attach_observations): the first chain of a device builds it and applies the weight multiplier, the others attach to
// it, and the builder goes on only when all of them have (its context must be alive while they attach).
class DeviceTraces {
    std::mutex mMutex;
    std::condition_variable mCv;
    std::map<int, hml_ctx*> mSource;
    std::map<int, int> mPending;
    std::map<int, bool> mFailed;

public:
    void expect(int device) { ++mPending[device]; }   // (before the threads start)
    void publish(int device, hml_ctx* ctx) {
        std::unique_lock<std::mutex> lock(mMutex);
        mSource[device] = ctx;
        mCv.notify_all();
        mCv.wait(lock, [&] { return mPending[device] <= 0; });
    }
    void fail(int device) {
        std::lock_guard<std::mutex> lock(mMutex);
        mFailed[device] = true;
        mCv.notify_all();
    }
    hml_ctx* waitForSource(int device) {
        std::unique_lock<std::mutex> lock(mMutex);
        mCv.wait(lock, [&] { return mSource.count(device) != 0 || mFailed[device]; });
        if (!mSource.count(device)) throw std::runtime_error("The chain that loads the observations on this device failed!");
        return mSource[device];
    }
    void attached(int device) {
        std::lock_guard<std::mutex> lock(mMutex);
        --mPending[device];
        mCv.notify_all();
    }
};

// One chain from its device context to its output files.  `index` > 0 (chains of `-chains N` beyond the first): the
// per-sweep side files carry the infix "chainK." and the (pooled) marginals are left to chain 0.
// `traces` (chains sharing GPUs): `builds` = this chain builds its device's construction, else it attaches to it.
static void runChain(const Job& job, vector<real_t>& inputValues, bool steal, int device, uint32_t chainId, int index, bool verbose,
                     Rendezvous* rendezvous, DeviceTraces* traces = nullptr, bool builds = true) {
    // an attaching chain reports to the builder of its device whatever happens to it (the builder waits for all of them)
    struct Attaching {
        DeviceTraces* t; int d; bool open;
        void close() { if (open) { open = false; t->attached(d); } }
        ~Attaching() { close(); }
    } attaching{traces, device, traces != nullptr && !builds};
    inputDevice() = device;
    rng_t RNG(job.seed, device, chainId);
    Transitions<DirichletVector> A(job.nrStates, RNG);
    Initial<Dirichlet> pi(job.nrStates, RNG);
    TransitionHyperParam<DirichletParamVector> tau_A(job.nrStates, job.trans, job.selfTrans);
    InitialHyperParam<DirichletParam> tau_pi(job.nrStates, job.initialAlpha);
    Mapping mapping(job.nrDataDim, job.thetaParams.size(), combinations);

    const string prefix = index == 0 ? job.opref : job.opref + "chain" + std::to_string(index) + ".";
    Records records(job.T, prefix, job.osuff, job.nrStates);
    auto wants = [&](const char* o) { return job.outputs.at(o); };
    records.setRecordStateSequence(wants("sequences"), job.overwrite);
    records.setRecordTheta(wants("parameters"), job.overwrite);
    records.setRecordBlocks(wants("blocks"), job.overwrite);
    records.setRecordCompression(wants("compression"), job.overwrite);
    records.setRecordSegments(wants("segments"), job.overwrite);
    if (index == 0) {
        records.setRecordMarginals(wants("marginals"), job.overwrite);
        records.setRecordMaxSegmentation(wants("maxsegmentation"), job.overwrite);
    } else {
        records.setRecordMarginals(false);
        records.setAccumulateMarginals(wants("marginals") || wants("maxsegmentation"));   // for the pool
    }

    typedef Statistics<IntegralArray, Normal> S;
    typedef Blocks<BreakpointArray> B;
    // upload + maxlet transform + weights + integral array (GPU); a lone chain takes the vector, several share it
    std::unique_ptr<S> iaHolder;
    if (traces && !builds) {
        iaHolder.reset(new S(traces->waitForSource(device), job.T, job.nrDataDim, S::attachInput));
        attaching.close();
    } else {
        try {
            iaHolder.reset(steal ? new S(inputValues, job.nrDataDim) : new S(static_cast<const vector<real_t>&>(inputValues), job.nrDataDim, S::keepInput));
        } catch (...) {
            if (traces) traces->fail(device);
            throw;
        }
    }
    S& ia = *iaHolder;
    B waveletBlocks(ia);
    if (!(traces && !builds)) {
        try {
            if (job.weightMultiplier != 1) waveletBlocks.scaleWeights(job.weightMultiplier);   // (the attached chains find the weights scaled)
        } catch (...) {
            if (traces) traces->fail(device);
            throw;
        }
        if (traces) traces->publish(device, RNG.ctx());
    }
    Emissions<S, B> y(ia, waveletBlocks);
    records.attach(y.ctx());

    vector<vector<real_t>> thetaParams = job.thetaParams;
    const double stdEstimate = ia.noiseEstimate();
    thetaParams[0] = autoPrior(thetaParams[0][0], thetaParams[0][1], y, stdEstimate);
    for (auto& p : thetaParams) p = thetaParams[0];
    ThetaHyperParam<NormalInverseGammaParam> tau_theta(thetaParams);
    Theta<NormalInverseGamma> theta(tau_theta, tau_A, tau_pi, job.useSelfTrans, RNG);

    // the scheme (reference main.cpp:383-452): a pending prior draw happens when the next token starts, whatever it is
    bool samplePrior = true, dynamic = true;
    if (verbose) cout << "Setting block structure to dynamic" << endl << flush;
    for (const Step& st : job.scheme) {
        if (samplePrior) {
            if (verbose) cout << "Sampling prior" << endl << flush;
            hml_check(hml_sample_prior(RNG.ctx()));
            samplePrior = false;
        }
        if (st.method == "P") { samplePrior = true; continue; }
        if (st.method == "S") {
            if (verbose) cout << "Setting block structure to static" << endl << flush;
            y.createBlocks(theta);
            dynamic = false;
            continue;
        }
        if (st.method == "D") {
            if (verbose) cout << "Setting block structure to dynamic" << endl << flush;
            dynamic = true;   // (sampleHMM switches the device back to per-sweep recompression)
            continue;
        }
        if (st.incomplete) throw std::runtime_error("Incomplete command line for -i!");
        if (st.method == "F") {
            if (verbose) cout << "Sampling Forward-Backward" << endl << flush;
            StateSequence<ForwardBackward> q(RNG);
            sampleHMM(y, q, theta, tau_theta, A, tau_A, pi, tau_pi, mapping, st.iterations, st.thinning, records, dynamic, job.useSelfTrans);
        } else if (st.method == "M") {
            if (verbose) cout << "Sampling mixture" << endl << flush;
            StateSequence<Mixture> q(RNG);
            sampleHMM(y, q, theta, tau_theta, A, tau_A, pi, tau_pi, mapping, st.iterations, st.thinning, records, dynamic, job.useSelfTrans);
        } else {
            throw std::runtime_error("Unknown sampling type " + st.method + "!");
        }
    }
    hml_check(hml_sync(RNG.ctx()));
    if (rendezvous && !rendezvous->arrive(index, RNG.ctx())) records.discardMarginals();   // pooling failed elsewhere
    records.close();
}

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
        args.registerFlags({"-chains"}, "1");
        args.registerFlags({"-compat"});
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
        // -compat: the reference-compatible mode of the library (include/hml.h, option "compat"): the reference's own
        // std::mt19937 stream, libm arithmetic and summation orders, so that -R SEED writes the reference's files
        if (args.isSet("-compat")) setenv("HML_COMPAT", "1", 1);
        const int device = args.parse<int>("-device");
        const uint32_t chain = args.parse<uint32_t>("-chain");
        const int nrChains = args.parse<int>("-chains");
        if (nrChains < 1) throw std::runtime_error("Number of chains must be positive!");

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

        // ---- sampling scheme, read once (reference main.cpp:364-377 validates the triples before anything runs; an
        // incomplete triple or an unknown method only fails when the loop reaches it, main.cpp:424-451)
        {
            size_t n = 0;
            for (const string& c : args.tokens("-i"))
                if (c != "P" && c != "S" && c != "D") n++;
            if (n % 3 != 0) throw std::runtime_error("Parameters for -i, excluding \"P\", \"S\" and \"D\", must be multiples of 3!");
        }
        vector<Step> scheme;
        {
            const size_t nrTokens = args.nrTokens("-i");
            for (size_t i = 0; i < nrTokens;) {
                Step st;
                st.method = args.parse<string>("-i", i);
                if (st.method == "P" || st.method == "S" || st.method == "D") { i++; }
                else if (i + 2 >= nrTokens) { st.incomplete = true; i = nrTokens; }
                else {
                    // (conversion errors surface here, before the first sweep; the reference parses them when the token is reached)
                    st.iterations = args.parse<size_t>("-i", i + 1);
                    st.thinning = args.parse<size_t>("-i", i + 2);
                    i += 3;
                }
                scheme.push_back(st);
            }
        }

        Job job;
        job.T = T; job.nrDataDim = nrDataDim; job.nrStates = nrStates; job.seed = rng_seed;
        job.opref = opref; job.osuff = osuff; job.overwrite = overwrite;
        job.weightMultiplier = weightMultiplier; job.useSelfTrans = useSelfTrans;
        job.thetaParams = thetaParams; job.trans = trans; job.selfTrans = selfTrans; job.initialAlpha = initialAlpha;
        job.scheme = scheme;
        for (const char* o : {"sequences", "parameters", "blocks", "compression", "marginals", "segments", "maxsegmentation"})
            job.outputs[o] = outputArgs.isSet(o);

        if (nrChains <= 1) {
            // the device context is created once every argument has been parsed and the input has been read
            runChain(job, inputValues, /*steal*/ true, device, chain, /*index*/ 0, verbose, nullptr);
        } else {
            // ---- chain-parallel (extension): chain k on GPU (device + k) mod #GPUs, each driven by its own host thread;
            // nothing is exchanged while sampling; the recorded marginals are pooled by one all-reduce (RCCL) at the end
            int nDev = 1;
            hml_check(hml_device_count(&nDev));
            Rendezvous rv(nrChains);
            DeviceTraces traces;   // chains beyond the first of a device attach to its construction
            for (int k = nDev; k < nrChains; ++k) traces.expect((device + k) % nDev);
            vector<std::thread> threads;
            vector<std::exception_ptr> errors(nrChains);
            for (int k = 0; k < nrChains; ++k)
                threads.emplace_back([&, k] {
                    try {
                        runChain(job, inputValues, /*steal*/ false, (device + k) % nDev, chain + (uint32_t)k, k, verbose && k == 0, &rv, &traces, /*builds*/ k < nDev);
                    } catch (...) {
                        errors[k] = std::current_exception();
                        rv.abandon();
                    }
                });
            // all chains have sampled (or one has failed): pool, then let them write their files
            vector<hml_ctx*> ctxs = rv.waitForAll();
            std::exception_ptr poolError;
            // (nothing to pool when neither the marginals nor their arg-max segmentation were asked for)
            const bool wantsPool = job.outputs.at("marginals") || job.outputs.at("maxsegmentation");
            if ((int)ctxs.size() == nrChains && wantsPool) {
                if (verbose) cout << "Pooling the marginals of " << nrChains << " chains" << endl << flush;
                try {
                    // The pooled files (PREFIXmarginalsSUFFIX, PREFIXmaxsegmentationSUFFIX) use COMMON labels - states by
                    // ascending mean - while every chain's parameters / sequences / segments files keep the chain's own
                    // labels: PREFIX[chainK.]relabelSUFFIX holds, tab-separated, the chain's label of pooled state 0, 1, ...
                    vector<int32_t> perms((size_t)nrChains * job.nrStates);
                    hml_check(hml_allreduce_marginals_perm(ctxs.data(), nrChains, perms.data()));
                    for (int k = 0; k < nrChains; ++k) {
                        const string fn = (k == 0 ? job.opref : job.opref + "chain" + std::to_string(k) + ".") + "relabel" + job.osuff;
                        if (!job.overwrite) { std::ifstream probe(fn); if (probe.good()) throw std::runtime_error("File " + fn + " already exists!"); }
                        std::ofstream out(fn);
                        if (!out) throw std::runtime_error("Cannot open file " + fn + " for writing!");
                        for (size_t j = 0; j < job.nrStates; ++j) out << (j ? "\t" : "") << perms[(size_t)k * job.nrStates + j];
                        out << "\n";
                    }
                } catch (...) { poolError = std::current_exception(); }
            }
            rv.release(poolError == nullptr && (int)ctxs.size() == nrChains);
            for (auto& t : threads) t.join();
            for (auto& e : errors) if (e) std::rethrow_exception(e);
            if (poolError) std::rethrow_exception(poolError);
        }
        if (verbose) cout << "Exit HaMMLET" << endl << flush;
        return 0;
    } catch (std::exception& e) {
        cout << flush;
        cerr << endl << flush << "[ERROR] " << e.what() << endl;
        cerr << "Terminating HaMMLET. The rest is silence." << endl << flush;
        return 1;
    }
}
