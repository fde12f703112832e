// reference_shaped_driver - a driver that builds and runs the sampler with the CALL SHAPES of the reference's
// src/main.cpp (the constructor signatures of lines 107-108, 152-166, 266-362 and the scheme loop of lines 383-452),
// compiled unedited against include/hammlet.  It is the text of INTEGRATION.md section B; the suite compiles it on the
// CPU and runs it on the GPU next to the `hammlet` driver, whose output files it must reproduce.
//
//   reference_shaped_driver INPUT PREFIX SUFFIX SEED STATES SELFTRANS(0|1) MULTIPLIER TOKEN...
//
// Scheme tokens are the reference's (M n t | F n t | S | D | P) plus lower-case m / f, which run the same sweeps
// through a loop written by hand in the shape of sampleHMM (src/HMM.hpp:99-121): StateSequence::sample with its eleven
// arguments, then theta.sample, pi.sample, A.sample, then records.record(theta).
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "hammlet/hammlet.hpp"

using namespace hammlet;
using namespace std;

template <typename Q, typename E, typename TH, typename TTH, typename TA, typename TTA, typename TP, typename TTP>
void handWrittenLoop(E& y, Q& q, TH& theta, TTH& tau_theta, TA& A, TTA& tau_A, TP& pi, TTP& tau_pi, const Mapping& mapping,
                     const size_t iterations, const size_t thinning, Records& records, const bool dynamic, const bool useSelfTransitions) {
    for (size_t i = 0; i < iterations; ++i) {
        if (dynamic) y.createBlocks(theta);
        bool doRecord = false;
        if (thinning > 0) doRecord = ((i + 1) % thinning == 0);
        q.sample(y, theta, tau_theta, A, tau_A, pi, tau_pi, mapping, records, doRecord, useSelfTransitions);
        theta.sample(tau_theta);
        pi.sample(tau_pi);
        A.sample(tau_A);
        if (doRecord) records.record(theta);
    }
}

int main(int argc, const char* argv[]) {
    try {
        if (argc < 9) throw runtime_error("usage: reference_shaped_driver INPUT PREFIX SUFFIX SEED STATES SELFTRANS MULTIPLIER TOKEN...");
        const string inputFile = argv[1], outputPrefix = argv[2], outputSuffix = argv[3];
        const size_t rng_seed = strtoull(argv[4], nullptr, 10);
        const size_t nrParams = strtoull(argv[5], nullptr, 10);
        const bool useSelfTrans = atoi(argv[6]) != 0;
        const real_t weightMultiplier = (real_t)atof(argv[7]);
        vector<string> tokens(argv + 8, argv + argc);

        rng_t RNG(rng_seed);

        const size_t nrDataDim = 1;
        const MappingType mappingType = combinations;
        Mapping mapping(nrDataDim, nrParams, mappingType);
        const size_t nrStates = mapping.nrStates();

        Transitions<DirichletVector> A(nrStates, RNG);
        TransitionHyperParam<DirichletParamVector> tau_A(nrStates, 0.5, 0.5);
        Initial<Dirichlet> pi(nrStates, RNG);
        InitialHyperParam<DirichletParam> tau_pi(nrStates, 0.5);

        vector<vector<real_t>> thetaParams(nrParams, vector<real_t>{0.2f, 0.9f});

        vector<real_t> inputValues;
        vector<SufficientStatistics<Normal>> stats;
        ifstream fin(inputFile);
        if (!fin) throw runtime_error("Cannot read from input file " + inputFile + "!");
        MaxletTransform(fin, inputValues, stats, nrDataDim);
        const size_t T = inputValues.size();

        // noise estimate from the finest detail coefficients, on the host like the reference's driver
        double stdEstimate = 0;
        size_t nrDetailCoeffs = 0;
        for (size_t i = 1; i < inputValues.size(); i += 2) {
            stdEstimate += inputValues[i];
            nrDetailCoeffs++;
        }
        stdEstimate /= nrDetailCoeffs;
        stdEstimate /= 0.797884560802865355879892119868763736951717262329869315331;

        HaarBreakpointWeights(inputValues);

        Records records(T, outputPrefix, outputSuffix, nrStates);
        records.setRecordStateSequence(true, true);
        records.setRecordTheta(true, true);
        records.setRecordBlocks(true, true);
        records.setRecordCompression(true, true);
        records.setRecordMarginals(true, true);

        for (auto& w : inputValues) w *= weightMultiplier;

        typedef Statistics<IntegralArray, Normal> S;
        typedef Blocks<BreakpointArray> B;
        S ia(stats, nrDataDim);
        B waveletBlocks(inputValues);
        Emissions<S, B> y(ia, waveletBlocks);

        thetaParams[0] = autoPrior(thetaParams[0][0], thetaParams[0][1], y, stdEstimate);
        for (auto& param : thetaParams) param = thetaParams[0];
        ThetaHyperParam<NormalInverseGammaParam> tau_theta(thetaParams);
        Theta<NormalInverseGamma> theta(tau_theta, nrDataDim, mappingType, RNG);

        bool samplePrior = true;
        bool dynamic = true;
        for (size_t i = 0; i < tokens.size();) {
            const string method = tokens[i];
            if (samplePrior) {
                theta.sample(tau_theta);
                pi.sample(tau_pi);
                A.sample(tau_A);
                samplePrior = false;
            }
            if (method == "P") { samplePrior = true; i++; continue; }
            if (method == "S") { y.createBlocks(theta); dynamic = false; i++; continue; }
            if (method == "D") { dynamic = true; i++; continue; }
            if (i + 2 >= tokens.size()) throw runtime_error("Incomplete command line for -i!");
            const size_t iterations = strtoull(tokens[i + 1].c_str(), nullptr, 10);
            const size_t thinning = strtoull(tokens[i + 2].c_str(), nullptr, 10);
            i += 3;
            if (method == "F") {
                StateSequence<ForwardBackward> q(RNG);
                sampleHMM(y, q, theta, tau_theta, A, tau_A, pi, tau_pi, mapping, iterations, thinning, records, dynamic, useSelfTrans);
            } else if (method == "M") {
                StateSequence<Mixture> q(RNG);
                sampleHMM(y, q, theta, tau_theta, A, tau_A, pi, tau_pi, mapping, iterations, thinning, records, dynamic, useSelfTrans);
            } else if (method == "f") {
                StateSequence<ForwardBackward> q(RNG);
                handWrittenLoop(y, q, theta, tau_theta, A, tau_A, pi, tau_pi, mapping, iterations, thinning, records, dynamic, useSelfTrans);
            } else if (method == "m") {
                StateSequence<Mixture> q(RNG);
                handWrittenLoop(y, q, theta, tau_theta, A, tau_A, pi, tau_pi, mapping, iterations, thinning, records, dynamic, useSelfTrans);
            } else {
                throw runtime_error("Unknown sampling type " + method + "!");
            }
        }
        // the container classes on their own: a trellis row appended and sampled, emission accessors
        Trellis trellis(nrStates, RNG);
        trellis.reserve(2);
        trellis.push_back(vector<real_t>(nrStates, 1.0f));
        vector<real_t> onehot(nrStates, 0.0f);
        onehot[nrStates - 1] = 1.0f;
        trellis.push_back(onehot);
        if (trellis.size() != 2 || trellis.sample(1) != nrStates - 1 || trellis.sample(0) >= nrStates)
            throw runtime_error("Trellis container misbehaves!");
        if (y.nrDim() != nrDataDim || y.size() != T) throw runtime_error("Emissions accessors misbehave!");
        return 0;
    } catch (exception& e) {
        cerr << endl << "[ERROR] " << e.what() << endl;
        cerr << "Terminating HaMMLET. The rest is silence." << endl;
        return 1;
    }
}
