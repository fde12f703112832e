// TEST INFRASTRUCTURE - force-included (`g++ -include oracle/shim.hpp`) in front of the UNMODIFIED reference
// translation unit (/root/reference/src/main.cpp; no reference file is copied or changed): every `mt19937` the
// reference names (src/Distribution.hpp:10,15: `using std::mt19937; typedef mt19937 rng_t;`) becomes the sequential
// Philox engine, so the reference's own consumption pattern - libstdc++'s discrete_distribution (src/Trellis.hpp:61-66),
// gamma_distribution and normal_distribution (src/Distribution.hpp:77-87,116-178) drawing from one shared engine -
// runs on the generator the GPU path uses.  The outputs are the "Philox goldens" (tests/golden/philox_*): they pin the
// generator and its word order against the reference itself, beyond the Random123 known-answer vectors.
#ifndef HML_ORACLE_SHIM_HPP
#define HML_ORACLE_SHIM_HPP

#include <random>

#include "philox_seq_engine.hpp"

namespace std { using ::PhiloxSeqEngine; }
#define mt19937 PhiloxSeqEngine

#endif
