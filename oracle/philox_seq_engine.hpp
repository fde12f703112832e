// TEST INFRASTRUCTURE - a sequential 32-bit uniform random bit generator whose n-th output is word n of Philox sub-stream
// (kind 0).  Shared by the CPU checker (hml_oracle.hpp, mode RNG_PHILOX_SEQ) and by shim.hpp, which puts it in the place
// of std::mt19937 (reference src/Distribution.hpp:15) when the UNMODIFIED reference is compiled for the Philox goldens.
#ifndef HML_PHILOX_SEQ_ENGINE_HPP
#define HML_PHILOX_SEQ_ENGINE_HPP

#include <cstdint>

#include "../hammlet_amd/csrc/hml_philox.h"

struct PhiloxSeqEngine {
    typedef uint32_t result_type;
    hml_key key;
    uint64_t n;
    hml_u32x4 buf;
    explicit PhiloxSeqEngine(uint64_t seed = 0) : key(hml_make_key(seed, 0)), n(0) {}
    static constexpr result_type min() { return 0; }
    static constexpr result_type max() { return 0xffffffffu; }
    result_type operator()() {
        if ((n & 3) == 0) buf = hml_philox4x32_10((uint32_t)(n >> 2), (uint32_t)(n >> 34), 0, 0, key.k0, key.k1);
        return buf.v[n++ & 3];
    }
};

#endif
