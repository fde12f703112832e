// libhammlet_hip.so, text reader translation unit: hml_text_* of include/hml.h (kernels hml_k_text.h, host side
// hml_text_reader.hpp).
#include <algorithm>
#include <cstring>
#include <vector>

#include "hml_host_common.hpp"

static int set_err(int code, const std::string& msg) { return hml_set_err(code, msg); }

#include "hml_text_reader.hpp"
