// Shared by the translation units of libhammlet_hip.so: the thread's last error message and the HIP call checks.
#ifndef HML_HOST_COMMON_HPP
#define HML_HOST_COMMON_HPP

#include <hip/hip_runtime.h>

#include <string>

#include "../../include/hml.h"

// stores the message hml_last_error() returns (hml_capi.hip) and passes the code through
int hml_set_err(int code, const std::string& msg);

#define HIPCHK(call)                                                                                  \
    do {                                                                                              \
        hipError_t e_ = (call);                                                                       \
        if (e_ != hipSuccess)                                                                         \
            return hml_set_err(HML_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));       \
    } while (0)

#define KLAUNCH_CHECK() HIPCHK(hipGetLastError())

#endif
