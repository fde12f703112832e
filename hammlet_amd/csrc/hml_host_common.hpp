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

// device memory that is released on every path out of a function (HIPCHK returns early)
struct DevBuf {
    void* p = nullptr;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { if (p) (void)hipFree(p); }
    template <class T> T* as() const { return static_cast<T*>(p); }
    template <class T> T* release() { T* q = static_cast<T*>(p); p = nullptr; return q; }
};

#endif
