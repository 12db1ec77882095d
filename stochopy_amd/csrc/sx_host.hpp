// Host-side helpers: error reporting and launch checks for the C ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <string>

namespace sx {

void set_error(const std::string &msg);

inline int hip_fail(hipError_t e, const char *what, const char *file, int line) {
    char buf[512];
    std::snprintf(buf, sizeof buf, "%s failed: %s (%s:%d)", what, hipGetErrorString(e), file, line);
    set_error(buf);
    return -(int)e - 1000;
}

}  // namespace sx

// an instantiated hipGraph of generations (opaque in the C ABI)
struct sx_graph {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    void *scratch = nullptr;  // device memory the graph's nodes own (freed with the graph)
};

#define SX_HIP(call)                                                          \
    do {                                                                      \
        hipError_t e__ = (call);                                              \
        if (e__ != hipSuccess) return sx::hip_fail(e__, #call, __FILE__, __LINE__); \
    } while (0)

#define SX_LAUNCH_CHECK() SX_HIP(hipGetLastError())

#define SX_REQUIRE(cond, msg)       \
    do {                            \
        if (!(cond)) {              \
            sx::set_error(msg);     \
            return -1;              \
        }                           \
    } while (0)
