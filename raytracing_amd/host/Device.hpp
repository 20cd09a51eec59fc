/* Device.hpp -- the one place the host classes touch the C-ABI (include/rtamd.h).
 * Errors follow the reference's convention for unrecoverable conditions: a message on stderr and
 * exit(EXIT_FAILURE) (R/src/Mesh.cpp:23-24).  There is no CPU path behind these calls. */
#pragma once
#include <cstdio>
#include <cstdlib>
#include "../../include/rtamd.h"

namespace rthost {

inline void require(int rc, const char* what) {
    if (rc != RT_OK) {
        std::fprintf(stderr, "%s failed (%d): %s\n", what, rc, rt_last_error());
        std::exit(EXIT_FAILURE);
    }
}

/* rt_init once per process; the device index comes from RT_DEVICE (default 0). */
inline void ensure_device() {
    static bool done = false;
    if (done) return;
    const char* env = std::getenv("RT_DEVICE");
    require(rt_init(env ? std::atoi(env) : 0), "rt_init");
    done = true;
}

}  // namespace rthost
