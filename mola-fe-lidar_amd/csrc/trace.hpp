// trace.hpp -- named ranges around the stages of an align for rocprofv3 --marker-trace (ROCTx), the counterpart of the
// reference's profiler entries around its ICP calls (`ProfilerEntry tle(profiler_, "run_one_icp")`,
// src/LidarOdometry.cpp:858; "doProcessNewObservation.2.icp_latest", cpp:296-297).  Off unless MOLA_ICP_ROCTX is set:
// the marker library (librocprofiler-sdk-roctx.so, else libroctx64.so) is then loaded with dlopen on first use; if it is
// not there the ranges are no-ops.  Host-side only; nothing here touches a stream.
#pragma once
#include <dlfcn.h>

#include <cstdlib>

namespace mola_icp_amd {

struct Roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    static const Roctx& get()
    {
        static const Roctx r = []() {
            Roctx x;
            if (!std::getenv("MOLA_ICP_ROCTX")) return x;
            for (const char* name : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
                if (void* h = dlopen(name, RTLD_NOW | RTLD_LOCAL)) {
                    x.push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
                    x.pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
                    if (x.push && x.pop) break;
                    x.push = nullptr; x.pop = nullptr;
                }
            }
            return x;
        }();
        return r;
    }
};

struct TraceRange {
    bool on;
    explicit TraceRange(const char* name) : on(Roctx::get().push != nullptr)
    {
        if (on) (void)Roctx::get().push(name);
    }
    ~TraceRange()
    {
        if (on) (void)Roctx::get().pop();
    }
    TraceRange(const TraceRange&) = delete;
    TraceRange& operator=(const TraceRange&) = delete;
};

}  // namespace mola_icp_amd
