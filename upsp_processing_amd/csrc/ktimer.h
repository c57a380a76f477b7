// Optional per-kernel timing with HIP events on the launch stream (bench / profiling).
#ifndef UPSP_KTIMER_H
#define UPSP_KTIMER_H
#include <hip/hip_runtime.h>

namespace upsp {
bool ktimer_on();
void ktimer_begin(const char *name, hipStream_t st);
void ktimer_end(hipStream_t st);

// RAII: brackets the launches issued while it is alive
struct KTimed {
    hipStream_t st;
    bool on;
    KTimed(const char *name, hipStream_t s) : st(s), on(ktimer_on())
    {
        if (on) ktimer_begin(name, st);
    }
    ~KTimed()
    {
        if (on) ktimer_end(st);
    }
};
}  // namespace upsp
#endif
