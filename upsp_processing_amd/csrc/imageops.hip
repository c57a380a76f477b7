// Image stages of the frame loop (register -> patch -> filter).  First cut: the
// stage plumbing only; kernels follow.
#include <hip/hip_runtime.h>

#include "pipeline.h"
#include "upsp_internal.h"

namespace upsp {

struct PatchTables {
    int nclusters = 0;
};
struct FrameScratch {
    int dummy = 0;
};

int patch_tables_create(int, int, int, const int32_t *, const int32_t *, const int32_t *,
                        const int32_t *, const int32_t *, const int32_t *, PatchTables **out)
{
    if (out) *out = nullptr;
    return fail(UPSP_ERR_INVALID, "patching kernels not built yet");
}
void patch_tables_free(PatchTables *t) { delete t; }
int frame_scratch_ensure(FrameScratch **, int, int, int, int, bool, bool)
{
    return fail(UPSP_ERR_INVALID, "registration / patch / filter kernels not built yet");
}
void frame_scratch_free(FrameScratch *s) { delete s; }
int run_frame_stages(FrameScratch *, int, const uint16_t *, int, int64_t, int, int,
                     const upsp_pipeline_opts &, const float *, const PatchTables *, float *, int,
                     const void **, int *, hipStream_t)
{
    return fail(UPSP_ERR_INVALID, "registration / patch / filter kernels not built yet");
}

}  // namespace upsp

using namespace upsp;
extern "C" {
int upsp_register_pixel_u16(const float *, const uint16_t *, int, int, int, double, int,
                            uint16_t *, float *, void *)
{
    return fail(UPSP_ERR_INVALID, "registration kernels not built yet");
}
int upsp_blur_f32(const float *, float *, int, int, int, int, void *)
{
    return fail(UPSP_ERR_INVALID, "filter kernels not built yet");
}
int upsp_patch_f32(float *, int, int, int, const int32_t *, const int32_t *, const int32_t *,
                   const int32_t *, const int32_t *, const int32_t *, void *)
{
    return fail(UPSP_ERR_INVALID, "patch kernels not built yet");
}
}
