// placeholder, replaced below
#include "sfm_ctx.h"
extern "C" {
int sfm_gpu_ransac(sfm_gpu_ctx* ctx, const double*, const double*, int, const double*, int, double, double, int, int, const int32_t*, int, uint8_t*, double*, int32_t*, int32_t*, int32_t*)
{ return ctx ? ctx->fail(SFM_GPU_ERR_UNSUPPORTED, "not built yet") : SFM_GPU_ERR_ARG; }
int sfm_gpu_match_pairs(sfm_gpu_ctx* ctx, sfm_gpu_image* const*, const int32_t*, int, const sfm_gpu_match_params*, int32_t*, int32_t*, sfm_gpu_dmatch*, int)
{ return ctx ? ctx->fail(SFM_GPU_ERR_UNSUPPORTED, "not built yet") : SFM_GPU_ERR_ARG; }
}
