// Context, error reporting and ABI version of libcr3dod.so.
#include "cr_common.h"
#include <string.h>
#include <stdlib.h>

static thread_local char g_err[512] = "";

void cr_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* cr_last_error(void) { return g_err; }
extern "C" int cr_abi_version(void) { return 5; }   // 5: z_type of cr_cube_select(_bwd) / cr_cube_decode_infer; 2: act_f32 argument on the activation entry points; 3: split mode (act_f32 = 2, w_split); 4: cr_roi_align_bwd_set, iou_boxes of cr_cubes_project_score

extern "C" int cr_ctx_create(int device, void* hip_stream, cr_ctx** out) {
    CR_CHECK_ARG(out != nullptr, "cr_ctx_create: out is NULL");
    CR_HIP(hipSetDevice(device));
    cr_ctx* c = (cr_ctx*)calloc(1, sizeof(cr_ctx));
    if (!c) { cr_set_error("cr_ctx_create: host alloc failed"); return CR_ENOMEM; }
    c->device = device;
    c->stream = (hipStream_t)hip_stream;
    c->ws_bytes = 128u << 20;
    hipError_t e = hipMalloc(&c->ws, c->ws_bytes);
    if (e != hipSuccess) {
        cr_set_error("cr_ctx_create: hipMalloc workspace failed: %s", hipGetErrorString(e));
        free(c);
        return CR_ENOMEM;
    }
    *out = c;
    return CR_OK;
}

extern "C" int cr_ctx_destroy(cr_ctx* ctx) {
    if (!ctx) return CR_OK;
    if (ctx->ws) (void)hipFree(ctx->ws);
    free(ctx);
    return CR_OK;
}

extern "C" int cr_ctx_set_stream(cr_ctx* ctx, void* hip_stream) {
    CR_CHECK_ARG(ctx != nullptr, "cr_ctx_set_stream: ctx is NULL");
    ctx->stream = (hipStream_t)hip_stream;
    return CR_OK;
}
