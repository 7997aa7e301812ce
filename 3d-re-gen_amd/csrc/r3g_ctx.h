// r3g_ctx.h -- per-device context behind the opaque r3g_ctx of include/r3g.h
#ifndef R3G_CTX_H
#define R3G_CTX_H
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "kernels.h"
#include "mc_kernels.h"

namespace r3g {

int fail(int code, const char* fmt, ...);
int hip_fail(hipError_t e, const char* what);

struct Ctx {
    int device = 0;
    int num_cu = 0;
    char* h_small = nullptr;  // pinned, 64 B: read-back of tiny device results
    // marching cubes
    char* mc_ws = nullptr;
    size_t mc_ws_bytes = 0;
    McWorkspaceLayout mc_lay{};
    const float* mc_grid = nullptr;
    int mc_n[3] = {0, 0, 0};
    double mc_level = 0.0;
    bool mc_counted = false;
    // mesh cleaners
    char* mesh_ws = nullptr;
    size_t mesh_ws_bytes = 0;
    // texture stage (z-buffer / inpainting workspace)
    char* tex_ws = nullptr;
    size_t tex_ws_bytes = 0;

    int reserve(char** buf, size_t* have, size_t need, const char* what);
    void* model = nullptr;  // r3g::Model (model.cpp)
    void release_model();
    void* unet = nullptr;   // r3g::Unet (unet.cpp)
    void release_unet();
};

}  // namespace r3g
#endif
