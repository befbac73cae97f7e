// rat_common.hip — error reporting, ABI version, slab reduction shared by the backward kernels.
#include "rat_device.h"
#include "../../include/rat_hip.h"

static thread_local std::string g_last_error;

const char* rat_set_error(const std::string& msg) {
    g_last_error = msg;
    return g_last_error.c_str();
}
int rat_fail(const std::string& msg) {
    rat_set_error(msg);
    return -1;
}
int rat_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return rat_fail(std::string(what) + ": " + hipGetErrorString(e));
    return 0;
}

extern "C" int rat_version(void) { return RAT_ABI_VERSION; }
extern "C" const char* rat_last_error(void) { return g_last_error.c_str(); }

// out[p] = sum over slabs (fixed order => bitwise reproducible for a fixed grid)
__global__ void rat_reduce_slabs_kernel(const float* slabs, int nslabs, int64_t stride, float* out, int64_t n) {
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int w = 0; w < nslabs; ++w) s += slabs[(int64_t)w * stride + p];
        out[p] = s;
    }
}

int rat_launch_reduce_slabs(const float* slabs, int nslabs, int64_t stride, float* const* outs_host,
                            const int64_t* offsets, const int64_t* sizes, int nouts, void* stream) {
    for (int i = 0; i < nouts; ++i) {
        if (!outs_host[i] || sizes[i] <= 0) continue;
        int blocks = (int)((sizes[i] + 255) / 256);
        if (blocks > 1024) blocks = 1024;
        RAT_LAUNCH(rat_reduce_slabs_kernel, blocks, 256, 0, stream, slabs + offsets[i], nslabs, stride, outs_host[i],
                   sizes[i]);
    }
    return rat_check_launch("rat_reduce_slabs");
}
