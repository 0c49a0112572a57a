// d3m_edge_grad.h -- host dispatch of the edge / silhouette gradient (KCU:245-503).
#pragma once
#include "d3m_backward.h"
#include "d3m_launch.h"

namespace d3m {

inline size_t edge_grad_workspace_bytes(int B, int F, int S) {
    (void)B; (void)F; (void)S;
    return 256;
}

template <class FS>
int run_edge_grad(FS fs, PixelMaps m, float* grad_faces, int B, float eps, void* ws, size_t ws_bytes, hipStream_t st,
                  int* last_err) {
    (void)ws; (void)ws_bytes;
    const long n = (long)B * fs.num_faces();
    LAUNCH("k_backward_pixel_map", k_backward_pixel_map<FS>, dim3((unsigned)((n + 255) / 256)), dim3(256), st, fs, m, grad_faces, B, eps);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { *last_err = (int)e; return 3; }
    return 0;
}

}  // namespace d3m
