// layout probe for v_mfma_f32_4x4x1_16b_f32 (dev only): which lane supplies A_b[i][0] / B_b[0][j], where D_b[i][j] lands
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void k(float* o, int mode) {
    const int l = threadIdx.x;
    const float a = mode == 0 ? (float)(l + 1) : 1.0f, b = mode == 0 ? 1.0f : (float)(l + 1);
    v4f c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; r++) o[l * 4 + r] = c[r];
}
int main() {
    float* d; (void)hipMalloc(&d, 2048); float h[2][256];
    for (int m = 0; m < 2; m++) { k<<<1, 64>>>(d, m); (void)hipMemcpy(h[m], d, 1024, hipMemcpyDeviceToHost); }
    int bad = 0;
    for (int l = 0; l < 64; l++) for (int r = 0; r < 4; r++) {
        const int la = (int)h[0][l * 4 + r] - 1, lb = (int)h[1][l * 4 + r] - 1;
        // expectation: D reg r of lane l = A from lane 4*(l/4) + r  times  B from lane l
        if (la != 4 * (l / 4) + r || lb != l) bad++;
        if (l < 6 || l > 61) printf("lane %d reg %d: a from lane %d, b from lane %d\n", l, r, la, lb);
    }
    printf("expected layout (A row i = lane%%4 of block lane/4; D[i = reg][j = lane%%4]): %s (%d mismatches)\n", bad ? "NO" : "YES", bad);
    return 0;
}
