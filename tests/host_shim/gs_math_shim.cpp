// Test-only host build of clm_gs_amd/csrc/gs_math.h (the product's per-element
// formulas) so the CPU-only suite can check them against autograd of the
// oracle.  Never loaded by the product.
#include "gs_math.h"
using namespace clmgs;
extern "C" {
void shim_project_fwd(int N, const float* means, const float* quats, const float* scales,
                      const float* viewmat, const float* K, float W, float H, float eps2d,
                      float near_plane, float far_plane, float radius_clip,
                      int* radii, float* means2d, float* depths, float* conics) {
  Cam cam = load_cam(viewmat, K);
  for (int i = 0; i < N; ++i) {
    Proj p = project_fwd(cam, means + 3 * i, quats + 4 * i, scales + 3 * i, W, H, eps2d,
                         near_plane, far_plane, radius_clip);
    radii[i] = p.radius; means2d[2 * i] = p.mx; means2d[2 * i + 1] = p.my; depths[i] = p.depth;
    conics[3 * i] = p.ca; conics[3 * i + 1] = p.cb; conics[3 * i + 2] = p.cc;
  }
}
void shim_project_bwd(int N, const float* means, const float* quats, const float* scales,
                      const float* viewmat, const float* K, float W, float H, float eps2d,
                      const int* radii, const float* v_means2d, const float* v_depths,
                      const float* v_conics, float* v_means, float* v_quats, float* v_scales) {
  Cam cam = load_cam(viewmat, K);
  for (int i = 0; i < N; ++i) {
    float vm[3] = {0, 0, 0}, vq[4] = {0, 0, 0, 0}, vs[3] = {0, 0, 0};
    if (radii[i] > 0)
      project_bwd(cam, means + 3 * i, quats + 4 * i, scales + 3 * i, W, H, eps2d,
                  v_means2d + 2 * i, v_depths[i], v_conics + 3 * i, vm, vq, vs);
    for (int k = 0; k < 3; ++k) { v_means[3 * i + k] = vm[k]; v_scales[3 * i + k] = vs[k]; }
    for (int k = 0; k < 4; ++k) v_quats[4 * i + k] = vq[k];
  }
}
void shim_sh_basis(int N, int deg, const float* dirs, float* B, float* Bx, float* By, float* Bz) {
  for (int i = 0; i < N; ++i) {
    float x = dirs[3 * i], y = dirs[3 * i + 1], z = dirs[3 * i + 2];
    float inv = 1.f / sqrtf(x * x + y * y + z * z);
    x *= inv; y *= inv; z *= inv;
    for (int k = 0; k < 16; ++k) B[16 * i + k] = Bx[16 * i + k] = By[16 * i + k] = Bz[16 * i + k] = 0.f;
    sh_basis(deg, x, y, z, B + 16 * i);
    sh_basis_grad(deg, x, y, z, Bx + 16 * i, By + 16 * i, Bz + 16 * i);
  }
}
}
