/* gsr_cpu.h - the CPU oracle behind the product library's call signatures (TEST INFRASTRUCTURE; see gsr_oracle.hpp header:
 * parity unpinned).  gsr_cpu_forward / gsr_cpu_backward take exactly the argument lists of gsr_forward / gsr_backward
 * (include/gsr.h) with HOST pointers, so a C host can run one call sequence against libgsr_hip.so and against
 * oracle/libgsr_oracle.so and compare.  fp32; GSR_FLAG_SH_PLANAR / COV_3X3 / EXTRA_MODE understood, the
 * scale + quaternion form is not.  `geom`: gsr_cpu_workspace_bytes(dims) bytes of host memory that keep the per-view state
 * between forward and backward (free it with gsr_cpu_release before reusing or dropping the buffer); `bin`: >= sizeof(GsrStatus)
 * bytes, receives num_pairs (the reference's num_rendered, 16x16 tiles) and max_list; `img`, `scratch`, `stream` are ignored. */
#ifndef GSR_CPU_H_
#define GSR_CPU_H_
#include "../include/gsr.h"
#ifdef __cplusplus
extern "C" {
#endif
size_t gsr_cpu_workspace_bytes(const GsrDims* dims);
void gsr_cpu_release(const GsrDims* dims, void* geom);
int gsr_cpu_forward(const GsrDims* dims, const GsrView* views, const float* means, const float* cov, const float* opacities,
                    const float* colors, const float* extra, float* out_color, float* out_extra, int32_t* radii, void* geom,
                    void* bin, void* img, void* stream);
int gsr_cpu_backward(const GsrDims* dims, const GsrView* views, const float* means, const float* cov, const float* opacities,
                     const float* colors, const float* extra, const void* geom, const void* bin, const void* img,
                     const float* dL_dcolor, const float* dL_dextra_img, void* scratch, float* dL_dmeans, float* dL_dcov,
                     float* dL_dopacities, float* dL_dcolors, float* dL_dextra, float* dL_dmeans2D, void* stream);
#ifdef __cplusplus
}
#endif
#endif
