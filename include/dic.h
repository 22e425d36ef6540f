/* libdic_hip.so - C ABI of the MI355X-native depth-soft captioning hot path.
 *
 * The reference (Kyo-suke-S/Depth_image_captioning_pub) has NO native / FFI boundary: its hot path
 * sits behind plain torch nn.Module objects.  These entry points are what a
 * ctypes binding of that path binds instead; each one cites the reference code it replaces
 * (paths relative to the reference root).  Conventions:
 *   - every pointer is a DEVICE pointer owned by the caller (torch allocates; tensor.data_ptr()),
 *     except `lengths`/`batch_sizes` style small int arrays, which are HOST pointers (the reference
 *     keeps `lengths` as a Python list: depth_models.py:153-154);
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*), no hidden synchronisation;
 *   - return 0 on success, a negative code on failure, never throw; dic_last_error() gives the text;
 *   - the library owns no device memory: scratch comes in as (workspace, workspace_bytes), sized by
 *     the matching *_workspace_bytes() query; one call at a time per workspace.
 *   - all floating point is IEEE fp32 (exact-fp32 MFMA), token ids are int64.
 */
#ifndef DIC_H_
#define DIC_H_
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DIC_L 196      /* 14x14 annotation cells     (Captioning_models/config.py:11) */
#define DIC_D 2048     /* dim_encoder                (config.py:14) */
#define DIC_A 128      /* dim_attention              (config.py:12) */
#define DIC_E 128      /* dim_embedding              (config.py:13) */
#define DIC_H 128      /* dim_hidden                 (config.py:15) */

int dic_version(void);
const char* dic_last_error(void);

/* ---- generic exact-fp32 MFMA contraction (building block; replaces the aten::addmm / aten::mm
 *      calls under every nn.Linear of the path, e.g. attention.py:84-87, depth_models.py:167,189,197)
 *      C[M,N] (+)= act(A*B^T + bias);  a_colk/b_colk = 1 means that operand is stored K-major
 *      (element (i,k) at p[k*ld + i]) instead of row-major (p[i*ld + k]). */
int dic_gemm_f32(int M, int N, int K, const float* A, long long lda, int a_colk, const float* B, long long ldb,
                 int b_colk, float* C, long long ldc, const float* bias, int act, int accumulate, int splitk,
                 float* workspace, size_t workspace_bytes, int force_tile, void* stream);

/* ---- convolution as implicit GEMM, NHWC activations, OHWI weights (replaces aten::conv2d under
 *      Depth_CNN_endoder.features, depth_models.py:19-23,36-47, and torchvision ResNet-152 under
 *      CNNEncoder_Atten.backbone, base_caption_models.py:23-30).  x may be NCHW when in_nchw=1
 *      (first layer: the reference feeds NCHW images).  bn_partial (nullable) receives per-M-tile
 *      column sums / sums of squares [mtiles][2][CO] for train-mode BatchNorm statistics. */
int dic_conv2d_fwd(const float* x, int B, int H, int W, int C, int in_nchw, const float* w_ohwi, const float* bias,
                   int CO, int KH, int KW, int stride, int pad, float* y_nhwc, float* bn_partial, int* mtiles_out,
                   int force_tile, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DIC_H_ */
