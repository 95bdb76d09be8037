/*
 * dce_oracle.h -- declarations of the CPU oracle (oracle/dce_oracle.c).
 * TEST INFRASTRUCTURE ONLY: included by tests (tests/c/abi_client.c) and by nothing in the product.
 */
#ifndef DCE_ORACLE_H
#define DCE_ORACLE_H
#include <stdint.h>

#define WIN   150
#define CH    54
#define NCLS  16
#define FEAT  4736

typedef struct {
    const float *c1w, *c1b;   /* block1.0  (64,54,3)  (64)   src/contact_cnn.py:11-15 */
    const float *c2w, *c2b;   /* block1.2  (64,64,3)  (64)   src/contact_cnn.py:17-21 */
    const float *c3w, *c3b;   /* block2.0  (128,64,3) (128)  src/contact_cnn.py:29-33 */
    const float *c4w, *c4b;   /* block2.2  (128,128,3)(128)  src/contact_cnn.py:35-39 */
    const float *f1w, *f1b;   /* fc.0      (2048,4736)(2048) src/contact_cnn.py:48-49 */
    const float *f2w, *f2b;   /* fc.3      (512,2048) (512)  src/contact_cnn.py:52-53 */
    const float *f3w, *f3b;   /* fc.6      (16,512)   (16)   src/contact_cnn.py:56-57 */
} oracle_weights;

#ifdef __cplusplus
extern "C" {
#endif
void    oracle_zscore_window(const float* rows, int64_t stride, float* out);
int32_t oracle_argmax16(const float* lg);
void    oracle_decimal2binary(int32_t cls, uint8_t out[4]);
int     oracle_forward_windows(const oracle_weights* W, const float* windows, int64_t n,
                               float* feat, float* h1, float* h2,
                               float* logits, int32_t* pred, uint8_t* contacts);
float   oracle_bf16_round(float x);
void    oracle_bf16_round_array(const float* in, int64_t n, float* out);
int     oracle_linear_rows(const float* x, int64_t rows, int in, const float* w, const float* b, int out, int relu, float* y);
int     oracle_linear_rows_tree(const float* x, int64_t rows, int in, const float* w, const float* b, int out, int relu, float* y);
int     oracle_forward_windows_bf16fc(const oracle_weights* W, const float* windows, int64_t n,
                                      float* feat, float* h1, float* h2,
                                      float* logits, int32_t* pred, uint8_t* contacts);
int     oracle_infer_sequence(const oracle_weights* W, const float* seq, int64_t T,
                              float* windows_out, float* logits, int32_t* pred, uint8_t* contacts);
int     oracle_layer_taps(const oracle_weights* W, const float* win,
                          float* conv1, float* conv2, float* pool1,
                          float* conv3, float* conv4, float* pool2);
#ifdef __cplusplus
}
#endif
#endif /* DCE_ORACLE_H */
