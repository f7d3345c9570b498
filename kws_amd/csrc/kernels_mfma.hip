// placeholder until the MFMA-tiled scan lands (next commit)
#include "common.h"
namespace fastgrnn {
bool mfma_supported(const fastgrnn_desc&, int) { return false; }
size_t mfma_forward_ws(const fastgrnn_desc&) { return 0; }
size_t mfma_backward_ws(const fastgrnn_desc&) { return 0; }
int mfma_forward(const fastgrnn_desc&, const fastgrnn_params&, const void*, const void*, void*, void*, void*, void*, hipStream_t) { return FASTGRNN_ERR_UNSUPPORTED; }
int mfma_backward(const fastgrnn_desc&, const fastgrnn_params&, const void*, const void*, const void*, const void*, const void*, const void*, const fastgrnn_grads&, void*, hipStream_t) { return FASTGRNN_ERR_UNSUPPORTED; }
}
