// Split-pair flavour of the tiled MFMA GEMM (ANYREF_MODE_PARITY16: f32 activations carried as two bf16 terms against
// exactly stored bf16 weights, common.h `sp16`); templates in gemm_impl.h, compiled beside gemm.hip / gemm_f16.hip.
#include "gemm_impl.h"

namespace anyref {

template void launch_gemm<sp16>(const GemmArgs&, hipStream_t);

}  // namespace anyref
