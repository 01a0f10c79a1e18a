// f16 flavour of the tiled MFMA GEMM (storage type of the SAM image encoder in the perf build); templates in gemm_impl.h.
#include "gemm_impl.h"

namespace anyref {

template void launch_gemm<f16>(const GemmArgs&, hipStream_t);

}  // namespace anyref
