// bf16-plane instances of the register-staged GEMM (gemm_regstage.h): fp32 operands are split into NS bf16 planes on their way
// into LDS and multiplied on the bf16 matrix cores with fp32 accumulation.
//   SUMK_PRECISION_BF16   (NS = 1): x1 = bf16(x), ONE v_mfma_f32_32x32x16_bf16 per 16 k -- plain mixed-precision arithmetic
//                                   (BASELINE config 2, "VASNet train ... bf16"): ~2^-9 relative per product;
//   SUMK_PRECISION_BF16X3 (NS = 2): hi + lo planes, 3 MFMAs, ~2^-16 relative;
//   SUMK_PRECISION_BF16X6 (NS = 3): three planes hold fp32 exactly, 6 MFMAs, fp32-grade.
// A translation unit of its own so that it compiles in parallel with the exact-fp32 instances (gemm_f32.hip).
#include "gemm_regstage.h"

namespace sumk {

template <int NS>
static int launch_planes(GemmLayout layout, GemmEpi epi, const GemmKArgs& ka, int tiles, int cfg, hipStream_t stream) {
  if (cfg == 1) return launch_layout<64, 64, 32, NS>(layout, epi, ka, tiles, stream);
  if (cfg == 2) return launch_layout<128, 64, 32, NS>(layout, epi, ka, tiles, stream);
  return launch_layout<128, 128, 32, NS>(layout, epi, ka, tiles, stream);
}

int launch_gemm_split(int precision, GemmLayout layout, GemmEpi epi, const GemmKArgs& ka, int tiles, int cfg, hipStream_t stream) {
  switch (precision) {
    case SUMK_PRECISION_BF16: return launch_planes<1>(layout, epi, ka, tiles, cfg, stream);
    case SUMK_PRECISION_BF16X3: return launch_planes<2>(layout, epi, ka, tiles, cfg, stream);
    case SUMK_PRECISION_BF16X6: return launch_planes<3>(layout, epi, ka, tiles, cfg, stream);
    default: set_error("gemm: unknown precision %d", precision); return SUMK_ERR_ARG;
  }
}

}  // namespace sumk
