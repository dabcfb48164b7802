// hscmp_mfma.h -- MFMA (matrix-core) variants of the correlation kernels (f32, gfx950).
// Placeholder until the MFMA kernels land: reports "unsupported" so that the generic kernels run.
#pragma once

#include "hscmp_kernels.h"

#include <vector>

namespace hscmp {

inline bool mfma_supported(int, int, int) { return false; }
inline void mfma_build_dict_image(const float*, int, int, int, std::vector<float>& out) { out.clear(); }
inline int mfma_launch_corr_init(hipStream_t, const DevParams&, const State<float>&, const float*) { return -1; }
inline int mfma_launch_iterate(hipStream_t, const DevParams&, const State<float>&, const float*) { return -1; }

}  // namespace hscmp
