// k_embed instantiations for table format SCONE_FMT_F32 (see scone_gather_impl.h, scone_embed_wave.h).
#include "scone_embed_wave.h"

namespace scone_gather {
int launch_f32(scone_handle *h, const embed_args &a, int src, int mode, int out_dtype, hipStream_t s) {
  return launch_table_fmt<SCONE_FMT_F32>(h, a, src, mode, out_dtype, s);
}
}  // namespace scone_gather
