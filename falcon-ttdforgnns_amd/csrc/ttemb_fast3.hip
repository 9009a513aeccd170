// Sorted / grouped MFMA path for 3-core tables (placeholder: not yet enabled).
#include "ttemb_common.h"

namespace ttemb {

bool fast3_supported(const DevShape&) { return false; }
int64_t fast3_workspace_bytes(const DevShape&, int32_t, int64_t, int64_t) { return 0; }
int launch_forward_fast3(const DevShape&, const CorePtrs&, const int64_t*, const int64_t*, int64_t,
                         const int32_t*, float*, void*, int64_t, hipStream_t) {
  return fail(TTEMB_E_UNSUPPORTED, "fast3 path not built");
}
int launch_backward_fast3(const DevShape&, const CorePtrs&, const int64_t*, const int64_t*, int64_t,
                          const int32_t*, const float*, const CorePtrsMut&, void*, int64_t, hipStream_t) {
  return fail(TTEMB_E_UNSUPPORTED, "fast3 path not built");
}

}  // namespace ttemb
