// fused.hip -- placeholder until the on-chip kernel lands: reports "unsupported" so that
// LDPC_PATH_AUTO resolves to the flood path and LDPC_PATH_FUSED fails loudly.
#include "fused.h"
namespace ldpc {
struct FusedState {};
bool fused_supported(const ldpc_code &, int, int) { return false; }
const char *fused_why_not(const ldpc_code &, int, int) { return "fused kernel not built"; }
FusedState *fused_create(const ldpc_code &, int, int, int) { set_error(LDPC_EUNSUPPORTED, "fused kernel not built"); return nullptr; }
void fused_destroy(FusedState *s) { delete s; }
int fused_decode(FusedState &, hipStream_t, int, int, const void *, int, uint8_t *, int32_t *, uint8_t *, double *, double *) { return set_error(LDPC_EUNSUPPORTED, "fused kernel not built"); }
int fused_step(FusedState &, hipStream_t, int, const double *, const double *, const double *, double *, double *, uint8_t *) { return set_error(LDPC_EUNSUPPORTED, "fused kernel not built"); }
}  // namespace ldpc
