// encoder.hip -- placeholder while the index path is brought up; replaced by the MFMA encoder.
#include "mx_common.h"
using namespace mx;
struct mx_encoder { int dummy; };
extern "C" {
size_t mx_encoder_weight_bytes(const mx_encoder_cfg *) { return 0; }
int mx_encoder_create(const mx_encoder_cfg *, const void *, size_t, int, mx_encoder **) { return fail(MX_EUNSUPPORTED, "encoder not built yet"); }
void mx_encoder_destroy(mx_encoder *) {}
int mx_encoder_encode(mx_encoder *, const int32_t *, const int32_t *, int, int, float *) { return fail(MX_EUNSUPPORTED, "encoder not built yet"); }
int mx_encoder_encode_device(mx_encoder *, const int32_t *, const int32_t *, int, int, float *) { return fail(MX_EUNSUPPORTED, "encoder not built yet"); }
int mx_encoder_set_profiling(mx_encoder *, int) { return fail(MX_EUNSUPPORTED, "encoder not built yet"); }
int mx_encoder_get_stats(mx_encoder *, mx_encoder_stats *) { return fail(MX_EUNSUPPORTED, "encoder not built yet"); }
int mx_encoder_reset_stats(mx_encoder *) { return fail(MX_EUNSUPPORTED, "encoder not built yet"); }
}
