// placeholder until the MFMA kernel lands (fails loudly; never computes on the CPU)
#include "amav_common.h"
using namespace amav;
extern "C" int amav_selfattn_forward(int, int, int, int, const float *, const float *, const float *, int64_t, float *,
                                     int64_t, float, void *) {
    return fail(AMAV_ERR_LAUNCH, "amav_selfattn_forward: kernel not built in this revision");
}
