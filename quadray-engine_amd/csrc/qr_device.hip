/* placeholder: replaced by the real backend once the oracle is pinned */
#include "qr_internal.h"
#include <hip/hip_runtime.h>

extern "C" int qr_render0(const void *s_inf, const qr_abi_desc *abi)
{
    (void)s_inf; (void)abi;
    return qr_fail(QR_ERR_DEVICE, "qr_render0: HIP backend not built yet");
}
