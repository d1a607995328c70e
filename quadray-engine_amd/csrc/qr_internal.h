/*
 * qr_internal.h - declarations shared by the host-side translation units of
 * libqrhip (walker, snapshot I/O, C-ABI glue).  Not part of the public ABI.
 */
#ifndef QR_INTERNAL_H
#define QR_INTERNAL_H

#include "qrhip.h"
#include "qr_scene.h"

#include <string>
#include <vector>
#include <cstdint>

/* flatten the rt_SIMD_INFOX graph into a qr_scene.h blob (qr_walker.cpp) */
int qr_flatten_impl(const void *s_inf, const qr_abi_desc *abi,
                    std::vector<uint8_t> &out, std::string &err);

/* thread-local error channel behind qr_last_error() (qr_capi_host.cpp) */
void qr_set_error(const std::string &msg);
int  qr_fail(int status, const std::string &msg);

#endif /* QR_INTERNAL_H */
