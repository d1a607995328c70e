/*
 * ref_shim.cpp - the reference-side forwarding TU of the drop-in boundary.
 *
 * Compiled by oracle/Makefile WITH the reference's own flags and headers and
 * linked INSTEAD OF core/tracer/tracer_128v8.cpp, so that the unmodified
 * reference engine reaches our C ABI through its normal dispatcher
 * (rt_Platform::render0, core/tracer/tracer.cpp:5992-6104, case 0x00000008)
 * whenever the 128x1v8 target is selected (`-n 1 -k 1 -s 8`).
 * This is the ~20 line binding INTEGRATION.md describes; reference sources
 * stay untouched.
 *
 * In this container there is no GPU, so the driver uses the shim in its
 * capture mode (qr_capture_snapshot); on a GPU host the same TU forwards to
 * qr_render0.
 */
#include "tracer.h"     /* rt_SIMD_INFOX, Q, RT_POINTER, ... (reference header) */
#include "system.h"     /* rt_Exception */
#include "qrhip.h"

#include <time.h>
extern "C" { extern const char *qr_shim_snapshot_path; extern int qr_shim_calls; extern int qr_shim_status; extern double qr_shim_ms; }

namespace simd_128v8
{

rt_void render0(rt_SIMD_INFOX *s_inf)
{
    qr_abi_desc abi = { sizeof(qr_abi_desc), Q, RT_POINTER, RT_ADDRESS, RT_ELEMENT, RT_ENDIAN, {0, 0} };
    struct timespec a, b;
    clock_gettime(CLOCK_MONOTONIC, &a);
    int rc = qr_shim_snapshot_path != RT_NULL
           ? qr_capture_snapshot(s_inf, &abi, qr_shim_snapshot_path)
           : qr_render0(s_inf, &abi);
    clock_gettime(CLOCK_MONOTONIC, &b);
    qr_shim_ms += (double)(b.tv_sec - a.tv_sec) * 1e3 + (double)(b.tv_nsec - a.tv_nsec) * 1e-6;      /* time inside the backend call */
    qr_shim_calls++;
    qr_shim_status = rc;
    if (rc != QR_OK)
    {
        throw rt_Exception(qr_last_error());
    }
}

/* side entry points of the namespace (engine.cpp:4100-4118 binds them to
 * simd_128v4 only, so nothing references these for 128v8) */

} /* namespace simd_128v8 */
