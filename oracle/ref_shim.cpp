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
#include <stdlib.h>
#include <pthread.h>
extern "C" { extern const char *qr_shim_snapshot_path; extern int qr_shim_calls; extern int qr_shim_status; extern double qr_shim_ms; }

/*
 * The engine's frame, written by the copy engines directly (qr_frame_register, include/qrhip.h): the binding registers a
 * frame the first time it sees it -- the frame belongs to the rt_Scene and lives as long as it does (engine.cpp:2829-2858,
 * 3815) -- and keeps at most four registrations; a frame that comes back with another size, or a fifth frame, replaces the
 * oldest.  QR_SHIM_PIN=0 switches it off (frames then go through the library's staging copy).  A maintainer whose
 * application frees scenes while it runs adds qr_frame_unregister(frame) to rt_Scene's destructor.
 * Bottom-up frames (negative stride) and any failure to register fall back to the staging path silently.
 */
static void shim_pin_frame(const rt_SIMD_INFOX *s_inf)
{
    static pthread_mutex_t lock = PTHREAD_MUTEX_INITIALIZER;
    static struct { void *p; unsigned long long bytes; } seen[4];
    static int n_seen = 0, off = -1;
    if (off < 0) { const char *e = getenv("QR_SHIM_PIN"); off = (e != RT_NULL && atoi(e) == 0) ? 1 : 0; }
    if (off || s_inf->frame == RT_NULL || s_inf->frm_row < s_inf->frm_w || s_inf->frm_h <= 0) return;
    void *p = (void *)s_inf->frame;
    const unsigned long long bytes = (unsigned long long)s_inf->frm_row * (unsigned long long)s_inf->frm_h * sizeof(rt_ui32);
    pthread_mutex_lock(&lock);
    int i;
    for (i = 0; i < n_seen; i++) if (seen[i].p == p) break;
    if (i < n_seen && seen[i].bytes != bytes) { qr_frame_unregister(p); seen[i] = seen[--n_seen]; i = n_seen; }
    if (i == n_seen)
    {
        if (n_seen == 4) { qr_frame_unregister(seen[0].p); seen[0] = seen[1]; seen[1] = seen[2]; seen[2] = seen[3]; n_seen = 3; }
        if (qr_frame_register(p, bytes) == QR_OK) { seen[n_seen].p = p; seen[n_seen].bytes = bytes; n_seen++; }
        else off = 1;                                       /* e.g. no device: the call below reports it */
    }
    pthread_mutex_unlock(&lock);
}

namespace simd_128v8
{

rt_void render0(rt_SIMD_INFOX *s_inf)
{
    qr_abi_desc abi = { sizeof(qr_abi_desc), Q, RT_POINTER, RT_ADDRESS, RT_ELEMENT, RT_ENDIAN, {0, 0} };
    struct timespec a, b;
    if (qr_shim_snapshot_path == RT_NULL) shim_pin_frame(s_inf);
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
