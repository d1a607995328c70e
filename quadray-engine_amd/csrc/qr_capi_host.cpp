/*
 * qr_capi_host.cpp - the part of the C ABI (include/qrhip.h) that needs no GPU:
 * error channel, version, graph flattening and snapshot capture.
 */
#include "qr_internal.h"

#include <algorithm>
#include <cstdio>
#include <ctime>
#include <cstdlib>
#include <cstring>

static thread_local std::string g_qr_error;

void qr_set_error(const std::string &msg) { g_qr_error = msg; }

int qr_fail(int status, const std::string &msg)
{
    g_qr_error = msg;
    return status;
}

extern "C" const char *qr_last_error(void) { return g_qr_error.c_str(); }

#ifndef QR_SRC_HASH
#define QR_SRC_HASH "unknown"
#endif
/* "... src <12 hex digits>": fingerprint of the sources this library was built from (csrc/Makefile SRC_HASH) */
extern "C" const char *qr_version(void) { return "qrhip 0.3 (gfx950) src " QR_SRC_HASH; }

extern "C" int qr_flatten(const void *s_inf, const qr_abi_desc *abi, void **blob, uint64_t *size)
{
    if (blob == nullptr || size == nullptr) return qr_fail(QR_ERR_ARG, "null output argument");
    std::vector<uint8_t> out;
    std::string err;
    int rc = qr_flatten_impl(s_inf, abi, out, err);
    if (rc != QR_OK) return qr_fail(rc, err);
    void *p = malloc(out.size());
    if (p == nullptr) return qr_fail(QR_ERR_NOMEM, "out of memory");
    memcpy(p, out.data(), out.size());
    *blob = p;
    *size = out.size();
    return QR_OK;
}

extern "C" void qr_free(void *blob) { free(blob); }

/* FNV-1a 64 over (pixel & 0xFFFFFF) as 4 little-endian bytes, row-major: the frame fingerprint
 * tests/golden/manifest.json records for every reference frame */
extern "C" uint64_t qr_frame_hash(const uint32_t *frame, uint64_t n_pixels)
{
    uint64_t h = 0xcbf29ce484222325ull;
    if (frame == nullptr) return h;
    for (uint64_t i = 0; i < n_pixels; i++)
    {
        const uint32_t v = frame[i] & 0x00FFFFFFu;
        for (int b = 0; b < 4; b++) { h ^= (v >> (8 * b)) & 0xFFu; h *= 0x100000001b3ull; }
    }
    return h;
}

static thread_local QrFlattenMap g_capture_map;

extern "C" int qr_capture_snapshot(const void *s_inf, const qr_abi_desc *abi, const char *path)
{
    if (path == nullptr) return qr_fail(QR_ERR_ARG, "null path");
    std::vector<uint8_t> out;
    std::string err;
    int rc = qr_flatten_impl(s_inf, abi, out, err, &g_capture_map);
    if (rc != QR_OK) return qr_fail(rc, err);
    if (const char *reps = getenv("QR_FLATTEN_TIMING"))     /* development aid: the flattener's time on this scene, host only */
    {
        std::vector<double> t;
        std::vector<uint8_t> again;
        for (int i = 0, n = atoi(reps); i < n; i++)
        {
            struct timespec a, b;
            clock_gettime(CLOCK_MONOTONIC, &a);
            qr_flatten_impl(s_inf, abi, again, err);
            clock_gettime(CLOCK_MONOTONIC, &b);
            t.push_back((double)(b.tv_sec - a.tv_sec) * 1e3 + (double)(b.tv_nsec - a.tv_nsec) * 1e-6);
        }
        std::sort(t.begin(), t.end());
        if (!t.empty()) fprintf(stderr, "flatten: median %.3f ms, min %.3f ms over %zu calls, %zu bytes, %s\n", t[t.size() / 2], t[0], t.size(),
                                again.size(), again == out ? "same bytes" : "DIFFERENT bytes");
    }
    FILE *f = fopen(path, "wb");
    if (f == nullptr) return qr_fail(QR_ERR_IO, std::string("cannot open ") + path);
    size_t n = fwrite(out.data(), 1, out.size(), f);
    int rc2 = fclose(f);
    if (n != out.size() || rc2 != 0) return qr_fail(QR_ERR_IO, std::string("short write to ") + path);
    return QR_OK;
}

extern "C" int qr_capture_index(int kind, const void *record)
{
    const std::vector<uint64_t> &v = kind == 0 ? g_capture_map.srf : g_capture_map.lgt;
    const uint64_t key = (uint64_t)(uintptr_t)record;
    if (key == 0 || (kind != 0 && kind != 1)) return -1;
    for (size_t i = 0; i < v.size(); i++) if (v[i] == key) return (int)i;
    return -1;
}
