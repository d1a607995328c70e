/*
 * qr_scene.h - flattened ("snapshot") form of the data one render0 call reads.
 *
 * The reference backend entry point `render0(rt_SIMD_INFOX*)`
 * (core/tracer/tracer.cpp:1081) receives a pointer graph of lane-broadcast
 * SIMD structures (core/tracer/tracer.h:127-1078).  This header defines the
 * pointer-free, de-broadcast equivalent that
 *   - the walker (csrc/qr_walker.cpp) produces from a live rt_SIMD_INFOX,
 *   - is stored on disk as a snapshot (*.qrs, little-endian, this exact layout),
 *   - the CPU oracle (oracle/qr_oracle.c) and the HIP backend consume.
 *
 * Every pointer of the reference becomes a dense index (-1 == NULL), every
 * lane-broadcast `rt_real x[S]` becomes one float, byte-offset axis maps
 * (core/engine/object.cpp:2489-2497) become axis indices 0..2.
 *
 * Plain C, no dependencies; shared by product code and by the test oracle
 * purely as a data-format definition.
 */
#ifndef QR_SCENE_H
#define QR_SCENE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QR_SNAPSHOT_MAGIC   0x31535251u /* "QRS1" */
#define QR_SNAPSHOT_VERSION 2u

#define QR_NULL (-1)

/* material property bits, same values as RT_PROP_* (tracer.h:61-72) */
#define QR_PROP_LIGHT    0x00000010
#define QR_PROP_METAL    0x00000020
#define QR_PROP_GAMMA    0x00000040
#define QR_PROP_FRESNEL  0x00000080
#define QR_PROP_NORMAL   0x00000100
#define QR_PROP_OPAQUE   0x00000200
#define QR_PROP_TRANSP   0x00000400
#define QR_PROP_TEXTURE  0x00000800
#define QR_PROP_REFLECT  0x00001000
#define QR_PROP_REFRACT  0x00002000
#define QR_PROP_DIFFUSE  0x00004000
#define QR_PROP_SPECULAR 0x00008000

/* surface tags, same values as RT_TAG_* (format.h:116-128) */
#define QR_TAG_ARRAY       (-1)
#define QR_TAG_PLANE         0
#define QR_TAG_SURFACE_MAX   9

/*
 * List element, flattened rt_ELEM (tracer.h:127-141).
 * Meaning of `data` depends on the list the element lives in:
 *   surface lists  : array/bvnode element -> index of the LAST element of its
 *                    sub-list (reference keeps `ptr | type`, engine.cpp:1690),
 *                    `kind` holds the low 2 type bits (1 == bounding volume);
 *                    plain surface -> 0.
 *   clipper lists  : clip side (+1 / -1, tracer.cpp:488-496), accum marker
 *                    (simd == QR_NULL, data == -1 enter / +1 leave,
 *                    tracer.h:79-80), trnode element -> index of last element.
 *   light lists    : index of the head of that light's shadow list
 *                    (engine.cpp:1126), simd == light index.
 */
typedef struct qr_elem
{
    int32_t simd;   /* surface index / light index / QR_NULL            */
    int32_t data;   /* see above                                        */
    int32_t next;   /* next element index or QR_NULL                    */
    int32_t kind;   /* low 2 bits of the reference's data field         */
} qr_elem;

/* flattened rt_SIMD_SURFACE (tracer.h:821-969), 64 x 4 bytes */
typedef struct qr_surface
{
    float    pos[3];        /* srf_POS_*                                 */
    uint32_t c_def;         /* srf_C_DEF, accum default mask             */
    float    min[3];        /* srf_MIN_*                                 */
    uint32_t minmax_t;      /* bits 0-2 min_t[x,y,z]!=0, 3-5 max_t       */
    float    max[3];        /* srf_MAX_*                                 */
    int32_t  conic;         /* msc_p[1]: 0,1 (cone),2 (hypercyl) 5801-07 */
    float    tci[3];        /* transform row i                           */
    int32_t  has_trm;       /* a_map[RT_L]: 0 none,1 scale,2 rot,3 both  */
    float    tcj[3];
    int32_t  shift;         /* a_sgn[RT_L]!=0: use IJK (trnode) fields   */
    float    tck[3];
    uint32_t axes;          /* map_i | map_j<<2 | map_k<<4 |
                               sgn_i<<8 | sgn_j<<9 | sgn_k<<10           */
    float    sci[4];        /* srf_SCI_X,Y,Z,W                           */
    float    scj[3];        /* srf_SCJ_X,Y,Z                             */
    uint32_t smask;         /* srf_SMASK (0x80000000)                    */
    float    d_eps;         /* srf_D_EPS                                 */
    float    t_eps;         /* srf_T_EPS                                 */
    int32_t  srf_t[4];      /* solver, material redirect, clip, tag      */
    int32_t  clip;          /* msc_p[2]: clipper list head               */
    int32_t  trnode;        /* msc_p[3]: trnode surface index            */
    int32_t  mat[2];        /* mat_p[0], mat_p[2]: outer/inner material  */
    int32_t  props[2];      /* mat_p[1], mat_p[3]: outer/inner props     */
    int32_t  lst[4];        /* lst_p[0..3]: lights o, surfaces o, l i, s i */
    int32_t  pad[16];
} qr_surface;

/* flattened rt_SIMD_MATERIAL (tracer.h:979-1078), 32 x 4 bytes */
typedef struct qr_material
{
    float    xscal, yscal, xoffs, yoffs;
    uint32_t xmask, ymask, yshft;
    int32_t  tex;           /* offset of texel (0,0) in the texel pool   */
    int32_t  t_map[2];      /* 0 -> TEX_U, 1 -> TEX_V                    */
    float    l_dff, l_spc;
    uint32_t l_pow;         /* fixed point 28.4                          */
    float    c_rfl, c_trn, c_rfr, rfr_2, c_rcp, ext_2;
    float    clamp;         /* 255.0                                     */
    uint32_t cmask;         /* 255                                       */
    float    emis[3];       /* mat_COL_R/G/B (tracer.h:1065-1071): emission, read by the path tracer only;
                             * the walker fills it for path-tracer snapshots (qr_frame.pt_on), 0 otherwise */
    int32_t  pad[8];
} qr_material;

/* flattened rt_SIMD_LIGHT (tracer.h:765-811), 16 x 4 bytes */
typedef struct qr_light
{
    float t_max;
    float pos[3];
    float col[3];
    float l_src;
    float a_qdr, a_lnr, a_cnt, a_rng;
    int32_t pad[4];
} qr_light;

/*
 * Everything scalar: rt_SIMD_CAMERA (tracer.h:677-755), the primary context
 * fields set by render_slice (engine.cpp:3588-3596) and the external
 * parameters of rt_SIMD_INFOX (tracer.h:154-216).
 */
typedef struct qr_frame
{
    /* camera */
    float    t_max;
    float    dir[3], hor[3], ver[3];
    float    hor_a[4], ver_a[4];    /* FSAA sub-sample offsets, period 4 */
    float    clamp;
    uint32_t cmask;
    float    l_amb;
    float    amb[3];                /* cam_COL_R/G/B                     */
    /* primary context */
    float    t_min;
    float    org[3];
    int32_t  ctx_flags;             /* ctx_PARAM(FLG): RT_PROP_GAMMA or 0 */
    /* info */
    int32_t  depth;
    int32_t  fsaa;                  /* 0 none, 1 2x, 2 4x                */
    int32_t  frm_w, frm_h, frm_row;
    int32_t  tile_w, tile_h, tls_row, tls_col;
    int32_t  clist;                 /* inf_LST head                      */
    int32_t  index, thnum;          /* row interleave of this call       */
    int32_t  pt_on;                 /* inf_PT_ON (tracer.h:216): captured in path-tracer mode */
    int32_t  pad[7];
} qr_frame;

typedef struct qr_header
{
    uint32_t magic;
    uint32_t version;
    uint32_t total_bytes;
    uint32_t header_bytes;
    /* element counts */
    uint32_t n_srf, n_mat, n_lgt, n_elm, n_tiles, n_texels;
    /* byte offsets from the start of the blob, each 16-byte aligned */
    uint32_t off_frame, off_srf, off_mat, off_lgt, off_elm, off_tiles, off_texels;
    /* record sizes, for forward compatibility checks */
    uint32_t sz_frame, sz_srf, sz_mat, sz_lgt, sz_elm;
    uint32_t pad[10];
} qr_header;

/* pointer view over a snapshot blob (no ownership) */
typedef struct qr_scene_view
{
    const qr_header   *hdr;
    const qr_frame    *frame;
    const qr_surface  *srf;
    const qr_material *mat;
    const qr_light    *lgt;
    const qr_elem     *elm;
    const int32_t     *tiles;   /* tls_col x tls_row list heads          */
    const uint32_t    *texels;  /* 0x00RRGGBB                            */
} qr_scene_view;

/* validate a blob and fill the view; returns 0 on success, <0 on error */
static inline int qr_scene_view_init(qr_scene_view *v, const void *blob, uint64_t size)
{
    const uint8_t *p = (const uint8_t *)blob;
    const qr_header *h = (const qr_header *)blob;
    if (size < sizeof(qr_header)) return -1;
    if (h->magic != QR_SNAPSHOT_MAGIC) return -2;
    if (h->version != QR_SNAPSHOT_VERSION) return -3;
    if (h->total_bytes > size) return -4;
    if (h->sz_frame != sizeof(qr_frame) || h->sz_srf != sizeof(qr_surface) ||
        h->sz_mat != sizeof(qr_material) || h->sz_lgt != sizeof(qr_light) ||
        h->sz_elm != sizeof(qr_elem)) return -5;
    if ((uint64_t)h->off_frame + sizeof(qr_frame) > h->total_bytes ||
        (uint64_t)h->off_srf + (uint64_t)h->n_srf * sizeof(qr_surface) > h->total_bytes ||
        (uint64_t)h->off_mat + (uint64_t)h->n_mat * sizeof(qr_material) > h->total_bytes ||
        (uint64_t)h->off_lgt + (uint64_t)h->n_lgt * sizeof(qr_light) > h->total_bytes ||
        (uint64_t)h->off_elm + (uint64_t)h->n_elm * sizeof(qr_elem) > h->total_bytes ||
        (uint64_t)h->off_tiles + (uint64_t)h->n_tiles * 4 > h->total_bytes ||
        (uint64_t)h->off_texels + (uint64_t)h->n_texels * 4 > h->total_bytes) return -6;
    v->hdr    = h;
    v->frame  = (const qr_frame *)(p + h->off_frame);
    v->srf    = (const qr_surface *)(p + h->off_srf);
    v->mat    = (const qr_material *)(p + h->off_mat);
    v->lgt    = (const qr_light *)(p + h->off_lgt);
    v->elm    = (const qr_elem *)(p + h->off_elm);
    v->tiles  = (const int32_t *)(p + h->off_tiles);
    v->texels = (const uint32_t *)(p + h->off_texels);
    if ((uint32_t)(v->frame->tls_row * v->frame->tls_col) != h->n_tiles) return -7;
    return 0;
}

#ifdef __cplusplus
}
#endif

#endif /* QR_SCENE_H */
