/*
 * qr_program.h - the COMPILED scene the render kernel walks (host + device definition).
 *
 * The snapshot (include/qr_scene.h) keeps the reference's linked lists: cells chained by `next`, the role of a
 * cell (plain surface, bounding-volume array head, trnode with a cached transform, ...) decided while walking
 * from tags and from per-ray state (`ctx_LOCAL(OBJ)`, tracer.cpp:1385-1417).  That state depends only on the
 * position in the list, so the upload pass (qr_compile.cpp) resolves it ONCE per list:
 *   - every distinct list becomes a contiguous run of 32-byte cells ending in an END cell (no `next` chasing);
 *   - a cell's opcode word says what the walk has to do there: solver, which diff / ray the solver reads
 *     (world, trnode-cached, own transform), whether a bounding-sphere cull applies, how shadows treat a hit;
 *   - arrays carry the byte offset just behind their last cell: a ray that misses a bounding volume becomes
 *     active again at that offset (one unsigned compare per cell decides who takes part), a wave no ray of which
 *     enters it jumps there;
 *   - cells the walk can neither hit nor needs for its state (markers without transform) are dropped.
 * Everything in the device blob is addressed by BYTE OFFSETS from the blob's base, so the kernel keeps one
 * 64-bit base in SGPRs instead of one pointer per array (the round-1 kernel spilled 88 SGPRs).
 * Offset 0 is the header and never a valid list / record: 0 == "none".
 */
#ifndef QR_PROGRAM_H
#define QR_PROGRAM_H

#include <stdint.h>
#include "qr_scene.h"

/* ---- surface-list cells --------------------------------------------------------------------------------- */

struct CCell                    /* 32 B, 32-byte aligned */
{
    uint32_t op;                /* QR_OPT_* type bit | flags below; 0 ends the list                       */
    uint32_t srf;               /* byte offset of the surface's DSurf                                    */
    union { uint32_t end;       /* BV: byte offset just behind the array's last cell (BV cells take 64 B: CBvExt follows) */
            float r2; };        /* cull cells: R^2 of the bounding sphere                                */
    float    r2x;               /* cull cells: 1.01 R^2 (origin counts as outside the sphere beyond it)  */
    float    cx, cy, cz, r;     /* conservative world-space bounding sphere (QR_OPF_CULL cells)          */
};

/* second slot of a QR_OPT_BV cell: the fields of the volume's surface record that AR_ptr reads */
struct CBvExt
{
    float pos[3];
    uint32_t mid;               /* byte offset of a later child of the array, near the middle of its cells: from there on the
                                 * rest of a walk can be handed to another lane (walk_pool); 0 = none (small array, or the
                                 * list carries trnode state)                                                            */
    float sci[4];
};

/* cell type: one bit each, so that the walk tests them with s_bitcmp in the order of their frequency
 * (a dense enum makes the compiler build a compare tree) */
#define QR_OPT_PLANE    (1u << 0)   /* PL_ptr, tracer.cpp:4062-4136                                       */
#define QR_OPT_QUADRIC  (1u << 1)   /* QD_ptr, 4378-4447                                                  */
#define QR_OPT_TWOPLANE (1u << 2)   /* TP_ptr, 4216-4277                                                  */
#define QR_OPT_BV       (1u << 3)   /* AR_ptr bounding volume, 3955-4054                                  */
#define QR_OPT_TRNODE   (1u << 4)   /* array element with a transform: fills the trnode cache, 1419-1556 */
#define QR_OPT_MASK     31u
#define QR_OPT_SOLVER   (QR_OPT_PLANE | QR_OPT_QUADRIC | QR_OPT_TWOPLANE)

#define QR_OPF_CULL    (1u << 5)    /* bounding-sphere cull applies                                       */
/* which diff / ray the cell reads: neither bit: world (diff = org - pos, ray = dir) */
#define QR_OPF_CACHED  (1u << 6)    /*   inside a trnode: diff = cached - pos, ray = cached                */
#define QR_OPF_OWN     (1u << 7)    /*   own transform: diff = M (org - pos), ray = M dir                  */
#define QR_OPF_LOCAL   (QR_OPF_CACHED | QR_OPF_OWN)
#define QR_OPF_FULLM   (1u << 8)    /* transform has rotation (a_map[L] != 1): full 3x3, else diagonal    */
#define QR_OPF_KX      (1u << 9)    /* plane / two-plane: axis k is x                                     */
#define QR_OPF_KY      (1u << 10)   /*                    axis k is y (neither: z)                        */
#define QR_OPF_SGNK    (1u << 11)   /* plane: sign of axis k                                              */
#define QR_OPF_IX      (1u << 12)   /* two-plane: axis i is x                                             */
#define QR_OPF_IY      (1u << 13)   /*            axis i is y (neither: z)                                */
#define QR_OPF_NOSHAD  (1u << 14)   /* CHECK_SHAD 549-589: a hit never occludes                           */
#define QR_OPF_SIDESHAD (1u << 15)  /*   ... occludes depending on the side hit (look at the props)       */
#define QR_OPF_CLIP    (1u << 16)   /* surface has custom clippers                                        */
#define QR_OPF_CONIC   (1u << 17)   /* conic singularity fix applies (cones, hyper-cylinders)             */
#define QR_OPF_BOX     (1u << 19)   /* cull cell (QR_OPF_CULL, solver cells only): the six cull slots hold an axis-aligned world-space
                                     * box {lo.xyz, hi.xyz} instead of the sphere {R^2, 1.01 R^2, c.xyz, R}: planes, cylinders, cones --
                                     * shapes a sphere fits badly.  Only in images whose lists are all short (packet walks serve them);
                                     * a per-lane walk that meets one (QR_DIV=1) does not cull on it */
#define QR_OPF_SPHBV   (1u << 18)   /* bounding volume is an untransformed world-space sphere that holds all its members'
                                     * bounds: its cull sphere is that sphere (x 1.0002), the r2x slot holds -1 (0x7F800000
                                     * = +inf otherwise), and the per-lane walk decides most rays from the sphere alone */

/* flags in the low bits of a list offset (list programs are 32-byte aligned), carried wherever a list is referenced */
#define QR_LISTF_DIV   1u       /* no cell has a clipper program: the per-lane walk may take this list      */
#define QR_LISTF_LONG  2u       /* a long hierarchy (bounding-volume arrays, many cells): rays part ways on it */
#define QR_LISTF_WORLD 4u       /* every cell reads the world-space ray: no trnode cell, no QR_OPF_CACHED / QR_OPF_OWN cell */
#define QR_LISTF_DDA   16u      /* a CDda record sits in the 64 bytes in front of the list program: nearest-hit rays that are
                                 * not coherent may walk the uniform grid instead of the list (walk_dda)                 */
#define QR_LISTF_GRID  8u       /* only in CLight::shadow: the offset is that of a CGrid, the shadow list depends on where the
                                 * surface was hit                                                                        */
#define QR_LIST_OFF(x) ((x) & ~31u)
#define QR_CLEAR_RUN_MAX 16     /* a schedule head below 256 is a run of that many footprints over empty tiles (0: one) */
#define QR_LONG_CELLS  192      /* a list needs this many cells (and four bounding volumes) to be flagged QR_LISTF_LONG */

/*
 * Shadow lists of a LARGE surface by hit position (ours; the reference keeps one shadow list per surface and light, so a
 * ground plane under 10 000 objects gets all of them for every shadow ray).  For an untransformed plane with a finite
 * clip rectangle the compiler cuts the rectangle into nx x ny cells and filters the surface's shadow list once per cell:
 * what can stand between the light and that cell (qr_compile.cpp, HullPred).  The kernel picks the cell from the local
 * hit: cell = floor((loc[comp] - org) * inv), clamped.  Cells overlap by a margin far above the rounding of `loc`.
 */
struct CGrid                    /* 32 B, 32-byte aligned */
{
    uint32_t table;             /* byte offset of nx * ny list offsets (row-major, b major; 0 = nothing can shadow that cell) */
    uint32_t nx, ny;            /* cells along component a / b, 1..QR_GRID_MAX each                                       */
    uint32_t comps;             /* a | b << 2: which components of the local hit span the plane                           */
    float    org_a, org_b;      /* local coordinate of the rectangle's low corner                                         */
    float    inv_a, inv_b;      /* cells per unit                                                                         */
};
#define QR_GRID_MAX 64u

/*
 * Uniform grid over the members of a long world-space list (ours).  A ray through a sparse cloud of thousands of small
 * objects enters 60 fat bounding spheres of the scene's hierarchy and tests 250 cells to find the two objects it comes
 * near; a grid of about one object per cell hands it those after ~35 cell steps.  The list program stays (packet walks
 * of coherent rays, shadow rays); the grid holds COPIES of the members' cells, 32 bytes each with the r2x slot replaced
 * by the byte offset of the original cell -- the position in list order that decides between equal depths
 * (tracer.cpp:1626: strict compare, first in the list wins).  refs [0, n_out) are the members too large for the grid
 * (or unbounded); they are tested first.  Bounding-volume elements of the list are not consulted: they only skip work
 * (tracer.cpp:3955-4054) and hold their members with a margin far above the rounding of their own test: the engine
 * derives an array's volume from its members' boxes (rt_Array::update_bounds, object.cpp:1830), synth.py does the same.
 */
struct CDda                     /* 64 B, immediately in front of the list program's first cell */
{
    float    org[3];  uint32_t dims;    /* low corner; nx | ny << 8 | nz << 16, 1..QR_GRID_MAX each                       */
    float    inv[3];  uint32_t cells;   /* cells per unit; byte offset of nx*ny*nz + 1 ref indices (cell i: [c[i], c[i+1])) */
    float    size[3]; uint32_t refs;    /* units per cell; byte offset of the refs (CCell copies)                          */
    uint32_t n_out;   uint32_t n_refs;  uint32_t pad[2];
};

/* ---- clipper programs (custom clipping, tracer.cpp:1931-2151) ------------------------------------------- */

struct CClip                    /* 32 B: one s_load_dwordx8 per clipper */
{
    uint32_t op;                /* QR_CL_* | flags                                                        */
    uint32_t srf;               /* byte offset of the clipper's DSurf                                     */
    uint32_t aux;               /* fast plane cell: the plane's position along its axis, sign folded in (float) */
    uint32_t sgn;               /* fast plane cell: 0x80000000 when the axis is negated, else 0           */
    uint32_t mx, my, mz;        /* fast plane cell: all-ones for the plane's axis, 0 for the others: the component is
                                 * (x & mx) | (y & my) | (z & mz), no decoding                              */
    uint32_t pad;
};
/* op == 0 ends the program; one type bit each */
#define QR_CLT_PLANE   (1u << 0)    /* PL_clp: f = +-(x_k)                                                */
#define QR_CLT_QUADJ   (1u << 1)    /* QD_clp with the linear term (scj)                                  */
#define QR_CLT_QUAD    (1u << 2)    /* QD_clp without                                                     */
#define QR_CLT_ENTER   (1u << 3)    /* accumulator enter marker, tracer.h:79                              */
#define QR_CLT_LEAVE   (1u << 4)    /* accumulator leave marker                                           */
#define QR_CLT_TRNODE  (1u << 5)    /* trnode of the clipper list: transform the hit once, cache          */
#define QR_CLT_TRSAME  (1u << 6)    /* the surface's own trnode: reuse the surface's local hit             */
#define QR_CLT_MASK    127u
#define QR_CLF_INNER   (1u << 7)    /* data < 0 (MINUS_INNER): keep f >= 0, else keep f <= 0              */
#define QR_CLF_CACHED  (1u << 8)    /* hit in the cached clipper trnode's space - pos                     */
#define QR_CLF_OWN     (1u << 9)    /* M (hit - pos); neither: hit - pos                                  */
#define QR_CLF_FULLM   (1u << 10)
#define QR_CLF_KX      (1u << 11)
#define QR_CLF_KY      (1u << 12)
#define QR_CLF_SGNK    (1u << 13)
#define QR_CLF_CDEF    (1u << 14)   /* ENTER: the owner's c_def mask is all ones                          */
/*
 * Fast plane cell (ours): a plane clipper that reads the hit as it stands -- in world space, or in the trnode space the
 * cells before it have put it into (QR_CLF_CACHED) -- without a transform of its own.  The reference keeps
 * f = +-(p_k - pos_k) <= 0 (MINUS_OUTER) or >= 0 (MINUS_INNER; APPLY_CLIP, tracer.cpp:488-496, PL_clp 4198-4208).  A
 * rounded difference of two floats has the sign of the exact one, so with a = +-p_k and aux = +-pos_k (the sign folded
 * in here) the tests are a <= aux and !(a < aux): one compare, no subtraction, no load of the clipper's record -- the
 * same decisions bit for bit (NaN: false / true, as cle / cge).  `aux` holds the float.
 */
#define QR_CLF_FASTPL  (1u << 15)
#define QR_CLF_LASTC   (1u << 16)   /* the cached trnode space ends behind this cell: fast plane cells read the world hit again */

/* ---- light lists ----------------------------------------------------------------------------------------- */

struct CLight { uint32_t lgt; uint32_t shadow; };     /* byte offsets of the qr_light (| QR_CLIGHT_LAST on the last entry) and of the shadow list (0 none) */
#define QR_CLIGHT_LAST 1u

/* ---- per-surface records --------------------------------------------------------------------------------- */

/* walk record, 128 B: a plane needs the first 48 bytes, every other solver the first 80 */
struct DSurf
{
    float pos[3]; uint32_t clip;        /*  0  clip: byte offset of the clipper program or 0               */
    float min[3]; float d_eps;          /*  4  unclipped axes hold -inf / +inf                               */
    float max[3]; float t_eps;          /*  8                                                              */
    float sci[4];                       /* 12                                                              */
    float scj[3]; uint32_t flags;       /* 16  DF_* below                                                    */
    float tci[3]; uint32_t trn;         /* 20  trn: byte offset of the trnode's DSurf (shading normals)      */
    float tcj[3]; int32_t props0;       /* 24                                                              */
    float tck[3]; int32_t props1;       /* 28                                                              */
};

#define DF_CONIC(f)   (((f) >> 6) & 3u)
#define DF_TRM(f)     (((f) >> 8) & 3u)
#define DF_SHIFT(f)   (((f) >> 10) & 1u)
#define DF_MAP(f, n)  (((f) >> (11 + 2 * (n))) & 3u)
#define DF_SGN(f, n)  ((((f) >> (17 + (n))) & 1u) ? 0x80000000u : 0u)
#define DF_SOLVER(f)  (((f) >> 20) & 3u)
#define DF_NKIND(f)   (((f) >> 22) & 3u)
#define DF_CKIND(f)   (((f) >> 24) & 3u)
#define DF_ARRAY(f)   (((f) >> 26) & 1u)
#define DF_CDEF(f)    (((f) >> 28) & 1u)

/* shading record, 32 B */
struct DShade
{
    uint32_t mat[2];            /* byte offsets of the outer / inner material                              */
    uint32_t lgt[2];            /* byte offsets of the outer / inner light list (0 none)                   */
    uint32_t lst[2];            /* byte offsets of the outer / inner surface list for secondary rays       */
    uint32_t srf;               /* byte offset of the DSurf                                                 */
    uint32_t pad;
};

/* blob layout: header at offset 0, the DSurf array right behind it */
#define QR_OFF_SRF 256u
/* blob header at offset 0 */
struct DevHeader
{
    qr_frame fr;                /* camera + frame geometry, read with scalar loads                          */
    uint32_t off_shade;         /* DShade array, indexed by surface index (hit -> shading)                  */
    uint32_t off_tiles;         /* tile heads as byte offsets of lists (0 = empty tile)                     */
    uint32_t off_order;         /* whole-frame wave schedule                                                */
    uint32_t n_blocks;
    uint32_t img_flags;         /* QR_IMG_*                                                                  */
    uint32_t img_bytes;         /* size of the image (the guarded diagnostic build checks cell offsets against it) */
    uint32_t pad[9];
};
#define QR_IMG_BOXES 1u         /* some cull cell carries a box (QR_OPF_BOX): packet walks prepare the slab test */

#endif /* QR_PROGRAM_H */
