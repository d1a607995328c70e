/*
 * qr_oracle.c - TEST INFRASTRUCTURE: scalar CPU restatement of the reference's
 * per-pixel rendering pipeline `render0` (core/tracer/tracer.cpp:1081-5405).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; the product (libqrhip.so) never does.
 *
 * One ray at a time ("a SIMD packet of width one"): every packed operation of
 * the reference becomes the same IEEE fp32 operation on one lane, in the same
 * order, with the same compare predicates; lane masks become 0 / 0xFFFFFFFF
 * words; packet-wide early-outs (CHECK_MASK NONE/FULL, rtbase.h:1209) become
 * plain branches.  Input is the flattened scene of include/qr_scene.h.
 *
 * Numeric rules:
 *   - no fused multiply-add (build with -ffp-contract=off),
 *   - rcp = 1.0f/x, rsq = 1.0f/sqrtf(x) (two roundings), rtconf.h:164-203,
 *   - ceq/clt/cle false on NaN; cne/cgt/cge true on NaN
 *     (rtarch_x32_128x1v4.h:721-824: predicates 0,1,2 / 4,6,5),
 *   - cvn = round-half-even, cvm = floor, both 0x80000000 when out of range
 *     (cvtps2dq semantics),
 *   - min(a,b) = a < b ? a : b (minps).
 *
 * PINNING: checked bit-for-bit against frames rendered by the unmodified
 * reference (oracle/_ref/qr_ref, built from /root/reference by oracle/Makefile)
 * for demo01-03 and test01-18; the committed fixtures under tests/golden/ are
 * those frames (tests/golden/make_golden.py is the generating script).
 */
#include "qr_scene.h"

#include <math.h>
#include <string.h>
#include <stdlib.h>
#include <stdint.h>

#ifdef _OPENMP
#include <omp.h>
#endif

typedef uint32_t u32;

/* ------------------------------------------------------------------------ */
/* lane primitives                                                           */
/* ------------------------------------------------------------------------ */

static inline u32   f2u(float f) { u32 u; memcpy(&u, &f, 4); return u; }
static inline float u2f(u32 u)   { float f; memcpy(&f, &u, 4); return f; }

#define MASK(c) ((c) ? 0xFFFFFFFFu : 0u)

static inline u32 ceq(float a, float b) { return MASK(a == b); }
static inline u32 cne(float a, float b) { return MASK(!(a == b)); }
static inline u32 clt(float a, float b) { return MASK(a < b); }
static inline u32 cle(float a, float b) { return MASK(a <= b); }
static inline u32 cgt(float a, float b) { return MASK(!(a <= b)); }   /* NLE */
static inline u32 cge(float a, float b) { return MASK(!(a < b)); }    /* NLT */

static inline float fand(float a, u32 m) { return u2f(f2u(a) & m); }
static inline float fxor(float a, u32 m) { return u2f(f2u(a) ^ m); }
static inline float fsel(u32 m, float a, float b) { return u2f((f2u(a) & m) | (f2u(b) & ~m)); }

static inline float rsq(float x) { return 1.0f / sqrtf(x); }

static inline int32_t cvt_floor(float x)     /* cvmps: round toward -inf */
{
    float f = floorf(x);
    if (!(f >= -2147483648.0f && f < 2147483648.0f)) return (int32_t)0x80000000u;
    return (int32_t)f;
}
static inline int32_t cvt_near(float x)      /* cvnps: round half to even */
{
    float f = nearbyintf(x);                 /* default rounding mode */
    if (!(f >= -2147483648.0f && f < 2147483648.0f)) return (int32_t)0x80000000u;
    return (int32_t)f;
}

#define SMASK 0x80000000u
#define AMASK_ABS 0x7FFFFFFFu

/* context flags, tracer.cpp:504-512 */
#define FLAG_SIDE 1
#define FLAG_PASS_THRU 2
#define FLAG_SHAD 4

/* ------------------------------------------------------------------------ */
/* state                                                                     */
/* ------------------------------------------------------------------------ */

typedef struct counts_t { uint64_t primary, shadow, reflect, refract, flops; } counts_t;

/*
 * Algorithmic-work counter (ours, test/bench infrastructure only): fp32 operations per executed
 * block for one lane, using the fixed weights of SURVEY.md section 8(d) (div and sqrt count 1,
 * compares and bit operations 0).  Deterministic per (snapshot, depth, mode).
 */
#define FL(T, n) ((T)->cnt.flops += (uint64_t)(n))

typedef struct scene_t
{
    qr_scene_view v;
    int depth;          /* inf_DEPTH start value */
} scene_t;

/*
 * One level of the reference's context stack (rt_SIMD_CONTEXT, tracer.h:426-662)
 * for a single lane.  vec[0..2] = world XYZ fields, vec[3..5] = IJK fields.
 */
typedef struct ctx_t
{
    float t_min;
    float org[3];
    float ray[6];
    float dff[6];
    float tex_uv[2];
    u32   c_buf;
    float tex[3];
    float col[3];
    u32   c_acc;
    float t_val, t_buf;
    u32   wmask;
    float nrm[6];
    float hit[3];
    float nw[6];            /* NEW_X..K: local hit / child ray */
    /* packed scalars */
    int   param_tag;        /* ctx_PARAM(PTR): 0 primary,1 shadow,2 rfl,3 rfr */
    int   param_flg;        /* ctx_PARAM(FLG) */
    int   param_obj;        /* ctx_PARAM(OBJ): originating surface or -1 */
    int   local_flg;        /* ctx_LOCAL(FLG) */
    int   local_obj;        /* ctx_LOCAL(OBJ): trnode's last element or -1 */
    /* quadric root state */
    u32   dmask, amask;
    int   xmisc_ptr;
    /* primary hit id (ours, not in the reference) */
    int   hit_id;
    /* deferred-shading mode (ours): the pending final hit of this context */
    int   pend_si, pend_side, pend_kind, in_final;
    float pend_t, pend_loc[3];
} ctx_t;

typedef struct tracer_t
{
    const scene_t *s;
    int depth;              /* inf_DEPTH, decremented around child packets */
    int deferred;           /* shade only the final hit of a list walk (see qro_render) */
    counts_t cnt;
} tracer_t;

static void trace_list(tracer_t *T, ctx_t *c, const float *parent_loc, int head);

/* ------------------------------------------------------------------------ */
/* helpers on surfaces                                                       */
/* ------------------------------------------------------------------------ */

static inline int ax_map(const qr_surface *s, int n) { return (int)((s->axes >> (2 * n)) & 3); }
static inline u32 ax_sgn(const qr_surface *s, int n) { return ((s->axes >> (8 + n)) & 1) ? SMASK : 0u; }

/* 3x3 transform of diff / ray / clip vector, tracer.cpp:1447-1479 (same order) */
static inline void xform(const qr_surface *s, const float *in, float *out)
{
    float x1 = in[0], x2 = in[1], x3 = in[2];
    float x4 = s->tci[0] * x1;
    float x5 = s->tcj[1] * x2;
    float x6 = s->tck[2] * x3;
    if (s->has_trm != 1)
    {
        x4 = x4 + s->tci[1] * x2;
        x4 = x4 + s->tci[2] * x3;
        x5 = x5 + s->tcj[0] * x1;
        x5 = x5 + s->tcj[2] * x3;
        x6 = x6 + s->tck[0] * x1;
        x6 = x6 + s->tck[1] * x2;
    }
    out[0] = x4; out[1] = x5; out[2] = x6;
}

/* ------------------------------------------------------------------------ */
/* CC_clp: depth test, hit point, conic fix, min/max, custom clippers        */
/* tracer.cpp:1597-2160.  Returns the refined lane mask.                     */
/* ------------------------------------------------------------------------ */

static u32 clip(tracer_t *T, ctx_t *c, int si, u32 m)
{
    const qr_scene_view *v = &T->s->v;
    const qr_surface *s = &v->srf[si];
    const int sh = s->shift ? 3 : 0;
    float t = c->t_val;
    float x4, x5, x6;

    FL(T, 15);
    /* depth testing, near plane clipping: 1602-1610 */
    m &= cgt(c->t_buf, t);
    m &= clt(c->t_min, t);

    /* world hit: 1612-1628 */
    x4 = c->ray[0] * t; x4 = x4 + c->org[0]; c->hit[0] = x4;
    x5 = c->ray[1] * t; x5 = x5 + c->org[1]; c->hit[1] = x5;
    x6 = c->ray[2] * t; x6 = x6 + c->org[2]; c->hit[2] = x6;

    if (s->has_trm != 0)
    {
        /* local hit in trnode space: 1635-1655 */
        x4 = c->ray[3] * t; x4 = x4 + c->dff[3]; c->nw[3] = x4;
        x5 = c->ray[4] * t; x5 = x5 + c->dff[4]; c->nw[4] = x5;
        x6 = c->ray[5] * t; x6 = x6 + c->dff[5]; c->nw[5] = x6;
    }
    else
    {
        /* 1665-1679 */
        x4 = x4 - s->pos[0]; c->nw[0] = x4;
        x5 = x5 - s->pos[1]; c->nw[1] = x5;
        x6 = x6 - s->pos[2]; c->nw[2] = x6;
    }

    /* conic singularity solver: 1706-1856 */
    if (s->conic != 0 && c->xmisc_ptr != 0)
    {
        const int mi = ax_map(s, 0), mj = ax_map(s, 1), mk = ax_map(s, 2);
        float x0, x1, x2, x3;
        u32 hmask;
        FL(T, 16);
        x1 = c->nw[sh + mi]; x1 = x1 * x1; x0 = x1;
        if (s->conic != 2)
        {
            x2 = c->nw[sh + mj]; x2 = x2 * x2; x0 = x0 + x2;
        }
        x3 = c->nw[sh + mk]; x3 = x3 * x3; x0 = x0 + x3;
        hmask = clt(x0, s->t_eps) & c->dmask;
        if (hmask != 0)
        {
            u32 sm = s->smask;
            float one = 1.0f, r4;
            u32 tside, u6, u5;
            x2 = 0.0f;
            x1 = fxor(fand(c->dff[sh + mi], sm), f2u(one));
            x3 = s->sci[mi];
            r4 = one;
            if (s->conic != 2)
            {
                x2 = fxor(fand(c->dff[sh + mj], sm), f2u(one));
                x3 = x3 + s->sci[mj];
                r4 = r4 + one;
            }
            x3 = x3 / s->sci[mk];
            x3 = fxor(x3, sm);
            x6 = x3;
            x3 = sqrtf(x3);
            x6 = x6 + r4;
            r4 = rsq(x6);
            r4 = r4 * s->t_eps;
            x1 = x1 * r4; x2 = x2 * r4; x3 = x3 * r4;

            tside = (c->local_flg & FLAG_SIDE) ? sm : 0u;   /* Iebx srf_SBASE + side*Q*16 */
            /* note: LOCAL(FLG) holds only the side here (set right before SUBROUTINE 3/5) */
            x3 = fxor(x3, f2u(c->dff[sh + mk]) & sm);
            u6 = (tside & c->amask) ^ c->amask;
            x3 = fxor(x3, u6);
            u5 = (tside | c->amask) ^ c->amask;
            x1 = fxor(x1, u5);
            x2 = fxor(x2, u5);

            c->nw[sh + mi] = fsel(hmask, x1, c->nw[sh + mi]);
            if (s->conic != 2)
            {
                c->nw[sh + mj] = fsel(hmask, x2, c->nw[sh + mj]);
            }
            c->nw[sh + mk] = fsel(hmask, x3, c->nw[sh + mk]);

            x4 = c->nw[sh + 0]; x5 = c->nw[sh + 1]; x6 = c->nw[sh + 2];
        }
    }

    /* axis min/max clipping: 1874-1927 */
    if (s->minmax_t & 0x01) m &= cle(s->min[0], x4);
    if (s->minmax_t & 0x08) m &= cge(s->max[0], x4);
    if (s->minmax_t & 0x02) m &= cle(s->min[1], x5);
    if (s->minmax_t & 0x10) m &= cge(s->max[1], x5);
    if (s->minmax_t & 0x04) m &= cle(s->min[2], x6);
    if (s->minmax_t & 0x20) m &= cge(s->max[2], x6);

    /* custom clipping: 1931-2151 */
    {
        int redx = QR_NULL;                 /* trnode's last element (caching) */
        const int local_lst = s->trnode;    /* ctx_LOCAL(LST) <- msc_p[3] */
        int e;
        for (e = s->clip; e != QR_NULL; e = v->elm[e].next)
        {
            const qr_elem *el = &v->elm[e];
            const qr_surface *k;
            float d[3], p[3];
            int ksh;
            if (el->simd == QR_NULL)
            {
                /* accum markers: 1948-1962 */
                if (el->data > 0) { m = ~m & c->c_acc; }            /* leave */
                else              { c->c_acc = m; m = s->c_def; }   /* enter */
                continue;
            }
            k = &v->srf[el->simd];
            if (k->srf_t[3] >= 0)
            {
                if (redx != QR_NULL)
                {
                    /* 1979-2004 */
                    FL(T, 3);
                    c->nrm[3] = c->nrm[0] - k->pos[0];
                    c->nrm[4] = c->nrm[1] - k->pos[1];
                    c->nrm[5] = c->nrm[2] - k->pos[2];
                    if (e == redx) redx = QR_NULL;
                    goto cc_trm;
                }
            }
            else
            {
                /* CC_arr: 2006-2037 */
                if (el->simd == local_lst)
                {
                    c->nrm[0] = c->nw[3] + s->pos[0];
                    c->nrm[1] = c->nw[4] + s->pos[1];
                    c->nrm[2] = c->nw[5] + s->pos[2];
                    redx = el->data;
                    continue;
                }
            }
            /* CC_dff: 2043-2125 */
            FL(T, 3);
            d[0] = c->hit[0] - k->pos[0];
            d[1] = c->hit[1] - k->pos[1];
            d[2] = c->hit[2] - k->pos[2];
            c->nrm[0] = d[0]; c->nrm[1] = d[1]; c->nrm[2] = d[2];
            if (k->has_trm != 0)
            {
                xform(k, d, p);
                FL(T, 18);
                if (k->srf_t[3] < 0)
                {
                    c->nrm[0] = p[0]; c->nrm[1] = p[1]; c->nrm[2] = p[2];
                    redx = el->data;
                    continue;
                }
                c->nrm[3] = p[0]; c->nrm[4] = p[1]; c->nrm[5] = p[2];
            }
        cc_trm:
            ksh = k->shift ? 3 : 0;
            {
                float f4 = 0.0f, f5, f6, f1, f2, f3;
                u32 r;
                switch (k->srf_t[2])
                {
                case 1: /* PL_clp 4198-4208 */
                    f4 = fxor(c->nrm[ksh + ax_map(k, 2)], ax_sgn(k, 2));
                    break;
                case 2: /* QD_clp 4910-4951 */
                    FL(T, 18);
                    f4 = c->nrm[ksh + 0]; f1 = k->scj[0]; f1 = f1 + f1; f1 = f1 * f4;
                    f4 = f4 * f4; f4 = f4 * k->sci[0]; f4 = f4 - f1;
                    f5 = c->nrm[ksh + 1]; f2 = k->scj[1]; f2 = f2 + f2; f2 = f2 * f5;
                    f5 = f5 * f5; f5 = f5 * k->sci[1]; f5 = f5 - f2;
                    f6 = c->nrm[ksh + 2]; f3 = k->scj[2]; f3 = f3 + f3; f3 = f3 * f6;
                    f6 = f6 * f6; f6 = f6 * k->sci[2]; f6 = f6 - f3;
                    f4 = f4 - k->sci[3]; f4 = f4 + f5; f4 = f4 + f6;
                    break;
                case 3: /* TP_clp 4341-4370 */
                    FL(T, 11);
                    f4 = c->nrm[ksh + 0]; f4 = f4 * f4; f4 = f4 * k->sci[0];
                    f5 = c->nrm[ksh + 1]; f5 = f5 * f5; f5 = f5 * k->sci[1];
                    f6 = c->nrm[ksh + 2]; f6 = f6 * f6; f6 = f6 * k->sci[2];
                    f4 = f4 - k->sci[3]; f4 = f4 + f5; f4 = f4 + f6;
                    break;
                default:
                    /* falls to CC_ret with Xmm4 as left by the code above;
                     * never reached for well-formed scenes */
                    continue;
                }
                /* APPLY_CLIP 488-496 */
                r = el->data < 0 ? cge(f4, 0.0f) : cle(f4, 0.0f);
                m &= r;
            }
        }
    }
    return m;
}

/* ------------------------------------------------------------------------ */
/* material / lighting / transparency / reflection                           */
/* tracer.cpp:2166-3947 for one lane that has hit surface `si` on `side`.    */
/* Returns 1 if the caller must stop traversing the list (shadow ray done).  */
/* ------------------------------------------------------------------------ */

enum { NRM_PLANE = 1, NRM_QUADRIC = 2, NRM_TWOPLANE = 3 };

static int shade(tracer_t *T, ctx_t *c, int si, int side, int kind)
{
    const qr_scene_view *v = &T->s->v;
    const qr_surface *s = &v->srf[si];
    const qr_frame *fr = v->frame;
    const int sh = s->shift ? 3 : 0;
    const u32 tside = side ? s->smask : 0u;
    const qr_material *mt;
    int props;
    float x0, x1, x2, x3, x4, x5, x6, x7;

    /* FETCH_PROP 597-604 */
    c->local_flg = side | s->props[side];
    props = c->local_flg;

    /* CHECK_SHAD 549-589 */
    if (c->param_flg & FLAG_SHAD)
    {
        if (props & QR_PROP_LIGHT) return 0;
        if ((props & QR_PROP_TRANSP) && !(props & QR_PROP_REFRACT)) return 0;
        c->c_buf = 0xFFFFFFFFu;
        return 1;
    }

    /*
     * Deferred mode (NOT how the reference works; it is the claim the HIP
     * backend relies on, checked here on the CPU): shading has no effect on
     * the list walk and fully overwrites the lane's colour, so only the last
     * hit that passes the depth test needs shading.  Keep the depth write
     * (PAINT_FRAG) and remember the hit.
     */
    if (T->deferred && !c->in_final)
    {
        c->t_buf = c->t_val;
        c->pend_si = si; c->pend_side = side; c->pend_kind = kind; c->pend_t = c->t_val;
        c->pend_loc[0] = c->nw[sh + 0]; c->pend_loc[1] = c->nw[sh + 1]; c->pend_loc[2] = c->nw[sh + 2];
        return 0;
    }

    FL(T, 16 + 4 + 6);      /* normal, texture lookup, ambient */
    /* surface-kind specific part: texture coords + normal */
    if (kind == NRM_PLANE)
    {
        /* PL_mat 4139-4193 */
        if (props & QR_PROP_TEXTURE)
        {
            c->tex_uv[0] = fxor(c->nw[sh + ax_map(s, 0)], ax_sgn(s, 0));
            c->tex_uv[1] = fxor(c->nw[sh + ax_map(s, 1)], ax_sgn(s, 1));
        }
        if (props & QR_PROP_NORMAL)
        {
            c->nrm[sh + ax_map(s, 0)] = 0.0f;                       /* MOVZR_ST stores +0 */
            c->nrm[sh + ax_map(s, 1)] = 0.0f;
            x6 = fxor(1.0f, tside);
            c->nrm[sh + ax_map(s, 2)] = fxor(x6, ax_sgn(s, 2));
        }
    }
    else if (props & QR_PROP_NORMAL)
    {
        /* QD_mat 4845-4905 / TP_mat 4280-4336 */
        x4 = c->nw[sh + 0]; x5 = c->nw[sh + 1]; x6 = c->nw[sh + 2];
        x4 = x4 * s->sci[0]; x5 = x5 * s->sci[1]; x6 = x6 * s->sci[2];
        if (kind == NRM_QUADRIC)
        {
            x4 = x4 - s->scj[0]; x5 = x5 - s->scj[1]; x6 = x6 - s->scj[2];
        }
        x1 = x4 * x4; x2 = x5 * x5; x3 = x6 * x6;
        x1 = x1 + x2; x1 = x1 + x3;
        x0 = rsq(x1);
        x0 = fxor(x0, tside);
        x4 = x4 * x0; x5 = x5 * x0; x6 = x6 * x0;
        c->nrm[sh + 0] = x4; c->nrm[sh + 1] = x5; c->nrm[sh + 2] = x6;
    }

    /* MT_nrm 2168-2263: transform normal by the trnode's transposed matrix */
    if ((props & QR_PROP_NORMAL) && s->has_trm != 0)
    {
        const qr_surface *tr = &v->srf[s->trnode];
        FL(T, 24);
        x1 = c->nrm[3]; x2 = c->nrm[4]; x3 = c->nrm[5];
        x4 = tr->tci[0] * x1;
        x5 = tr->tcj[1] * x2;
        x6 = tr->tck[2] * x3;
        if (tr->has_trm != 1)
        {
            x4 = x4 + tr->tcj[0] * x2;
            x4 = x4 + tr->tck[0] * x3;
            x5 = x5 + tr->tci[1] * x1;
            x5 = x5 + tr->tck[1] * x3;
            x6 = x6 + tr->tci[2] * x1;
            x6 = x6 + tr->tcj[2] * x2;
        }
        if (tr->has_trm == 1 || tr->has_trm != 2)
        {
            /* MT_trn: renormalize */
            x1 = x4 * x4; x2 = x5 * x5; x3 = x6 * x6;
            x1 = x1 + x2; x1 = x1 + x3;
            x0 = rsq(x1);
            x4 = x4 * x0; x5 = x5 * x0; x6 = x6 * x0;
        }
        c->nrm[0] = x4; c->nrm[1] = x5; c->nrm[2] = x6;
    }

    /* MT_mat 2267-2327: keep local hit in NRM_I/J/K for child contexts */
    c->nrm[3] = c->nw[sh + 0];
    c->nrm[4] = c->nw[sh + 1];
    c->nrm[5] = c->nw[sh + 2];

    mt = &v->mat[s->mat[side]];

    {
        u32 texel_off = 0;
        if (props & QR_PROP_TEXTURE)
        {
            int32_t iu, iv;
            x4 = c->tex_uv[mt->t_map[0]];
            x5 = c->tex_uv[mt->t_map[1]];
            x4 = x4 - mt->xoffs; x5 = x5 - mt->yoffs;
            x4 = x4 * mt->xscal; x5 = x5 * mt->yscal;
            iu = cvt_floor(x4) & (int32_t)mt->xmask;
            iv = cvt_floor(x5) & (int32_t)mt->ymask;
            texel_off = (u32)iu + ((u32)iv << (mt->yshft & 31));
        }
        /* PAINT_FRAG 653-662 */
        c->t_buf = c->t_val;
        c->c_buf = v->texels[mt->tex + (int32_t)texel_off];
    }
    if (c->param_tag == 0) c->hit_id = (si << 1) | side;

    /* PAINT_COLX 664-673 */
    {
        int k;
        static const int shft[3] = { 16, 8, 0 };
        for (k = 0; k < 3; k++)
        {
            float t = (float)(int32_t)((c->c_buf >> shft[k]) & mt->cmask);
            t = t / mt->clamp;
            if (props & QR_PROP_GAMMA) t = t * t;
            c->tex[k] = t;
        }
    }

    /* ---------------- lights: 2709-3179 ---------------- */
    if (props & QR_PROP_LIGHT)
    {
        /* LT_set */
        c->col[0] = c->tex[0]; c->col[1] = c->tex[1]; c->col[2] = c->tex[2];
    }
    else
    {
        int le;
        c->col[0] = c->tex[0] * fr->amb[0];
        c->col[1] = c->tex[1] * fr->amb[1];
        c->col[2] = c->tex[2] * fr->amb[2];

        for (le = s->lst[side * 2]; le != QR_NULL; le = v->elm[le].next)
        {
            const qr_light *lg = &v->lgt[v->elm[le].simd];
            float dot, r2;
            u32 lit;
            FL(T, 8);

            x1 = lg->pos[0] - c->hit[0]; c->nw[0] = x1; x1 = x1 * c->nrm[0];
            x2 = lg->pos[1] - c->hit[1]; c->nw[1] = x2; x2 = x2 * c->nrm[1];
            x3 = lg->pos[2] - c->hit[2]; c->nw[2] = x3; x3 = x3 * c->nrm[2];
            x0 = x1; x0 = x0 + x2; x0 = x0 + x3;
            if (!clt(0.0f, x0)) continue;           /* LT_amb */
            dot = x0;

            /* shadow packet: 2794-2852 */
            {
                ctx_t ch;
                memset(&ch, 0, sizeof(ch));
                ch.param_flg = c->local_flg | FLAG_SHAD;
                ch.param_obj = si;
                ch.param_tag = 1;
                ch.wmask = 0xFFFFFFFFu;
                ch.t_buf = lg->t_max;
                ch.c_buf = 0;
                ch.t_min = 0.0f;
                ch.org[0] = c->hit[0]; ch.org[1] = c->hit[1]; ch.org[2] = c->hit[2];
                ch.ray[0] = c->nw[0];  ch.ray[1] = c->nw[1];  ch.ray[2] = c->nw[2];
                ch.local_obj = QR_NULL; ch.pend_si = QR_NULL;
                T->depth -= 1;
                T->cnt.shadow++;
                trace_list(T, &ch, &c->nrm[3], v->elm[le].data);
                T->depth += 1;
                if (ch.c_buf != 0) continue;        /* shadowed */
            }
            lit = 0xFFFFFFFFu;
            FL(T, 17);

            x1 = c->nw[0]; x4 = x1 * x1;
            x2 = c->nw[1]; x5 = x2 * x2;
            x3 = c->nw[2]; x6 = x3 * x3;
            x4 = x4 + x5; x4 = x4 + x6;
            r2 = x4;
            x0 = dot;

            if (props & QR_PROP_DIFFUSE)
            {
                x0 = fand(x0, lit);
                x6 = x4;
                x5 = rsq(x4);
                x4 = x5 * x6;
                x6 = x6 * lg->a_qdr;
                x4 = x4 * lg->a_lnr;
                x6 = x6 + lg->a_cnt;
                x6 = x6 + x4;
                x4 = rsq(x6);
                x6 = x0;
                x0 = x0 * x4;
                x0 = x0 * x5;
                x0 = x0 * mt->l_dff;
            }
            else
            {
                x6 = x0;
                x0 = 0.0f;
            }

            {
                int plain = 0;
                float spec = 0.0f;
                if (props & QR_PROP_SPECULAR)
                {
                    u32 m2;
                    FL(T, 32);
                    x4 = x6; x5 = x6;
                    x4 = x4 * c->nrm[0]; x1 = x1 - x4; x1 = x1 - x4;
                    x5 = x5 * c->nrm[1]; x2 = x2 - x5; x2 = x2 - x5;
                    x6 = x6 * c->nrm[2]; x3 = x3 - x6; x3 = x3 - x6;
                    x4 = c->ray[0]; x1 = x1 * x4; x4 = x4 * x4;
                    x5 = c->ray[1]; x2 = x2 * x5; x5 = x5 * x5;
                    x6 = c->ray[2]; x3 = x3 * x6; x6 = x6 * x6;
                    x6 = x6 + x4; x6 = x6 + x5;
                    x1 = x1 + x2; x1 = x1 + x3;
                    m2 = clt(0.0f, x1) & lit;
                    x1 = fand(x1, m2);
                    if (m2 != 0)
                    {
                        u32 pw;
                        x4 = r2;
                        x5 = rsq(x6); x1 = x1 * x5;
                        x5 = rsq(x4); x1 = x1 * x5;
                        /* fixed-point 28.4 power: 2981-3039 */
                        pw = mt->l_pow & 0xF;
                        x2 = x1; x4 = x1; x1 = 1.0f;
                        if (pw != 0)
                        {
                            do
                            {
                                u32 bit;
                                x4 = sqrtf(x4);
                                bit = pw & 0x8;
                                pw = (pw << 1) & 0xF;
                                if (bit) x1 = x1 * x4;
                            }
                            while (pw != 0);
                        }
                        pw = mt->l_pow >> 4;
                        if (pw != 0)
                        {
                            x3 = x1; x1 = 1.0f;
                            do
                            {
                                u32 bit = pw & 1;
                                pw >>= 1;
                                if (bit) x1 = x1 * x2;
                                x2 = x2 * x2;
                            }
                            while (pw != 0);
                            x1 = x1 * x3;
                        }
                        x1 = x1 * mt->l_spc;
                        if (props & QR_PROP_METAL) { x0 = x0 + x1; }
                        else { plain = 1; spec = x1; }
                    }
                }
                if (!plain)
                {
                    /* metal / common: 3051-3086 */
                    FL(T, 10);
                    x1 = c->tex[0] * lg->col[0];
                    x2 = c->tex[1] * lg->col[1];
                    x3 = c->tex[2] * lg->col[2];
                    x1 = x1 * x0; x2 = x2 * x0; x3 = x3 * x0;
                    c->col[0] = x1 + c->col[0];
                    c->col[1] = x2 + c->col[1];
                    c->col[2] = x3 + c->col[2];
                }
                else
                {
                    /* LT_mtl plain: 3090-3149 */
                    x7 = spec;
                    x1 = c->tex[0]; x2 = c->tex[1]; x3 = c->tex[2];
                    x4 = lg->col[0]; x5 = lg->col[1]; x6 = lg->col[2];
                    x1 = x1 * x0; x2 = x2 * x0; x3 = x3 * x0;
                    x1 = x1 * x4; x2 = x2 * x5; x3 = x3 * x6;
                    x4 = x4 * x7; x5 = x5 * x7; x6 = x6 * x7;
                    x1 = x1 + x4; x2 = x2 + x5; x3 = x3 + x6;
                    c->col[0] = x1 + c->col[0];
                    c->col[1] = x2 + c->col[1];
                    c->col[2] = x3 + c->col[2];
                }
            }
        }
    }

    /* ---------------- transparency: 3185-3598 ---------------- */
    {
        float c_trn = mt->c_trn, c_rfl = mt->c_rfl;     /* ctx_C_TRN / ctx_C_RFL */
        u32 m_trn = 0xFFFFFFFFu;
        float r1 = 0.0f, r2 = 0.0f, r3 = 0.0f;         /* Xmm1..3 into TR_mix */

        if (!(props & QR_PROP_OPAQUE))
        {
            int do_rfi = (props & QR_PROP_REFRACT) || (props & QR_PROP_FRESNEL);
            int tir_all = 0;
            x0 = x4 = x6 = x7 = 0.0f;
            if (do_rfi)
            {
                /* TR_rfi 3212-3324 */
                FL(T, 65);
                x1 = c->ray[0]; x7 = x1 * x1; x0 = x7;
                x2 = c->ray[1]; x7 = x2 * x2; x0 = x0 + x7;
                x3 = c->ray[2]; x7 = x3 * x3; x0 = x0 + x7;
                x7 = rsq(x0);
                x1 = x1 * x7; x2 = x2 * x7; x3 = x3 * x7;
                x7 = x1 * c->nrm[0]; x0 = x7;
                x7 = x2 * c->nrm[1]; x0 = x0 + x7;
                x7 = x3 * c->nrm[2]; x0 = x0 + x7;
                x4 = x0;
                x6 = mt->c_rfr;
                x0 = x0 * x6;
                x7 = x0 * x0;
                x7 = x7 + 1.0f;
                x7 = x7 - mt->rfr_2;
                if (props & QR_PROP_FRESNEL)
                {
                    m_trn = cle(0.0f, x7) & m_trn;
                    if (m_trn == 0)
                    {
                        /* TR_tir 3280-3295 */
                        c_trn = 0.0f;
                        c_rfl = mt->c_rfl + mt->c_trn;
                        tir_all = 1;
                    }
                }
                if (!tir_all)
                {
                    x7 = sqrtf(x7);
                    x0 = x0 + x7;
                    if (props & QR_PROP_REFRACT)
                    {
                        x5 = c->nrm[0] * x0; x1 = x1 * x6; x1 = x1 - x5; c->nw[0] = x1;
                        x5 = c->nrm[1] * x0; x2 = x2 * x6; x2 = x2 - x5; c->nw[1] = x2;
                        x5 = c->nrm[2] * x0; x3 = x3 * x6; x3 = x3 - x5; c->nw[2] = x3;
                    }
                    else
                    {
                        c->nw[0] = c->ray[0]; c->nw[1] = c->ray[1]; c->nw[2] = c->ray[2];
                    }
                }
            }
            else
            {
                /* TR_rfe 3336-3347 */
                c->nw[0] = c->ray[0]; c->nw[1] = c->ray[1]; c->nw[2] = c->ray[2];
            }

            if (!tir_all)
            {
                /* TR_ini 3349-3426 */
                if (props & QR_PROP_FRESNEL)
                {
                    float f;
                    u32 u;
                    x1 = x4;
                    x2 = x1; x2 = x2 * x6; x2 = x2 - x7;
                    x7 = x7 * x6;
                    x3 = x1;
                    x1 = x1 + x7;
                    x3 = x3 - x7;
                    x0 = x0 / x2;
                    x1 = x1 / x3;
                    x0 = x0 * x0; x1 = x1 * x1;
                    x0 = x0 + x1;
                    x0 = x0 * -0.5f;
                    x0 = fand(x0, AMASK_ABS);
                    f = fand(x0, m_trn);
                    f = f * mt->c_trn;
                    u = ~m_trn & f2u(mt->c_trn);
                    f = u2f(f2u(f) | u);
                    c_trn = mt->c_trn - f;
                    c_rfl = mt->c_rfl + f;
                }
                /* TR_frn 3468-3552 */
                if (m_trn != 0 && T->depth != 0)
                {
                    ctx_t ch;
                    memset(&ch, 0, sizeof(ch));
                    ch.param_flg = c->local_flg | FLAG_PASS_THRU;
                    ch.param_obj = si;
                    ch.param_tag = 3;
                    ch.wmask = 0xFFFFFFFFu;
                    ch.t_buf = fr->t_max;
                    ch.t_min = 0.0f;
                    ch.org[0] = c->hit[0]; ch.org[1] = c->hit[1]; ch.org[2] = c->hit[2];
                    ch.ray[0] = c->nw[0];  ch.ray[1] = c->nw[1];  ch.ray[2] = c->nw[2];
                    ch.local_obj = QR_NULL; ch.pend_si = QR_NULL;
                    ch.hit_id = -1;
                    T->depth -= 1;
                    T->cnt.refract++;
                    trace_list(T, &ch, &c->nrm[3], s->lst[(1 - side) * 2 + 1]);
                    T->depth += 1;
                    r1 = ch.col[0] * c_trn;
                    r2 = ch.col[1] * c_trn;
                    r3 = ch.col[2] * c_trn;
                }
            }
        }

        /* TR_mix 3564-3598 */
        x0 = 1.0f - mt->c_trn;
        x0 = x0 - mt->c_rfl;
        x0 = fand(x0, cle(0.0f, x0));
        c->col[0] = r1 + c->col[0] * x0;
        c->col[1] = r2 + c->col[1] * x0;
        c->col[2] = r3 + c->col[2] * x0;

        /* ---------------- reflections: 3604-3930 ---------------- */
        if ((props & QR_PROP_REFLECT) ||
            (!(props & QR_PROP_OPAQUE) && (props & QR_PROP_FRESNEL)))
        {
            /* RF_ini */
            FL(T, 24);
            x1 = c->ray[0]; x4 = c->nrm[0]; x7 = x1 * x1; x0 = x7;
            x2 = c->ray[1]; x5 = c->nrm[1]; x7 = x2 * x2; x0 = x0 + x7;
            x3 = c->ray[2]; x6 = c->nrm[2]; x7 = x3 * x3; x0 = x0 + x7;
            x7 = rsq(x0);
            x1 = x1 * x7; x2 = x2 * x7; x3 = x3 * x7;
            x7 = x1 * x4; x0 = x7;
            x7 = x2 * x5; x0 = x0 + x7;
            x7 = x3 * x6; x0 = x0 + x7;
            x4 = x4 * x0; x1 = x1 - x4; x1 = x1 - x4; c->nw[0] = x1;
            x5 = x5 * x0; x2 = x2 - x5; x2 = x2 - x5; c->nw[1] = x2;
            x6 = x6 * x0; x3 = x3 - x6; x3 = x3 - x6; c->nw[2] = x3;

            if ((props & QR_PROP_FRESNEL) && (props & QR_PROP_OPAQUE))
            {
                if (props & QR_PROP_METAL)
                {
                    /* Fresnel for metals fast: 3729-3751 */
                    FL(T, 16);
                    x6 = mt->c_rcp;
                    x4 = x0; x4 = x4 * x6; x4 = x4 + x4;
                    x0 = x0 * x0;
                    x6 = x6 * x6;
                    x6 = x6 + mt->ext_2;
                    x1 = x0; x1 = x1 * x6;
                    x0 = x0 + x6;
                    x1 = x1 + 1.0f;
                    x2 = x0; x3 = x1;
                    x0 = x0 + x4; x1 = x1 + x4;
                    x2 = x2 - x4; x3 = x3 - x4;
                    x0 = x0 / x2; x1 = x1 / x3;
                    x0 = x0 + x1;
                    x0 = x0 * -0.5f;
                    x0 = fand(x0, AMASK_ABS);
                }
                else
                {
                    /* RF_mtl: Fresnel for opaque plain 3765-3798 */
                    FL(T, 16);
                    x4 = x0;
                    x6 = mt->c_rfr;
                    x0 = x0 * x6;
                    x7 = x0 * x0;
                    x7 = x7 + 1.0f;
                    x7 = x7 - mt->rfr_2;
                    x7 = sqrtf(x7);
                    x0 = x0 + x7;
                    x1 = x4;
                    x2 = x1; x2 = x2 * x6; x2 = x2 - x7;
                    x7 = x7 * x6;
                    x3 = x1;
                    x1 = x1 + x7;
                    x3 = x3 - x7;
                    x0 = x0 / x2; x1 = x1 / x3;
                    x0 = x0 * x0; x1 = x1 * x1;
                    x0 = x0 + x1;
                    x0 = x0 * -0.5f;
                    x0 = fand(x0, AMASK_ABS);
                }
                /* RF_pre 3806-3815 */
                x0 = x0 - 1.0f;
                x0 = x0 * mt->c_rfl;
                c_rfl = mt->c_rfl + x0;
            }

            /* RF_frn 3819-3908 */
            r1 = r2 = r3 = 0.0f;
            if (T->depth != 0)
            {
                ctx_t ch;
                memset(&ch, 0, sizeof(ch));
                ch.param_flg = c->local_flg;            /* | RT_FLAG_PASS_BACK (0) */
                ch.param_obj = si;
                ch.param_tag = 2;
                ch.wmask = 0xFFFFFFFFu;
                ch.t_buf = fr->t_max;
                ch.t_min = 0.0f;
                ch.org[0] = c->hit[0]; ch.org[1] = c->hit[1]; ch.org[2] = c->hit[2];
                ch.ray[0] = c->nw[0];  ch.ray[1] = c->nw[1];  ch.ray[2] = c->nw[2];
                ch.local_obj = QR_NULL; ch.pend_si = QR_NULL;
                ch.hit_id = -1;
                T->depth -= 1;
                T->cnt.reflect++;
                trace_list(T, &ch, &c->nrm[3], s->lst[side * 2 + 1]);
                T->depth += 1;
                r1 = ch.col[0] * c_rfl;
                r2 = ch.col[1] * c_rfl;
                r3 = ch.col[2] * c_rfl;
            }
            c->col[0] = r1 + c->col[0];
            c->col[1] = r2 + c->col[1];
            c->col[2] = r3 + c->col[2];
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------ */
/* quadric roots QD_rts..QD_mtr, tracer.cpp:4449-4842, for one lane          */
/* a,b,c,d as left by QD_ptr / TP_ptr (Xmm1, Xmm4, Xmm6, Xmm3)               */
/* Returns 1 when traversal must stop (shadow ray satisfied).                */
/* ------------------------------------------------------------------------ */

static int quadric_roots(tracer_t *T, ctx_t *c, int si, float a, float b, float cc, float d)
{
    const qr_surface *s = &T->s->v.srf[si];
    const u32 sm = s->smask;
    u32 xmask, dmask, m_pos, m_neg;
    float sd, bd, t1n, t1d, t2n, t2d, t1, t2;
    u32 t1msk = 0, t2msk = 0;
    int order_inner_first, pass, sides;
    int kind = s->srf_t[1] == 1 ? NRM_PLANE : s->srf_t[1] == 2 ? NRM_QUADRIC : NRM_TWOPLANE;

    xmask = cle(0.0f, d) & c->wmask;
    if (xmask == 0) return 0;
    FL(T, 10);

    b = fxor(b, sm);                                /* -b */
    dmask = clt(d, s->d_eps) & xmask;
    c->dmask = dmask;

    /* b-mixed quads: 4518-4547 */
    sd = fxor(sqrtf(d), sm & f2u(b));
    bd = b + sd;
    m_pos = cle(0.0f, sd);
    m_neg = cgt(0.0f, sd);
    t2n = u2f((f2u(cc) & m_neg) | (f2u(bd) & m_pos));   /* Xmm6 */
    t1n = u2f((f2u(bd) & m_neg) | (f2u(cc) & m_pos));   /* Xmm4 */
    t2d = u2f((f2u(bd) & m_neg) | (f2u(a) & m_pos));    /* Xmm3 */
    t1d = u2f((f2u(a) & m_neg) | (f2u(bd) & m_pos));    /* Xmm1 */
    /* Xmm0 = (a & m_pos) | (a & m_neg) */
    a = u2f((f2u(a) & m_pos) | (f2u(a) & m_neg));

    /* root sorting for near-zero determinant: 4572-4623 */
    c->xmisc_ptr = 0;
    t1 = t1n; t2 = t2n;
    if (dmask != 0)
    {
        u32 z, f;
        float tdf, eps;
        c->xmisc_ptr = 1;
        c->amask = sm & f2u(a);
        z = ceq(t1n, 0.0f);
        t1d = u2f((f2u(t1d) & ~z) | (z & f2u(1.0f)));
        z = ceq(t2n, 0.0f);
        t2d = u2f((f2u(t2d) & ~z) | (z & f2u(1.0f)));
        t1 = t1n / t1d;
        t2 = t2n / t2d;
        t1msk = cne(t1d, 0.0f);
        t2msk = cne(t2d, 0.0f);
        tdf = t1 - t2;
        tdf = fxor(tdf, c->amask);
        f = cle(0.0f, tdf);
        tdf = fand(tdf, f);
        eps = u2f(f & f2u(s->t_eps));
        eps = eps * t1;
        eps = fand(eps, AMASK_ABS);
        tdf = tdf * -0.5f;
        tdf = tdf - eps;
        tdf = fxor(tdf, c->amask);
        tdf = fand(tdf, t1msk & t2msk & dmask);
        t1 = t1 + tdf;
        t2 = t2 - tdf;
    }

    /* a-mixed quads: per lane the order is decided by the sign of a, 4646-4658 */
    order_inner_first = (cgt(0.0f, a) & xmask) != 0;
    sides = 2;                                       /* ctx_XMISC(FLG) */

    for (pass = 0; pass < 2 && sides > 0; pass++)
    {
        int inner = order_inner_first ? (pass == 0) : (pass == 1);
        u32 m;
        float t;
        sides--;

        /* CHECK_SIDE 531-540 */
        if (si == c->param_obj)
        {
            int f = c->param_flg & (FLAG_SIDE | FLAG_PASS_THRU);
            if (f == 1 - inner || f == 2 + inner) continue;
        }
        if (!inner)
        {
            if (c->xmisc_ptr == 0) { t1 = t1n / t1d; t1msk = cne(t1d, 0.0f); }
            t = t1; m = xmask & t1msk;
        }
        else
        {
            if (c->xmisc_ptr == 0) { t2 = t2n / t2d; t2msk = cne(t2d, 0.0f); }
            t = t2; m = xmask & t2msk;
        }
        c->t_val = t;
        c->local_flg = inner;                        /* RT_FLAG_SIDE_OUTER/INNER */
        m = clip(T, c, si, m);
        if (m == 0) continue;
        if (shade(T, c, si, inner, kind)) return 1;
        /* overdraw optimisation 4733-4740: this lane is done with the surface */
        break;
    }
    return 0;
}

/* ------------------------------------------------------------------------ */
/* OO_cyc: object list traversal, tracer.cpp:1341-1592, 3955-4054, 5137-5140 */
/* ------------------------------------------------------------------------ */

static void trace_list(tracer_t *T, ctx_t *c, const float *parent_loc, int head)
{
    const qr_scene_view *v = &T->s->v;
    int e = head;

    while (e != QR_NULL)
    {
        const qr_elem *el = &v->elm[e];
        const int si = el->simd;
        const qr_surface *s = &v->srf[si];
        const int sh = s->shift ? 3 : 0;
        int solver;

        /* secondary ray leaving this very surface: reuse the parent's local
         * hit as diff, 1352-1373 */
        if (si == c->param_obj && parent_loc != NULL)
        {
            c->dff[sh + 0] = parent_loc[0];
            c->dff[sh + 1] = parent_loc[1];
            c->dff[sh + 2] = parent_loc[2];
        }

        if (s->srf_t[3] >= 0 && c->local_obj != QR_NULL)
        {
            /* transform caching from trnode: 1385-1417 */
            FL(T, 3);
            if (si != c->param_obj)
            {
                c->dff[3] = c->dff[0] - s->pos[0];
                c->dff[4] = c->dff[1] - s->pos[1];
                c->dff[5] = c->dff[2] - s->pos[2];
            }
            if (e == c->local_obj) c->local_obj = QR_NULL;
        }
        else
        {
            /* OO_dff 1419-1556 */
            int do_ray = 1;
            if (si != c->param_obj)
            {
                float d[3], p[3];
                d[0] = c->org[0] - s->pos[0];
                d[1] = c->org[1] - s->pos[1];
                d[2] = c->org[2] - s->pos[2];
                FL(T, 3);
                c->dff[0] = d[0]; c->dff[1] = d[1]; c->dff[2] = d[2];
                if (s->has_trm == 0)
                {
                    do_ray = 0;                     /* -> OO_trm */
                }
                else
                {
                    xform(s, d, p);
                    FL(T, s->has_trm == 1 ? 3 : 15);
                    if (s->srf_t[3] < 0)
                    {
                        c->dff[0] = p[0]; c->dff[1] = p[1]; c->dff[2] = p[2];
                        c->local_obj = el->data;    /* trnode's last element */
                    }
                    else
                    {
                        c->dff[3] = p[0]; c->dff[4] = p[1]; c->dff[5] = p[2];
                    }
                }
            }
            if (do_ray)
            {
                xform(s, &c->ray[0], &c->ray[3]);   /* OO_ray 1508-1554 */
                FL(T, s->has_trm == 1 ? 3 : 15);
            }
        }

        /* OO_trm */
        if ((el->kind & 3) == 1)
        {
            /* AR_ptr: bounding volume 3955-4054 */
            float x0, x1, x2, x3, x4, x5, x6, x7;
            FL(T, 25);
            x1 = c->ray[sh + 0]; x0 = s->sci[0] * x1; x5 = c->dff[sh + 0]; x7 = s->sci[0] * x5;
            x3 = x1; x1 = x1 * x0; x3 = x3 * x7; x5 = x5 * x7;
            x2 = c->ray[sh + 1]; x0 = s->sci[1] * x2; x6 = c->dff[sh + 1]; x7 = s->sci[1] * x6;
            x4 = x2; x2 = x2 * x0; x4 = x4 * x7; x6 = x6 * x7;
            x1 = x1 + x2; x3 = x3 + x4; x5 = x5 + x6;
            x2 = c->ray[sh + 2]; x0 = s->sci[2] * x2; x6 = c->dff[sh + 2]; x7 = s->sci[2] * x6;
            x4 = x2; x2 = x2 * x0; x4 = x4 * x7; x6 = x6 * x7;
            x1 = x1 + x2; x3 = x3 + x4; x5 = x5 + x6;
            x5 = x5 - s->sci[3];
            x5 = x5 * x1;
            x3 = x3 * x3;
            x3 = x3 - x5;
            if ((cle(0.0f, x3) & c->wmask) == 0)
            {
                /* AR_skp: skip array's contents */
                e = el->data;
                if (e == c->local_obj) c->local_obj = QR_NULL;
            }
            e = v->elm[e].next;
            continue;
        }

        solver = s->srf_t[0];
        if (solver == 1)
        {
            /* PL_ptr 4062-4136 */
            if (si != c->param_obj)
            {
                const int mk = ax_map(s, 2);
                const u32 sg = ax_sgn(s, 2);
                float dk = fxor(c->dff[sh + mk], sg);
                float rk = fxor(c->ray[sh + mk], sg);
                u32 m;
                FL(T, 1);
                dk = fxor(dk, s->smask);
                m = cne(0.0f, rk) & c->wmask;
                c->t_val = dk / rk;
                m = clip(T, c, si, m);
                if (m != 0)
                {
                    int inner = clt(rk, 0.0f) ? 0 : 1;
                    c->local_flg = inner;
                    if (shade(T, c, si, inner, NRM_PLANE)) return;
                }
            }
        }
        else if (solver == 2)
        {
            /* QD_ptr 4378-4447 */
            float x0, x1, x2, x3, x4, x5, x6, x7;
            FL(T, 31);
            x1 = c->ray[sh + 0]; x0 = s->sci[0] * x1; x5 = c->dff[sh + 0]; x7 = s->sci[0] * x5;
            x7 = x7 - s->scj[0]; x3 = x1; x1 = x1 * x0; x3 = x3 * x7; x7 = x7 - s->scj[0]; x5 = x5 * x7;
            x2 = c->ray[sh + 1]; x0 = s->sci[1] * x2; x6 = c->dff[sh + 1]; x7 = s->sci[1] * x6;
            x7 = x7 - s->scj[1]; x4 = x2; x2 = x2 * x0; x4 = x4 * x7; x7 = x7 - s->scj[1]; x6 = x6 * x7;
            x1 = x1 + x2; x3 = x3 + x4; x5 = x5 + x6;
            x2 = c->ray[sh + 2]; x0 = s->sci[2] * x2; x6 = c->dff[sh + 2]; x7 = s->sci[2] * x6;
            x7 = x7 - s->scj[2]; x4 = x2; x2 = x2 * x0; x4 = x4 * x7; x7 = x7 - s->scj[2]; x6 = x6 * x7;
            x1 = x1 + x2; x3 = x3 + x4; x5 = x5 + x6;
            x5 = x5 - s->sci[3];
            x6 = x5; x5 = x5 * x1; x4 = x3; x3 = x3 * x3; x3 = x3 - x5;
            if (quadric_roots(T, c, si, x1, x4, x6, x3)) return;
        }
        else if (solver == 3)
        {
            /* TP_ptr 4216-4277 */
            const int mi = ax_map(s, 0), mk = ax_map(s, 2);
            float x0, x1, x2, x3, x4, x5, x6, x7;
            FL(T, 21);
            x1 = c->ray[sh + mi]; x5 = c->dff[sh + mi]; x3 = s->sci[mi];
            x2 = c->ray[sh + mk]; x6 = c->dff[sh + mk]; x4 = s->sci[mk];
            x0 = x5; x7 = x6;
            x6 = x6 * x1; x5 = x5 * x2; x5 = x5 - x6; x5 = x5 * x5; x5 = x5 * x3; x5 = x5 * x4;
            x5 = fand(x5, AMASK_ABS);
            x6 = x3; x3 = x3 * x0; x4 = x4 * x7; x3 = x3 * x1; x4 = x4 * x2; x3 = x3 + x4;
            x4 = s->sci[mk];
            x0 = x0 * x0; x7 = x7 * x7; x0 = x0 * x6; x7 = x7 * x4; x0 = x0 + x7;
            x1 = x1 * x1; x2 = x2 * x2; x1 = x1 * x6; x2 = x2 * x4; x1 = x1 + x2;
            if (quadric_roots(T, c, si, x1, x3, x0, x5)) return;
        }
        /* solver 0: trnode marker, skipped (1574-1584) */

        e = el->next;
    }

    if (T->deferred && c->pend_si != QR_NULL)
    {
        /* shade the final hit once, from the same inputs the eager path had */
        const qr_surface *s = &v->srf[c->pend_si];
        const int sh = s->shift ? 3 : 0;
        const float t = c->pend_t;
        float x;
        x = c->ray[0] * t; c->hit[0] = x + c->org[0];
        x = c->ray[1] * t; c->hit[1] = x + c->org[1];
        x = c->ray[2] * t; c->hit[2] = x + c->org[2];
        c->nw[sh + 0] = c->pend_loc[0]; c->nw[sh + 1] = c->pend_loc[1]; c->nw[sh + 2] = c->pend_loc[2];
        c->t_val = t;
        c->in_final = 1;
        shade(T, c, c->pend_si, c->pend_side, c->pend_kind);
    }
}

/* ------------------------------------------------------------------------ */
/* frame loop: YY_cyc / XX_cyc / XX_end, tracer.cpp:1142-1339, 5161-5343     */
/* ------------------------------------------------------------------------ */

static void sample(tracer_t *T, int x, int y, int k, float col[3], int *hit_id)
{
    const qr_scene_view *v = &T->s->v;
    const qr_frame *fr = v->frame;
    ctx_t c;
    float hs, vs, x1, x2, x3, x4, x5, x6;
    int ai = 0, tile;

    if (fr->fsaa == 1) ai = (x & 1) * 2 + k;        /* engine.cpp:3489-3510 */
    if (fr->fsaa == 2) ai = k;                      /* engine.cpp:3525-3546 */

    memset(&c, 0, sizeof(c));
    c.t_buf = fr->t_max;
    c.t_min = fr->t_min;
    c.org[0] = fr->org[0]; c.org[1] = fr->org[1]; c.org[2] = fr->org[2];
    c.wmask = 0xFFFFFFFFu;
    c.param_tag = 0;
    c.param_flg = fr->ctx_flags;
    c.param_obj = QR_NULL;
    c.local_obj = QR_NULL;
    c.pend_si = QR_NULL;
    c.hit_id = -1;

    /* primary ray 1287-1322 */
    FL(T, 16);
    hs = (float)x + fr->hor_a[ai]; hs = hs + 0.0f;
    vs = (float)y + fr->ver_a[ai]; vs = vs + 0.0f;
    x1 = fr->hor[0] * hs; x2 = fr->hor[1] * hs; x3 = fr->hor[2] * hs;
    x4 = fr->ver[0] * vs; x5 = fr->ver[1] * vs; x6 = fr->ver[2] * vs;
    x1 = x1 + x4; x2 = x2 + x5; x3 = x3 + x6;
    c.ray[0] = x1 + fr->dir[0];
    c.ray[1] = x2 + fr->dir[1];
    c.ray[2] = x3 + fr->dir[2];

    tile = (y / fr->tile_h) * fr->tls_row + (x / fr->tile_w);
    T->cnt.primary++;
    trace_list(T, &c, NULL, v->tiles[tile]);
    col[0] = c.col[0]; col[1] = c.col[1]; col[2] = c.col[2];
    *hit_id = c.hit_id;
}

static inline float clamp1(float x) { return x < 1.0f ? x : 1.0f; }     /* minps */

static u32 pixel(tracer_t *T, int x, int y, int *hit_id)
{
    const qr_frame *fr = T->s->v.frame;
    const int ns = 1 << fr->fsaa;
    float s[4][3], col[3];
    int k, ch, id = -1;
    u32 out = 0;
    static const int shft[3] = { 16, 8, 0 };

    for (k = 0; k < ns; k++)
    {
        int h;
        sample(T, x, y, k, s[k], &h);
        if (k == 0) id = h;
        for (ch = 0; ch < 3; ch++) s[k][ch] = clamp1(s[k][ch]);
    }
    FL(T, 6 + 2 * fr->fsaa);
    /* AA_cyc 5241-5308: per pass halve, then add neighbours */
    for (ch = 0; ch < 3; ch++)
    {
        if (ns == 1) col[ch] = s[0][ch];
        else if (ns == 2) col[ch] = s[0][ch] * 0.5f + s[1][ch] * 0.5f;
        else
        {
            float p0 = s[0][ch] * 0.5f + s[1][ch] * 0.5f;
            float p1 = s[2][ch] * 0.5f + s[3][ch] * 0.5f;
            col[ch] = p0 * 0.5f + p1 * 0.5f;
        }
    }
    /* FRAME_SIMD 988-1006 */
    for (ch = 0; ch < 3; ch++)
    {
        float t = col[ch];
        if (fr->ctx_flags & QR_PROP_GAMMA) t = sqrtf(t);
        t = t * fr->clamp;
        out |= ((u32)cvt_near(t) & fr->cmask) << shft[ch];
    }
    *hit_id = id;
    return out;
}

/* ------------------------------------------------------------------------ */
/* public entry points (ctypes)                                              */
/* ------------------------------------------------------------------------ */

/*
 * Render the snapshot `blob` into `frame` (compact stride frm_w).
 * depth < 0 keeps the snapshot's depth.  rows [row_begin,row_end) with
 * (y % thnum) == index; pass 0, frm_h, and the snapshot's index/thnum (or 0,1).
 * ids (optional) receives (surface<<1|side) of the visible primary hit.
 * counts (optional) = {primary, shadow, reflect, refract} rays.
 * deferred != 0: shade only the final hit of each list walk (same pixels, fewer
 * rays: the "useful ray" count the HIP backend also reports).
 */
static uint64_t g_last_flops;

/* fp32 operation count (SURVEY.md 8(d) weights) of the most recent qro_render/qro_render2 call */
uint64_t qro_last_flops(void) { return g_last_flops; }

int qro_render2(const void *blob, uint64_t size, uint32_t *frame, int32_t *ids,
                int depth, int row_begin, int row_end, int index, int thnum,
                int threads, uint64_t counts[4], int deferred)
{
    scene_t S;
    int rc = qr_scene_view_init(&S.v, blob, size);
    int y, w, h;
    uint64_t c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0;
    if (rc != 0) return rc;
    S.depth = depth >= 0 ? depth : S.v.frame->depth;
    w = S.v.frame->frm_w; h = S.v.frame->frm_h;
    if (row_begin < 0) row_begin = 0;
    if (row_end > h) row_end = h;
    if (thnum <= 0) { thnum = 1; index = 0; }
    (void)threads;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel for schedule(dynamic, 4) reduction(+:c0,c1,c2,c3,c4)
    for (y = row_begin; y < row_end; y++)
    {
        tracer_t T;
        int x;
        if ((y % thnum) != index) continue;
        T.s = &S; T.depth = S.depth; T.deferred = deferred;
        memset(&T.cnt, 0, sizeof(T.cnt));
        for (x = 0; x < w; x++)
        {
            int id;
            u32 p = pixel(&T, x, y, &id);
            frame[(size_t)y * w + x] = p;
            if (ids) ids[(size_t)y * w + x] = id;
        }
        c0 += T.cnt.primary; c1 += T.cnt.shadow; c2 += T.cnt.reflect; c3 += T.cnt.refract; c4 += T.cnt.flops;
    }
    g_last_flops = c4;
    if (counts) { counts[0] = c0; counts[1] = c1; counts[2] = c2; counts[3] = c3; }
    return 0;
}

/* reference semantics: every hit that passes the depth test is shaded (overdraw) */
int qro_render(const void *blob, uint64_t size, uint32_t *frame, int32_t *ids,
               int depth, int row_begin, int row_end, int index, int thnum,
               int threads, uint64_t counts[4])
{
    return qro_render2(blob, size, frame, ids, depth, row_begin, row_end, index, thnum, threads, counts, 0);
}

int qro_info(const void *blob, uint64_t size, int32_t out[8])
{
    qr_scene_view v;
    int rc = qr_scene_view_init(&v, blob, size);
    if (rc != 0) return rc;
    out[0] = v.frame->frm_w; out[1] = v.frame->frm_h; out[2] = v.frame->fsaa; out[3] = v.frame->depth;
    out[4] = (int32_t)v.hdr->n_srf; out[5] = (int32_t)v.hdr->n_elm; out[6] = v.frame->index; out[7] = v.frame->thnum;
    return 0;
}

/* FNV-1a 64 over (pixel & 0xFFFFFF) as 4 little-endian bytes, row-major */
uint64_t qro_hash(const uint32_t *frame, uint64_t n)
{
    uint64_t h = 0xcbf29ce484222325ull, i;
    for (i = 0; i < n; i++)
    {
        uint32_t v = frame[i] & 0x00FFFFFFu;
        int b;
        for (b = 0; b < 4; b++) { h ^= (v >> (8 * b)) & 0xFF; h *= 0x100000001b3ull; }
    }
    return h;
}
