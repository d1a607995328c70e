/*
 * qr_shade.hpp - device code, shading of the final hit of a walk (tracer.cpp:2166-3930 without the child
 * packets): normal, texture, lights with shadow traversals, refraction + Fresnel, reflection + Fresnel.
 * Included by qr_kernel.hpp after qr_walk.hpp.
 */
#ifndef QR_SHADE_HPP
#define QR_SHADE_HPP

/* ------------------------------------------------------------------------ */
/* shading of the final hit, tracer.cpp:2166-3930 without the child packets  */
/* ------------------------------------------------------------------------ */

/*
 * One level of a ray's recursion (the reference's context stack, tracer.h:426-665), 16 dwords in four 16-byte
 * quarters.  The two shallowest levels of every lane live in LDS -- a one-wave workgroup has 10 KB of it to itself at
 * 16 waves per CU, and most recursion ends at depth 1-2 -- the deeper ones in scratch.  A return reads q0 and q1;
 * q2 and q3 are only read when a node with a refraction child also has a reflection child to start.
 */
struct Frame
{
    float col[3]; int meta;             /* q0: colour so far; si << 4 | side << 3 | rf << 2 | phase (1 TR, 2 RF) */
    float c_trn, c_rfl, x0, hit0;       /* q1 */
    float rdir[3]; float hit1;          /* q2: the reflection child's direction */
    float loc[3]; float hit2;           /* q3: the local hit (the child's ploc) */
};
#ifndef QR_LDS_NARROW_LEVELS
#define QR_LDS_NARROW_LEVELS 4  /* levels whose first two quarters live in LDS (2 KB per level and wave); the kernel instance with the
                                 * per-lane walks adds 2.5 KB for walk_pool / walk_dda: 10.5 KB per wave, 12 waves per CU fit */
#endif
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Shaded
{
    V3 col;                 /* local colour after lights                      */
    V3 hit;                 /* world hit = child origin                       */
    V3 loc;                 /* local hit = child's ploc                       */
    V3 tdir;                /* refraction child direction (ctx_NEW after TR)  */
    V3 rdir;                /* reflection child direction (ctx_NEW after RF)  */
    float c_trn, c_rfl, x0;
    bool want_tr;           /* refraction child exists (M_TRN, not opaque)    */
    bool want_rf;           /* reflection pass applies (RF_ini reached)       */
    u32  lst_tr, lst_rf;    /* byte offsets of the children's list programs */
    /* path tracer (PT instance only): the diffuse bounce, tracer.cpp:2339-2640 */
    bool want_pt;           /* a bounce ray was sampled                       */
    V3 pdir;                /* its direction                                  */
    V3 ptw;                 /* l_dff * texture colour (after Russian roulette): weight of what it returns */
    V3 ptex; float pldff;   /* the two factors apart (the eager path tracer multiplies in the reference's order) */
    u32 lst_pt;             /* its list: the surface's own side, LST_P(SRF)   */
    bool rr_dead;           /* Russian roulette ended the sample here: no bounce, and no refraction / reflection either */
};

/*
 * Path tracer: the reference's 24-bit LCG (GET_RANDOM, tracer.cpp:1013-1027; constants engine.cpp:866-874): the
 * state advances, its upper 24 bits / 2^24 are the number.  One state per pixel sample, seeded like
 * rt_Scene::reset_pseed (engine.cpp:3670-3685) and kept between frames.
 */
__device__ __forceinline__ float pt_random(u32 &state)
{
    state = state * 214013u + 2531011u;
    return (float)(int32_t)((state >> 8) & 0xFFFFFFu) / 16777216.0f;
}
/* the reference's power series for sin / cos (tracer.cpp:1031-1057), used as they are: their error at +-pi is part of
 * the distribution of the bounce directions */
__device__ __forceinline__ float pt_sin(float x)
{
    /* fmaps3ld is a fused multiply-add on the reference's AVX2 / AVX-512 targets (vfmadd231ps,
     * rtarch_x86_512x1v2.h:572-581, RT_SIMD_COMPAT_FMA 1): one rounding per term */
    const float t = x * x; float d = x, s = x * t;
    d = __builtin_fmaf(s, -0.1666666666666666666666666666666666666666666f, d); s = s * t;
    d = __builtin_fmaf(s, +0.0083333333333333333333333333333333333333333f, d); s = s * t;
    d = __builtin_fmaf(s, -0.0001984126984126984126984126984126984126984f, d); s = s * t;
    d = __builtin_fmaf(s, +0.0000027557319223985890652557319223985890652f, d);
    return d;
}
__device__ __forceinline__ float pt_cos(float x)
{
    const float t = x * x; float d = 1.0f, s = t;
    d = __builtin_fmaf(s, -0.5f, d); s = s * t;
    d = __builtin_fmaf(s, +0.0416666666666666666666666666666666666666666f, d); s = s * t;
    d = __builtin_fmaf(s, -0.0013888888888888888888888888888888888888888f, d); s = s * t;
    d = __builtin_fmaf(s, +0.0000248015873015873015873015873015873015873f, d);
    return d;
}

/* LT_amb .. LT_end for one light that reaches the hit, tracer.cpp:2879-3156: attenuated diffuse term, specular term raised to
 * the material's 28.4 fixed-point power, metal / plain colour blend, added to `col` */
__device__ __forceinline__ void light_terms(const char *__restrict__ G, u32 mo, int props, const qr_light *__restrict__ lg,
                                            const V3 &L, float dot, const Ray &r, const V3 &nrm, const V3 &tex, V3 &col)
{
    const qr_material *__restrict__ mt = (const qr_material *)(G + mo);
    float x0, x1, x2, x3, x4, x5, x6, x7;
    x1 = L.x; x4 = x1 * x1;
    x2 = L.y; x5 = x2 * x2;
    x3 = L.z; x6 = x3 * x3;
    x4 = x4 + x5; x4 = x4 + x6;
    const float r2 = x4;
    x0 = dot;
    QR_FLOPS(10);
    if (props & QR_PROP_DIFFUSE)
    {
        QR_FLOPS(17);
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
    bool plain = false;
    float spec = 0.0f;
    if (props & QR_PROP_SPECULAR)
    {
        x4 = x6; x5 = x6;
        x4 = x4 * nrm.x; x1 = x1 - x4; x1 = x1 - x4;
        x5 = x5 * nrm.y; x2 = x2 - x5; x2 = x2 - x5;
        x6 = x6 * nrm.z; x3 = x3 - x6; x3 = x3 - x6;
        x4 = r.dir.x; x1 = x1 * x4; x4 = x4 * x4;
        x5 = r.dir.y; x2 = x2 * x5; x5 = x5 * x5;
        x6 = r.dir.z; x3 = x3 * x6; x6 = x6 * x6;
        x6 = x6 + x4; x6 = x6 + x5;
        x1 = x1 + x2; x1 = x1 + x3;
        if (clt(0.0f, x1))
        {
            QR_FLOPS(32);
            x4 = r2;
            x5 = rsq(x6); x1 = x1 * x5;
            x5 = rsq(x4); x1 = x1 * x5;
            /* fixed-point 28.4 power, 2981-3039 */
            const u32 lpow = mt->l_pow;
            u32 pw = lpow & 0xF;
            x2 = x1; x4 = x1; x1 = 1.0f;
            while (pw != 0)
            {
                x4 = __builtin_sqrtf(x4);
                const u32 bit = pw & 0x8;
                pw = (pw << 1) & 0xF;
                if (bit) x1 = x1 * x4;
            }
            pw = lpow >> 4;
            if (pw != 0)
            {
                x3 = x1; x1 = 1.0f;
                do
                {
                    const u32 bit = pw & 1;
                    pw >>= 1;
                    if (bit) x1 = x1 * x2;
                    x2 = x2 * x2;
                }
                while (pw != 0);
                x1 = x1 * x3;
            }
            x1 = x1 * mt->l_spc;
            if (props & QR_PROP_METAL) { x0 = x0 + x1; }
            else { plain = true; spec = x1; }
        }
    }
    if (!plain)
    {
        x1 = tex.x * lg->col[0];
        x2 = tex.y * lg->col[1];
        x3 = tex.z * lg->col[2];
        x1 = x1 * x0; x2 = x2 * x0; x3 = x3 * x0;
        col.x = x1 + col.x; col.y = x2 + col.y; col.z = x3 + col.z;
    }
    else
    {
        x7 = spec;
        x1 = tex.x; x2 = tex.y; x3 = tex.z;
        x4 = lg->col[0]; x5 = lg->col[1]; x6 = lg->col[2];
        x1 = x1 * x0; x2 = x2 * x0; x3 = x3 * x0;
        x1 = x1 * x4; x2 = x2 * x5; x3 = x3 * x6;
        x4 = x4 * x7; x5 = x5 * x7; x6 = x6 * x7;
        x1 = x1 + x4; x2 = x2 + x5; x3 = x3 + x6;
        col.x = x1 + col.x; col.y = x2 + col.y; col.z = x3 + col.z;
    }
}

struct Counters { u32 primary, shadow, reflect, refract; };

/* state of the enclosing recursion that only has to survive a shade() call */
struct Outer { V3 ret; int hit_id, sp, mode; };

template <bool COUNT, bool DIVK, bool PT = false>
__device__ __forceinline__ void shade(const Ctx &cx, bool act, bool coherent, const Ray &r, const Hit &h,
                                      Shaded &o, Counters &cnt, u32 *rng = nullptr, int depth_left = 0, int pt_stage = 0)
{
    /* pt_stage (PT only): 3 = 2 for a sample the roulette of stage 1 ended (Shaded::rr_dead);
     * 0 everything in one call (the statistical path tracer); 1 up to the bounce sampling; 2 only the
     * transparency / reflection part -- the eager path tracer calls the two stages around the bounce's subtree, because
     * the Fresnel split draws its number after that subtree has drawn its own (tracer.cpp: 2339-2703 before 3428-3466) */
    /* per-lane (divergent) material data: vector loads at byte offsets from the blob base; everything
     * below is lane-private except the wave-wide shadow traversals in the light loop */
    const BaseP B = cx.B;
    const char *__restrict__ G = cx.G;
    const FrmP fr = c_frm(B);
    const u32 hsrf = act ? h.srf : QR_OFF_SRF;          /* lanes without a hit read surface 0: harmless */
    const int side = h.side;
    const DShade *__restrict__ sd = (const DShade *)(G + (cx.off_shade + ((hsrf - QR_OFF_SRF) >> 2)));   /* 32 B per 128 B */
    const DSurf *__restrict__ s = (const DSurf *)(G + hsrf);

    V3 nrm = {0, 0, 1};
    V3 tex = {0, 0, 0};
    V3 col = {0, 0, 0};
    V3 hit = {0, 0, 0};
    int props = 0;
    u32 mo = 0;                 /* byte offset of the hit side's material */
    u32 le = 0;                 /* byte offset of the current light-list entry, 0 = none */

    if (act)
    {
        const float t = h.t;
        float x0, x1, x2, x3, x4, x5, x6;
        x4 = r.dir.x * t; hit.x = x4 + r.org.x;
        x5 = r.dir.y * t; hit.y = x5 + r.org.y;
        x6 = r.dir.z * t; hit.z = x6 + r.org.z;

        props = side | (side ? s->props1 : s->props0);
        mo = sd->mat[side];
        const u32 fl = s->flags;
        const u32 tside = side ? QR_SMASK : 0u;
        const int has_trm = (int)DF_TRM(fl);
        const int nkind = (int)DF_NKIND(fl);
        float tu = 0.0f, tv = 0.0f;
        V3 ln = {0, 0, 0};                          /* normal in surface space */

        if (nkind == 1)
        {
            /* PL_mat 4139-4193 */
            if (props & QR_PROP_TEXTURE)
            {
                tu = fxor(vget(h.loc, (int)DF_MAP(fl, 0)), DF_SGN(fl, 0));
                tv = fxor(vget(h.loc, (int)DF_MAP(fl, 1)), DF_SGN(fl, 1));
            }
            x6 = fxor(1.0f, tside);
            vset(ln, (int)DF_MAP(fl, 2), fxor(x6, DF_SGN(fl, 2)));
        }
        else
        {
            /* QD_mat 4845-4905 / TP_mat 4280-4336 */
            x4 = h.loc.x * s->sci[0]; x5 = h.loc.y * s->sci[1]; x6 = h.loc.z * s->sci[2];
            if (nkind == 2)
            {
                x4 = x4 - s->scj[0]; x5 = x5 - s->scj[1]; x6 = x6 - s->scj[2];
            }
            x1 = x4 * x4; x2 = x5 * x5; x3 = x6 * x6;
            x1 = x1 + x2; x1 = x1 + x3;
            x0 = rsq(x1);
            x0 = fxor(x0, tside);
            ln.x = x4 * x0; ln.y = x5 * x0; ln.z = x6 * x0;
        }
        nrm = ln;
        if (has_trm != 0)
        {
            QR_FLOPS(24);
            /* MT_nrm 2184-2263: transposed trnode matrix */
            const DSurf *__restrict__ tr = (const DSurf *)(G + s->trn);
            const int ttrm = (int)DF_TRM(tr->flags);
            x1 = ln.x; x2 = ln.y; x3 = ln.z;
            x4 = tr->tci[0] * x1;
            x5 = tr->tcj[1] * x2;
            x6 = tr->tck[2] * x3;
            if (ttrm != 1)
            {
                x4 = x4 + tr->tcj[0] * x2;
                x4 = x4 + tr->tck[0] * x3;
                x5 = x5 + tr->tci[1] * x1;
                x5 = x5 + tr->tck[1] * x3;
                x6 = x6 + tr->tci[2] * x1;
                x6 = x6 + tr->tcj[2] * x2;
            }
            if (ttrm != 2)
            {
                x1 = x4 * x4; x2 = x5 * x5; x3 = x6 * x6;
                x1 = x1 + x2; x1 = x1 + x3;
                x0 = rsq(x1);
                x4 = x4 * x0; x5 = x5 * x0; x6 = x6 * x0;
            }
            nrm.x = x4; nrm.y = x5; nrm.z = x6;
        }

        /* MT_tex 2293-2327, PAINT_FRAG / PAINT_COLX 653-673 */
        const qr_material *__restrict__ mt = (const qr_material *)(G + mo);
        u32 toff = 0;
        if (props & QR_PROP_TEXTURE)
        {
            x4 = mt->t_map[0] ? tv : tu;
            x5 = mt->t_map[1] ? tv : tu;
            x4 = x4 - mt->xoffs; x5 = x5 - mt->yoffs;
            x4 = x4 * mt->xscal; x5 = x5 * mt->yscal;
            const int32_t iu = cvt_floor(x4) & (int32_t)mt->xmask;
            const int32_t iv = cvt_floor(x5) & (int32_t)mt->ymask;
            toff = (u32)iu + ((u32)iv << (mt->yshft & 31));
        }
        const u32 texel = *(const u32 *)(G + ((u32)mt->tex + toff * 4u));
        const u32 cmask = mt->cmask;
        const float clampv = mt->clamp;
        tex.x = (float)(int32_t)((texel >> 16) & cmask) / clampv;
        tex.y = (float)(int32_t)((texel >> 8) & cmask) / clampv;
        tex.z = (float)(int32_t)(texel & cmask) / clampv;
        if (props & QR_PROP_GAMMA) { tex.x = tex.x * tex.x; tex.y = tex.y * tex.y; tex.z = tex.z * tex.z; }

        if (props & QR_PROP_LIGHT)
        {
            col = tex;                              /* LT_set */
        }
        else
        {
            col.x = tex.x * fr->fr.amb[0];
            col.y = tex.y * fr->fr.amb[1];
            col.z = tex.z * fr->fr.amb[2];
            le = sd->lgt[side];
        }
    }

    o.want_pt = false; o.pdir = {0, 0, 0}; o.ptw = {0, 0, 0}; o.lst_pt = 0; o.ptex = {0, 0, 0}; o.pldff = 0.0f;
    o.rr_dead = PT && pt_stage == 3;
    if constexpr (PT)
    {
        /* path tracer, tracer.cpp:2339-2690: no light loop; the local colour is the material's emission, a diffuse
         * surface samples one bounce over the cosine-weighted hemisphere */
        le = 0;
        col = {0, 0, 0};
        if (act)
        {
            const qr_material *__restrict__ mt = (const qr_material *)(G + mo);
            V3 t3 = tex;
            bool go = (props & QR_PROP_DIFFUSE) != 0 && pt_stage < 2;
            if (go && depth_left <= QR_MAX_DEPTH - 5)
            {
                /* Russian roulette from the sixth level on: survive with the largest colour component */
                float p = tex.x > tex.y ? tex.x : tex.y; p = p > tex.z ? p : tex.z;
                const float u = pt_random(*rng);
                go = u < p;
                /* the survivors' mask stays the lane mask of the transparency and reflection blocks too (ctx_F_PRB,
                 * tracer.cpp:2364-2366, 3193-3198): a sample the roulette ends spawns no child of any kind */
                o.rr_dead = !go;
                const float ip = 1.0f / p;
                t3.x = tex.x * ip; t3.y = tex.y * ip; t3.z = tex.z * ip;
            }
            if (go)
            {
                /* orthonormal basis around the normal: u = normalize(n x ray), v = n x u */
                V3 u, v;
                u.x = nrm.y * r.dir.z - nrm.z * r.dir.y;
                u.y = nrm.z * r.dir.x - nrm.x * r.dir.z;
                u.z = nrm.x * r.dir.y - nrm.y * r.dir.x;
                const float il = rsq(u.x * u.x + u.y * u.y + u.z * u.z);
                u.x = u.x * il; u.y = u.y * il; u.z = u.z * il;
                v.x = nrm.y * u.z - nrm.z * u.y;
                v.y = nrm.z * u.x - nrm.x * u.z;
                v.z = nrm.x * u.y - nrm.y * u.x;
                const float r1 = pt_random(*rng);
                const float s1 = __builtin_sqrtf(r1), c1 = __builtin_sqrtf(1.0f - r1);
                const float r2 = pt_random(*rng);
                const float pi = 3.14159265358979323846f;
                const float phi = (r2 + r2) * pi - pi;
                const float cp = pt_cos(phi) * s1, sp = pt_sin(phi) * s1;
                o.pdir.x = nrm.x * c1 + u.x * cp + v.x * sp;
                o.pdir.y = nrm.y * c1 + u.y * cp + v.y * sp;
                o.pdir.z = nrm.z * c1 + u.z * cp + v.z * sp;
                o.want_pt = true;
                o.ptw.x = t3.x * mt->l_dff; o.ptw.y = t3.y * mt->l_dff; o.ptw.z = t3.z * mt->l_dff;
                o.ptex = t3; o.pldff = mt->l_dff;
                o.lst_pt = sd->lst[side];
            }
            col.x = mt->emis[0]; col.y = mt->emis[1]; col.z = mt->emis[2];
        }
    }

    QR_PROF_HIT(24);                    /* shade() calls */
    if (pt_stage != 2 && pt_stage != 3) QR_FLOPS_M(16 + 4 + 6, __popcll(__ballot(act)));            /* normal, texture look-up, ambient */
    /*
     * Shadow rays of SECONDARY hits, regrouped over the lanes of the wave (DIVK instance, rounds that are not all primary
     * rays).  The loop below walks the lights one position at a time and sends each light's shadow rays through the
     * traversal alone -- on the 10 000-object scene 20 of 64 lanes at the start of such a walk (half of a round's 30 hits
     * face a given light), 9.6 stepping on average, four walks per round.  Here the shadow rays of up to four light
     * positions of ALL hits of the round -- (hit, light) pairs -- are queued in LDS and dealt out to the lanes 64 at a time:
     * a lane traces a ray of ANOTHER lane's hit (the hit from its owner's registers through ds_bpermute, the light from its
     * list entry) and reports occlusion in a bit of the owner's LDS word.  The owner then adds its lights' terms in list
     * order with the arithmetic of the loop below, so colours keep every bit (tests: all synthetic-scene parity tests).
     */
    if constexpr (DIVK && !PT)
    {
        if (!coherent)
        {
            __shared__ unsigned short lq[256];      /* (owner lane << 2) | light position in the chunk */
            __shared__ u32 locc[64];                /* per owner: bit j = its j-th light of the chunk is occluded */
            const int lane = (int)(threadIdx.x & 63u);
            while (any_lane(le != 0))
            {
                /* ---- enumerate: which of its next four lights does each hit face (LT_cyc 2764-2790) ---- */
                u32 lmbits = 0, cur = le;
                int nl = 0, n_q = 0;
                locc[lane] = 0u;
#pragma nounroll
                for (int j = 0; j < 4; j++)
                {
                    if (!any_lane(cur != 0)) break;
                    const bool has = cur != 0;
                    const CLight cl = *(const CLight *)(G + cur);
                    const qr_light *__restrict__ lg = (const qr_light *)(G + (has ? (cl.lgt & ~QR_CLIGHT_LAST) : 0u));
                    bool lm = false;
                    if (has)
                    {
                        QR_FLOPS(8);
                        float x1, x2, x3, x0;
                        x1 = lg->pos[0] - hit.x; x1 = x1 * nrm.x;
                        x2 = lg->pos[1] - hit.y; x2 = x2 * nrm.y;
                        x3 = lg->pos[2] - hit.z; x3 = x3 * nrm.z;
                        x0 = x1; x0 = x0 + x2; x0 = x0 + x3;
                        lm = clt(0.0f, x0);
                        nl++;
                    }
                    if (COUNT) { if (lm) cnt.shadow++; }
                    if (QR_KNOB(2)) lm = false;
                    const lm_t m = LM(lm);
                    if (lm) { lq[n_q + lanes_below(m)] = (unsigned short)((lane << 2) | j); lmbits |= 1u << j; }
                    n_q += __popcll(m);
                    cur = (has && !(cl.lgt & QR_CLIGHT_LAST)) ? cur + (u32)sizeof(CLight) : 0u;
                }
                __syncthreads();
                /* ---- trace: 64 (hit, light) pairs at a time ---- */
                if (!QR_KNOB(1))
                {
#pragma nounroll
                    for (int b0 = 0; b0 < n_q; b0 += 64)
                    {
                        const bool ta = b0 + lane < n_q;
                        const u32 q = lq[ta ? b0 + lane : 0];
                        const int own = ta ? (int)(q >> 2) : lane;
                        const u32 jj = q & 3u;
                        /* the owner's hit, from its registers (every lane executes the shuffles) */
                        Ray sr;
                        sr.org.x = __shfl(hit.x, own); sr.org.y = __shfl(hit.y, own); sr.org.z = __shfl(hit.z, own);
                        sr.ploc.x = __shfl(h.loc.x, own); sr.ploc.y = __shfl(h.loc.y, own); sr.ploc.z = __shfl(h.loc.z, own);
                        sr.osrf = (u32)__shfl((int)hsrf, own); sr.oflg = __shfl(side, own);
                        const u32 ole = (u32)__shfl((int)le, own);
                        const CLight cl = *(const CLight *)(G + (ta ? ole + jj * (u32)sizeof(CLight) : 0u));
                        const qr_light *__restrict__ lg = (const qr_light *)(G + (ta ? (cl.lgt & ~QR_CLIGHT_LAST) : 0u));
                        sr.dir.x = lg->pos[0] - sr.org.x; sr.dir.y = lg->pos[1] - sr.org.y; sr.dir.z = lg->pos[2] - sr.org.z;
                        sr.tmin = 0.0f; sr.tmax = lg->t_max;
                        u32 sl = ta ? cl.shadow : 0u;
                        if (any_lane((sl & QR_LISTF_GRID) != 0))
                        {
                            if (sl & QR_LISTF_GRID)
                            {
                                const CGrid *__restrict__ gr = (const CGrid *)(G + (sl & ~31u));
                                const u32 cp = gr->comps;
                                const float la = (cp & 3u) == 0 ? sr.ploc.x : ((cp & 3u) == 1 ? sr.ploc.y : sr.ploc.z);
                                const float lb = ((cp >> 2) & 3u) == 0 ? sr.ploc.x : (((cp >> 2) & 3u) == 1 ? sr.ploc.y : sr.ploc.z);
                                int ia = cvt_floor((la - gr->org_a) * gr->inv_a), ib = cvt_floor((lb - gr->org_b) * gr->inv_b);
                                const int nx = (int)gr->nx, ny = (int)gr->ny;
                                ia = ia < 0 ? 0 : (ia >= nx ? nx - 1 : ia);
                                ib = ib < 0 ? 0 : (ib >= ny ? ny - 1 : ib);
                                sl = *(const u32 *)(G + (gr->table + (u32)(ib * nx + ia) * 4u));
                            }
                        }
                        sr.list = sl;
                        Hit sh; bool occ;
#ifdef QR_WAVETIME
                        const unsigned long long wt_s0 = __builtin_amdgcn_s_memrealtime();
#endif
                        traverse<true, DIVK>(B, ta, false, sr, sh, occ
#ifdef QR_STATS
                                             , cx.stats
#endif
                                             );
#ifdef QR_WAVETIME
                        if (__ffsll((long long)__ballot(true)) - 1 == (int)(threadIdx.x & 63u)) qr_wt_shadow += __builtin_amdgcn_s_memrealtime() - wt_s0;
#endif
                        if (ta && occ) atomicOr(&locc[own], 1u << jj);
                    }
                }
                __syncthreads();
                const u32 ob = locc[lane];
                /* ---- the lights' terms, in list order ---- */
#pragma nounroll
                for (int j = 0; j < 4; j++)
                {
                    if (!any_lane(j < nl)) break;
                    if (j < nl)
                    {
                        const CLight cl = *(const CLight *)(G + (le + (u32)j * (u32)sizeof(CLight)));
                        const qr_light *__restrict__ lg = (const qr_light *)(G + (cl.lgt & ~QR_CLIGHT_LAST));
                        if (((lmbits & ~ob) >> j) & 1u)
                        {
                            V3 L;
                            float x1, x2, x3, x0;
                            x1 = lg->pos[0] - hit.x; L.x = x1; x1 = x1 * nrm.x;
                            x2 = lg->pos[1] - hit.y; L.y = x2; x2 = x2 * nrm.y;
                            x3 = lg->pos[2] - hit.z; L.z = x3; x3 = x3 * nrm.z;
                            x0 = x1; x0 = x0 + x2; x0 = x0 + x3;
                            light_terms(G, mo, props, lg, L, x0, r, nrm, tex, col);
                        }
                    }
                }
                le = cur;
                __syncthreads();                /* locc is cleared again at the top */
            }
        }
    }
    /* lights, 2758-3156: wave-wide loop, per-lane light-list entries */
    while (any_lane(le != 0))
    {
        QR_PROF_HIT(25);                /* light rounds */
        const bool has = le != 0;
        const CLight cl = *(const CLight *)(G + le);            /* lanes without a light read the header: harmless */
        const qr_light *__restrict__ lg = (const qr_light *)(G + (has ? (cl.lgt & ~QR_CLIGHT_LAST) : 0u));
        V3 L = {0, 0, 0};
        float dot = 0.0f;
        bool lm = false;
        if (has)
        {
            QR_FLOPS(8);
            float x1, x2, x3, x0;
            x1 = lg->pos[0] - hit.x; L.x = x1; x1 = x1 * nrm.x;
            x2 = lg->pos[1] - hit.y; L.y = x2; x2 = x2 * nrm.y;
            x3 = lg->pos[2] - hit.z; L.z = x3; x3 = x3 * nrm.z;
            x0 = x1; x0 = x0 + x2; x0 = x0 + x3;
            dot = x0;
            lm = clt(0.0f, x0);
        }
        Ray sr;
        sr.org = hit; sr.dir = L; sr.tmin = 0.0f; sr.tmax = lg->t_max;
        u32 sl = has ? cl.shadow : 0u;
        if constexpr (DIVK)
        {
            /* shadow lists by hit position (CGrid, qr_program.h): large clipped planes of scenes with long lists */
            if (any_lane((sl & QR_LISTF_GRID) != 0))
            {
                if (sl & QR_LISTF_GRID)
                {
                    const CGrid *__restrict__ gr = (const CGrid *)(G + (sl & ~31u));
                    const u32 cp = gr->comps;
                    const float la = (cp & 3u) == 0 ? h.loc.x : ((cp & 3u) == 1 ? h.loc.y : h.loc.z);
                    const float lb = ((cp >> 2) & 3u) == 0 ? h.loc.x : (((cp >> 2) & 3u) == 1 ? h.loc.y : h.loc.z);
                    int ia = cvt_floor((la - gr->org_a) * gr->inv_a), ib = cvt_floor((lb - gr->org_b) * gr->inv_b);
                    const int nx = (int)gr->nx, ny = (int)gr->ny;
                    ia = ia < 0 ? 0 : (ia >= nx ? nx - 1 : ia);
                    ib = ib < 0 ? 0 : (ib >= ny ? ny - 1 : ib);
                    sl = *(const u32 *)(G + (gr->table + (u32)(ib * nx + ia) * 4u));
                }
            }
        }
        sr.list = sl; sr.osrf = hsrf; sr.oflg = side; sr.ploc = h.loc;
        Hit sh; bool occ;
        if (COUNT) { if (lm) cnt.shadow++; }
        if (QR_KNOB(2)) lm = false;
        if (QR_KNOB(1)) occ = false; else
        {
#ifdef QR_WAVETIME
            const unsigned long long wt_s0 = __builtin_amdgcn_s_memrealtime();
#endif
            traverse<true, DIVK>(B, lm, coherent, sr, sh, occ
#ifdef QR_STATS
                                  , cx.stats
#endif
                                  );
#ifdef QR_WAVETIME
            if (__ffsll((long long)__ballot(true)) - 1 == (int)(threadIdx.x & 63u)) qr_wt_shadow += __builtin_amdgcn_s_memrealtime() - wt_s0;
#endif
        }
        if (lm && !occ) light_terms(G, mo, props, lg, L, dot, r, nrm, tex, col);
        le = (has && !(cl.lgt & QR_CLIGHT_LAST)) ? le + (u32)sizeof(CLight) : 0u;
    }

    o.col = col; o.hit = hit; o.loc = h.loc;
    o.tdir = {0, 0, 0}; o.rdir = {0, 0, 0};
    o.c_trn = 0.0f; o.c_rfl = 0.0f; o.x0 = 0.0f;
    o.want_tr = false; o.want_rf = false;
    o.lst_tr = 0; o.lst_rf = 0;

    if (act && pt_stage != 1 && o.rr_dead)
    {
        /* TR_end -> TR_mix with nothing added (3556-3583), reflections skipped (M_RFL empty, 3615-3617) */
        const qr_material *__restrict__ mt = (const qr_material *)(G + mo);
        float x0 = 1.0f - mt->c_trn;
        x0 = x0 - mt->c_rfl;
        o.x0 = cle(0.0f, x0) ? x0 : 0.0f;
    }
    else if (act && pt_stage != 1)
    {
        const qr_material *__restrict__ mt = (const qr_material *)(G + mo);
        const float m_trn_c = mt->c_trn, m_rfl_c = mt->c_rfl;
        float c_trn = m_trn_c, c_rfl = m_rfl_c;
        float x0 = 0.0f, x1, x2, x3, x4 = 0.0f, x5, x6 = 0.0f, x7 = 0.0f;
        bool m_trn = true;
        bool total_refl = false;

        /* transparency 3185-3552 */
        if (!(props & QR_PROP_OPAQUE))
        {
            QR_FLOPS(65);
            const bool do_rfi = (props & QR_PROP_REFRACT) || (props & QR_PROP_FRESNEL);
            bool tir = false;
            V3 nd = r.dir;
            if (do_rfi)
            {
                x1 = r.dir.x; x7 = x1 * x1; x0 = x7;
                x2 = r.dir.y; x7 = x2 * x2; x0 = x0 + x7;
                x3 = r.dir.z; x7 = x3 * x3; x0 = x0 + x7;
                x7 = rsq(x0);
                x1 = x1 * x7; x2 = x2 * x7; x3 = x3 * x7;
                x7 = x1 * nrm.x; x0 = x7;
                x7 = x2 * nrm.y; x0 = x0 + x7;
                x7 = x3 * nrm.z; x0 = x0 + x7;
                x4 = x0;
                x6 = mt->c_rfr;
                x0 = x0 * x6;
                x7 = x0 * x0;
                x7 = x7 + 1.0f;
                x7 = x7 - mt->rfr_2;
                if (props & QR_PROP_FRESNEL)
                {
                    m_trn = cle(0.0f, x7);
                    if (!m_trn)
                    {
                        c_trn = 0.0f;
                        c_rfl = m_rfl_c + m_trn_c;
                        tir = true;
                        total_refl = true;
                    }
                }
                if (!tir)
                {
                    x7 = __builtin_sqrtf(x7);
                    x0 = x0 + x7;
                    if (props & QR_PROP_REFRACT)
                    {
                        x5 = nrm.x * x0; x1 = x1 * x6; nd.x = x1 - x5;
                        x5 = nrm.y * x0; x2 = x2 * x6; nd.y = x2 - x5;
                        x5 = nrm.z * x0; x3 = x3 * x6; nd.z = x3 - x5;
                    }
                }
            }
            if (!tir)
            {
                if (props & QR_PROP_FRESNEL)
                {
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
                    x0 = fabs_bits(x0);
                    const float f = x0 * m_trn_c;   /* m_trn is true here */
                    c_trn = m_trn_c - f;
                    c_rfl = m_rfl_c + f;
                }
                o.want_tr = m_trn;
                o.tdir = nd;
                o.lst_tr = sd->lst[1 - side];
            }
        }

        bool rf_ok = true;
        if constexpr (PT)
        {
            /* path tracer, 3428-3466: from the third level on a Fresnel surface follows ONE of the two children, the
             * reflection with probability P = 0.25 + 0.5 c_rfl / (c_trn + c_rfl), weights divided by the probabilities */
            /* total inner reflection leaves through TR_tir -> TR_end (3279-3295) before the split: no number drawn */
            if (!(props & QR_PROP_OPAQUE) && (props & QR_PROP_FRESNEL) && depth_left <= QR_MAX_DEPTH - 2 && !total_refl)
            {
                const float u = pt_random(*rng);
                const float P = 0.25f + 0.5f * (c_rfl / (c_trn + c_rfl));
                if (u < P) { o.want_tr = false; c_trn = 0.0f; c_rfl = c_rfl / P; }
                else       { rf_ok = false; c_rfl = 0.0f; c_trn = c_trn / (1.0f - P); }
            }
        }

        /* TR_mix factor 3564-3573 */
        x0 = 1.0f - m_trn_c;
        x0 = x0 - m_rfl_c;
        x0 = cle(0.0f, x0) ? x0 : 0.0f;
        o.x0 = x0;

        /* reflections 3604-3815 */
        if ((props & QR_PROP_REFLECT) ||
            (!(props & QR_PROP_OPAQUE) && (props & QR_PROP_FRESNEL)))
        {
            QR_FLOPS(24);
            x1 = r.dir.x; x4 = nrm.x; x7 = x1 * x1; x0 = x7;
            x2 = r.dir.y; x5 = nrm.y; x7 = x2 * x2; x0 = x0 + x7;
            x3 = r.dir.z; x6 = nrm.z; x7 = x3 * x3; x0 = x0 + x7;
            x7 = rsq(x0);
            x1 = x1 * x7; x2 = x2 * x7; x3 = x3 * x7;
            x7 = x1 * x4; x0 = x7;
            x7 = x2 * x5; x0 = x0 + x7;
            x7 = x3 * x6; x0 = x0 + x7;
            x4 = x4 * x0; x1 = x1 - x4; x1 = x1 - x4; o.rdir.x = x1;
            x5 = x5 * x0; x2 = x2 - x5; x2 = x2 - x5; o.rdir.y = x2;
            x6 = x6 * x0; x3 = x3 - x6; x3 = x3 - x6; o.rdir.z = x3;

            if ((props & QR_PROP_FRESNEL) && (props & QR_PROP_OPAQUE))
            {
                QR_FLOPS(16);
                if (props & QR_PROP_METAL)
                {
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
                    x0 = fabs_bits(x0);
                }
                else
                {
                    x4 = x0;
                    x6 = mt->c_rfr;
                    x0 = x0 * x6;
                    x7 = x0 * x0;
                    x7 = x7 + 1.0f;
                    x7 = x7 - mt->rfr_2;
                    x7 = __builtin_sqrtf(x7);
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
                    x0 = fabs_bits(x0);
                }
                x0 = x0 - 1.0f;
                x0 = x0 * m_rfl_c;
                c_rfl = m_rfl_c + x0;
            }
            o.want_rf = rf_ok;
            o.lst_rf = sd->lst[side];
        }
        o.c_trn = c_trn;
        o.c_rfl = c_rfl;
    }
}

#endif /* QR_SHADE_HPP */
