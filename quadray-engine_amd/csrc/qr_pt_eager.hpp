/*
 * qr_pt_eager.hpp - the path tracer in the reference's EAGER shading order.
 *
 * render0 shades a hit the moment it passes the depth test, while the list is still being walked (tracer.cpp:
 * CC_clp -> *_mat -> ... -> back to OO_cyc); a later, closer surface shades again and overwrites the colour.  In
 * path-tracer mode shading draws random numbers and recurses, so the numbers a sample has consumed depend on that
 * order.  The deferred-shading kernel (render_wave) cannot reproduce it; this per-lane machine does: every lane walks
 * ITS list cell by cell (the per-lane walk of qr_walk.hpp), and a lane whose candidate is accepted leaves the walk,
 * shades that hit at once -- bounce sampling, the bounce's whole subtree, then the Fresnel split, the refraction
 * subtree, the reflection subtree, in the reference's order -- stores the colour as the level's current one and goes on
 * with the walk where it stopped.  Levels live in scratch (48 words each).  Slow by design: only the path tracer's
 * bit-exact mode uses it (qr_scene_set_pt(scn, 2)).
 */
#ifndef QR_PT_EAGER_HPP
#define QR_PT_EAGER_HPP

#define QR_EF_WORDS 48

__device__ __forceinline__ V3 pt_eager(const Ctx &cx, const Ray &primary, bool inside, int depth, float t_inf, u32 &rng, bool rt_shading)
{
    const BaseP B = cx.B;
    enum { M_WALK = 0, M_END = 1, M_SHADE1 = 2, M_SHADE2 = 3, M_DONE = 4 };
    float ef[QR_MAX_DEPTH + 1][QR_EF_WORDS];

    /* the current level */
    int level = 0;
    Ray cur = primary;
    u32 pos = inside ? (primary.list & ~31u) : 0u;
    WalkState w;
    w.txyz = {0, 0, 0}; w.trijk = {0, 0, 0}; w.tbuf = cur.tmax; w.resume = 0;
    float dd = cur.dir.x * cur.dir.x + cur.dir.y * cur.dir.y + cur.dir.z * cur.dir.z;
    w.tbd = w.tbuf * dd;
    V3 col = {0, 0, 0};                         /* colour of the last hit shaded at this level */
    Hit h; h.t = cur.tmax; h.srf = 0; h.side = 0; h.loc = {0, 0, 0};
    V3 acc = {0, 0, 0};                         /* the hit being shaded: colour so far */
    V3 ptex = {0, 0, 0}; float pldff = 0.0f;
    float c_trn = 0.0f, c_rfl = 0.0f, x0 = 0.0f;
    V3 rdir = {0, 0, 0}; u32 lst_rf = 0; int want_rf = 0;
    int phase = 0;                              /* which child the level waits for: 1 bounce, 2 refraction, 3 reflection */
    int mode = inside ? (pos != 0 ? M_WALK : M_END) : M_DONE;
    bool rr_dead = false;
    u32 p_op = 0, p_srf = 0;
    V3 ret = {0, 0, 0};
    Counters cnt = {0, 0, 0, 0};

    auto push = [&]() {
        float *f = ef[level];
        f[0] = cur.org.x; f[1] = cur.org.y; f[2] = cur.org.z; f[3] = cur.dir.x; f[4] = cur.dir.y; f[5] = cur.dir.z;
        f[6] = cur.tmin; f[7] = u2f(cur.list); f[8] = u2f(cur.osrf); f[9] = u2f((u32)cur.oflg);
        f[10] = cur.ploc.x; f[11] = cur.ploc.y; f[12] = cur.ploc.z;
        f[13] = u2f(pos); f[14] = w.tbuf; f[15] = w.txyz.x; f[16] = w.txyz.y; f[17] = w.txyz.z;
        f[18] = w.trijk.x; f[19] = w.trijk.y; f[20] = w.trijk.z;
        f[21] = col.x; f[22] = col.y; f[23] = col.z;
        f[24] = h.t; f[25] = u2f(h.srf); f[26] = u2f((u32)h.side); f[27] = h.loc.x; f[28] = h.loc.y; f[29] = h.loc.z;
        f[30] = acc.x; f[31] = acc.y; f[32] = acc.z;
        f[33] = ptex.x; f[34] = ptex.y; f[35] = ptex.z; f[36] = pldff;
        f[37] = c_trn; f[38] = c_rfl; f[39] = x0; f[40] = rdir.x; f[41] = rdir.y; f[42] = rdir.z;
        f[43] = u2f(lst_rf); f[44] = u2f((u32)want_rf); f[45] = u2f((u32)phase);
    };
    auto pop = [&]() {
        const float *f = ef[level];
        cur.org = {f[0], f[1], f[2]}; cur.dir = {f[3], f[4], f[5]};
        cur.tmin = f[6]; cur.list = f2u(f[7]); cur.osrf = f2u(f[8]); cur.oflg = (int)f2u(f[9]);
        cur.ploc = {f[10], f[11], f[12]};
        pos = f2u(f[13]); w.tbuf = f[14]; w.txyz = {f[15], f[16], f[17]}; w.trijk = {f[18], f[19], f[20]};
        col = {f[21], f[22], f[23]};
        h.t = f[24]; h.srf = f2u(f[25]); h.side = (int)f2u(f[26]); h.loc = {f[27], f[28], f[29]};
        acc = {f[30], f[31], f[32]};
        ptex = {f[33], f[34], f[35]}; pldff = f[36];
        c_trn = f[37]; c_rfl = f[38]; x0 = f[39]; rdir = {f[40], f[41], f[42]};
        lst_rf = f2u(f[43]); want_rf = (int)f2u(f[44]); phase = (int)f2u(f[45]);
        dd = cur.dir.x * cur.dir.x + cur.dir.y * cur.dir.y + cur.dir.z * cur.dir.z;
        w.tbd = w.tbuf * dd; w.resume = 0;
        rr_dead = false;                                    /* a level that spawned a child had survived */
    };
    /* a child ray of the hit being shaded starts its own level */
    auto spawn = [&](V3 dir, u32 list, int flg, int ph) {
        phase = ph;
        const V3 hp = { cur.dir.x * h.t + cur.org.x, cur.dir.y * h.t + cur.org.y, cur.dir.z * h.t + cur.org.z };
        const u32 srf = h.srf; const V3 loc = h.loc;
        push();
        level++;
        cur.org = hp; cur.dir = dir; cur.tmin = 0.0f; cur.tmax = t_inf;
        cur.list = list; cur.osrf = srf; cur.oflg = flg; cur.ploc = loc;
        pos = list & ~31u;
        w.txyz = {0, 0, 0}; w.trijk = {0, 0, 0}; w.tbuf = t_inf; w.resume = 0;
        dd = dir.x * dir.x + dir.y * dir.y + dir.z * dir.z;
        w.tbd = w.tbuf * dd;
        col = {0, 0, 0};
        h.t = t_inf; h.srf = 0; h.side = 0; h.loc = {0, 0, 0};
        mode = pos != 0 ? M_WALK : M_END;
    };
    /* refraction done (or absent) with colour c: reflection child or the hit is finished */
    auto after_tr = [&](V3 c) {
        if (want_rf && (depth - level) != 0) { acc = c; spawn(rdir, lst_rf, h.side, 3); }
        else
        {
            if (want_rf) { c.x = 0.0f + c.x; c.y = 0.0f + c.y; c.z = 0.0f + c.z; }
            col = c; mode = M_WALK;
        }
    };

    while (any_lane(mode != M_DONE))
    {
        /* ---- walk: every walking lane up to its next accepted hit or the end of its list ---- */
        for (;;)
        {
            const lm_t walking = LM(mode == M_WALK);
            if (walking == 0) break;
            const lm_t pend = LM(p_op != 0) & walking;
            const lm_t adv = walking & ~pend;
            if (adv != 0 && __popcll(pend) < 4)
            {
                if (lane_of(adv))
                {
                    const u32x4 a0 = *(const QR_CONST u32x4 *)(B + pos), a1 = *(const QR_CONST u32x4 *)(B + pos + 16),
                                b0 = *(const QR_CONST u32x4 *)(B + pos + 32), b1 = *(const QR_CONST u32x4 *)(B + pos + 48);
                    const u32 op = a0.x, srf_off = a0.y;
                    u32 next = pos + 32;
                    if (op == 0) { mode = M_END; next = pos; }
                    else if (op & QR_OPT_SOLVER) { p_op = op; p_srf = srf_off; }      /* no cull: ours only removes work */
                    else if (op & QR_OPT_BV)
                    {
                        next = pos + 64;
                        V3 df, ry;
                        cell_space(B, op, srf_off, u2f(b0.x), u2f(b0.y), u2f(b0.z), cur, w, df, ry);
                        if (!bv_hit(ry, df, u2f(b1.x), u2f(b1.y), u2f(b1.z), u2f(b1.w))) next = a0.z;
                    }
                    else
                    {
                        const u32x4 p0 = *(const QR_CONST u32x4 *)(B + srf_off);
                        V3 d;
                        d.x = cur.org.x - u2f(p0.x); d.y = cur.org.y - u2f(p0.y); d.z = cur.org.z - u2f(p0.z);
                        w.txyz = xform(B, srf_off, (op & QR_OPF_FULLM) != 0, d);
                        w.trijk = xform(B, srf_off, (op & QR_OPF_FULLM) != 0, cur.dir);
                    }
                    (void)a1;
                    pos = next;
                }
            }
            else
            {
                if (lane_of(pend))
                {
                    SurfS s;
                    ld_surf_lane(B, p_srf, s);
                    const float tb = w.tbuf;
                    solve_cell<false, true, false, true>(B, p_op, p_srf, s, cur, dd, w, h);     /* with the cell's clipper program, per lane */
                    p_op = 0;
                    if (w.tbuf != tb) mode = M_SHADE1;      /* accepted: shaded before the walk goes on */
                }
            }
        }

        /* ---- a list has ended: its colour goes to the level that waits for it ---- */
        if (any_lane(mode == M_END))
        {
            if (mode == M_END)
            {
                if (level == 0) { ret = col; mode = M_DONE; }
                else
                {
                    const V3 r = col;
                    level--;
                    pop();
                    if (phase == 1)
                    {
                        /* PT_ret 2598-2620: child * l_dff * texture, + emission (acc) */
                        float x1 = r.x * pldff, x2 = r.y * pldff, x3 = r.z * pldff;
                        x1 = x1 * ptex.x; x2 = x2 * ptex.y; x3 = x3 * ptex.z;
                        acc.x = x1 + acc.x; acc.y = x2 + acc.y; acc.z = x3 + acc.z;
                        mode = M_SHADE2;
                    }
                    else if (phase == 2)
                    {
                        V3 c;
                        c.x = r.x * c_trn + acc.x * x0; c.y = r.y * c_trn + acc.y * x0; c.z = r.z * c_trn + acc.z * x0;
                        after_tr(c);
                    }
                    else
                    {
                        col.x = r.x * c_rfl + acc.x; col.y = r.y * c_rfl + acc.y; col.z = r.z * c_rfl + acc.z;
                        mode = M_WALK;
                    }
                }
            }
        }

        /* ---- stage 1 of a fresh hit: texture, normal, emission, bounce sampling ---- */
        if (any_lane(mode == M_SHADE1))
        {
            Shaded o;
            /* rt_shading (self-test of this machine): the ray tracer's shading, lights and shadows included, in the
             * eager order -- it draws no numbers, so the frame must be the ray tracer's */
            if (rt_shading) { shade<false, false, false>(cx, mode == M_SHADE1, false, cur, h, o, cnt); o.want_pt = false; }
            else shade<false, false, true>(cx, mode == M_SHADE1, false, cur, h, o, cnt, &rng, depth - level, 1);
            if (mode == M_SHADE1)
            {
                acc = o.col;                                /* the material's emission */
                ptex = o.ptex; pldff = o.pldff;
                rr_dead = o.rr_dead;                        /* no spawn follows a dead sample: stage 2 comes next, same level */
                if (o.want_pt && (depth - level) != 0) spawn(o.pdir, o.lst_pt, h.side, 1);
                else mode = M_SHADE2;
            }
        }

        /* ---- stage 2: Fresnel split, refraction child, reflection child ---- */
        if (any_lane(mode == M_SHADE2))
        {
            Shaded o;
            if (rt_shading) shade<false, false, false>(cx, mode == M_SHADE2, false, cur, h, o, cnt);
            else shade<false, false, true>(cx, mode == M_SHADE2, false, cur, h, o, cnt, &rng, depth - level, rr_dead ? 3 : 2);
            if (mode == M_SHADE2)
            {
                c_trn = o.c_trn; c_rfl = o.c_rfl; x0 = o.x0; rdir = o.rdir; lst_rf = o.lst_rf; want_rf = o.want_rf ? 1 : 0;
                if (o.want_tr && (depth - level) != 0) spawn(o.tdir, o.lst_tr, h.side | FLAG_PASS_THRU, 2);
                else
                {
                    V3 c;
                    c.x = 0.0f + acc.x * x0; c.y = 0.0f + acc.y * x0; c.z = 0.0f + acc.z * x0;
                    after_tr(c);
                }
            }
        }
    }
    return ret;
}

#endif /* QR_PT_EAGER_HPP */
