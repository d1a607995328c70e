/*
 * qr_hierarchy.h - host-side object hierarchy: animators and the hierarchical transform update that turn a
 * tree of objects (arrays, surfaces, cameras, lights, each with scale / rotation / position) into the
 * transform fields of the snapshot records (include/qr_scene.h) one frame reads.
 *
 * In the reference this is phase 0.5 of rt_Scene::render plus the update_fields of every object:
 *   rt_Object::update_status / update_matrix   core/engine/object.cpp:175-389
 *   rt_Array::update_status / update_matrix / update_object   object.cpp:1669-1756
 *   rt_Node / rt_Array / rt_Surface / rt_Quadric / shape update_fields   object.cpp:813-843, 1761-1825,
 *       2472-2503, 3012-3063, 3120-3910
 *   rt_Light::update_fields object.cpp:649-667, camera vectors of rt_Scene::render engine.cpp:3029-3050, 3256-3260
 *   rt_Surface::update_minmax / update_bounds, the shapes' adjust_minmax, rt_Node::update_bbgeom,
 *       rt_Array::update_bounds   object.cpp:849-1091, 1830-2318, 2508-2845, 2957-3956   (qr_hierarchy_bounds)
 *   matrix_from_transform / matrix_mul_matrix / matrix_inverse   core/engine/rtgeom.cpp:59-203
 *   animators: rt_FUNC_ANIM3D of rt_OBJECT, called from rt_Object::update_status (object.cpp:182-190)
 *
 * A drop-in backend never needs it (the engine updates its own hierarchy before it calls render0); it is for
 * hosts that hold a snapshot and want the next frame of an animated scene without the engine: animate the
 * nodes, qr_hierarchy_apply the result to the snapshot, rebuild the lists (qr_snapshot_build_lists_c), render.
 *
 * Scope (what a frame-to-frame update may change): positions and rotations anywhere in the tree as long as the set
 * of transform nodes ("trnodes": nodes with a non-trivial rotation) and every node's axis mapping and scalers stay
 * the same, and no array with a bounding volume moves -- unless the node tables carry the bounds inputs and
 * QR_HIER_BOUNDS is given: then clip boxes and bounding volumes are recomputed like the engine does (rt_Surface::
 * update_minmax, rt_Array::update_bounds).
 *
 * A changing SET of transform nodes (round 4) -- any object may start or stop turning.  Needs both node tables (`base`),
 * QR_HIER_BOUNDS and QR_HIER_RESET_TILES, and returns a LARGER snapshot:
 *   - the global list holds one element per ARRAY that is the transform node of surfaces, in front of its members, which stand
 *     together (rt_SceneThread::insert, engine.cpp:1148-1214).  When surfaces change that array -- an array under a turning
 *     array starts to turn itself, an array returns to a right angle, the rotating light arrays of the demo scenes -- the list
 *     is regrouped: every group where its first member stood, members in list order, other surfaces where they were;
 *   - every clipper list is regrouped the same way between its accum markers (engine.cpp:1845-1947: transform-node markers in
 *     front of the clippers that share an array);
 *   - an array that becomes a transform node and has no record gets one: the k-th such array in NODE ORDER holds record
 *     n_srf + k of the returned snapshot (enter it into the table before patching that snapshot again);
 *   - a SURFACE that starts or stops being its OWN transform node (a right angle <-> any angle) changes no list: its record
 *     takes / loses the matrix;
 *   - a textured plane's axis scalers enter the texture scale and offset of its two materials (rt_Plane::update_fields,
 *     object.cpp:2893-2938): they follow, from qr_node.tex when the table carries it, else read back from the record where
 *     the base scalers are 1 (exact there); otherwise QR_ERR_UNSUP naming the node.
 * The ORDER of the regrouped lists is the previous order, not the view order the engine's insert (engine.cpp:1216-1645 over
 * bbox_sort, rtgeom.cpp:1244) would give them this frame: order decides nothing but exact depth ties.  The engine's removal
 * of fully hidden surfaces from its camera list (RT_OPTS_REMOVE) is undone, not redone: with `base` and QR_HIER_RESET_TILES a
 * list that lacks surfaces of the table gets them back behind the others.
 *
 * Plain C ABI; fp32 arithmetic in the reference's operation order: the results are bit-identical to the engine's
 * (tests/test_hierarchy.py, fixtures dumped from the engine by oracle/ref_driver.cpp --tree).
 */
#ifndef QR_HIERARCHY_H
#define QR_HIERARCHY_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* object tags, same values as RT_TAG_* (format.h:116-133) */
#define QR_NODE_ARRAY   (-1)
#define QR_NODE_CAMERA  100
#define QR_NODE_LIGHT   101

/* the optimisation flags the update depends on, same values as RT_OPTS_* (format.h:43-44) */
#define QR_OPTS_FSCALE  (1 << 3)
#define QR_OPTS_TARRAY  (1 << 4)
#define QR_OPTS_ADJUST  (1 << 7)    /* bounding boxes shrink to what outer clippers leave (rt_Surface::update_minmax) */

/* One object of the hierarchy (rt_OBJECT + what the engine's object keeps of it).  Parents precede children. */
typedef struct qr_node
{
    int32_t parent;         /* index of the parent array, -1 for the root                           */
    int32_t tag;            /* QR_NODE_ARRAY, surface tag 0..8 (qr_scene.h), QR_NODE_CAMERA / _LIGHT  */
    float   scl[3];         /* rt_TRANSFORM3D (format.h:186-191): scalers,                           */
    float   rot[3];         /*   rotation in degrees around x, y, z,                                 */
    float   pos[3];         /*   position                                                            */
    float   shape[3];       /* surface parameters: cylinder / sphere rad; cone rat; paraboloid / paracylinder par;
                             * hyperboloid / hypercylinder rat, hyp; hyperparaboloid pr1, pr2 (format.h:496-727) */
    int32_t srf;            /* snapshot record of a surface / of an array's transform node, -1 none   */
    int32_t inb, bvb;       /* arrays: records of the inner / outer bounding volume, -1 none (never written:
                             * they only tell qr_hierarchy_apply that the array has a bounding volume)  */
    int32_t lgt;            /* lights: qr_light index, -1 none                                        */
    int32_t anim;           /* slot of qr_hierarchy_animate's tables, -1 no animator                  */
    float   pov;            /* cameras: distance of the screen plane (rt_Camera::pov)                 */
    /* inputs of the bounds update (qr_hierarchy_bounds, QR_HIER_BOUNDS); a table without them leaves them 0 / -1 */
    int32_t bvnode;         /* the array whose bounding volume holds this node (rt_Node::bvnode: the nearest array up the
                             * tree with RT_REL_BOUND_ARRAY / _INDEX covering it), -1 none                            */
    int32_t nverts;         /* surfaces: corners of the bounding box the engine allots when the scene description bounds
                             * the shape (4 plane, 8 quadric; the constructors' conditions, object.cpp:2870-2876, 3096-3101,
                             * 3190-3196, ...), 0 when it does not                                                   */
    float   lmin[3], lmax[3];   /* surfaces: the axis clippers of the scene description (rt_SURFACE::min / max, local axes
                                 * I, J, K; -/+FLT_MAX = none, as RT_INF)                                            */
    /* planes: texture scale (x, y) and position (x, y) of the outer, then of the inner material before the plane's axis
     * scalers enter them (rt_Material::scl, rt_SIDE::pos; rt_Plane::update_fields, object.cpp:2893-2938 multiplies the
     * scalers in).  has_tex 0: not given -- then a textured plane's scalers may only change away from 1 */
    float   tex[8];
    int32_t has_tex;
    int32_t pad_[3];
} qr_node;

/* What the bounds update computes per node (members of rt_BOUND / rt_SHAPE and of rt_Array's three boxes). */
typedef struct qr_node_bounds
{
    float   bmin[3], bmax[3];   /* surface: bounding box in its sub-world space; array: bvbox (world space)          */
    float   cmin[3], cmax[3];   /* surface: clipping box (sides that clip, the others at -/+FLT_MAX)                   */
    float   mid[3], rad;        /* centre and radius of that bounding box's corners (rad = FLT_MAX: unbounded)         */
    int32_t nverts;             /* corners computed (0: none)                                                          */
    float   inmin[3], inmax[3], inmid[3], inrad;     /* arrays: inbox (the transform node's space)                    */
    float   trmin[3], trmax[3], trrad;               /* arrays: trbox                                                   */
    int32_t inb_form, bvb_form; /* arrays: what the records of the bounding volumes hold -- 0 nothing (left alone),
                                 * 1 the ellipsoid through the box's corners, 2 (inb only) the sphere around its centre */
} qr_node_bounds;

/* What the update computes per node (the members of rt_Object it fills). */
typedef struct qr_node_state
{
    float   mtx[16];        /* rt_Object::mtx, row-major 4x4, row 3 = position; relative to the trnode when there is
                             * one and the node is not it                                             */
    int32_t map[4];         /* axis mapping of a trivial transform: local axis i is sub-world axis map[i] */
    int32_t sgn[4];         /*   with sign sgn[i]                                                     */
    float   scl[4];         /*   and scale scl[map[i]]                                                */
    int32_t trnode;         /* node with the non-trivial transform this node lives under (itself possible), -1 */
    int32_t obj_has_trm;    /* bit 0 scaling, bit 1 rotation somewhere up the hierarchy               */
    int32_t mtx_has_trm;    /* same for the node's own matrix                                         */
    int32_t pad;
} qr_node_state;

/* the hierarchical update: nodes[0..n) -> out[0..n).  opts: the scene's RT_OPTS_* word (only FSCALE / TARRAY matter). */
int qr_hierarchy_update(const qr_node *nodes, int32_t n, uint32_t opts, qr_node_state *out);

/*
 * Animators.  fns[k] is called for every node with anim == k as fns[k](time, last, &node.scl[0], users[k]) -- the
 * nine floats scl, rot, pos -- where last is the time of the node's previous update, 0 at the first one
 * (node_time[i] == -1), and not at all when node_time[i] == time; node_time[i] becomes time (object.cpp:182-195).
 */
typedef void (*qr_anim_fn)(int64_t time, int64_t last_time, float *trm, void *user);
int qr_hierarchy_animate(qr_node *nodes, int32_t n, int64_t time, int64_t *node_time,
                         const qr_anim_fn *fns, void *const *users, int32_t n_fns);

/* the two animators of the reference's demo scenes, as qr_anim_fn with a qr_anim_params user record:
 * spin:  rot[axis] += (time - last) / 50 * rate, minus 360 once when >= 360  (data/scenes/scn_demo01.h:513-524, 550-561)
 * swing: rot[axis] = rate * sin(time / period)                                (data/scenes/scn_demo03.h:464-470) */
typedef struct qr_anim_params { int32_t axis; float rate; float period; int32_t pad; } qr_anim_params;
void qr_anim_spin(int64_t time, int64_t last_time, float *trm, void *user);
void qr_anim_swing(int64_t time, int64_t last_time, float *trm, void *user);

/*
 * Bounding and clipping boxes of every node for the transforms of `nodes` (rt_Surface::update_minmax / update_bounds,
 * rt_Array::update_bounds, object.cpp:1830-2318, 2534-2845): surfaces from their description's axis clippers, their shape
 * and -- with QR_OPTS_ADJUST -- the cuts of their outer clippers (read from the snapshot's clipper lists), arrays from their
 * members.  `blob` is the snapshot the nodes' record indices refer to.
 */
int qr_hierarchy_bounds(const void *blob, uint64_t size, const qr_node *nodes, int32_t n, uint32_t opts, qr_node_bounds *out);

#define QR_HIER_BOUNDS      2u  /* qr_hierarchy_apply: the node tables carry the bounds inputs (bvnode, nverts, lmin, lmax):
                                 * clip boxes of surfaces (min / max / which sides clip) and the records of arrays'
                                 * bounding volumes are recomputed, so arrays with bounding volumes may move and axis
                                 * mappings / scalers may change */
#define QR_HIER_RESET_TILES 1u  /* qr_hierarchy_apply: point every tile at the global list (the camera may have moved;
                                 * QR_UPLOAD_REBIN_TILES bins again on the GPU) */
#define QR_HIER_REGROUP     4u  /* qr_hierarchy_apply (with `base`, _RESET_TILES and _BOUNDS): rebuild the global list and every
                                 * clipper list from their leaves even if no surface changes its transform node -- what a
                                 * change would trigger by itself; on an unchanged scene the result has the input's structure */

/*
 * Writes the transform fields the nodes' new state implies into a copy of the snapshot: surfaces and array nodes
 * (pos, tci/tcj/tck of transform nodes, has_trm, shift, axes, trnode, quadric coefficients sci/scj), lights (pos), and
 * for node `camera` (-1: leave the camera alone) the frame's org, dir, hor, ver.  `base` (may be NULL) are the nodes the
 * snapshot was captured with: with them the call checks the scope stated at the top and fails with QR_ERR_UNSUP outside
 * it.  Per-surface lists and tile lists of the snapshot are NOT rebuilt: run qr_snapshot_build_lists_c on the result.
 * The caller releases *out_blob with qr_free.
 */
int qr_hierarchy_apply(const void *blob, uint64_t size, const qr_node *base, const qr_node *next, int32_t n,
                       uint32_t opts, int32_t camera, uint32_t flags, void **out_blob, uint64_t *out_size);

#ifdef __cplusplus
}
#endif

#endif /* QR_HIERARCHY_H */
